// Issue rate of v_fma_f64 and of the float64 transcendentals (v_rcp_f64 / v_rsq_f64 / v_sqrt_f64, and a float32 reciprocal behind
// two conversions) at 1-4 waves per SIMD, eight independent chains per lane.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate tools/valu_rate.hip && /tmp/valu_rate
// Measured on MI355X (20 000 x 8 instructions per wave, ms):
//   waves/CU  4: fma_f64 0.68 | rcp_f64 1.43 | rsq_f64 1.40 | sqrt_f64 1.37      -> 4.25 ns per wave-instruction per SIMD
//   waves/CU  8: fma_f64 0.87 | rcp_f64 2.42 | rsq_f64 2.36 | sqrt_f64 2.33      -> 2.72 ns
//   waves/CU 12: fma_f64 1.17 | rcp_f64 3.51                                     -> 2.44 ns
//   waves/CU 16: fma_f64 1.44 | rcp_f64 4.63                                     -> 2.25 ns (the asymptote; transcendentals ~3.2x)
// i.e. one wave alone reaches half of the float64 FMA rate, two waves 83 % of it, and the transcendentals issue at about a third
// of the FMA rate.  ik_kernel (two waves per SIMD by register count) spends 1 593 VALU instructions per solve and 10.5 us per
// solve and wave: 1 593 x 2 x 2.72 ns = 8.7 us, so it runs at ~82 % of what two waves of independent FMAs achieve.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ void k(double *o, int iters) {
  double a[8];
  for (int i = 0; i < 8; i++) a[i] = 1.0 + threadIdx.x * 1e-3 + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (OP == 0) a[i] = fma(a[i], 1.0000001, 1e-9);
      if (OP == 1) a[i] = __builtin_amdgcn_rcp(a[i]);
      if (OP == 2) a[i] = __builtin_amdgcn_rsq(a[i]);
      if (OP == 3) { float f = (float)a[i]; f = __builtin_amdgcn_rcpf(f); a[i] = f; }
      if (OP == 4) a[i] = __builtin_amdgcn_sqrt(a[i]);
    }
  }
  double s = 0;
  for (int i = 0; i < 8; i++) s += a[i];
  o[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP>
float run(int waves_per_cu, int iters, double *d) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  dim3 grid(256 * waves_per_cu), block(64);
  k<OP><<<grid, block>>>(d, 10);
  hipEventRecord(a);
  k<OP><<<grid, block>>>(d, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms;
}
int main() {
  double *d;
  hipMalloc(&d, 256 * 16 * 64 * 8);
  const int iters = 20000;
  for (int w : {4, 8, 12, 16}) {
    float t0 = run<0>(w, iters, d), t1 = run<1>(w, iters, d), t2 = run<2>(w, iters, d), t3 = run<3>(w, iters, d), t4 = run<4>(w, iters, d);
    printf("waves/CU %d: fma_f64 %.2f ms | rcp_f64 %.2f (x%.1f) | rsq_f64 %.2f (x%.1f) | cvt+rcp_f32+cvt %.2f (x%.1f) | sqrt_f64 %.2f (x%.1f)\n", w, t0, t1, t1 / t0, t2, t2 / t0, t3, t3 / t0, t4, t4 / t0);
  }
  return 0;
}
