// A cross-lane reduction on the matrix core: with A = all ones, v_mfma_f64_16x16x4_f64 returns in EVERY lane the sum of its B
// operand over the four 16-lane groups (same local lane): D[row][col] = sum_k 1 * B[k][col], B[k][col] = lane 16 k + col,
// D's column = lane & 15.  That is box_qp_struct's group_sum4 (ik_kernel.hip.h) in one instruction instead of ~8 VALU
// instructions per value (permlane16/32 swaps).  This probe checks the semantics -- full EXEC, partial EXEC (the Schur sum runs
// with only the core rows enabled) -- and times ten sums per iteration both ways.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_probe tools/mfma_groupsum_probe.hip && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double msum(double v) {
  double4v c = {0.0, 0.0, 0.0, 0.0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(1.0, v, c, 0, 0, 0);
  return c[0];
}
__device__ __forceinline__ double psum(double v) {
  {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    v = __hiloint2double((int)rh[0], (int)rl[0]) + __hiloint2double((int)rh[1], (int)rl[1]);
  }
  {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    v = __hiloint2double((int)rh[0], (int)rl[0]) + __hiloint2double((int)rh[1], (int)rl[1]);
  }
  return v;
}
__global__ void check(double *o) {
  const int lane = threadIdx.x;
  const double v = 1.0 + lane * 0.5 + (lane >> 4) * 1000.0;
  o[lane] = msum(v);
  double keep = -7.0;
  if ((lane & 15) >= 7) keep = msum(v);  // partial EXEC: lanes with local index < 7 are off (and must keep their -7)
  o[64 + lane] = keep;
  o[128 + lane] = psum(v);
}
template <int MODE>
__global__ void timing(double *o, int iters) {
  double x[10];
  for (int i = 0; i < 10; i++) x[i] = threadIdx.x * 0.001 + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 10; i++) x[i] = (MODE ? msum(x[i]) : psum(x[i])) * 0.25 + 1e-3;
  }
  double s = 0;
  for (int i = 0; i < 10; i++) s += x[i];
  o[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  double *d, h[192];
  hipMalloc(&d, 1 << 24);
  check<<<1, 64>>>(d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad_full = 0, bad_part = 0, bad_perm = 0;
  for (int l = 0; l < 64; l++) {
    const int a = l & 15;
    double e = 0;
    for (int g = 0; g < 4; g++) e += 1.0 + (16 * g + a) * 0.5 + g * 1000.0;
    bad_full += h[l] != e;
    bad_part += h[64 + l] != (a >= 7 ? e : -7.0);
    bad_perm += h[128 + l] != e;
  }
  printf("mfma sum, all lanes: %d wrong | only local lanes >= 7 enabled: %d wrong (lane 3 holds %g, lane 9 holds %g, expected %g) | permlane sum: %d wrong\n",
         bad_full, bad_part, h[64 + 3], h[64 + 9], h[9], bad_perm);
  for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
    const int blocks = 256 * 4 * waves_per_simd, iters = 20000;
    float ms[2];
    for (int mode = 0; mode < 2; ++mode) {
      hipEvent_t a, b;
      hipEventCreate(&a); hipEventCreate(&b);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        if (mode) timing<1><<<blocks, 64>>>(d, iters); else timing<0><<<blocks, 64>>>(d, iters);
        hipEventRecord(b);
        hipEventSynchronize(b);
        hipEventElapsedTime(&ms[mode], a, b);
      }
    }
    printf("%d wave(s) per SIMD, 10 sums per iteration: permlane %.1f ns, mfma %.1f ns per iteration and wave\n", waves_per_simd,
           ms[0] * 1e6 / iters, ms[1] * 1e6 / iters);
  }
  return 0;
}
