// v_fmac_f64_dpp with row_newbcast on gfx950: acc -= bcast_K(src) * u in one instruction -- semantics check, including which lanes
// may serve as the broadcast's source under a partial EXEC mask (an enabled lane: yes; a disabled lane: reads as 0).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/dpp_fmac_probe tools/dpp_fmac_probe.hip && /tmp/dpp_fmac_probe
// MI355X: "all enabled: 0 wrong | source enabled, some lanes off: 0 wrong | source lane disabled: 48 wrong (... read as 0)".
// llvm-mc accepts the DPP form only for v_fmac_f64 among the float64 ALU ops (v_add / v_mul / v_fma_f64: "dpp variant of this
// instruction is not supported"), and only with row_newbcast ("DP ALU dpp only supports row_newbcast").
#include <hip/hip_runtime.h>
#include <cstdio>
template <int K>
__device__ __forceinline__ void fmac_bcast_neg(double &acc, double src, double u) {
  asm volatile("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(u), "n"(K));
}
__global__ void k(double *o) {
  const int lane = threadIdx.x;
  double acc = lane, src = 100.0 + lane, u = 2.0;
  fmac_bcast_neg<3>(acc, src, u);           // all lanes enabled
  o[lane] = acc;
  double acc2 = lane;
  if ((lane & 15) > 3) fmac_bcast_neg<5>(acc2, src, u);   // source lane 5 enabled, lanes 0..3 disabled
  o[64 + lane] = acc2;
  double acc3 = lane;
  if ((lane & 15) > 3) fmac_bcast_neg<2>(acc3, src, u);   // source lane 2 DISABLED
  o[128 + lane] = acc3;
}
int main() {
  double *d, h[192];
  hipMalloc(&d, sizeof(h));
  k<<<1, 64>>>(d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad1 = 0, bad2 = 0, bad3 = 0;
  for (int l = 0; l < 64; l++) {
    double e1 = l - (100.0 + (l & 48) + 3) * 2.0;
    double e2 = (l & 15) > 3 ? l - (100.0 + (l & 48) + 5) * 2.0 : l;
    double e3 = (l & 15) > 3 ? l - (100.0 + (l & 48) + 2) * 2.0 : l;
    bad1 += h[l] != e1; bad2 += h[64 + l] != e2; bad3 += h[128 + l] != e3;
  }
  printf("all enabled: %d wrong | source enabled, some lanes off: %d wrong | source lane disabled: %d wrong (lane 20 got %g, expected %g if readable, %g if read as 0)\n",
         bad1, bad2, bad3, h[128 + 20], 20 - (100.0 + 16 + 2) * 2.0, 20.0);
  return 0;
}
