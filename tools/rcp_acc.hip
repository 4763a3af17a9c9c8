// Accuracy of the v_rcp_f64 / v_rsq_f64 seeds and of the refinement steps used by the kernels' fast_rcp / fast_rsqrt /
// fast_sqrt / fast_div (same formulas as gmr_amd/csrc/ik_kernel.hip.h), relative to long double on the host.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/rcp_acc tools/rcp_acc.hip && /tmp/rcp_acc
// Measured on MI355X: seeds 4.6e-8 / 5.2e-8; one Newton step 2.2e-15 / 4.1e-15; fast_rcp 1.11e-16, fast_rsqrt 1.38e-16,
// fast_sqrt 1.11e-16, fast_div 1.11e-16.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
constexpr int K = 8;
__global__ void k(const double *x, const double *y, double *o, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = x[i], a = y[i];
  const double r0 = __builtin_amdgcn_rcp(v), s0 = __builtin_amdgcn_rsq(v);
  o[K * i] = r0;
  const double r1 = fma(r0, fma(-v, r0, 1.0), r0);
  o[K * i + 1] = r1;
  const double e = fma(-v, r0, 1.0);
  o[K * i + 2] = fma(r0, fma(e, e, e), r0);  // fast_rcp
  o[K * i + 3] = s0;
  const double s1 = s0 * fma(-0.5 * v * s0, s0, 1.5);
  o[K * i + 4] = s1;
  const double es = fma(-v * s0, s0, 1.0);
  o[K * i + 5] = fma(s0, es * fma(0.375, es, 0.5), s0);  // fast_rsqrt
  double s = v * s1;
  o[K * i + 6] = fma(0.5 * s1, fma(-s, s, v), s);  // fast_sqrt
  const double q = a * r1;
  o[K * i + 7] = fma(r1, fma(-q, v, a), q);  // fast_div(a, v)
}
int main() {
  const int n = 1 << 20;
  std::vector<double> x(n), y(n), o((size_t)K * n);
  unsigned long long st = 88172645463325252ull;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (st >> 11) * (1.0 / 9007199254740992.0); };
  for (int i = 0; i < n; i++) { x[i] = std::exp((rnd() - 0.5) * 40.0); y[i] = std::exp((rnd() - 0.5) * 40.0); }
  double *dx, *dy, *dout;
  hipMalloc(&dx, n * 8); hipMalloc(&dy, n * 8); hipMalloc(&dout, (size_t)K * n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipMemcpy(dy, y.data(), n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, dy, dout, n);
  hipMemcpy(o.data(), dout, (size_t)K * n * 8, hipMemcpyDeviceToHost);
  double e[K] = {};
  for (int i = 0; i < n; i++) {
    const long double v = x[i], rc = 1.0L / v, rs = 1.0L / sqrtl(v), ref[K] = {rc, rc, rc, rs, rs, rs, sqrtl(v), (long double)y[i] / v};
    for (int j = 0; j < K; j++) e[j] = fmax(e[j], (double)fabsl(((long double)o[(size_t)K * i + j] - ref[j]) / ref[j]));
  }
  printf("max relative error over %d samples in [2e-9, 5e8] (double eps/2 = 1.11e-16)\n", n);
  printf("  v_rcp_f64 seed %.3e | one Newton step %.3e | fast_rcp (third order) %.3e\n", e[0], e[1], e[2]);
  printf("  v_rsq_f64 seed %.3e | one Newton step %.3e | fast_rsqrt (third order) %.3e\n", e[3], e[4], e[5]);
  printf("  fast_sqrt %.3e | fast_div %.3e\n", e[6], e[7]);
  return 0;
}
