// tree_chain.hip.h -- shared machinery of the two input-adapter kernels (bvh_kernel.hip.h, smplx_kernel.hip.h).
//
// Both adapters chain local joint transforms down a kinematic tree for every frame of a clip
// (reference: quat_fk, general_motion_retargeting/utils/lafan_vendor/utils.py:88-103; the orientation chaining of
// get_smplx_data_offline_fast, general_motion_retargeting/utils/smpl.py:179-196).  The reference walks the joints in
// hierarchy order per frame; here a wavefront takes whole frames with LANE = JOINT, so that
//   * a frame's input row and output row are contiguous across the lanes (every load / store instruction of the wavefront
//     touches one dense run of the row: no per-lane strides of a whole frame, nothing is re-read from the output arrays), and
//   * the chain is evaluated by pointer jumping, as in ik_kernel's FK: every item holds its pose relative to an ancestor and,
//     each round, composes it with that ancestor's own relative pose -- ceil(log2(depth + 1)) rounds instead of `depth` levels.
// Items of a wavefront: slot s = 64 k + lane, k < K.  Skeletons of up to 32 joints put 64 / Jp frames side by side in one
// wavefront (Jp = joints rounded up to a power of two), larger ones one frame with K = ceil(J / 64) joints per lane.
// The ancestor's pose travels through a small structure-of-arrays exchange buffer in LDS (one 8-byte word per component and
// slot: conflict-free ds_write_b64, at most two-way conflicts on the indexed ds_read_b64); a workgroup is one wavefront, so the
// write and the read of a round need no barrier, only their program order (wave_lds_sync).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ik_kernel.hip.h"  // lean float64 math: kc, fast_rsqrt, fast_sqrt, fast_rcp, sincos_small, qmul, qrot

namespace gmr {

constexpr int kChainMaxRounds = 8;  // pointer jumping doubles the folded distance per round: trees of depth < 256
constexpr unsigned kNoAnc = 0xffu;

struct ChainGeom {
  int jp;      // slots per frame in the wavefront's exchange buffer
  int groups;  // frames a wavefront processes side by side (64 K / jp)
};

__host__ __device__ inline ChainGeom chain_geom(int n_joints) {
  ChainGeom g;
  if (n_joints > 32) {
    g.jp = 64 * ((n_joints + 63) / 64);
    g.groups = 1;
  } else {
    int p = 1;
    while (p < n_joints) p <<= 1;
    g.jp = p;
    g.groups = 64 / p;
  }
  return g;
}

// Order this wavefront's own LDS traffic: the stores of a round before the indexed loads that follow them, and those loads before
// the next round's stores.  A workgroup here is ONE wavefront and the LDS unit executes a wavefront's DS instructions in order, so
// nothing has to be waited for -- a wavefront-scope fence pair and a scheduling barrier keep the compiler from moving LDS accesses
// across this point.  `__syncthreads()` would be correct too but drains vmcnt as well: the prefetched rows of the next frames and
// the previous frames' stores, i.e. one full HBM round trip per round (measured: 67 % of the wave cycles parked, 0.33 -> of the
// HBM roofline; profiles/r03_adapters_*).
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The pointer-jumping plan of this lane's K items, one byte per round (slot of the ancestor whose pose is folded in that round,
// kNoAnc = already relative to the world), built once per wavefront from the parent slots by the same doubling the poses go
// through.  `xi` is a 64 K-entry LDS scratch.  Returns the number of rounds (wave-uniform).
template <int K>
__device__ __forceinline__ int chain_plan(const int (&parent_slot)[K], int lane, int *xi, unsigned long long (&plan)[K]) {
  int cur[K];
#pragma unroll
  for (int k = 0; k < K; ++k) { cur[k] = parent_slot[k]; plan[k] = ~0ull; }
  int rounds = 0;
#pragma unroll
  for (int r = 0; r < kChainMaxRounds; ++r) {
    bool any = false;
#pragma unroll
    for (int k = 0; k < K; ++k) any |= cur[k] >= 0;
    if (__ballot(any) == 0) break;
    rounds = r + 1;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const unsigned long long b = cur[k] >= 0 ? (unsigned long long)cur[k] : (unsigned long long)kNoAnc;
      plan[k] = (plan[k] & ~(0xffull << (8 * r))) | (b << (8 * r));
      xi[64 * k + lane] = cur[k];
    }
    wave_lds_sync();
#pragma unroll
    for (int k = 0; k < K; ++k) cur[k] = cur[k] >= 0 ? xi[cur[k]] : -1;
    wave_lds_sync();
  }
  return rounds;
}

__device__ __forceinline__ unsigned chain_anc(unsigned long long plan, int r) { return (unsigned)(plan >> (8 * r)) & 0xffu; }

// sin and cos without range reduction for |x| <= 1.6 (the kernels of sincos_fk, ik_kernel.hip.h: argument halved, fdlibm
// polynomials on |x / 2| <= 0.8, one double-angle step): 23 operations instead of sincos_small's 45.
__device__ __forceinline__ void sincos_short(double x, double *sn, double *cs) {
  const double h = 0.5 * x, z = h * h;
  const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, kc(1.58969099521155010221e-10), kc(-2.50507602534068634195e-08)), kc(2.75573137070700676789e-06)),
                                       kc(-1.98412698298579493134e-04)), kc(8.33333333332248946124e-03)), kc(-1.66666666666666324348e-01));
  const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, kc(-1.13596475577881948265e-11), kc(2.08757232129817482790e-09)), kc(-2.75573143513906633035e-07)),
                                       kc(2.48015872894767294178e-05)), kc(-1.38888888888741095749e-03)), kc(4.16666666666666019037e-02));
  const double s1 = fma(h * z, ps, h);
  const double c1 = fma(z * z, pc, fma(-0.5, z, 1.0));
  *sn = 2.0 * s1 * c1;
  *cs = fma(-2.0 * s1, s1, 1.0);
}
// N half-angles at once: the short kernels for every argument within 1.6 (angles within +-183 degrees -- what mocap channels and
// rotation vectors hold), the range-reducing ones for the others.  Which kernel an element gets depends on ITS value only: one ballot
// skips the slow path when no lane needs it, but a lane's result never depends on what its neighbours hold -- a column selection or another
// packing of frames into wavefronts must not change a single bit of a joint's output.
template <int N>
__device__ __forceinline__ void sincos_n(const double (&x)[N], double (&sn)[N], double (&cs)[N]) {
  bool big = false;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    big |= !(fabs(x[i]) <= 1.6);
    sincos_short(x[i], &sn[i], &cs[i]);
  }
  if (__ballot(big) != 0) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      double s, c;
      sincos_small(x[i], &s, &c);
      if (!(fabs(x[i]) <= 1.6)) { sn[i] = s; cs[i] = c; }
    }
  }
}

// exp of a rotation vector as scipy's Rotation.from_rotvec builds it (xyzw): k = sin(a/2)/a with the series below 1e-3.
// `sn`, `cs`: sine and cosine of a / 2.
__device__ __forceinline__ void rotvec_to_quat_xyzw(const double rv[3], double a2, double a, double sn, double cs, double q[4]) {
  const double k = a <= 1e-3 ? 0.5 - a2 * (1.0 / 48.0) + a2 * a2 * (1.0 / 3840.0) : sn * fast_rcp(a);
  q[0] = k * rv[0]; q[1] = k * rv[1]; q[2] = k * rv[2]; q[3] = cs;
}

}  // namespace gmr
