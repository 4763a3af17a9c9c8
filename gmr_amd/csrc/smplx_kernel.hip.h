// smplx_kernel.hip.h -- SMPL-X key-point adapter: frame-rate alignment + global joint orientations (float64).
//
// Replaces the numeric part of get_smplx_data_offline_fast (reference general_motion_retargeting/utils/smpl.py:109-198)
// downstream of the licensed SMPL-X body model (which stays external): per output frame t = linspace(0, T-1, T')[k],
// every joint's axis-angle rotation is slerped between frames floor(t) and floor(t)+1 exactly as `slerp` does
// (:75-107: shorter arc, linear blend above dot 0.9995), joint positions are interpolated linearly (:162-168), and
// orientations are chained down `parents` (:179-196): R_0 = global_orient, R_i = R_parent(i) * exp(pose_i).
//
// One wavefront per run of output frames, lane = joint (tree_chain.hip.h): the two source rows of a frame are read densely by
// the joints' lanes, the slerp of all joints runs side by side, the orientations are chained by pointer jumping through the LDS
// exchange buffer, and each output row is written once.  `out_col` emits only the columns a consumer names (the 14 joints an IK
// config reads); joints that are neither emitted nor an ancestor of an emitted one take no lane at all (SmplSkeleton).
#pragma once

#include "tree_chain.hip.h"

namespace gmr {

constexpr int kSmplMaxJoints = 64;

// The skeleton the kernel works on holds only the LIVE joints -- those emitted or an ancestor of an emitted one --, renumbered densely
// (parents before children, the root first): with the 14 columns an smplx_to_*.json config reads that is ~22 of SMPL-X's 55 joints, so
// two frames share a wavefront (chain_geom) where the full tree takes one.  `src` leads back to the joint's place in the input rows.
struct SmplSkeleton {
  int n_joints, joints_stride, resample, n_out;  // n_joints: live joints; joints_stride: joints per frame in the position array
  int pose_stride;                               // joints per frame in full_pose (the model's joint count)
  short parent[kSmplMaxJoints];    // parent of live joint c (an index into the live joints), -1 for the root
  short out_col[kSmplMaxJoints];   // output column of live joint c, -1 = not emitted (a mere ancestor)
  short src[kSmplMaxJoints];       // the model's index of live joint c
};

__device__ __forceinline__ void quat_mul_xyzw(const double a[4], const double b[4], double o[4]) {
  o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  o[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
  o[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
}

// acos on [0, 1): 4 atan(u) with u = t / (1 + sqrt(1 + t^2)), t = sqrt((1 - x) / (1 + x)) -- two half-angle steps bring the argument
// under tan(pi/8), where fdlibm's atan kernel needs no reduction (the polynomial of so3_log_factor, ik_kernel.hip.h)
__device__ __forceinline__ double acos_unit(double x) {
  const double t = fast_sqrt((1.0 - x) * fast_rcp(1.0 + x));
  const double u = t * fast_rcp(1.0 + fast_sqrt(fma(t, t, 1.0)));
  const double z = u * u, w = z * z;
  const double s1 = z * fma(w, fma(w, fma(w, fma(w, fma(w, kc(1.62858201153657823623e-02), kc(4.97687799461593236017e-02)), kc(6.66107313738753120669e-02)),
                                          kc(9.09088713343650656196e-02)), kc(1.42857142725034663711e-01)), kc(3.33333333333329318027e-01));
  const double s2 = w * fma(w, fma(w, fma(w, fma(w, kc(-3.65315727442169155270e-02), kc(-5.83357013379057348645e-02)), kc(-7.69187620504482999495e-02)),
                                   kc(-1.11111104054623557880e-01)), kc(-1.99999999998764832476e-01));
  return 4.0 * (u - u * (s1 + s2));
}

// smpl.py:75-107 on two unit quaternions with dot >= 0; the caller has decided (wave-uniformly) whether any lane needs the
// trigonometric arm.  sin(theta_0) = sqrt(1 - dot^2) on theta_0 = acos(dot) in (0, pi/2].
__device__ __forceinline__ void slerp_xyzw(const double q1[4], const double q2[4], double dot, double t, bool trig, double q[4]) {
  double s0 = 1.0 - t, s1 = t;
  if (trig) {
    const double d = dot <= 0.9995 ? dot : 0.5;  // (lanes on the linear arm run the arithmetic on a harmless value)
    const double th = acos_unit(d) * t;
    double st, ct;
    sincos_small(th, &st, &ct);
    const double r = st * fast_rsqrt(fma(-d, d, 1.0));  // sin(theta) / sin(theta_0)
    if (dot <= 0.9995) { s0 = ct - d * r; s1 = r; }
  }
  double n = 0.0;
#pragma unroll
  for (int i = 0; i < 4; i++) { q[i] = s0 * q1[i] + s1 * q2[i]; n += q[i] * q[i]; }
  n = fast_rsqrt(n);  // Rotation.from_quat normalises
#pragma unroll
  for (int i = 0; i < 4; i++) q[i] *= n;
}

// global_orient [T][3], full_pose [T][J][3] (axis-angle), joints [T][joints_stride][3]  ->
// pos_out [T_out][n_out][3], quat_out [T_out][n_out][4] wxyz.  resample = 0 copies frames 1:1 (T_out == T).
// A wavefront handles output frames [blockIdx.x * chunk, ... + chunk), 64 / jp of them per iteration.
// TIn: the input arrays' element type -- double, or float as a body model emits it (promoted to double on load, which is what the
// reference's scipy / numpy calls do with float32 input: exact, and half the bytes).
template <typename TIn>
__global__ void __launch_bounds__(64) smplx_keypoints_kernel(SmplSkeleton sk, const TIn *__restrict__ global_orient,
                                                            const TIn *__restrict__ full_pose, const TIn *__restrict__ joints,
                                                            int64_t T, int64_t T_out, int chunk, double *__restrict__ pos_out,
                                                            double *__restrict__ quat_out) {
  __shared__ double xb[4][64];
  __shared__ int xi[64];
  const int lane = threadIdx.x;
  const int J = sk.n_joints, NO = sk.n_out;
  const ChainGeom geo = chain_geom(J);
  const int jp = geo.jp, G = geo.groups;
  const int j = G > 1 ? (lane & (jp - 1)) : lane;
  const int grp = G > 1 ? lane / jp : 0;
  const bool has = j < J;
  const int par = has ? (int)sk.parent[j] : -1;
  const int ocol = has ? (int)sk.out_col[j] : -1;
  const int jo = has ? (int)sk.src[j] : 0;  // where this joint's numbers are in the input rows
  int pslot[1] = {par >= 0 ? lane - j + par : -1};
  unsigned long long plan[1];
  const int rounds = chain_plan<1>(pslot, lane, xi, plan);
  const bool resample = sk.resample != 0;
  const double step = T_out > 1 ? (double)(T - 1) / (double)(T_out - 1) : 0.0;  // np.linspace(0, T-1, T_out)

  const int64_t f_begin = (int64_t)blockIdx.x * chunk;
  const int64_t f_end = f_begin + chunk < T_out ? f_begin + chunk : T_out;

  struct Row { double r1[3], r2[3], p1[3], p2[3], alpha; };
  auto load = [&](int64_t fb, Row &w) {
    const int64_t k = fb + grp;
    const bool ok = has && k < f_end;
    int64_t i1 = k, i2 = k;
    w.alpha = 0.0;
    if (resample) {
      const double t = k == T_out - 1 && T_out > 1 ? (double)(T - 1) : (double)k * step;  // (np.linspace ends on `stop` exactly)
      i1 = (int64_t)floor(t);
      if (i1 > T - 1) i1 = T - 1;
      i2 = i1 + 1 < T ? i1 + 1 : T - 1;
      w.alpha = t - (double)i1;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) { w.r1[c] = 0.0; w.r2[c] = 0.0; w.p1[c] = 0.0; w.p2[c] = 0.0; }
    if (ok) {
      const TIn *a = jo == 0 ? global_orient + i1 * 3 : full_pose + (i1 * sk.pose_stride + jo) * 3;
      w.r1[0] = a[0]; w.r1[1] = a[1]; w.r1[2] = a[2];
      if (resample) {
        const TIn *b = jo == 0 ? global_orient + i2 * 3 : full_pose + (i2 * sk.pose_stride + jo) * 3;
        w.r2[0] = b[0]; w.r2[1] = b[1]; w.r2[2] = b[2];
      }
      if (ocol >= 0) {
        const TIn *pa = joints + (i1 * sk.joints_stride + jo) * 3;
        w.p1[0] = pa[0]; w.p1[1] = pa[1]; w.p1[2] = pa[2];
        if (resample) {
          const TIn *pb = joints + (i2 * sk.joints_stride + jo) * 3;
          w.p2[0] = pb[0]; w.p2[1] = pb[1]; w.p2[2] = pb[2];
        }
      }
    }
  };
  Row cur;
  load(f_begin, cur);
  for (int64_t fb = f_begin; fb < f_end; fb += G) {
    Row nxt;  // the next iteration's rows, requested before this one's arithmetic (defaults past the end)
    load(fb + G, nxt);

    double q[4];  // xyzw, as scipy holds it
    double q1[4];
    const double a2_1 = cur.r1[0] * cur.r1[0] + cur.r1[1] * cur.r1[1] + cur.r1[2] * cur.r1[2], a_1 = fast_sqrt(a2_1);
    if (resample) {
      double q2[4];
      const double a2_2 = cur.r2[0] * cur.r2[0] + cur.r2[1] * cur.r2[1] + cur.r2[2] * cur.r2[2], a_2 = fast_sqrt(a2_2);
      const double h[2] = {0.5 * a_1, 0.5 * a_2};
      double sn[2], cs[2];
      sincos_n<2>(h, sn, cs);
      rotvec_to_quat_xyzw(cur.r1, a2_1, a_1, sn[0], cs[0], q1);
      rotvec_to_quat_xyzw(cur.r2, a2_2, a_2, sn[1], cs[1], q2);
      double dot = q1[0] * q2[0] + q1[1] * q2[1] + q1[2] * q2[2] + q1[3] * q2[3];
      if (dot < 0.0) { dot = -dot; q2[0] = -q2[0]; q2[1] = -q2[1]; q2[2] = -q2[2]; q2[3] = -q2[3]; }
      const bool trig = __ballot(dot <= 0.9995) != 0;  // rare between neighbouring mocap frames: decided per wavefront
      slerp_xyzw(q1, q2, dot, cur.alpha, trig, q);
      // the reference stores the interpolated rotation as a rotation vector and re-reads it (as_rotvec / from_rotvec): w >= 0
      if (q[3] < 0.0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    } else {
      const double h[1] = {0.5 * a_1};
      double sn[1], cs[1];
      sincos_n<1>(h, sn, cs);
      rotvec_to_quat_xyzw(cur.r1, a2_1, a_1, sn[0], cs[0], q);
    }
    // orientation chaining by pointer jumping
    bool dirty = true;
#pragma unroll
    for (int r = 0; r < kChainMaxRounds; ++r) {
      if (r >= rounds) break;
      if (dirty) { xb[0][lane] = q[0]; xb[1][lane] = q[1]; xb[2][lane] = q[2]; xb[3][lane] = q[3]; }
      wave_lds_sync();
      const unsigned a = chain_anc(plan[0], r);
      dirty = a != kNoAnc;
      if (dirty) {
        const double aq[4] = {xb[0][a], xb[1][a], xb[2][a], xb[3][a]};
        double o[4];
        quat_mul_xyzw(aq, q, o);
        q[0] = o[0]; q[1] = o[1]; q[2] = o[2]; q[3] = o[3];
      }
      wave_lds_sync();
    }
    const int64_t k = fb + grp;
    if (ocol >= 0 && k < f_end) {
      double *oq = quat_out + (k * NO + ocol) * 4, *op = pos_out + (k * NO + ocol) * 3;
      oq[0] = q[3]; oq[1] = q[0]; oq[2] = q[1]; oq[3] = q[2];
#pragma unroll
      for (int c = 0; c < 3; c++) op[c] = cur.p1[c] + cur.alpha * (cur.p2[c] - cur.p1[c]);
    }
    cur = nxt;
  }
}

}  // namespace gmr
