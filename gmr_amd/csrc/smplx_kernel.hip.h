// smplx_kernel.hip.h -- SMPL-X key-point adapter: frame-rate alignment + global joint orientations (float64).
//
// Replaces the numeric part of get_smplx_data_offline_fast (reference general_motion_retargeting/utils/smpl.py:109-198)
// downstream of the licensed SMPL-X body model (which stays external): per output frame t = linspace(0, T-1, T')[k],
// every joint's axis-angle rotation is slerped between frames floor(t) and floor(t)+1 exactly as `slerp` does
// (:75-107: shorter arc, linear blend above dot 0.9995), joint positions are interpolated linearly (:162-168), and
// orientations are chained down `parents` (:179-196): R_0 = global_orient, R_i = R_parent(i) * exp(pose_i).
// One output frame per lane; a joint's parent orientation is read back from the output array.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gmr {

constexpr int kSmplMaxJoints = 64;

struct SmplSkeleton {
  int n_joints, joints_stride, resample, pad;  // joints_stride: joints per frame in the position array (>= n_joints)
  short parent[kSmplMaxJoints];
};

__device__ __forceinline__ void rotvec_to_quat_xyzw(const double *rv, double q[4]) {
  const double a2 = rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2];
  const double a = sqrt(a2);
  double k;  // sin(a/2)/a with scipy's small-angle series
  if (a <= 1e-3) k = 0.5 - a2 / 48.0 + a2 * a2 / 3840.0;
  else k = sin(0.5 * a) / a;
  q[0] = k * rv[0]; q[1] = k * rv[1]; q[2] = k * rv[2]; q[3] = cos(0.5 * a);
}
__device__ __forceinline__ void quat_mul_xyzw(const double a[4], const double b[4], double o[4]) {
  o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  o[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
  o[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
}
__device__ __forceinline__ void slerp_xyzw(const double *rv1, const double *rv2, double t, double q[4]) {
  double q1[4], q2[4];
  rotvec_to_quat_xyzw(rv1, q1);
  rotvec_to_quat_xyzw(rv2, q2);
  double dot = q1[0] * q2[0] + q1[1] * q2[1] + q1[2] * q2[2] + q1[3] * q2[3];
  if (dot < 0.0) { dot = -dot; for (int i = 0; i < 4; i++) q2[i] = -q2[i]; }
  double s0, s1;
  if (dot > 0.9995) { s0 = 1.0 - t; s1 = t; }
  else {
    const double th0 = acos(dot), th = th0 * t, st = sin(th), st0 = sin(th0);
    s0 = cos(th) - dot * st / st0;
    s1 = st / st0;
  }
  double n = 0.0;
  for (int i = 0; i < 4; i++) { q[i] = s0 * q1[i] + s1 * q2[i]; n += q[i] * q[i]; }
  n = 1.0 / sqrt(n);  // Rotation.from_quat normalises
  for (int i = 0; i < 4; i++) q[i] *= n;
}

// global_orient [T][3], full_pose [T][J][3] (axis-angle), joints [T][joints_stride][3]  ->
// pos_out [T_out][J][3], quat_out [T_out][J][4] wxyz.  resample = 0 copies frames 1:1 (T_out == T).
__global__ void __launch_bounds__(128) smplx_keypoints_kernel(SmplSkeleton sk, const double *__restrict__ global_orient,
                                                             const double *__restrict__ full_pose, const double *__restrict__ joints,
                                                             int64_t T, int64_t T_out, double *__restrict__ pos_out,
                                                             double *__restrict__ quat_out) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= T_out) return;
  const int J = sk.n_joints;
  int64_t i1 = k, i2 = k;
  double alpha = 0.0;
  if (sk.resample) {  // np.linspace(0, T-1, T_out)[k]
    const double t = T_out > 1 ? (double)k * ((double)(T - 1) / (double)(T_out - 1)) : 0.0;
    i1 = (int64_t)floor(t);
    if (i1 > T - 1) i1 = T - 1;
    i2 = i1 + 1 < T ? i1 + 1 : T - 1;
    alpha = t - (double)i1;
  }
  for (int j = 0; j < J; ++j) {
    const double *r1 = j == 0 ? global_orient + i1 * 3 : full_pose + (i1 * J + j) * 3;
    const double *r2 = j == 0 ? global_orient + i2 * 3 : full_pose + (i2 * J + j) * 3;
    double lq[4], gq[4];
    slerp_xyzw(r1, r2, alpha, lq);
    if (j == 0) { gq[0] = lq[0]; gq[1] = lq[1]; gq[2] = lq[2]; gq[3] = lq[3]; }
    else {
      const double *pq = quat_out + (k * J + sk.parent[j]) * 4;  // stored wxyz
      const double p[4] = {pq[1], pq[2], pq[3], pq[0]};
      quat_mul_xyzw(p, lq, gq);
    }
    double *oq = quat_out + (k * J + j) * 4;
    oq[0] = gq[3]; oq[1] = gq[0]; oq[2] = gq[1]; oq[3] = gq[2];
    const double *p1 = joints + (i1 * sk.joints_stride + j) * 3, *p2 = joints + (i2 * sk.joints_stride + j) * 3;
    double *op = pos_out + (k * J + j) * 3;
    for (int c = 0; c < 3; c++) op[c] = p1[c] + alpha * (p2[c] - p1[c]);
  }
}

}  // namespace gmr
