// bvh_kernel.hip.h -- BVH skeleton FK for the LAFAN1-style input adapter (float64).
//
// Replaces the numeric part of load_lafan1_file (reference general_motion_retargeting/utils/lafan1.py:8-40):
// utils.euler_to_quat (lafan_vendor/utils.py:56-75), utils.quat_fk (:88-103), the Y-up -> Z-up rotation
// (lafan1.py:20-21,31-32), centimetres -> metres (:32) and the synthesised LeftFootMod / RightFootMod entries
// (foot position + toe orientation, :36-39).  One frame per lane; global poses are written to the output arrays and a
// joint's parent pose is read back from there (L1/L2 resident: joints are in hierarchy order).
// remove_quat_discontinuities (extract.py:164) only flips quaternion signs along time and is not applied: every
// consumer of the orientations (scipy Rotation in update_targets, the SE3 log) is sign-insensitive.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gmr {

constexpr int kBvhMaxJoints = 160;
constexpr int kBvhMaxExtra = 8;

struct BvhSkeleton {
  int n_joints, n_extra, order[3], pad;
  short parent[kBvhMaxJoints];
  short extra_pos_src[kBvhMaxExtra], extra_rot_src[kBvhMaxExtra];
};

__device__ __forceinline__ void bvh_qmul(const double a[4], const double b[4], double o[4]) {
  o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  o[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  o[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}
__device__ __forceinline__ void bvh_axis_quat(double ang, int axis, double q[4]) {
  double s, c;
  sincos(0.5 * ang, &s, &c);
  q[0] = c; q[1] = axis == 0 ? s : 0.0; q[2] = axis == 1 ? s : 0.0; q[3] = axis == 2 ? s : 0.0;
}

// pos_out [T][J+E][3] metres, Z-up; quat_out [T][J+E][4] wxyz.  rot = [[1,0,0],[0,0,-1],[0,1,0]] (a +90 deg turn about x).
__global__ void __launch_bounds__(128) bvh_fk_kernel(BvhSkeleton sk, const double *__restrict__ local_pos,
                                                     const double *__restrict__ euler_rad, int64_t T, double scale,
                                                     double *__restrict__ pos_out, double *__restrict__ quat_out) {
  const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= T) return;
  const int J = sk.n_joints, B = J + sk.n_extra;
  // work in the file's own frame first (stored in the output arrays), convert in a second sweep
  for (int j = 0; j < J; ++j) {
    const double *e = euler_rad + (f * J + j) * 3, *lp = local_pos + (f * J + j) * 3;
    double q0[4], q1[4], q2[4], t[4], lq[4];
    bvh_axis_quat(e[0], sk.order[0], q0);
    bvh_axis_quat(e[1], sk.order[1], q1);
    bvh_axis_quat(e[2], sk.order[2], q2);
    bvh_qmul(q1, q2, t);
    bvh_qmul(q0, t, lq);
    double *gq = quat_out + (f * B + j) * 4, *gp = pos_out + (f * B + j) * 3;
    if (j == 0) {
      gq[0] = lq[0]; gq[1] = lq[1]; gq[2] = lq[2]; gq[3] = lq[3];
      gp[0] = lp[0]; gp[1] = lp[1]; gp[2] = lp[2];
    } else {
      const int p = sk.parent[j];
      const double *pq = quat_out + (f * B + p) * 4, *pp = pos_out + (f * B + p) * 3;
      const double a[4] = {pq[0], pq[1], pq[2], pq[3]};
      double o[4];
      bvh_qmul(a, lq, o);
      // quat_mul_vec: v + 2 w (u x v) + 2 u x (u x v)
      const double tx = 2.0 * (a[2] * lp[2] - a[3] * lp[1]), ty = 2.0 * (a[3] * lp[0] - a[1] * lp[2]), tz = 2.0 * (a[1] * lp[1] - a[2] * lp[0]);
      gp[0] = pp[0] + lp[0] + a[0] * tx + (a[2] * tz - a[3] * ty);
      gp[1] = pp[1] + lp[1] + a[0] * ty + (a[3] * tx - a[1] * tz);
      gp[2] = pp[2] + lp[2] + a[0] * tz + (a[1] * ty - a[2] * tx);
      gq[0] = o[0]; gq[1] = o[1]; gq[2] = o[2]; gq[3] = o[3];
    }
  }
  const double rq[4] = {0.70710678118654757, 0.70710678118654757, 0.0, 0.0};
  for (int j = 0; j < J; ++j) {
    double *gq = quat_out + (f * B + j) * 4, *gp = pos_out + (f * B + j) * 3;
    const double a[4] = {gq[0], gq[1], gq[2], gq[3]};
    double o[4];
    bvh_qmul(rq, a, o);
    gq[0] = o[0]; gq[1] = o[1]; gq[2] = o[2]; gq[3] = o[3];
    const double x = gp[0], y = gp[1], z = gp[2];
    gp[0] = x * scale; gp[1] = -z * scale; gp[2] = y * scale;  // p @ rot.T / 100
  }
  for (int k = 0; k < sk.n_extra; ++k) {
    const double *sp = pos_out + (f * B + sk.extra_pos_src[k]) * 3, *sq = quat_out + (f * B + sk.extra_rot_src[k]) * 4;
    double *gp = pos_out + (f * B + J + k) * 3, *gq = quat_out + (f * B + J + k) * 4;
    gp[0] = sp[0]; gp[1] = sp[1]; gp[2] = sp[2];
    gq[0] = sq[0]; gq[1] = sq[1]; gq[2] = sq[2]; gq[3] = sq[3];
  }
}

}  // namespace gmr
