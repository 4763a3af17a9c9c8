// ik_kernel.hip.h -- the frames-batched two-stage IK solver for gfx950 (CDNA4).
//
// One wavefront (64 lanes) owns one work item: a run of consecutive frames of one clip,
// solved in time order with warm start exactly as the reference's caller loop does
// (`for frame in frames: retargeter.retarget(frame)`, scripts/smplx_to_robot_dataset.py:84-89;
// GeneralMotionRetargeting.retarget, motion_retarget.py:139-185).  Thousands of items run
// concurrently, one per wavefront; nothing is exchanged between them.
//
// Lane roles change per phase (all state that crosses phases lives in LDS):
//   FK            lane = body        level-synchronous quaternion chain down the joint tree
//   residual      lane = task        SE3 log of T_body^-1 T_target, |e| by wave reduction
//   task blocks   lane = task        6x6 "task inertia" B_t = A_t' W^2 A_t and g_t = A_t' W^2 e_t
//   composites    lane = (node,elt)  B^c = sum of B_t below a joint (composite-rigid-body style)
//   H, c          lane = dof (row)   H[i][j] = S_j . (B^c_i S_i) for j an ancestor of i; row i in VGPRs
//   box QP        lane = dof (row)   in-register Cholesky, broadcast-style forward/backward solves,
//                                    primal active set with warm-started working set
//   integrate     lane = dof
// All arithmetic is float64: gfx950 issues v_fma_f64 at the same rate as unpacked v_fma_f32, and
// the reference's `curr_error - next_error > 1e-3` loop test (motion_retarget.py:153,172) is
// decided in float64 on the CPU.
//
// The math (mink.FrameTask / ConfigurationLimit / solve_ik, MuJoCo mj_kinematics / mj_jacBody /
// mj_integratePos; none of which is vendored in the reference) is stated in SURVEY.md Appendix A
// and restated independently, in dense textbook form, by oracle/gmr_oracle.c.  Here the same
// H = damping I + sum_t [(W J_t)'(W J_t) + lm |W e_t|^2 I], c = sum_t (W J_t)'(W e_t) is assembled
// without ever forming a Jacobian: with S_k = [x_k x a_k ; a_k] the world-frame screw of dof k,
// J_t[:,k] = A_t S_k for every dof k above task body t, A_t = -Jl^-1(e_t) [R_t' , -R_t'[x_t]x ; 0 , R_t'],
// hence H[k][l] = S_k' (sum_{t below l} A_t' W^2 A_t) S_l  for k above-or-equal l.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gmr_amd.h"

namespace gmr {

typedef unsigned long long u64;

constexpr int kBT = 27;  // doubles per task block: LL(6) LA(9) AA(6) g(6)
constexpr double kLieEps = 1e-10;  // mink.lie.utils.get_epsilon(float64)

// Device view of a compiled model: pointers into one device allocation (see api.hip).
struct DevModel {
  int nbody, nq, nv, nslot, root_slot, maxdepth, n_act, pad0;
  int ntask[2], use_table[2], ncomp[2];
  // per body [nbody]
  const int *parent, *jtype, *qadr, *depth;
  const double *bpos, *bquat, *axis;  // [nb][3], [nb][4] unit wxyz, [nb][3]
  const double *qpos0;                // [nq]
  // per slot [nslot]
  const double *sscale, *spoff, *sroff;
  const int *sfoot;
  // per task, table-major [2][GMR_MAX_TASKS]
  const int *tbody, *tslot;
  const double *twp, *twr;
  // per active dof [64]
  const int *abody, *akind, *aqadr, *alimited;  // akind: 0..2 root translation, 3..5 root rotation, 6 hinge
  const u64 *aanc;                              // active dofs strictly above (lower index)
  const double *arange;                         // [64][2]
  const int *acomp;                             // [2][64] composite node of the dof per table
  const unsigned *compmask;                     // [2][GMR_MAX_TASKS*2] tasks summed into each composite
};

struct LdsLayout {
  int q, tp, tq, S, xpos, xquat, B, Bc, H, total_doubles;  // H aliases [xpos, xquat, B, Bc] (dead during the QP)
};

struct IkLaunch {
  const void *hpos, *hquat;
  const int *slot_col;
  const gmr_work_item *items;
  const double *qinit;
  double *qfinal, *qout;
  int *iters;
  int in_f64, n_cols, n_items, pad;
  gmr_ik_params prm;
};

// ------------------------------------------------------------------ wave helpers (wave = 64)
__device__ __forceinline__ double rdlane(double v, int lane) {  // lane: wave-uniform
  int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ u64 rdlane_u64(u64 v, int lane) {
  unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)v, lane);
  unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(v >> 32), lane);
  return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double fast_rsqrt(double x) {  // ~1 ulp after two Newton steps
  double r = __builtin_amdgcn_rsq(x);
  r = r * (1.5 - 0.5 * x * r * r);
  r = r * (1.5 - 0.5 * x * r * r);
  return r;
}

// ------------------------------------------------------------------ quaternion / matrix helpers (wxyz)
__device__ __forceinline__ void qmul(const double a[4], const double b[4], double o[4]) {
  double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  double z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  o[0] = w; o[1] = x; o[2] = y; o[3] = z;
}
__device__ __forceinline__ void qnormalize(double q[4]) {
  double r = fast_rsqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  q[0] *= r; q[1] *= r; q[2] *= r; q[3] *= r;
}
__device__ __forceinline__ void q2mat(const double q[4], double R[9]) {  // mju_quat2Mat
  double q00 = q[0] * q[0], q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3];
  double q11 = q[1] * q[1], q12 = q[1] * q[2], q13 = q[1] * q[3];
  double q22 = q[2] * q[2], q23 = q[2] * q[3], q33 = q[3] * q[3];
  R[0] = q00 + q11 - q22 - q33; R[4] = q00 - q11 + q22 - q33; R[8] = q00 - q11 - q22 + q33;
  R[1] = 2 * (q12 - q03); R[2] = 2 * (q13 + q02);
  R[3] = 2 * (q12 + q03); R[5] = 2 * (q23 - q01);
  R[6] = 2 * (q13 - q02); R[7] = 2 * (q23 + q01);
}
__device__ __forceinline__ void mv(const double R[9], const double v[3], double o[3]) {
  o[0] = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
  o[1] = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
  o[2] = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
}
__device__ __forceinline__ void mtv(const double R[9], const double v[3], double o[3]) {
  o[0] = R[0] * v[0] + R[3] * v[1] + R[6] * v[2];
  o[1] = R[1] * v[0] + R[4] * v[1] + R[7] * v[2];
  o[2] = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
}
__device__ __forceinline__ void cross(const double a[3], const double b[3], double o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

// per-lane constants of the body this lane owns in the FK phase
struct BodyConst {
  int parent, depth, jtype, qadr;
  double pos[3], quat[4], axis[3];
};

// ------------------------------------------------------------------ FK (mj_kinematics), lane = body
// xpos_j = xpos_p + R(xquat_p) pos_j ;  xquat_j = normalize(xquat_p (x) quat_j (x) [cos t/2, sin t/2 axis_j])
__device__ __forceinline__ void fk_phase(const DevModel &m, const BodyConst &bc, int lane, const double *q, double *xpos,
                                         double *xquat) {
  double ql[4] = {1, 0, 0, 0};
  const bool has = lane < m.nbody;
  if (has) {
    if (bc.jtype == GMR_JNT_FREE) {
      double r[4] = {q[3], q[4], q[5], q[6]};
      qnormalize(r);
      xpos[0] = q[0]; xpos[1] = q[1]; xpos[2] = q[2];
      xquat[0] = r[0]; xquat[1] = r[1]; xquat[2] = r[2]; xquat[3] = r[3];
    } else if (bc.jtype == GMR_JNT_HINGE) {
      double s, c;
      sincos(0.5 * q[bc.qadr], &s, &c);
      double jq[4] = {c, s * bc.axis[0], s * bc.axis[1], s * bc.axis[2]};
      qmul(bc.quat, jq, ql);
    } else {
      ql[0] = bc.quat[0]; ql[1] = bc.quat[1]; ql[2] = bc.quat[2]; ql[3] = bc.quat[3];
    }
  }
  __syncthreads();
  for (int d = 1; d <= m.maxdepth; ++d) {
    if (has && bc.depth == d) {
      const int p = bc.parent;
      double qp[4] = {xquat[4 * p], xquat[4 * p + 1], xquat[4 * p + 2], xquat[4 * p + 3]};
      double R[9], t[3], qo[4];
      q2mat(qp, R);
      mv(R, bc.pos, t);
      qmul(qp, ql, qo);
      qnormalize(qo);
      xpos[3 * lane] = xpos[3 * p] + t[0];
      xpos[3 * lane + 1] = xpos[3 * p + 1] + t[1];
      xpos[3 * lane + 2] = xpos[3 * p + 2] + t[2];
      xquat[4 * lane] = qo[0]; xquat[4 * lane + 1] = qo[1]; xquat[4 * lane + 2] = qo[2]; xquat[4 * lane + 3] = qo[3];
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ residual, lane = task
// e = Log(T_wb^-1 T_wt) in [v; w] order (mink FrameTask.compute_error, via motion_retarget.py:188-200).
// Returns |e|^2 contribution of this lane; e is kept in registers for the assembly.
__device__ __forceinline__ double task_residual(int body, int slot, const double *xpos, const double *xquat, const double *tp,
                                                const double *tq, double e[6]) {
  const double qb[4] = {xquat[4 * body], xquat[4 * body + 1], xquat[4 * body + 2], xquat[4 * body + 3]};
  const double qt[4] = {tq[4 * slot], tq[4 * slot + 1], tq[4 * slot + 2], tq[4 * slot + 3]};
  const double qc[4] = {qb[0], -qb[1], -qb[2], -qb[3]};
  double qr[4], R[9], d[3], t[3];
  qmul(qc, qt, qr);
  q2mat(qb, R);
  d[0] = tp[3 * slot] - xpos[3 * body];
  d[1] = tp[3 * slot + 1] - xpos[3 * body + 1];
  d[2] = tp[3 * slot + 2] - xpos[3 * body + 2];
  mtv(R, d, t);
  // SO3 log, short side (mink.lie.so3.SO3.log)
  const double w = qr[0], n2 = qr[1] * qr[1] + qr[2] * qr[2] + qr[3] * qr[3];
  double f;
  if (n2 < kLieEps) {
    f = 2.0 / w - 2.0 / 3.0 * n2 / (w * w * w);
  } else {
    const double n = sqrt(n2);
    if (fabs(w) < kLieEps) f = (w > 0 ? 1.0 : -1.0) * M_PI / n;
    else f = 2.0 * atan2(w < 0 ? -n : n, fabs(w)) / n;
  }
  const double om[3] = {f * qr[1], f * qr[2], f * qr[3]};
  const double th2 = om[0] * om[0] + om[1] * om[1] + om[2] * om[2];
  double c2;
  if (th2 < kLieEps) c2 = 1.0 / 12.0;
  else {
    const double th = sqrt(th2);
    double s, c;
    sincos(0.5 * th, &s, &c);
    c2 = (1.0 - 0.5 * th * c / s) / th2;
  }
  // V^-1 t = t - 1/2 om x t + c2 om x (om x t)
  double a[3], b[3];
  cross(om, t, a);
  cross(om, a, b);
  e[0] = t[0] - 0.5 * a[0] + c2 * b[0];
  e[1] = t[1] - 0.5 * a[1] + c2 * b[1];
  e[2] = t[2] - 0.5 * a[2] + c2 * b[2];
  e[3] = om[0]; e[4] = om[1]; e[5] = om[2];
  return e[0] * e[0] + e[1] * e[1] + e[2] * e[2] + th2;
}

// ------------------------------------------------------------------ task block, lane = task
// A_t = -[[U, V],[0, U]] with U = Jso3^-1 R', V = Bo R' - U [x_b]x (see header); writes
// LL = wp^2 U'U, LA = wp^2 U'V, AA = wp^2 V'V + wr^2 U'U, g = A_t' W^2 e  -> out[27]; returns |W e|^2.
__device__ __forceinline__ double task_block(int body, const double *xpos, const double *xquat, const double e[6], double wp,
                                             double wr, double *out) {
  const double *u = e, *ph = e + 3;
  const double th2 = ph[0] * ph[0] + ph[1] * ph[1] + ph[2] * ph[2];
  const double pu = ph[0] * u[0] + ph[1] * u[1] + ph[2] * u[2];
  double kap = 0, bet = 0;
  const bool small = th2 < kLieEps;  // mink SE3.ljacinv returns the identity below this threshold
  if (!small) {
    const double th = sqrt(th2);
    double s, c;
    sincos(0.5 * th, &s, &c);
    const double cot = c / s;
    kap = (1.0 - 0.5 * th * cot) / th2;
    const double delta = th / (4.0 * s * s) - 0.5 * cot;
    bet = (kap - delta / (2.0 * th)) / th2;
  }
  // A = I - 1/2 [ph]x + kap (ph ph' - th2 I);  Bo = -1/2 [u]x + kap (ph u' + u ph' - 2 pu I) - 2 bet pu (ph ph' - th2 I)
  double A[9], Bo[9];
  const double k2 = -2.0 * bet * pu;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const double pp = ph[i] * ph[j] - (i == j ? th2 : 0.0);
      A[3 * i + j] = (i == j ? 1.0 : 0.0) + kap * pp;
      Bo[3 * i + j] = small ? 0.0 : kap * (ph[i] * u[j] + u[i] * ph[j] - (i == j ? 2.0 * pu : 0.0)) + k2 * pp;
    }
  if (!small) {
    A[1] += 0.5 * ph[2]; A[2] -= 0.5 * ph[1]; A[3] -= 0.5 * ph[2]; A[5] += 0.5 * ph[0]; A[6] += 0.5 * ph[1]; A[7] -= 0.5 * ph[0];
    Bo[1] += 0.5 * u[2]; Bo[2] -= 0.5 * u[1]; Bo[3] -= 0.5 * u[2]; Bo[5] += 0.5 * u[0]; Bo[6] += 0.5 * u[1]; Bo[7] -= 0.5 * u[0];
  }
  const double qb[4] = {xquat[4 * body], xquat[4 * body + 1], xquat[4 * body + 2], xquat[4 * body + 3]};
  const double xb[3] = {xpos[3 * body], xpos[3 * body + 1], xpos[3 * body + 2]};
  double R[9], U[9], V[9];
  q2mat(qb, R);
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {  // (M R')[i][j] = sum_k M[i][k] R[j][k]
      U[3 * i + j] = A[3 * i] * R[3 * j] + A[3 * i + 1] * R[3 * j + 1] + A[3 * i + 2] * R[3 * j + 2];
      V[3 * i + j] = Bo[3 * i] * R[3 * j] + Bo[3 * i + 1] * R[3 * j + 1] + Bo[3 * i + 2] * R[3 * j + 2];
    }
  // V -= U [xb]x ; column j of U[x]x = U (e_j-th column of skew) : (U [x]x)[i][:] = (U[i][1] x2 - U[i][2] x1, U[i][2] x0 - U[i][0] x2, U[i][0] x1 - U[i][1] x0)
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const double a = U[3 * i], b = U[3 * i + 1], c = U[3 * i + 2];
    V[3 * i] -= b * xb[2] - c * xb[1];
    V[3 * i + 1] -= c * xb[0] - a * xb[2];
    V[3 * i + 2] -= a * xb[1] - b * xb[0];
  }
  const double wp2 = wp * wp, wr2 = wr * wr;
  // symmetric 3x3 products, index order xx xy xz yy yz zz
  int n = 0;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = i; j < 3; j++) {
      const double uu = U[i] * U[j] + U[3 + i] * U[3 + j] + U[6 + i] * U[6 + j];
      const double vv = V[i] * V[j] + V[3 + i] * V[3 + j] + V[6 + i] * V[6 + j];
      out[n] = wp2 * uu;
      out[15 + n] = wp2 * vv + wr2 * uu;
      n++;
    }
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) out[6 + 3 * i + j] = wp2 * (U[i] * V[j] + U[3 + i] * V[3 + j] + U[6 + i] * V[6 + j]);
  // g = A_t' W^2 e = -[U'(wp2 e_v) ; V'(wp2 e_v) + U'(wr2 e_w)]
  const double ev[3] = {wp2 * e[0], wp2 * e[1], wp2 * e[2]}, ew[3] = {wr2 * e[3], wr2 * e[4], wr2 * e[5]};
#pragma unroll
  for (int j = 0; j < 3; j++) {
    out[21 + j] = -(U[j] * ev[0] + U[3 + j] * ev[1] + U[6 + j] * ev[2]);
    out[24 + j] = -(V[j] * ev[0] + V[3 + j] * ev[1] + V[6 + j] * ev[2] + U[j] * ew[0] + U[3 + j] * ew[1] + U[6 + j] * ew[2]);
  }
  return wp2 * (e[0] * e[0] + e[1] * e[1] + e[2] * e[2]) + wr2 * th2;
}

// y = B [m; a] for a 6x6 symmetric block stored as LL(6) LA(9) AA(6)
__device__ __forceinline__ void sym6_mul(const double *B, const double m[3], const double a[3], double fl[3], double fa[3]) {
  const double *LL = B, *LA = B + 6, *AA = B + 15;
  fl[0] = LL[0] * m[0] + LL[1] * m[1] + LL[2] * m[2] + LA[0] * a[0] + LA[1] * a[1] + LA[2] * a[2];
  fl[1] = LL[1] * m[0] + LL[3] * m[1] + LL[4] * m[2] + LA[3] * a[0] + LA[4] * a[1] + LA[5] * a[2];
  fl[2] = LL[2] * m[0] + LL[4] * m[1] + LL[5] * m[2] + LA[6] * a[0] + LA[7] * a[1] + LA[8] * a[2];
  fa[0] = LA[0] * m[0] + LA[3] * m[1] + LA[6] * m[2] + AA[0] * a[0] + AA[1] * a[1] + AA[2] * a[2];
  fa[1] = LA[1] * m[0] + LA[4] * m[1] + LA[7] * m[2] + AA[1] * a[0] + AA[3] * a[1] + AA[4] * a[2];
  fa[2] = LA[2] * m[0] + LA[5] * m[1] + LA[8] * m[2] + AA[2] * a[0] + AA[4] * a[1] + AA[5] * a[2];
}

// ------------------------------------------------------------------ exact box QP, lane = dof (row of H)
// min 1/2 x'Hx + c'x, lo <= x <= hi.  H (dense, symmetric, NVP x NVP, column-major Hm[j*NVP+i]) sits in
// LDS; each lane pulls its row into VGPRs, the wave factors it in registers (right-looking Cholesky kept
// symmetric so that after step k lane k's registers j>k hold L[j][k], which makes BOTH triangular solves
// broadcast-style), and a primal active-set loop pins/releases bounds until the KKT conditions hold --
// the unique optimum DAQP returns for mink.solve_ik's QP.  `status` (0 free, 1 at lo, 2 at hi, 3 padding)
// persists across calls as the warm-started working set.  Returns the iteration count, negative if capped.
template <int NVP>
__device__ __forceinline__ int box_qp(int lane, int n_act, const double *Hm, double ci, double lo, double hi, int &status,
                                      double &x_out) {
  const bool real_row = lane < n_act;
  const bool in_mat = lane < NVP;
  const u64 real_mask = n_act >= 64 ? ~0ull : ((1ull << n_act) - 1ull);
  double x = 0.0;
  if (real_row) {
    if (status == 1) x = lo;
    else if (status == 2) x = hi;
    else if (lo > 0.0) { x = lo; status = 1; }
    else if (hi < 0.0) { x = hi; status = 2; }
  }
  const double gtol = 1e-10 * (1.0 + wave_max(real_row ? fabs(ci) : 0.0));
  constexpr int kMaxIt = 6 * NVP + 16;
  int it = 0;
  for (; it < kMaxIt; ++it) {
    const u64 fixed = __ballot(status != 0);
    const u64 fixed_real = fixed & real_mask;
    // right-hand side: free rows -c_i - sum_{j fixed} H_ij x_j ; fixed rows x_i
    double b = -ci;
    for (u64 mm = fixed_real; mm; mm &= mm - 1) {
      const int j = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(mm));
      const double xj = rdlane(x, j);
      if (in_mat) b -= Hm[j * NVP + lane] * xj;
    }
    if (status != 0) b = x;
    double R[NVP];
#pragma unroll
    for (int j = 0; j < NVP; j++) {
      double v = in_mat ? Hm[j * NVP + lane] : 0.0;
      const bool fj = (fixed >> j) & 1ull;
      if (fj || status != 0) v = (j == lane) ? 1.0 : 0.0;
      R[j] = v;
    }
    double myinv = 1.0;
#pragma unroll
    for (int k = 0; k < NVP; k++) {
      const double dkk = rdlane(R[k], k);
      const double inv = fast_rsqrt(dkk);
      double l = R[k] * inv;
      if (lane == k) { myinv = inv; l = dkk * inv - 1.0; }
      if (lane < k) l = 0.0;
#pragma unroll
      for (int j = k + 1; j < NVP; j++) {
        const double ljk = rdlane(l, j);
        R[j] -= l * ljk;
      }
      if (lane > k) R[k] = l;
    }
    // forward L y = b (rows below k use L[i][k] = R_i[k]); backward L' z = y (rows above k use L[k][i] = R_i[k])
#pragma unroll
    for (int k = 0; k < NVP; k++) {
      const double yk = rdlane(b * myinv, k);
      const double coef = lane > k ? R[k] : 0.0;
      b = lane == k ? yk : b - coef * yk;
    }
#pragma unroll
    for (int k = NVP - 1; k >= 0; k--) {
      const double zk = rdlane(b * myinv, k);
      const double coef = lane < k ? R[k] : 0.0;
      b = lane == k ? zk : b - coef * zk;
    }
    const double z = b;
    // ratio test along x -> z over the free variables
    double a = 2.0;
    int side = 0;
    if (status == 0) {
      const double d = z - x;
      if (z > hi + 1e-14 && d > 0) { a = (hi - x) / d; side = 2; }
      else if (z < lo - 1e-14 && d < 0) { a = (lo - x) / d; side = 1; }
    }
    const double amin = wave_min(a);
    if (amin <= 1.0) {
      const u64 who = __ballot(a == amin);
      const int blk = (int)__builtin_ctzll(who);
      const double al = fmax(amin, 0.0);
      if (status == 0) x += al * (z - x);
      if (lane == blk) { x = side == 2 ? hi : lo; status = side; }
      continue;
    }
    if (status == 0) x = z;
    if (!fixed_real) { ++it; break; }
    // multipliers of the working set: g_j = c_j + sum_i H_ji x_i
    double worst = gtol;
    int rel = -1;
    for (u64 mm = fixed_real; mm; mm &= mm - 1) {
      const int j = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(mm));
      const double hx = in_mat ? Hm[j * NVP + lane] * x : 0.0;
      const double g = wave_sum(hx) + rdlane(ci, j);
      const int sj = __builtin_amdgcn_readlane(status, j);
      const double viol = sj == 1 ? -g : g;
      if (viol > worst) { worst = viol; rel = j; }
    }
    if (rel < 0) { ++it; break; }
    if (lane == rel) status = 0;
  }
  x_out = x;
  return it < kMaxIt ? it : -it;
}

// ------------------------------------------------------------------ the kernel
template <int NVP>
__global__ void __launch_bounds__(64, 2) ik_kernel(DevModel m, IkLaunch L, LdsLayout lay) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x;
  double *q = lds + lay.q, *xpos = lds + lay.xpos, *xquat = lds + lay.xquat, *tp = lds + lay.tp, *tq = lds + lay.tq;
  double *Bt = lds + lay.B, *Bc = lds + lay.Bc, *S = lds + lay.S, *Hm = lds + lay.H;
  const gmr_work_item w = L.items[blockIdx.x];
  const gmr_ik_params prm = L.prm;

  BodyConst bc;
  bc.parent = -1; bc.depth = -1; bc.jtype = 0; bc.qadr = -1;
#pragma unroll
  for (int i = 0; i < 3; i++) { bc.pos[i] = 0; bc.axis[i] = 0; }
  bc.quat[0] = 1; bc.quat[1] = bc.quat[2] = bc.quat[3] = 0;
  if (lane < m.nbody) {
    bc.parent = m.parent[lane]; bc.depth = m.depth[lane]; bc.jtype = m.jtype[lane]; bc.qadr = m.qadr[lane];
#pragma unroll
    for (int i = 0; i < 3; i++) { bc.pos[i] = m.bpos[3 * lane + i]; bc.axis[i] = m.axis[3 * lane + i]; }
#pragma unroll
    for (int i = 0; i < 4; i++) bc.quat[i] = m.bquat[4 * lane + i];
  }
  // active-dof constants of this lane (row of the QP)
  const bool real_row = lane < m.n_act;
  int a_body = 0, a_kind = -1, a_qadr = 0, a_lim = 0;
  u64 a_anc = 0;
  double a_rlo = 0, a_rhi = 0, a_axis[3] = {0, 0, 0};
  if (real_row) {
    a_body = m.abody[lane]; a_kind = m.akind[lane]; a_qadr = m.aqadr[lane]; a_lim = m.alimited[lane];
    a_anc = m.aanc[lane]; a_rlo = m.arange[2 * lane]; a_rhi = m.arange[2 * lane + 1];
#pragma unroll
    for (int i = 0; i < 3; i++) a_axis[i] = m.axis[3 * a_body + i];
  }
  // slot constants (target preparation), lane = slot
  const bool is_slot = lane < m.nslot;
  double s_scale = 1, s_poff[3] = {0, 0, 0}, s_roff[4] = {1, 0, 0, 0};
  int s_col = 0, s_foot = 0;
  if (is_slot) {
    s_scale = m.sscale[lane]; s_col = L.slot_col[lane]; s_foot = m.sfoot[lane];
#pragma unroll
    for (int i = 0; i < 3; i++) s_poff[i] = m.spoff[3 * lane + i];
#pragma unroll
    for (int i = 0; i < 4; i++) s_roff[i] = m.sroff[4 * lane + i];
  }
  const double root_scale = m.sscale[m.root_slot];
  const int root_col = L.slot_col[m.root_slot];

  for (int i = lane; i < m.nq; i += 64) q[i] = w.init_row >= 0 ? L.qinit[(size_t)w.init_row * m.nq + i] : m.qpos0[i];
  int status = real_row ? 0 : 3;
  __syncthreads();

  const int nfr = w.n_burn + w.n_out;
  for (int kf = 0; kf < nfr; ++kf) {
    const int64_t f = w.frame_begin + kf;
    // ---- target preparation (update_targets: scale_human_data + offset_human_data, table-1 offsets) ----
    {
      double hp[3] = {0, 0, 0}, hq[4] = {1, 0, 0, 0}, rp[3];
      const int64_t base = f * L.n_cols;
      if (L.in_f64) {
        const double *P = (const double *)L.hpos, *Q = (const double *)L.hquat;
#pragma unroll
        for (int i = 0; i < 3; i++) rp[i] = P[(base + root_col) * 3 + i];
        if (is_slot) {
#pragma unroll
          for (int i = 0; i < 3; i++) hp[i] = P[(base + s_col) * 3 + i];
#pragma unroll
          for (int i = 0; i < 4; i++) hq[i] = Q[(base + s_col) * 4 + i];
        }
      } else {
        const float *P = (const float *)L.hpos, *Q = (const float *)L.hquat;
#pragma unroll
        for (int i = 0; i < 3; i++) rp[i] = (double)P[(base + root_col) * 3 + i];
        if (is_slot) {
#pragma unroll
          for (int i = 0; i < 3; i++) hp[i] = (double)P[(base + s_col) * 3 + i];
#pragma unroll
          for (int i = 0; i < 4; i++) hq[i] = (double)Q[(base + s_col) * 4 + i];
        }
      }
      double pz = INFINITY;
      double p[3] = {0, 0, 0}, qo[4] = {1, 0, 0, 0}, R[9], g[3];
      if (is_slot) {
#pragma unroll
        for (int i = 0; i < 3; i++)
          p[i] = (lane == m.root_slot) ? root_scale * rp[i] : (hp[i] - rp[i]) * s_scale + root_scale * rp[i];
        const double hn = 1.0 / sqrt(hq[0] * hq[0] + hq[1] * hq[1] + hq[2] * hq[2] + hq[3] * hq[3]);
#pragma unroll
        for (int i = 0; i < 4; i++) hq[i] *= hn;
        qmul(hq, s_roff, qo);
        qnormalize(qo);
        q2mat(qo, R);
        mv(R, s_poff, g);
#pragma unroll
        for (int i = 0; i < 3; i++) p[i] += g[i];
        if (s_foot) pz = p[2];
      }
      if (prm.offset_to_ground) {
        const double lowest = wave_min(pz);
        p[2] = p[2] - lowest + 0.1;
      }
      if (is_slot) {
#pragma unroll
        for (int i = 0; i < 3; i++) tp[3 * lane + i] = p[i];
#pragma unroll
        for (int i = 0; i < 4; i++) tq[4 * lane + i] = qo[i];
      }
    }
    __syncthreads();

    int solves = 0, qpflag = 0;
    for (int tab = 0; tab < 2; ++tab) {
      if (!m.use_table[tab]) continue;
      const int nt = m.ntask[tab];
      const bool is_task = lane < nt;
      int t_body = 0, t_slot = 0;
      double t_wp = 0, t_wr = 0;
      if (is_task) {
        t_body = m.tbody[tab * GMR_MAX_TASKS + lane]; t_slot = m.tslot[tab * GMR_MAX_TASKS + lane];
        t_wp = m.twp[tab * GMR_MAX_TASKS + lane]; t_wr = m.twr[tab * GMR_MAX_TASKS + lane];
      }
      const int a_comp = real_row ? m.acomp[tab * 64 + lane] : 0;
      const int ncomp = m.ncomp[tab];

      double e[6] = {0, 0, 0, 0, 0, 0};
      fk_phase(m, bc, lane, q, xpos, xquat);
      double curr = sqrt(wave_sum(is_task ? task_residual(t_body, t_slot, xpos, xquat, tp, tq, e) : 0.0));
      int num_iter = 0;
      bool first = true;
      for (;;) {
        // ---- per-task 6x6 blocks ----
        double mu = 0.0;
        if (is_task) mu = task_block(t_body, xpos, xquat, e, t_wp, t_wr, Bt + kBT * lane);
        const double diag = prm.damping + prm.lm_damping * wave_sum(mu);
        // ---- screws S_i (world frame, about the origin) ----
        double Si[6] = {0, 0, 0, 0, 0, 0};
        if (real_row) {
          if (a_kind < 3) {
            Si[a_kind] = 1.0;  // root translation: world aligned
          } else {
            const double qb[4] = {xquat[4 * a_body], xquat[4 * a_body + 1], xquat[4 * a_body + 2], xquat[4 * a_body + 3]};
            const double xb[3] = {xpos[3 * a_body], xpos[3 * a_body + 1], xpos[3 * a_body + 2]};
            double R[9], ax[3], mo[3];
            q2mat(qb, R);
            if (a_kind < 6) { ax[0] = R[a_kind - 3]; ax[1] = R[a_kind]; ax[2] = R[a_kind + 3]; }  // root-body-frame axes
            else mv(R, a_axis, ax);
            cross(xb, ax, mo);
            Si[0] = mo[0]; Si[1] = mo[1]; Si[2] = mo[2]; Si[3] = ax[0]; Si[4] = ax[1]; Si[5] = ax[2];
          }
#pragma unroll
          for (int k = 0; k < 6; k++) S[6 * lane + k] = Si[k];
        }
        __syncthreads();
        // ---- composites: Bc[c] = sum of task blocks below the joint ----
        for (int idx = lane; idx < ncomp * kBT; idx += 64) {
          const int c = idx / kBT, el = idx - c * kBT;
          double s = 0.0;
          for (unsigned mm = m.compmask[tab * 2 * GMR_MAX_TASKS + c]; mm; mm &= mm - 1) s += Bt[kBT * __builtin_ctz(mm) + el];
          Bc[idx] = s;
        }
        __syncthreads();
        double Fi[6] = {0, 0, 0, 0, 0, 0}, ci = 0.0, lo = -1e30, hi = 1e30;
        if (real_row) {
          const double *B = Bc + kBT * a_comp;
          sym6_mul(B, Si, Si + 3, Fi, Fi + 3);
          ci = Si[0] * B[21] + Si[1] * B[22] + Si[2] * B[23] + Si[3] * B[24] + Si[4] * B[25] + Si[5] * B[26];
          if (a_lim) {  // mink ConfigurationLimit: -gain (q - lower) <= dq <= gain (upper - q)
            const double qv = q[a_qadr];
            lo = -prm.limit_gain * (qv - a_rlo);
            hi = prm.limit_gain * (a_rhi - qv);
          }
        }
        __syncthreads();  // Bc / poses are dead from here: H overwrites them
        for (int idx = lane; idx < NVP * NVP; idx += 64) Hm[idx] = 0.0;
        __syncthreads();
        if (real_row) {  // H[i][j] = S_j . F_i for every dof j above i (and the mirror), H[i][i] = S_i . F_i + diag
          for (u64 mm = a_anc; mm; mm &= mm - 1) {
            const int j = (int)__builtin_ctzll(mm);
            const double *Sj = S + 6 * j;
            const double d = Sj[0] * Fi[0] + Sj[1] * Fi[1] + Sj[2] * Fi[2] + Sj[3] * Fi[3] + Sj[4] * Fi[4] + Sj[5] * Fi[5];
            Hm[j * NVP + lane] = d;
            Hm[lane * NVP + j] = d;
          }
          Hm[lane * NVP + lane] =
              Si[0] * Fi[0] + Si[1] * Fi[1] + Si[2] * Fi[2] + Si[3] * Fi[3] + Si[4] * Fi[4] + Si[5] * Fi[5] + diag;
        } else if (lane < NVP) {
          Hm[lane * NVP + lane] = 1.0;
        }
        __syncthreads();
        double dq;
        const int qit = box_qp<NVP>(lane, m.n_act, Hm, ci, lo, hi, status, dq);
        if (qit < 0) qpflag = 1;
        // ---- integrate (mj_integratePos) ----
        {
          const double wx = rdlane(dq, 3), wy = rdlane(dq, 4), wz = rdlane(dq, 5);
          if (real_row) {
            if (a_kind < 3) q[a_kind] += dq;
            else if (a_kind == 6) q[a_qadr] += dq;
          }
          if (lane == 0) {
            const double ang = sqrt(wx * wx + wy * wy + wz * wz);
            if (ang > 0) {
              double s, c;
              sincos(0.5 * ang, &s, &c);
              s /= ang;
              const double dqt[4] = {c, s * wx, s * wy, s * wz}, q0[4] = {q[3], q[4], q[5], q[6]};
              double o[4];
              qmul(q0, dqt, o);
              const double n = 1.0 / sqrt(o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3]);
              q[3] = o[0] * n; q[4] = o[1] * n; q[5] = o[2] * n; q[6] = o[3] * n;
            }
          }
        }
        __syncthreads();
        ++solves;
        fk_phase(m, bc, lane, q, xpos, xquat);
        const double next = sqrt(wave_sum(is_task ? task_residual(t_body, t_slot, xpos, xquat, tp, tq, e) : 0.0));
        if (!first) ++num_iter;
        first = false;
        if (!(curr - next > prm.tol && num_iter < prm.max_iter)) break;
        curr = next;
      }
    }
    if (kf >= w.n_burn) {
      for (int i = lane; i < m.nq; i += 64) L.qout[(size_t)f * m.nq + i] = q[i];
      if (L.iters && lane == 0) L.iters[f] = solves | (qpflag << 30);
    }
    __syncthreads();
  }
  if (w.final_row >= 0 && L.qfinal)
    for (int i = lane; i < m.nq; i += 64) L.qfinal[(size_t)w.final_row * m.nq + i] = q[i];
}

}  // namespace gmr
