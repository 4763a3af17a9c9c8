/* gmr_oracle.c -- CPU restatement (float64, plain C) of the GMR retarget hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under gmr_amd/ (the product) may import,
 * link or call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, as the checker / the timed CPU comparator.
 *
 * PARITY STATUS
 *   - FK in the KinematicsModel convention (oracle_fk_kin): PINNED against
 *     golden vectors generated from the reference's own
 *     general_motion_retargeting/kinematics_model.py (tests/golden/).
 *   - IK (everything that goes through mink / mujoco / daqp): PARITY UNPINNED.
 *     Those packages are third-party, unpinned in the reference's setup.py:16-20,
 *     absent from /root/reference and not installed here, and the reference
 *     holds no golden qpos.  What follows restates GMR's own code line by line
 *     (cited) and the *published* algorithms of mink (differential IK with
 *     FrameTask / ConfigurationLimit / solve_ik), MuJoCo (mj_kinematics,
 *     mj_jacBody, mj_integratePos) and an exact dense box-QP (what DAQP
 *     returns).  It is pinned only by mathematical invariants (finite-difference
 *     Jacobians, KKT residuals, reachable-target recovery) in tests/.
 *
 * Deliberately written in the "textbook" form the reference executes -- a dense
 * 6 x nv Jacobian per task, a dense nv x nv H, Cholesky from scratch per
 * active-set change -- so that it is an independent check of the HIP kernels,
 * which use a different (composite-inertia) assembly of the same H and c.
 *
 * Conventions: quaternions are wxyz unless a name says xyzw.  Tangent vectors
 * are [v; w] (translation first), as in mink.  qpos = [p(3), quat wxyz(4), hinges].
 * dq = [dp world(3), dw root-body frame(3), dtheta].
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/gmr_blob.h"

#define MAXV 96   /* nv upper bound for stack arrays  */
#define MAXT GMR_MAX_TASKS
#define MAXS GMR_MAX_SLOTS
#define MAXB 128

typedef struct oracle_model {
  gmr_blob_header h;
  uint8_t *blob;
  const int32_t *parent, *jnt_type, *qpos_adr, *dof_adr, *jnt_limited;
  const double *body_pos, *body_quat, *body_quat_raw, *jnt_axis, *jnt_range, *qpos0;
  const double *slot_scale, *slot_pos_off, *slot_rot_off;
  const int32_t *slot_is_foot;
  const int32_t *task_body[2], *task_slot[2];
  const double *task_wp[2], *task_wr[2];
} oracle_model;

/* ------------------------------------------------------------------ small math */
static void quat_mul(const double a[4], const double b[4], double o[4]) {
  /* Hamilton product, wxyz (same product as reference rot_utils.py:27-56). */
  double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  double z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  o[0] = w; o[1] = x; o[2] = y; o[3] = z;
}
static void quat_conj(const double a[4], double o[4]) { o[0] = a[0]; o[1] = -a[1]; o[2] = -a[2]; o[3] = -a[3]; }
static void quat_normalize(double q[4]) {
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
static void quat_to_mat(const double q[4], double R[9]) {
  /* mju_quat2Mat */
  double q00 = q[0] * q[0], q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3];
  double q11 = q[1] * q[1], q12 = q[1] * q[2], q13 = q[1] * q[3];
  double q22 = q[2] * q[2], q23 = q[2] * q[3], q33 = q[3] * q[3];
  R[0] = q00 + q11 - q22 - q33; R[4] = q00 - q11 + q22 - q33; R[8] = q00 - q11 - q22 + q33;
  R[1] = 2 * (q12 - q03); R[2] = 2 * (q13 + q02);
  R[3] = 2 * (q12 + q03); R[5] = 2 * (q23 - q01);
  R[6] = 2 * (q13 - q02); R[7] = 2 * (q23 + q01);
}
static void mat_vec(const double R[9], const double v[3], double o[3]) {
  double a = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
  double b = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
  double c = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
  o[0] = a; o[1] = b; o[2] = c;
}
static void matT_vec(const double R[9], const double v[3], double o[3]) {
  double a = R[0] * v[0] + R[3] * v[1] + R[6] * v[2];
  double b = R[1] * v[0] + R[4] * v[1] + R[7] * v[2];
  double c = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
  o[0] = a; o[1] = b; o[2] = c;
}
static void cross3(const double a[3], const double b[3], double o[3]) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
static void skew3(const double v[3], double S[9]) {
  S[0] = 0; S[1] = -v[2]; S[2] = v[1];
  S[3] = v[2]; S[4] = 0; S[5] = -v[0];
  S[6] = -v[1]; S[7] = v[0]; S[8] = 0;
}
static void mm3(const double A[9], const double B[9], double C[9]) {
  double T[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) T[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
  memcpy(C, T, sizeof(T));
}
static void tr3(const double A[9], double T[9]) {
  double B[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) B[3 * i + j] = A[3 * j + i];
  memcpy(T, B, sizeof(B));
}

/* ------------------------------------------------------------------ model */
oracle_model *oracle_model_create(const void *blob, size_t nbytes) {
  if (nbytes < sizeof(gmr_blob_header)) return NULL;
  gmr_blob_header h;
  memcpy(&h, blob, sizeof(h));
  if (h.magic != GMR_BLOB_MAGIC || h.version != GMR_BLOB_VERSION || h.total_bytes != nbytes) return NULL;
  if (h.nbody < 1 || h.nbody > MAXB || h.nv > MAXV || h.ntask[0] > MAXT || h.ntask[1] > MAXT || h.nslot > MAXS) return NULL;
  oracle_model *m = (oracle_model *)calloc(1, sizeof(*m));
  m->h = h;
  m->blob = (uint8_t *)malloc(nbytes);
  memcpy(m->blob, blob, nbytes);
#define P(T, off) ((const T *)(m->blob + (off)))
  m->parent = P(int32_t, h.off_parent); m->jnt_type = P(int32_t, h.off_jnt_type);
  m->qpos_adr = P(int32_t, h.off_qpos_adr); m->dof_adr = P(int32_t, h.off_dof_adr);
  m->jnt_limited = P(int32_t, h.off_jnt_limited);
  m->body_pos = P(double, h.off_body_pos); m->body_quat = P(double, h.off_body_quat);
  m->body_quat_raw = P(double, h.off_body_quat_raw); m->jnt_axis = P(double, h.off_jnt_axis);
  m->jnt_range = P(double, h.off_jnt_range); m->qpos0 = P(double, h.off_qpos0);
  m->slot_scale = P(double, h.off_slot_scale); m->slot_pos_off = P(double, h.off_slot_pos_off);
  m->slot_rot_off = P(double, h.off_slot_rot_off); m->slot_is_foot = P(int32_t, h.off_slot_is_foot);
  for (int k = 0; k < 2; k++) {
    m->task_body[k] = P(int32_t, h.off_task_body[k]); m->task_slot[k] = P(int32_t, h.off_task_slot[k]);
    m->task_wp[k] = P(double, h.off_task_wp[k]); m->task_wr[k] = P(double, h.off_task_wr[k]);
  }
#undef P
  return m;
}
void oracle_model_destroy(oracle_model *m) {
  if (!m) return;
  free(m->blob);
  free(m);
}

/* ------------------------------------------------------------------ FK, MuJoCo convention
 * mj_kinematics for a tree of hinge joints with joint pos = 0 under a free root
 * (what mink.Configuration.update() runs; call sites motion_retarget.py:75,150):
 *   xpos_j  = xpos_p + R(xquat_p) * body_pos_j
 *   xquat_j = xquat_p (x) body_quat_j (x) [cos(t/2), sin(t/2) * axis_j]                */
void oracle_fk_mj(const oracle_model *m, const double *qpos, double *xpos, double *xquat) {
  int nb = m->h.nbody;
  for (int b = 0; b < nb; b++) {
    if (b == 0) {
      memcpy(xpos, qpos, 3 * sizeof(double));
      memcpy(xquat, qpos + 3, 4 * sizeof(double));
      quat_normalize(xquat);
      continue;
    }
    int p = m->parent[b];
    double R[9], t[3], q[4];
    quat_to_mat(xquat + 4 * p, R);
    mat_vec(R, m->body_pos + 3 * b, t);
    for (int i = 0; i < 3; i++) xpos[3 * b + i] = xpos[3 * p + i] + t[i];
    quat_mul(xquat + 4 * p, m->body_quat + 4 * b, q);
    if (m->jnt_type[b] == GMR_JNT_HINGE) {
      double th = qpos[m->qpos_adr[b]], s = sin(0.5 * th), jq[4] = {cos(0.5 * th), 0, 0, 0}, o[4];
      for (int i = 0; i < 3; i++) jq[1 + i] = s * m->jnt_axis[3 * b + i];
      quat_mul(q, jq, o);
      memcpy(q, o, sizeof(o));
    }
    quat_normalize(q);
    memcpy(xquat + 4 * b, q, sizeof(q));
  }
}

/* ------------------------------------------------------------------ Lie group pieces (mink.lie) */
#define LIE_EPS 1e-10 /* mink.lie.utils.get_epsilon(float64) */

/* SO3.log of a unit quaternion (short side), as in mink/jaxlie. */
static void so3_log(const double q[4], double w_out[3]) {
  double w = q[0], n2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3], f;
  if (n2 < LIE_EPS) {
    f = 2.0 / w - 2.0 / 3.0 * n2 / (w * w * w);
  } else {
    double n = sqrt(n2);
    if (fabs(w) < LIE_EPS) f = (w > 0 ? 1.0 : -1.0) * M_PI / n;
    else f = 2.0 * atan2(w < 0 ? -n : n, fabs(w)) / n;
  }
  for (int i = 0; i < 3; i++) w_out[i] = f * q[1 + i];
}

/* SE3.log of T = (q, t): [V^-1 t ; omega]. */
static void se3_log(const double q[4], const double t[3], double e[6]) {
  double om[3];
  so3_log(q, om);
  double th2 = om[0] * om[0] + om[1] * om[1] + om[2] * om[2];
  double K[9], K2[9], Vi[9];
  skew3(om, K);
  mm3(K, K, K2);
  double c2;
  if (th2 < LIE_EPS) c2 = 1.0 / 12.0;
  else {
    double th = sqrt(th2), hth = 0.5 * th;
    c2 = (1.0 - 0.5 * th * cos(hth) / sin(hth)) / th2;
  }
  for (int i = 0; i < 9; i++) Vi[i] = (i % 4 == 0 ? 1.0 : 0.0) - 0.5 * K[i] + c2 * K2[i];
  mat_vec(Vi, t, e);
  e[3] = om[0]; e[4] = om[1]; e[5] = om[2];
}

/* SO3 left-Jacobian inverse. */
static void so3_ljacinv(const double th[3], double J[9]) {
  double t2 = th[0] * th[0] + th[1] * th[1] + th[2] * th[2], K[9], K2[9], c2;
  skew3(th, K);
  mm3(K, K, K2);
  if (t2 < LIE_EPS) c2 = 1.0 / 12.0;
  else {
    double t = sqrt(t2);
    c2 = 1.0 / t2 - (1.0 + cos(t)) / (2.0 * t * sin(t));
  }
  for (int i = 0; i < 9; i++) J[i] = (i % 4 == 0 ? 1.0 : 0.0) - 0.5 * K[i] + c2 * K2[i];
}

/* Barfoot's Q(rho, phi) block of the SE(3) left Jacobian (mink.lie.se3._getQ). */
static void se3_Q(const double e[6], double Q[9]) {
  const double *rho = e, *phi = e + 3;
  double t2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  double A = 0.5, B, C, D;
  if (t2 < LIE_EPS) {
    B = 1.0 / 6.0 + 1.0 / 120.0 * t2;
    C = -1.0 / 24.0 + 1.0 / 720.0 * t2;
    D = -1.0 / 60.0;
  } else {
    double t = sqrt(t2), s = sin(t), c = cos(t);
    B = (t - s) / (t2 * t);
    C = (1.0 - t2 / 2.0 - c) / (t2 * t2);
    D = (2.0 * t - 3.0 * s + t * c) / (2.0 * t2 * t2 * t);
  }
  double V[9], W[9], VW[9], WV[9], WVW[9], VWW[9], VWWt[9], T1[9], T2[9];
  skew3(rho, V);
  skew3(phi, W);
  mm3(V, W, VW);
  tr3(VW, WV); /* (VW)^T = W^T V^T = WV for skew V, W */
  mm3(WV, W, WVW);
  mm3(VW, W, VWW);
  tr3(VWW, VWWt);
  mm3(WVW, W, T1);
  mm3(W, WVW, T2);
  for (int i = 0; i < 9; i++)
    Q[i] = A * V[i] + B * (WV[i] + VW[i] + WVW[i]) - C * (VWW[i] - VWWt[i] - 3.0 * WVW[i]) + D * (T1[i] + T2[i]);
}

/* SE3 left-Jacobian inverse, tangent order [rho; phi]; identity below the
 * small-angle threshold exactly as mink.lie.se3.SE3.ljacinv does. */
static void se3_ljacinv(const double e[6], double J[36]) {
  const double *phi = e + 3;
  double t2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  memset(J, 0, 36 * sizeof(double));
  if (t2 < LIE_EPS) {
    for (int i = 0; i < 6; i++) J[7 * i] = 1.0;
    return;
  }
  double Ji[9], Q[9], T[9], B[9];
  so3_ljacinv(phi, Ji);
  se3_Q(e, Q);
  mm3(Ji, Q, T);
  mm3(T, Ji, B);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      J[6 * i + j] = Ji[3 * i + j];
      J[6 * (i + 3) + (j + 3)] = Ji[3 * i + j];
      J[6 * i + (j + 3)] = -B[3 * i + j];
    }
}

/* ------------------------------------------------------------------ FrameTask pieces
 * mink.FrameTask.compute_error:  e = log(T_wb^-1 T_wt)  (body twist, [v; w]).
 * Newer mink releases flip the sign of e and of the Jacobian together; H, c and
 * |e| -- everything GMR consumes (motion_retarget.py:147-151,188-200) -- are
 * invariant to that choice.                                                       */
static void task_error(const double *xpos_b, const double *xquat_b, const double *tpos, const double *tquat, double e[6]) {
  double qc[4], qr[4], R[9], d[3], t[3];
  quat_conj(xquat_b, qc);
  quat_mul(qc, tquat, qr);
  quat_to_mat(xquat_b, R);
  for (int i = 0; i < 3; i++) d[i] = tpos[i] - xpos_b[i];
  matT_vec(R, d, t);
  se3_log(qr, t, e);
}

/* mink.Configuration.get_frame_jacobian(frame_type="body"): mj_jacBody rotated
 * into the body frame.  Jb is 6 x nv row-major, rows [v; w].                     */
static void body_jacobian(const oracle_model *m, const double *xpos, const double *xquat, int body, double *Jb) {
  int nv = m->h.nv;
  memset(Jb, 0, 6 * nv * sizeof(double));
  double Rb[9];
  quat_to_mat(xquat + 4 * body, Rb);
  const double *pb = xpos + 3 * body;
  for (int b = body; b >= 0; b = m->parent[b]) {
    if (m->jnt_type[b] == GMR_JNT_HINGE) {
      int k = m->dof_adr[b];
      double R[9], ax[3], r[3], jp[3], lp[3], lr[3];
      quat_to_mat(xquat + 4 * b, R);
      mat_vec(R, m->jnt_axis + 3 * b, ax); /* world axis (invariant under the hinge's own rotation) */
      for (int i = 0; i < 3; i++) r[i] = pb[i] - xpos[3 * b + i];
      cross3(ax, r, jp);
      matT_vec(Rb, jp, lp);
      matT_vec(Rb, ax, lr);
      for (int i = 0; i < 3; i++) { Jb[i * nv + k] = lp[i]; Jb[(i + 3) * nv + k] = lr[i]; }
    } else if (m->jnt_type[b] == GMR_JNT_FREE) {
      int k0 = m->dof_adr[b];
      double R0[9];
      quat_to_mat(xquat + 4 * b, R0);
      for (int k = 0; k < 3; k++) {
        double ek[3] = {0, 0, 0}, lp[3];
        ek[k] = 1.0;
        matT_vec(Rb, ek, lp); /* translational dofs: world aligned */
        for (int i = 0; i < 3; i++) Jb[i * nv + k0 + k] = lp[i];
        double ax[3] = {R0[k], R0[3 + k], R0[6 + k]}, r[3], jp[3], lr[3];
        for (int i = 0; i < 3; i++) r[i] = pb[i] - xpos[3 * b + i];
        cross3(ax, r, jp); /* rotational dofs: root-body-frame axes */
        matT_vec(Rb, jp, lp);
        matT_vec(Rb, ax, lr);
        for (int i = 0; i < 3; i++) { Jb[i * nv + k0 + 3 + k] = lp[i]; Jb[(i + 3) * nv + k0 + 3 + k] = lr[i]; }
      }
    }
  }
}

/* mink.FrameTask.compute_jacobian: J = -Jlog(T_tb) J_b = -ljacinv(e) J_b = de/dq. */
static void task_jacobian(const oracle_model *m, const double *xpos, const double *xquat, int body, const double e[6], double *J) {
  int nv = m->h.nv;
  double Jb[6 * MAXV], L[36];
  body_jacobian(m, xpos, xquat, body, Jb);
  se3_ljacinv(e, L);
  for (int i = 0; i < 6; i++)
    for (int k = 0; k < nv; k++) {
      double s = 0;
      for (int j = 0; j < 6; j++) s += L[6 * i + j] * Jb[j * nv + k];
      J[i * nv + k] = -s;
    }
}

/* exported for the finite-difference test */
void oracle_task_error_and_jacobian(const oracle_model *m, const double *qpos, int body, const double *tpos, const double *tquat, double *e, double *J) {
  double xpos[3 * MAXB], xquat[4 * MAXB];
  oracle_fk_mj(m, qpos, xpos, xquat);
  task_error(xpos + 3 * body, xquat + 4 * body, tpos, tquat, e);
  task_jacobian(m, xpos, xquat, body, e, J);
}

/* ------------------------------------------------------------------ manifold integrate
 * mj_integratePos with dt folded in (v = dq/dt then q (+)= v dt: motion_retarget.py:146-150). */
void oracle_integrate(const oracle_model *m, double *qpos, const double *dq) {
  for (int b = 0; b < m->h.nbody; b++) {
    if (m->jnt_type[b] == GMR_JNT_FREE) {
      int qa = m->qpos_adr[b], da = m->dof_adr[b];
      for (int i = 0; i < 3; i++) qpos[qa + i] += dq[da + i];
      const double *w = dq + da + 3;
      double ang = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
      if (ang > 0) { /* mju_quatIntegrate: q <- normalize(q (x) axisangle(w/|w|, |w|)) */
        double s = sin(0.5 * ang) / ang, dqt[4] = {cos(0.5 * ang), s * w[0], s * w[1], s * w[2]}, o[4];
        quat_mul(qpos + qa + 3, dqt, o);
        quat_normalize(o);
        memcpy(qpos + qa + 3, o, sizeof(o));
      }
    } else if (m->jnt_type[b] == GMR_JNT_HINGE) {
      qpos[m->qpos_adr[b]] += dq[m->dof_adr[b]];
    }
  }
}

/* ------------------------------------------------------------------ exact box QP
 * min 1/2 x'Hx + c'x  s.t. lo <= x <= hi, H symmetric positive definite (dense,
 * row-major n x n).  Primal active set, Cholesky from scratch per change.  This
 * is the unique optimum DAQP returns for the QP mink.solve_ik builds.
 * Returns the number of active-set iterations, or -1 on failure.               */
static int chol_solve(int n, const double *A, const double *b, double *x) {
  /* A: n x n SPD (copied), solves A x = b */
  double L[MAXV * MAXV];
  for (int i = 0; i < n; i++)
    for (int j = 0; j <= i; j++) {
      double s = A[i * n + j];
      for (int k = 0; k < j; k++) s -= L[i * n + k] * L[j * n + k];
      if (i == j) {
        if (s <= 0) return -1;
        L[i * n + i] = sqrt(s);
      } else L[i * n + j] = s / L[j * n + j];
    }
  double y[MAXV];
  for (int i = 0; i < n; i++) {
    double s = b[i];
    for (int k = 0; k < i; k++) s -= L[i * n + k] * y[k];
    y[i] = s / L[i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    double s = y[i];
    for (int k = i + 1; k < n; k++) s -= L[k * n + i] * x[k];
    x[i] = s / L[i * n + i];
  }
  return 0;
}

int oracle_box_qp(int n, const double *H, const double *c, const double *lo, const double *hi, double *x) {
  int status[MAXV]; /* 0 free, -1 at lo, +1 at hi */
  double cmax = 0;
  for (int i = 0; i < n; i++) {
    double v = 0;
    status[i] = 0;
    if (lo[i] > 0) { v = lo[i]; status[i] = -1; }
    if (hi[i] < 0) { v = hi[i]; status[i] = 1; }
    x[i] = v;
    if (fabs(c[i]) > cmax) cmax = fabs(c[i]);
  }
  const double gtol = 1e-10 * (1.0 + cmax);
  for (int it = 0; it < 20 * n + 20; it++) {
    int fidx[MAXV], nf = 0;
    for (int i = 0; i < n; i++) if (status[i] == 0) fidx[nf++] = i;
    double z[MAXV];
    if (nf > 0) {
      double A[MAXV * MAXV], b[MAXV], zf[MAXV];
      for (int a = 0; a < nf; a++) {
        int i = fidx[a];
        double s = -c[i];
        for (int j = 0; j < n; j++) if (status[j] != 0) s -= H[i * n + j] * x[j];
        b[a] = s;
        for (int bb = 0; bb < nf; bb++) A[a * nf + bb] = H[i * n + fidx[bb]];
      }
      if (chol_solve(nf, A, b, zf) != 0) return -1;
      for (int a = 0; a < nf; a++) z[fidx[a]] = zf[a];
    }
    double alpha = 1.0;
    int block = -1, side = 0;
    for (int a = 0; a < nf; a++) {
      int i = fidx[a];
      double d = z[i] - x[i];
      if (z[i] > hi[i] + 1e-14 && d > 0) {
        double s = (hi[i] - x[i]) / d;
        if (s < alpha) { alpha = s; block = i; side = 1; }
      } else if (z[i] < lo[i] - 1e-14 && d < 0) {
        double s = (lo[i] - x[i]) / d;
        if (s < alpha) { alpha = s; block = i; side = -1; }
      }
    }
    if (block >= 0) {
      if (alpha < 0) alpha = 0;
      for (int a = 0; a < nf; a++) { int i = fidx[a]; x[i] += alpha * (z[i] - x[i]); }
      x[block] = side > 0 ? hi[block] : lo[block];
      status[block] = side;
      continue;
    }
    for (int a = 0; a < nf; a++) { int i = fidx[a]; x[i] = z[i]; }
    /* multipliers of the active bounds */
    double worst = gtol;
    int rel = -1;
    for (int i = 0; i < n; i++) {
      if (status[i] == 0) continue;
      double g = c[i];
      for (int j = 0; j < n; j++) g += H[i * n + j] * x[j];
      double viol = status[i] < 0 ? -g : g; /* at lo need g >= 0; at hi need g <= 0 */
      if (viol > worst) { worst = viol; rel = i; }
    }
    if (rel < 0) return it + 1;
    status[rel] = 0;
  }
  return -1;
}

/* ------------------------------------------------------------------ target preparation
 * GeneralMotionRetargeting.update_targets: scale_human_data then offset_human_data with
 * the TABLE-1 offsets (motion_retarget.py:117-124, 209-250), optional
 * offset_human_data_to_ground (:252-270).  hp/hq: raw human pos/quat per slot.   */
/* hscale: per-clip factor on every human_scale_table entry (actual_human_height of this clip over the height the model was
 * compiled with, motion_retarget.py:36-43); 1.0 for the model's own ratio. */
void oracle_prepare_targets_scaled(const oracle_model *m, const double *hp, const double *hq, int offset_to_ground, double hscale, double *tp, double *tq) {
  int ns = m->h.nslot, rs = m->h.root_slot;
  const double *root = hp + 3 * rs;
  double sroot[3];
  for (int i = 0; i < 3; i++) sroot[i] = hscale * m->slot_scale[rs] * root[i]; /* :215 scaled about the world origin */
  for (int s = 0; s < ns; s++) {
    double p[3], q[4], R[9], g[3];
    if (s == rs) memcpy(p, sroot, sizeof(p));
    else for (int i = 0; i < 3; i++) p[i] = (hp[3 * s + i] - root[i]) * (hscale * m->slot_scale[s]) + sroot[i]; /* :225,230 */
    memcpy(q, hq + 4 * s, sizeof(q));
    quat_normalize(q); /* scipy Rotation.from_quat normalises */
    quat_mul(q, m->slot_rot_off + 4 * s, tq + 4 * s); /* :241 */
    quat_normalize(tq + 4 * s);
    quat_to_mat(tq + 4 * s, R);
    mat_vec(R, m->slot_pos_off + 3 * s, g); /* :244-248 */
    for (int i = 0; i < 3; i++) tp[3 * s + i] = p[i] + g[i];
  }
  if (offset_to_ground) {
    double lowest = INFINITY;
    for (int s = 0; s < ns; s++)
      if (m->slot_is_foot[s] && tp[3 * s + 2] < lowest) lowest = tp[3 * s + 2];
    for (int s = 0; s < ns; s++) tp[3 * s + 2] = tp[3 * s + 2] - lowest + 0.1;
  }
}

void oracle_prepare_targets(const oracle_model *m, const double *hp, const double *hq, int offset_to_ground, double *tp, double *tq) {
  oracle_prepare_targets_scaled(m, hp, hq, offset_to_ground, 1.0, tp, tq);
}

/* ------------------------------------------------------------------ one stage quantities */
static double stage_error(const oracle_model *m, int tab, const double *xpos, const double *xquat, const double *tp, const double *tq, double *e_all) {
  /* error1()/error2(): 2-norm of the concatenated, UNWEIGHTED 6-vectors (motion_retarget.py:188-200). */
  double s2 = 0;
  for (int t = 0; t < m->h.ntask[tab]; t++) {
    int b = m->task_body[tab][t], s = m->task_slot[tab][t];
    task_error(xpos + 3 * b, xquat + 4 * b, tp + 3 * s, tq + 4 * s, e_all + 6 * t);
    for (int i = 0; i < 6; i++) s2 += e_all[6 * t + i] * e_all[6 * t + i];
  }
  return sqrt(s2);
}

/* mink.solve_ik's QP for one table at the current configuration:
 *   H = damping I + sum_t [ (W J)'(W J) + lm |W e|^2 I ],  c = sum_t (W J)'(W e)
 *   box: -gain (q - lower) <= dq <= gain (upper - q) on limited hinges.          */
/* Experiment switch (tests / tools only): 1 = emulate a mixed-precision assembly -- the weighted Jacobian rows and residuals
 * rounded to float32 and H, c accumulated in float32 (what packed v_pk_fma_f32 task blocks / composites / H pairs would carry),
 * everything else (FK, residual, LM damping, QP, termination test) in float64.  0 = the reference's float64 throughout. */
static int g_mixed_assembly = 0;
void oracle_set_mixed_assembly(int on) { g_mixed_assembly = on; }

void oracle_build_qp(const oracle_model *m, int tab, const gmr_ik_params *prm, const double *qpos, const double *xpos, const double *xquat,
                     const double *tp, const double *tq, const double *e_all, double *H, double *c, double *lo, double *hi) {
  int nv = m->h.nv;
  memset(H, 0, nv * nv * sizeof(double));
  memset(c, 0, nv * sizeof(double));
  double diag = prm->damping;
  for (int t = 0; t < m->h.ntask[tab]; t++) {
    int b = m->task_body[tab][t];
    double J[6 * MAXV], w[6], we2 = 0;
    const double *e = e_all + 6 * t;
    task_jacobian(m, xpos, xquat, b, e, J);
    for (int i = 0; i < 6; i++) {
      w[i] = i < 3 ? m->task_wp[tab][t] : m->task_wr[tab][t];
      we2 += (w[i] * e[i]) * (w[i] * e[i]);
    }
    diag += prm->lm_damping * we2;
    if (g_mixed_assembly) {
      for (int i = 0; i < 6; i++) {
        if (w[i] == 0) continue;
        const double *Ji = J + i * nv;
        float wei = (float)(w[i] * e[i]);
        for (int k = 0; k < nv; k++) {
          if (Ji[k] == 0) continue;
          float a = (float)(w[i] * Ji[k]);
          c[k] = (double)((float)c[k] + a * wei);
          for (int l = 0; l < nv; l++) H[k * nv + l] = (double)((float)H[k * nv + l] + a * (float)(w[i] * Ji[l]));
        }
      }
      continue;
    }
    for (int i = 0; i < 6; i++) {
      double w2 = w[i] * w[i];
      if (w2 == 0) continue;
      const double *Ji = J + i * nv;
      for (int k = 0; k < nv; k++) {
        if (Ji[k] == 0) continue;
        double a = w2 * Ji[k];
        c[k] += a * e[i];
        for (int l = 0; l < nv; l++) H[k * nv + l] += a * Ji[l];
      }
    }
  }
  /* a root that is not a full free joint (gmr_blob.h root_dof_mask; the planar base of assets/galaxea_r1pro/r1_pro.xml:102-104):
   * the reference's model simply has no such dofs, i.e. its H is the sub-matrix without them.  Decoupling them here (zero row,
   * column and gradient, diagonal = the damping) solves that sub-problem exactly and leaves dq = 0 on the absent dofs. */
  if (m->h.root_dof_mask)
    for (int k = 0; k < 6; k++)
      if (!((m->h.root_dof_mask >> k) & 1)) {
        for (int l = 0; l < nv; l++) H[k * nv + l] = H[l * nv + k] = 0.0;
        c[k] = 0.0;
      }
  for (int k = 0; k < nv; k++) H[k * nv + k] += diag;
  (void)tp; (void)tq;
  for (int k = 0; k < nv; k++) { lo[k] = -1e30; hi[k] = 1e30; }
  for (int b = 0; b < m->h.nbody; b++) {
    if (m->jnt_type[b] != GMR_JNT_HINGE || !m->jnt_limited[b]) continue;
    int k = m->dof_adr[b];
    double q = qpos[m->qpos_adr[b]];
    lo[k] = -prm->limit_gain * (q - m->jnt_range[2 * b]);
    hi[k] = prm->limit_gain * (m->jnt_range[2 * b + 1] - q);
  }
}

/* One IK stage of retarget() (motion_retarget.py:143-161 / :163-182). Returns #solves. */
static int run_stage(const oracle_model *m, int tab, const gmr_ik_params *prm, double *qpos, const double *tp, const double *tq, double *err_out) {
  int nv = m->h.nv, solves = 0;
  double xpos[3 * MAXB], xquat[4 * MAXB], e[6 * MAXT];
  double *H = (double *)malloc(sizeof(double) * nv * nv), c[MAXV], lo[MAXV], hi[MAXV], dq[MAXV];
  oracle_fk_mj(m, qpos, xpos, xquat);
  double curr = stage_error(m, tab, xpos, xquat, tp, tq, e), next;
  int num_iter = 0;
  for (;;) {
    oracle_build_qp(m, tab, prm, qpos, xpos, xquat, tp, tq, e, H, c, lo, hi);
    if (oracle_box_qp(nv, H, c, lo, hi, dq) < 0) { free(H); return -1000; }
    oracle_integrate(m, qpos, dq);
    solves++;
    oracle_fk_mj(m, qpos, xpos, xquat);
    next = stage_error(m, tab, xpos, xquat, tp, tq, e);
    if (solves > 1) num_iter++;
    if (!(curr - next > prm->tol && num_iter < prm->max_iter)) break;
    curr = next;
  }
  free(H);
  if (err_out) *err_out = next;
  return solves;
}

/* GeneralMotionRetargeting.retarget for one frame (motion_retarget.py:139-185).
 * hp/hq: [nslot][3|4] raw human data of the frame (already gathered per slot). */
static int root_target_slot(const oracle_model *m) {
  /* GMR_INIT_ROOT_TARGET (gmr_blob.h): slot of the first used table's task on the floating base (body 0), or -1 */
  for (int k = 0; k < 2; k++)
    for (int t = 0; m->h.use_table[k] && t < m->h.ntask[k]; t++)
      if (m->task_body[k][t] == 0) return m->task_slot[k][t];
  return -1;
}

/* init_root != 0: before solving, place the floating base on the prepared target of the root body's task. */
int oracle_retarget_frame_ex(const oracle_model *m, const gmr_ik_params *prm, double *qpos, const double *hp, const double *hq, double hscale,
                             int init_root, double *errs) {
  double tp[3 * MAXS], tq[4 * MAXS];
  int total = 0;
  oracle_prepare_targets_scaled(m, hp, hq, prm->offset_to_ground, hscale, tp, tq);
  if (init_root) {
    int rts = root_target_slot(m);
    if (rts >= 0 && m->h.root_dof_mask == 0x23) { /* planar base: the target's place and heading only */
      const double *t = tq + 4 * rts;
      double S = 2.0 * (t[0] * t[3] + t[1] * t[2]), C = 1.0 - 2.0 * (t[2] * t[2] + t[3] * t[3]), n2 = S * S + C * C, ch = 1.0, sh = 0.0;
      if (n2 > 1e-24) { /* ZYX yaw as a half-angle pair */
        double cy = C / sqrt(n2), sy = S / sqrt(n2);
        if (cy >= 0.0) { ch = sqrt(0.5 * (1.0 + cy)); sh = 0.5 * sy / ch; }
        else { sh = sqrt(0.5 * (1.0 - cy)); if (sy < 0.0) sh = -sh; ch = 0.5 * sy / sh; }
      }
      qpos[0] = tp[3 * rts]; qpos[1] = tp[3 * rts + 1];
      qpos[3] = ch; qpos[4] = 0.0; qpos[5] = 0.0; qpos[6] = sh;
    } else if (rts >= 0) { memcpy(qpos, tp + 3 * rts, 3 * sizeof(double)); memcpy(qpos + 3, tq + 4 * rts, 4 * sizeof(double)); }
  }
  for (int tab = 0; tab < 2; tab++) {
    if (!m->h.use_table[tab]) continue;
    int s = run_stage(m, tab, prm, qpos, tp, tq, errs ? errs + tab : NULL);
    if (s < 0) return s;
    total += s;
  }
  return total;
}

int oracle_retarget_frame(const oracle_model *m, const gmr_ik_params *prm, double *qpos, const double *hp, const double *hq, double *errs) {
  return oracle_retarget_frame_ex(m, prm, qpos, hp, hq, 1.0, 0, errs);
}

static void gather_frame(const void *pos, const void *quat, int in_f64, int n_cols, const int32_t *slot_col, int ns, int64_t f, double *hp, double *hq) {
  for (int s = 0; s < ns; s++) {
    int64_t cidx = f * n_cols + slot_col[s];
    for (int i = 0; i < 3; i++) hp[3 * s + i] = in_f64 ? ((const double *)pos)[3 * cidx + i] : (double)((const float *)pos)[3 * cidx + i];
    for (int i = 0; i < 4; i++) hq[4 * s + i] = in_f64 ? ((const double *)quat)[4 * cidx + i] : (double)((const float *)quat)[4 * cidx + i];
  }
}

/* Batch driver with the same work-item semantics as gmr_ik_solve (include/gmr_amd.h):
 * the caller loop `for frame in frames: retarget(frame)` of
 * scripts/smplx_to_robot_dataset.py:84-89 per item, warm start carried inside an item.
 * n_threads > 1 runs items on that many OpenMP threads (clip-parallel CPU baseline). */
int oracle_ik_solve(const oracle_model *m, const gmr_ik_params *prm, const void *pos, const void *quat, int in_f64, int n_cols,
                    const int32_t *slot_col, const gmr_work_item *items, int n_items, const double *qpos_init, double *qpos_final,
                    double *qpos_out, int32_t *iters_out, int32_t *frames_done, int n_threads) {
  int nq = m->h.nq, ns = m->h.nslot, fail = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads > 0 ? n_threads : 1) reduction(| : fail)
#endif
  for (int it = 0; it < n_items; it++) {
    const gmr_work_item *w = items + it;
    double q[MAXV + 1], hp[3 * MAXS], hq[4 * MAXS];
    memcpy(q, w->init_row >= 0 ? qpos_init + (size_t)w->init_row * nq : m->qpos0, nq * sizeof(double));
    int out_done = 0, kc = 0, left = 0, nfr = w->n_burn + w->n_out;
    double hscale = w->height_scale != 0.0 ? w->height_scale : 1.0;
    for (int k = 0; k < nfr; k++) {
      if (w->check_stride > 0 && left == 0) { /* verification walk (gmr_blob.h): adopt a consistent chunk, solve an inconsistent one */
        double *B = qpos_final + (size_t)(w->burn_row + kc) * nq, d = 0.0;
        int len = w->check_stride < nfr - k ? w->check_stride : nfr - k;
        /* the base quaternion is compared up to sign (q and -q are one rotation; a chunk started on a target carries the target's
         * sign, the sequence its own): a chunk adopted with the other sign has its stored quaternions turned to the sequence's */
        double sgn = (q[3] * B[3] + q[4] * B[4] + q[5] * B[5] + q[6] * B[6]) < 0.0 ? -1.0 : 1.0;
        for (int i = 0; i < nq; i++) d = fmax(d, fabs(q[i] - ((i >= 3 && i < 7) ? sgn * B[i] : B[i])));
        if (d < prm->check_tol) {
          memcpy(q, qpos_final + (size_t)(w->final_row + kc) * nq, nq * sizeof(double));
          if (sgn < 0.0) {
            for (int i = 3; i < 7; i++) { q[i] = -q[i]; B[i] = -B[i]; }
            for (int r = 0; r < len; r++)
              for (int i = 3; i < 7; i++) qpos_out[(size_t)(w->frame_begin + k + r) * nq + i] = -qpos_out[(size_t)(w->frame_begin + k + r) * nq + i];
          }
          kc++;
          k += len - 1;
          continue;
        }
        memcpy(B, q, nq * sizeof(double));
        left = len;
      }
      int64_t f = w->frame_begin + k;
      if (w->check_stride == 0 && k == w->n_burn && w->burn_row >= 0 && qpos_final) memcpy(qpos_final + (size_t)w->burn_row * nq, q, nq * sizeof(double));
      gather_frame(pos, quat, in_f64, n_cols, slot_col, ns, f, hp, hq);
      int s = oracle_retarget_frame_ex(m, prm, q, hp, hq, hscale, k == 0 && w->init_row == GMR_INIT_ROOT_TARGET, NULL);
      if (s < 0) fail |= 1;
      if (k >= w->n_burn) {
        memcpy(qpos_out + (size_t)f * nq, q, nq * sizeof(double));
        if (iters_out) iters_out[f] = s;
        out_done++;
        if (w->check_stride > 0 && --left == 0) {
          memcpy(qpos_final + (size_t)(w->final_row + kc) * nq, q, nq * sizeof(double));
          kc++;
        }
      }
    }
    if (frames_done) frames_done[it] = out_done;
    if (w->check_stride == 0 && w->final_row >= 0 && qpos_final) memcpy(qpos_final + (size_t)w->final_row * nq, q, nq * sizeof(double));
  }
  (void)n_threads;
  return fail ? -1 : 0;
}

/* ------------------------------------------------------------------ FK, KinematicsModel convention (float32)
 * KinematicsModel.forward_kinematics (kinematics_model.py:213-246) with
 * torch_utils.quat_mul :117-138, quat_rotate :65-75, axis_angle_to_quat :353-359.
 * xyzw quaternions, raw (un-normalised) XML body quats, float32 arithmetic; the
 * hinge quaternion is built in float64 from a float32 sin/cos exactly as torch's
 * type promotion does (axis is a float64 tensor, kinematics_model.py:133-134).   */
static void quat_mul_xyzw_f(const float a[4], const float b[4], float o[4]) {
  float x1 = a[0], y1 = a[1], z1 = a[2], w1 = a[3], x2 = b[0], y2 = b[1], z2 = b[2], w2 = b[3];
  float ww = (z1 + x1) * (x2 + y2), yy = (w1 - y1) * (w2 + z2), zz = (w1 + y1) * (w2 - z2);
  float xx = ww + yy + zz, qq = 0.5f * (xx + (z1 - x1) * (x2 - y2));
  o[3] = qq - ww + (z1 - y1) * (y2 - z2);
  o[0] = qq - xx + (x1 + w1) * (x2 + w2);
  o[1] = qq - yy + (w1 - x1) * (y2 + z2);
  o[2] = qq - zz + (z1 + y1) * (w2 - x2);
}
static void quat_rotate_xyzw_f(const float q[4], const float v[3], float o[3]) {
  float w = q[3], k = 2.0f * w * w - 1.0f;
  float cx = q[1] * v[2] - q[2] * v[1], cy = q[2] * v[0] - q[0] * v[2], cz = q[0] * v[1] - q[1] * v[0];
  float d = q[0] * v[0] + q[1] * v[1] + q[2] * v[2];
  o[0] = v[0] * k + cx * w * 2.0f + q[0] * d * 2.0f;
  o[1] = v[1] * k + cy * w * 2.0f + q[1] * d * 2.0f;
  o[2] = v[2] * k + cz * w * 2.0f + q[2] * d * 2.0f;
}
/* Joint.dof_to_rot for a hinge (kinematics_model.py:21-36 -> torch_utils.axis_angle_to_quat :353-359): float32 sin / cos of the
 * half angle, the product with the float64 axis and the renormalisation in float64 (type promotion), rounded to float32. */
static void hinge_quat_f(const double *ax, float ang, float jq[4]) {
  float th = ang / 2.0f;
  double s = (double)sinf(th), cw = (double)cosf(th);
  double q4[4] = {ax[0] * s, ax[1] * s, ax[2] * s, cw};
  double n = sqrt(q4[0] * q4[0] + q4[1] * q4[1] + q4[2] * q4[2] + q4[3] * q4[3]);
  if (n < 1e-9) n = 1e-9;
  for (int i = 0; i < 4; i++) jq[i] = (float)(q4[i] / n);
}
/* fitted_shape (kinematics_model.py:225): local translation x shape[j] in float32; shape is [nb][3] (a per-body scalar is passed
 * repeated three times) or NULL. */
void oracle_fk_kin_shape(const oracle_model *m, const float *root_pos, const float *root_rot_xyzw, const float *dof, const float *shape,
                         int64_t n_frames, float *body_pos, float *body_rot);
void oracle_fk_kin(const oracle_model *m, const float *root_pos, const float *root_rot_xyzw, const float *dof, int64_t n_frames,
                   float *body_pos, float *body_rot) {
  oracle_fk_kin_shape(m, root_pos, root_rot_xyzw, dof, NULL, n_frames, body_pos, body_rot);
}
void oracle_fk_kin_shape(const oracle_model *m, const float *root_pos, const float *root_rot_xyzw, const float *dof, const float *shape,
                         int64_t n_frames, float *body_pos, float *body_rot) {
  int nb = m->h.nbody, ndof = m->h.nq - 7;
  for (int64_t f = 0; f < n_frames; f++) {
    float *P = body_pos + (size_t)f * nb * 3, Rtmp[4 * MAXB], *R = body_rot ? body_rot + (size_t)f * nb * 4 : Rtmp;
    memcpy(P, root_pos + 3 * f, 3 * sizeof(float));
    memcpy(R, root_rot_xyzw + 4 * f, 4 * sizeof(float));
    for (int j = 1; j < nb; j++) {
      int p = m->parent[j];
      float jq[4] = {0, 0, 0, 1};
      if (m->jnt_type[j] == GMR_JNT_HINGE) hinge_quat_f(m->jnt_axis + 3 * j /* unit: normalize(axis) */, dof[(size_t)f * ndof + (m->qpos_adr[j] - 7)], jq);
      float lt[3] = {(float)m->body_pos[3 * j], (float)m->body_pos[3 * j + 1], (float)m->body_pos[3 * j + 2]};
      if (shape)
        for (int i = 0; i < 3; i++) lt[i] = lt[i] * shape[3 * j + i];
      const double *qr = m->body_quat_raw + 4 * j;
      float lr[4] = {(float)qr[1], (float)qr[2], (float)qr[3], (float)qr[0]}, wt[3], t[4];
      quat_rotate_xyzw_f(R + 4 * p, lt, wt);
      for (int i = 0; i < 3; i++) P[3 * j + i] = P[3 * p + i] + wt[i];
      quat_mul_xyzw_f(lr, jq, t);
      quat_mul_xyzw_f(R + 4 * p, t, R + 4 * j);
    }
  }
}

/* KinematicsModel.dof_to_rot (kinematics_model.py:172-182): [T, ndof] -> [T, nb-1, 4] xyzw; bodies without a hinge get the identity. */
void oracle_dof_to_rot(const oracle_model *m, const float *dof, int64_t n_frames, float *joint_rot) {
  int nb = m->h.nbody, ndof = m->h.nq - 7;
  for (int64_t f = 0; f < n_frames; f++)
    for (int j = 1; j < nb; j++) {
      float *o = joint_rot + ((size_t)f * (nb - 1) + (j - 1)) * 4;
      o[0] = o[1] = o[2] = 0.0f; o[3] = 1.0f;
      if (m->jnt_type[j] == GMR_JNT_HINGE) hinge_quat_f(m->jnt_axis + 3 * j, dof[(size_t)f * ndof + (m->qpos_adr[j] - 7)], o);
    }
}
/* KinematicsModel.rot_to_dof (kinematics_model.py:184-197) with Joint.rot_to_dof :38-53 and torch_utils.quat_to_axis_angle :320-341,
 * quat_pos :313-318: [T, nb-1, 4] -> [T, ndof], clamped to the joint limits (float32 tensors of the XML `range`). */
void oracle_rot_to_dof(const oracle_model *m, const float *joint_rot, int64_t n_frames, float *dof) {
  int nb = m->h.nbody, ndof = m->h.nq - 7;
  for (int64_t f = 0; f < n_frames; f++)
    for (int j = 1; j < nb; j++) {
      if (m->jnt_type[j] != GMR_JNT_HINGE) continue;
      const float *q0 = joint_rot + ((size_t)f * (nb - 1) + (j - 1)) * 4;
      float z = q0[3] < 0.0f ? 1.0f : 0.0f, sg = 1.0f - 2.0f * z;
      float q[4] = {sg * q0[0], sg * q0[1], sg * q0[2], sg * q0[3]};
      float len = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
      float ang = 2.0f * atan2f(len, q[3]);
      float ax[3] = {q[0] / len, q[1] / len, q[2] / len};
      if (!(len > 1e-5f)) { ang = 0.0f; ax[0] = ax[1] = 0.0f; ax[2] = 1.0f; }
      /* the reference dots with the RAW axis tensor (float64, not normalised: kinematics_model.py:133-134,47); only the sign is used */
      const double *a = m->jnt_axis + 3 * j;
      double dot = (double)ax[0] * a[0] + (double)ax[1] * a[1] + (double)ax[2] * a[2];
      if (dot < 0.0) ang = -ang;
      float lo = (float)m->jnt_range[2 * j], hi = (float)m->jnt_range[2 * j + 1];
      ang = ang < lo ? lo : ang;
      ang = ang > hi ? hi : ang;
      dof[(size_t)f * ndof + (m->qpos_adr[j] - 7)] = ang;
    }
}
/* KinematicsModel.convert_local_rot_to_global (kinematics_model.py:199-211): [T, nb, 4] -> [T, nb, 4], row 0 is the root rotation. */
void oracle_local_rot_to_global(const oracle_model *m, const float *local_rot, int64_t n_frames, float *global_rot) {
  int nb = m->h.nbody;
  for (int64_t f = 0; f < n_frames; f++) {
    const float *L = local_rot + (size_t)f * nb * 4;
    float *G = global_rot + (size_t)f * nb * 4;
    memcpy(G, L, 4 * sizeof(float));
    for (int j = 1; j < nb; j++) quat_mul_xyzw_f(G + 4 * m->parent[j], L + 4 * j, G + 4 * j);
  }
}

/* raw accessors for tests */
int oracle_nq(const oracle_model *m) { return m->h.nq; }
int oracle_nv(const oracle_model *m) { return m->h.nv; }
int oracle_nbody(const oracle_model *m) { return m->h.nbody; }
double oracle_stage_error(const oracle_model *m, int tab, const double *qpos, const double *tp, const double *tq, double *e_all) {
  double xpos[3 * MAXB], xquat[4 * MAXB];
  oracle_fk_mj(m, qpos, xpos, xquat);
  return stage_error(m, tab, xpos, xquat, tp, tq, e_all);
}
void oracle_build_qp_at(const oracle_model *m, int tab, const gmr_ik_params *prm, const double *qpos, const double *tp, const double *tq,
                        double *H, double *c, double *lo, double *hi) {
  double xpos[3 * MAXB], xquat[4 * MAXB], e[6 * MAXT];
  oracle_fk_mj(m, qpos, xpos, xquat);
  stage_error(m, tab, xpos, xquat, tp, tq, e);
  oracle_build_qp(m, tab, prm, qpos, xpos, xquat, tp, tq, e, H, c, lo, hi);
}
