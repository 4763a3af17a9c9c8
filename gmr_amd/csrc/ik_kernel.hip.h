// ik_kernel.hip.h -- the frames-batched two-stage IK solver for gfx950 (CDNA4).
//
// One wavefront (64 lanes) owns one work item: a run of consecutive frames of one clip,
// solved in time order with warm start exactly as the reference's caller loop does
// (`for frame in frames: retargeter.retarget(frame)`, scripts/smplx_to_robot_dataset.py:84-89;
// GeneralMotionRetargeting.retarget, motion_retarget.py:139-185).  Thousands of items run
// concurrently, one per wavefront; nothing is exchanged between them.
//
// Lane roles change per phase (all state that crosses phases lives in LDS):
//   FK            lane = body        pointer jumping up the joint tree: ceil(log2(depth+1)) rounds of lane permutes
//   residual      lane = task        SE3 log of T_body^-1 T_target, |e| by a DPP wave reduction
//   task blocks   lane = task        6x6 "task inertia" B_t = A_t' W^2 A_t and g_t = A_t' W^2 e_t
//   composites    lane = (half,elt)  B^c = sum of B_t below a joint (composite-rigid-body style), two composites per pass
//                                    from a host-made plan of LDS offsets
//   F, c          lane = dof         F_i = B^c_i S_i, c_i = S_i . g^c_i
//   H             lane = pair        H[i][j] = S_j . F_i for every structurally non-zero pair, from a host-made plan
//   box QP        lane = QP row      structured (core + limbs in four 16-lane groups, LDL' with DPP row broadcasts, no LDS)
//                                    or generic dense (rows in VGPRs, LDS-broadcast Cholesky); primal active set with
//                                    warm-started working set
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

#include <type_traits>

#include "../../include/gmr_amd.h"

namespace gmr {

typedef unsigned long long u64;

// Measurement instruments (per-phase cycle stamps, ISA phase marks, the phase-duplication builds of tools/build_variant.sh) live in
// ik_variants.hip.h and exist in variant builds only (-DGMR_IK_VARIANTS); the shipped kernel sees empty macros.
#ifdef GMR_IK_VARIANTS
#include "ik_variants.hip.h"
#else
#define GMR_STAMP(i) do { } while (0)
#define GMR_STAMP_DECL() do { } while (0)
#define GMR_STAMP_QP() do { } while (0)
#define GMR_STAMP_FLUSH() do { } while (0)
#define GMR_DUP(p)
#define GMR_DUP_QP_TWICE()
#define GMR_DUP_FK_TWICE()
#endif
#ifndef GMR_QP_GROUP
#define GMR_QP_GROUP 6
#endif
#ifndef GMR_IK_STAGE_TREE
#define GMR_IK_STAGE_TREE 1  // joint tree staged in LDS per wavefront; 0 = re-read from L2 (saves 3.3 KB LDS for G1)
#endif
#ifndef GMR_IK_WAVES_PER_SIMD
#define GMR_IK_WAVES_PER_SIMD 2  // the structured variants (every registry robot): 2 wavefronts per SIMD at <= 256 registers.  The
#endif                            // generic-QP variants ask for 1: a dense 40-64-entry row of H per lane plus the rest of the kernel does
                                  // not fit 256 registers, and with 512 (256 VGPRs + AGPRs as their overflow) nothing goes to scratch
                                  // memory -- 480 bytes per lane did for NVP 64 (profiles/r03_generic_qp_*)

// Doubles per task / composite block: LL(6) LA(9) AA(6) g(6) = 27 used, the block sums move 28 (14 lanes x 16 bytes), and the
// stride is 30 = 15 sixteen-byte slots, odd, so that the blocks of different composites start on different LDS slots: the F
// phase (every dof lane reads element c of ITS composite) then has no bank conflicts (stride 28 = 14 slots: 91 extra LDS
// cycles per solve on G1).  Stored "by column": element 3k + s, s = 0..2,
// (Round 2's mixed-precision variant -- this assembly in float32: -0.5 % time, 5.4e-5 rad -- lives in the history up to commit d74eaa7,
// profiles/experiment_log_r01_r02.md, round-2 item 7.)
typedef double blk_t;
constexpr int kBT = 30, kBTLanes = 14;
// k: 0 LL(s,s)  1 LL(s,s+1)  2..4 LA(s,s), LA(s,s+1), LA(s,s+2)  5 AA(s,s)  6 AA(s,s+1)  7 gl_s  8 ga_s   (indices mod 3)
// -- the order in which three lanes per task (one per column s) produce it in task_block_quad
__host__ __device__ constexpr int bt_ll(int i, int j) { return i == j ? i : ((j - i + 3) % 3 == 1 ? 3 + i : 3 + j); }
__host__ __device__ constexpr int bt_la(int i, int j) { return 3 * (2 + (j - i + 3) % 3) + i; }
__host__ __device__ constexpr int bt_aa(int i, int j) { return i == j ? 15 + i : ((j - i + 3) % 3 == 1 ? 18 + i : 18 + j); }
constexpr int kHPlanRegsSQ = 6;  // rounds of the H pair plan held in registers (even); structured back end only -- the generic one has no registers to spare
constexpr int kCompRegsSQ = 3;   // passes of the composite plan whose addresses stay in registers for a whole stage (ditto)
constexpr int kMaxCompPass = 32;  // passes of the composite plan (two composites per pass, children before parents)
constexpr double kLieEps = 1e-10;  // mink.lie.utils.get_epsilon(float64)

// A compiled model as the kernels see it: ONE struct of fixed-capacity arrays in device memory (filled by api.hip), reached
// through a single kernel-argument pointer.  Every table is base + compile-time offset, so the model costs two SGPRs instead
// of two per table (with ~45 tables passed by value the kernel spilled hundreds of SGPRs into VGPR lanes).
constexpr int kMaxPairsPadded = (GMR_MAX_BODIES * (GMR_MAX_BODIES - 1) / 2 + 127) / 128 * 128;
struct DevModel {
  // root_tslot: slot whose prepared target the root body tracks (-1 = none); same_tasks: both tables used, same (body, slot) per task
  int nbody, nq, nv, nslot, root_slot, n_act, root_tslot, same_tasks;
  int ntask[2], use_table[2], ncomp[2], ncpass[2];  // ncpass: composite passes per table
  int npairp, fkrounds, sq_ok, sq_nlimb;              // npairp: entries of hplan (a multiple of 128)
  int root_planar, pad_[3];                           // 1: planar base (gmr_blob.h root_dof_mask 0x23): rows 2-4 of the QP are null dofs (akind 7)
  // per active dof [64]
  int abody[64], akind[64], aqadr[64], alimited[64];  // akind: 0..2 root translation, 3..5 root rotation, 6 hinge, 7 null (a root
                                                      // dof the model lacks: takes the hinge path with the root's zero axis)
  double arange[128];                                 // [64][2]
  int acomp[2 * 64];                                  // composite node of the dof per table
  // per task, table-major [2][GMR_MAX_TASKS]
  int tbody[2 * GMR_MAX_TASKS], tslot[2 * GMR_MAX_TASKS];
  double twp[2 * GMR_MAX_TASKS], twr[2 * GMR_MAX_TASKS];
  // per body [nbody]
  int jtype[GMR_MAX_BODIES], qadr[GMR_MAX_BODIES];
  double bpos[3 * GMR_MAX_BODIES], bquat[4 * GMR_MAX_BODIES], axis[3 * GMR_MAX_BODIES];  // bquat: unit wxyz
  u64 fkanc[GMR_MAX_BODIES];                          // byte r = ancestor folded in FK round r (0xff: none)
  double qpos0[GMR_MAX_BODIES + 8];                   // [nq]
  // per slot [nslot]
  double sscale[GMR_MAX_SLOTS], spoff[3 * GMR_MAX_SLOTS], sroff[4 * GMR_MAX_SLOTS];
  int sfoot[GMR_MAX_SLOTS];
  // structured QP (box_qp_struct): 4 groups of 16 lanes, each = one bin of limb dofs + a copy of the core dofs
  signed char sq_gdof[64], sq_owner[64];              // dof of a structured lane (-1 padding); 1 if the lane owns that dof
  int sq_lane_of_dof[64], sq_diag[64];                // per dof: its owner lane; LDS index of its diagonal entry
  // H assembly plan, one entry per structurally non-zero off-diagonal pair (dof j strictly above dof i), padded to a multiple
  // of 64 with entries that land in the dummy slots: LDS byte offsets {S_j | F_i << 16, H[i][j] | H[j][i] << 16}
  uint2 hplan[kMaxPairsPadded];
  // composite plan per table, pass and quarter-wave: LDS byte offsets of four source blocks and the destination block
  // {s0|s1<<16, s2|s3<<16, dst, -}; absent sources point at the zero block, an idle quarter at a scratch block
  uint4 comp_plan[2 * kMaxCompPass * 4];
};

struct LdsLayout {
  // offsets in doubles.  zero: block of zeros; hplan / cplan: the staged H-pair and composite plans; Lb (generic QP broadcast
  // rows) aliases S; Bc aliases the poses; H aliases [B | poses / Bc] (all dead during the QP)
  int zero, hplan, cplan, q, tp, tq, S, F, Lb, bodyc, xpos, xquat, B, Bc, H, total_doubles;
};

struct IkLaunch {
  const void *hpos, *hquat;
  const int *slot_col;
  const gmr_work_item *items;
  const int *order;  // [n_items] caller's index of each (length-sorted) item, for frames_done
  int *frames_done;  // [n_items] or NULL
  const double *qinit;
  double *qfinal, *qout;
  int *iters;
  int in_f64, n_cols, n_items, pad;
  gmr_ik_params prm;
  u64 *dbg;  // [16] phase cycle sums, diagnostic builds only
  const int *perm;  // [n_items] or NULL: workgroup b runs item perm[b] (a launch order made on the device, gmr_ik_plan_order)
  int *cost;        // ik_probe_kernel only: [n_items] solves spent on each item, indexed like frames_done
};

// Opaque to the optimiser: values derived from it cannot be hoisted out of the enclosing loop.  Used on indices of
// per-solve table look-ups so that LICM does not turn them into dozens of VGPRs that stay live across the QP.
__device__ __forceinline__ int launder(int v) {
  asm volatile("" : "+v"(v));
  return v;
}
// Same for a wave-uniform value (stays in an SGPR): conditions on it are re-evaluated where they are used instead of being
// hoisted out of the loop as a set of long-lived lane masks.
__device__ __forceinline__ int launder_uniform(int v) {
  asm volatile("" : "+s"(v));
  return v;
}

// A float64 constant kept in a scalar register pair at its point of use.  Polynomial coefficients otherwise get hoisted out
// of the solve loop into VGPR pairs (two dozen of them live across every phase) and each Horner step then costs a v_mov_b64
// plus a v_fmac_f64; from an SGPR pair the step is one v_fma_f64.
__device__ __forceinline__ double kc(double c) {
  asm volatile("" : "+s"(c));
  return c;
}
// A compile-time lane mask (the same 32-bit pattern in both halves of the wave) materialised from a literal where it is used,
// not hoisted into scalar registers that then spill to VGPR lanes.
template <unsigned HALF>
__device__ __forceinline__ unsigned long long kmask() {
  unsigned h;
  asm volatile("s_mov_b32 %0, %1" : "=s"(h) : "i"(HALF));
  return ((unsigned long long)h << 32) | h;
}
// LDS reads of the right width.  Left alone, the compiler pairs neighbouring 8-byte LDS reads into ds_read2_b64, which the
// LDS serves at half the rate of ds_read_b64 / ds_read_b128 (8 cycles per wave for 16 bytes per lane against 2 for 8 and 4 for
// 16, MI355X_MICROARCH "LDS").  lds1: one ds_read_b64 that is never paired (volatile); lds2: one ds_read_b128 (16-byte aligned).
typedef const volatile __attribute__((address_space(3))) double lds_volatile_double;
__device__ __forceinline__ double lds1(const double *p) { return *(lds_volatile_double *)p; }  // p must point into LDS
__device__ __forceinline__ double2 lds2(const double *p) { return *reinterpret_cast<const double2 *>(__builtin_assume_aligned(p, 16)); }

// ------------------------------------------------------------------ wave helpers (wave = 64)
__device__ __forceinline__ double rdlane(double v, int lane) {  // lane: wave-uniform
  int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
// One DPP hop of a 64-bit value: lanes without a source keep `ident` (row_shr / row_bcast, bound_ctrl off).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_hop(double v, double ident) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(ident), __double2loint(v), CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(ident), __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
// Wave reductions in the VALU (6 DPP hops, no LDS round trips); result broadcast from lane 63.
// Lanes without a source keep `old`: the value itself for min / max (idempotent), so no identity has to be materialised.
#define GMR_WAVE_REDUCE(NAME, OP)                                       \
  __device__ __forceinline__ double NAME(double v) {                    \
    v = OP(v, dpp_hop<0x111, 0xf>(v, v)); /* row_shr:1 */               \
    v = OP(v, dpp_hop<0x112, 0xf>(v, v)); /* row_shr:2 */               \
    v = OP(v, dpp_hop<0x114, 0xf>(v, v)); /* row_shr:4 */               \
    v = OP(v, dpp_hop<0x118, 0xf>(v, v)); /* row_shr:8 */               \
    v = OP(v, dpp_hop<0x142, 0xa>(v, v)); /* row_bcast:15 -> rows 1,3 */ \
    v = OP(v, dpp_hop<0x143, 0xc>(v, v)); /* row_bcast:31 -> rows 2,3 */ \
    return rdlane(v, 63);                                               \
  }
__device__ __forceinline__ double op_min(double a, double b) { return fmin(a, b); }
__device__ __forceinline__ double op_max(double a, double b) { return fmax(a, b); }
GMR_WAVE_REDUCE(wave_min, op_min)
GMR_WAVE_REDUCE(wave_max, op_max)
// Sum: bound_ctrl makes a missing source read as 0 and every row is written, so there is no `old` operand at all.  Rows that
// receive an extra partial sum (row 2) are never read: only lane 63 is.
template <int CTRL>
__device__ __forceinline__ double dpp_hop0(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
// Two sums at once for the quad layout of the task lanes (lane 4t + 3 carries a_t, lane 4t + 2 carries b_t): stride-4 hops
// leave the row sums of a in lane 15 and of b in lane 14 of each row; seven hops instead of twelve for two wave_sum.
__device__ __forceinline__ void quad_sum2(double v, double &sa, double &sb) {
  v += dpp_hop0<0x114>(v);  // row_shr:4
  v += dpp_hop0<0x118>(v);  // row_shr:8
  double w = dpp_hop0<0x111>(v);  // row_shr:1: lane 15 <- lane 14
  v += dpp_hop0<0x142>(v);
  w += dpp_hop0<0x142>(w);
  v += dpp_hop0<0x143>(v);
  w += dpp_hop0<0x143>(w);
  sa = rdlane(v, 63);
  sb = rdlane(w, 63);
}
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_hop0<0x111>(v);  // row_shr:1
  v += dpp_hop0<0x112>(v);  // row_shr:2
  v += dpp_hop0<0x114>(v);  // row_shr:4
  v += dpp_hop0<0x118>(v);  // row_shr:8: lane 15 of each row holds the row sum
  v += dpp_hop0<0x142>(v);  // row_bcast:15: row r += sum of row r-1
  v += dpp_hop0<0x143>(v);  // row_bcast:31: rows 2,3 += lane 31 = rows 0+1  -> lane 63 = total
  return rdlane(v, 63);
}

// ------------------------------------------------------------------ lean float64 math (<= ~1 ulp, no slow paths)
// The v_rcp_f64 / v_rsq_f64 seeds are good to 4.6e-8 / 5.2e-8 (measured on gfx950, tools/rcp_acc.hip), so a
// third-order step (error^3) reaches the rounding level in one go, and where a residual correction follows anyway
// (sqrt, div) a second-order step in front of it is enough.
__device__ __forceinline__ double rsqrt_step2(double x) {  // ~4e-15
  const double r = __builtin_amdgcn_rsq(x);
  return r * fma(-0.5 * x * r, r, 1.5);
}
__device__ __forceinline__ double fast_rsqrt(double x) {
  const double r = __builtin_amdgcn_rsq(x);
  const double e = fma(-x * r, r, 1.0);  // 1 - x r^2;  x^-1/2 = r (1 - e)^-1/2 = r (1 + e/2 + 3 e^2 / 8 + ...)
  return fma(r, e * fma(0.375, e, 0.5), r);
}
__device__ __forceinline__ double fast_sqrt(double x) {  // x >= 0
  const double r = rsqrt_step2(x);
  double s = x * r;
  s = fma(0.5 * r, fma(-s, s, x), s);
  return x > 0.0 ? s : 0.0;
}
__device__ __forceinline__ double fast_rcp(double x) {
  const double r = __builtin_amdgcn_rcp(x);
  const double e = fma(-x, r, 1.0);  // 1/x = r / (1 - e) = r (1 + e + e^2 + ...)
  return fma(r, fma(e, e, e), r);
}
__device__ __forceinline__ double fast_div(double a, double b) {
  double r = __builtin_amdgcn_rcp(b);
  r = fma(r, fma(-b, r, 1.0), r);  // ~2e-15, squared by the correction below
  const double q = a * r;
  return fma(r, fma(-q, b, a), q);
}
// sin and cos for |x| <~ 4 (joint half-angles, |dw|/2): Cody-Waite by pi/2, fdlibm kernels.
__device__ __forceinline__ void sincos_small(double x, double *sn, double *cs) {
  const double k = rint(x * 6.36619772367581382433e-01);
  double r = fma(-k, 1.57079632673412561417e+00, x);
  r = fma(-k, 6.07710050630396597660e-11, r);
  r = fma(-k, 2.02226624879595063154e-21, r);
  const double z = r * r;
  const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, kc(1.58969099521155010221e-10), kc(-2.50507602534068634195e-08)), kc(2.75573137070700676789e-06)),
                                       kc(-1.98412698298579493134e-04)), kc(8.33333333332248946124e-03)), kc(-1.66666666666666324348e-01));
  const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, kc(-1.13596475577881948265e-11), kc(2.08757232129817482790e-09)), kc(-2.75573143513906633035e-07)),
                                       kc(2.48015872894767294178e-05)), kc(-1.38888888888741095749e-03)), kc(4.16666666666666019037e-02));
  const double s = fma(r * z, ps, r);
  const double c = fma(z * z, pc, fma(-0.5, z, 1.0));
  const int n = (int)k & 3;
  const double s1 = (n & 1) ? c : s, c1 = (n & 1) ? s : c;
  *sn = (n & 2) ? -s1 : s1;
  *cs = ((n + 1) & 2) ? -c1 : c1;
}
// sin and cos of the half-angles of one FK call.  When no lane's argument exceeds 1.6 in magnitude (every joint angle within
// +-3.2 rad: the config robots inside their limits; decided per call with one compare and a ballot, so any state is handled)
// the argument is halved once more, the fdlibm kernels run without range reduction or quadrant selection on |x/2| <= 0.8
// (max error 3.8e-16 there), and one double-angle step recovers sin x, cos x: 23 operations instead of 45.
__device__ __forceinline__ void sincos_fk(double x, double *sn, double *cs) {
  if (__ballot(fabs(x) > 1.6) == 0) {  // wave-uniform
    const double h = 0.5 * x, z = h * h;
    const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, kc(1.58969099521155010221e-10), kc(-2.50507602534068634195e-08)), kc(2.75573137070700676789e-06)),
                                         kc(-1.98412698298579493134e-04)), kc(8.33333333332248946124e-03)), kc(-1.66666666666666324348e-01));
    const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, kc(-1.13596475577881948265e-11), kc(2.08757232129817482790e-09)), kc(-2.75573143513906633035e-07)),
                                         kc(2.48015872894767294178e-05)), kc(-1.38888888888741095749e-03)), kc(4.16666666666666019037e-02));
    const double s1 = fma(h * z, ps, h);
    const double c1 = fma(z * z, pc, fma(-0.5, z, 1.0));
    *sn = 2.0 * s1 * c1;
    *cs = fma(-2.0 * s1, s1, 1.0);
  } else {
    sincos_small(x, sn, cs);
  }
}
// 2 atan2(y, x) / y for y = sin(phi) >= 0, x = cos(phi) >= 0 of ONE angle (x^2 + y^2 = 1: the vector and scalar norms of a unit
// quaternion), phi in [0, pi/2] -- the factor that turns the quaternion's vector part into the rotation vector (SO3 log).
// Two half-angle steps need no range selection: cos(phi/2) = sqrt((1 + x) / 2), tan(phi/4) = sin(phi/2) / (1 + cos(phi/2)) =
// y / (2 c (1 + c)) <= tan(pi/8) = 0.4142, inside the interval where fdlibm's atan kernel needs no argument reduction, so
// atan2(y, x) = 4 t (1 - s(t^2)) with t = y r / 2, r = 1 / (c (1 + c)), and the quotient by y is 4 r (1 - s): no division by y,
// finite and exact to rounding down to y = 0 (mink's small-angle series 2/w - 2 n^2 / (3 w^3) agrees with it to O(n^4), its
// |w| ~ 0 case pi / n is the same expression at x = 0).
__device__ __forceinline__ double so3_log_factor(double y, double x) {
  const double c = fast_sqrt(fma(0.5, x, 0.5));
  const double r = fast_rcp(c * (1.0 + c));
  const double t = 0.5 * y * r;
  const double z = t * t, w = z * z;
  const double s1 = z * fma(w, fma(w, fma(w, fma(w, fma(w, kc(1.62858201153657823623e-02), kc(4.97687799461593236017e-02)), kc(6.66107313738753120669e-02)),
                                          kc(9.09088713343650656196e-02)), kc(1.42857142725034663711e-01)), kc(3.33333333333329318027e-01));
  const double s2 = w * fma(w, fma(w, fma(w, fma(w, kc(-3.65315727442169155270e-02), kc(-5.83357013379057348645e-02)), kc(-7.69187620504482999495e-02)),
                                   kc(-1.11111104054623557880e-01)), kc(-1.99999999998764832476e-01));
  return 4.0 * r * (1.0 - (s1 + s2));
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
// renormalise a quaternion that is already unit to rounding (product of unit quaternions): one Newton step from r = 1
__device__ __forceinline__ void qrenorm(double q[4]) {
  const double r = fma(-0.5, q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3], 1.5);
  q[0] *= r; q[1] *= r; q[2] *= r; q[3] *= r;
}
__device__ __forceinline__ void q2mat(const double q[4], double R[9]) {  // mju_quat2Mat for a unit quaternion
  // diagonal as 1 - 2 (y^2 + z^2) (= w^2 + x^2 - y^2 - z^2 when |q| = 1), doubled components shared: 21 operations
  const double x2 = q[1] + q[1], y2 = q[2] + q[2], z2 = q[3] + q[3];
  const double xx = x2 * q[1], yy = y2 * q[2], zz = z2 * q[3];
  const double wx = x2 * q[0], wy = y2 * q[0], wz = z2 * q[0];
  R[0] = (1.0 - yy) - zz; R[4] = (1.0 - xx) - zz; R[8] = (1.0 - xx) - yy;
  R[1] = fma(x2, q[2], -wz); R[3] = fma(x2, q[2], wz);
  R[2] = fma(x2, q[3], wy); R[6] = fma(x2, q[3], -wy);
  R[5] = fma(y2, q[3], -wx); R[7] = fma(y2, q[3], wx);
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

// ------------------------------------------------------------------ FK (mj_kinematics), lane = body
// xpos_j = xpos_p + R(xquat_p) pos_j ;  xquat_j = xquat_p (x) quat_j (x) [cos t/2, sin t/2 axis_j]
// evaluated by pointer jumping instead of walking the tree level by level: every lane holds its body's pose
// relative to an ancestor and, each round, composes it with that ancestor's own relative pose (which doubles
// the distance folded), so a chain of depth d needs ceil(log2 d) rounds instead of d.  The ancestor of each
// round is static per body (m.fkanc, one byte per round, 0xff = already in the world frame); its pose is
// fetched from its lane's registers by ds_bpermute.
// The joint tree (per-body pos / quat / axis / joint type / ancestor plan) is staged in LDS once per wavefront.
__device__ __forceinline__ void qrot(const double q[4], const double v[3], double o[3]) {  // R(q) v, q unit
  // v + 2 (w c + qv x c), c = qv x v: 18 operations
  const double cx = q[2] * v[2] - q[3] * v[1], cy = q[3] * v[0] - q[1] * v[2], cz = q[1] * v[1] - q[2] * v[0];
  o[0] = fma(2.0, q[0] * cx + (q[2] * cz - q[3] * cy), v[0]);
  o[1] = fma(2.0, q[0] * cy + (q[3] * cx - q[1] * cz), v[1]);
  o[2] = fma(2.0, q[0] * cz + (q[1] * cy - q[2] * cx), v[2]);
}
constexpr int kBodyC = 11;  // doubles per body in the LDS-staged joint tree: pos(3) quat(4) axis(3) {fkanc(6 bytes), jtype, qadr}
// Stage this lane's body of the joint tree into LDS (once per wavefront).
// The model is always read through the CONSTANT address space (it is read-only for every launch): a group launch gets its model
// pointer from memory, and loads through a generic (flat) pointer count as divergent, loads through a global pointer that is not a
// `restrict` kernel argument as clobberable -- either way every wave-uniform branch, mask and scalar constant derived from the
// model would be a VMEM load into VGPRs instead of a scalar load.
using DevModelG = const DevModel __attribute__((address_space(4)));
// HIP's vector classes cannot be copied out of a qualified address space: read those members through a plain pointer (per-lane data).
template <class T>
__device__ __forceinline__ T ld_plain(const T __attribute__((address_space(4))) *p) { return *(const T *)(uintptr_t)p; }

__device__ __forceinline__ void stage_tree(DevModelG &m, int lane, double *bodyc) {
  if (lane < m.nbody) {
    double *bcst = bodyc + kBodyC * lane;
#pragma unroll
    for (int i = 0; i < 3; i++) { bcst[i] = m.bpos[3 * lane + i]; bcst[7 + i] = m.axis[3 * lane + i]; }
#pragma unroll
    for (int i = 0; i < 4; i++) bcst[3 + i] = m.bquat[4 * lane + i];
    bcst[10] = __longlong_as_double((long long)((m.fkanc[lane] & 0x0000ffffffffffffull) | ((u64)(m.jtype[lane] & 0xff) << 48) | ((u64)(m.qadr[lane] & 0xff) << 56)));
  }
}
// bodyc != nullptr: joint tree staged in LDS (default).  nullptr: re-read from the L2-resident model each call -- the
// variant for builds that trade LDS for occupancy (GMR_IK_STAGE_TREE=0).
// STEP: the root quaternion of q is first advanced by the body-frame rotation w (mj_integratePos for the free joint:
// q <- normalize(q (x) exp(w)), written back to q[3..6]) -- the half angle |w|/2 shares the one sincos evaluation with the
// hinges' half angles instead of costing a second, single-lane one in a separate integrate phase.
// FkJump: the first four pointer-jumping rounds of this lane's body, fixed per wavefront -- the ds_bpermute byte address of the
// ancestor's lane and, per round, the mask of lanes that still fold (a scalar pair: EXEC is narrowed without a compare).
struct FkJump {
  int addr[4];
  u64 act[4];
};
template <bool STAGED, bool STEP = false, bool JUMP = false>
__device__ __forceinline__ void fk_phase(DevModelG &m, const double *bodyc, int nbody, int nrounds, int lane, double *q,
                                         double *xpos, double *xquat, double wx = 0.0, double wy = 0.0, double wz = 0.0,
                                         const FkJump *jp = nullptr) {
  const bool has = lane < nbody;
  int jtype, qadr;
  u64 ancs;
  double pos[3], bq[4], ax[3];
  if constexpr (STAGED) {
    const double *bcst = bodyc + kBodyC * (has ? lane : 0);
    const u64 packed = (u64)__double_as_longlong(bcst[10]);
    jtype = (int)((packed >> 48) & 0xff); qadr = (int)(packed >> 56);
    ancs = has ? (packed | 0xffff000000000000ull) : ~0ull;
#pragma unroll
    for (int i = 0; i < 3; i++) { pos[i] = bcst[i]; ax[i] = bcst[7 + i]; }
#pragma unroll
    for (int i = 0; i < 4; i++) bq[i] = bcst[3 + i];
  } else {
    const int b = launder(has ? lane : 0);
    jtype = m.jtype[b]; qadr = m.qadr[b];
    ancs = has ? m.fkanc[b] : ~0ull;
#pragma unroll
    for (int i = 0; i < 3; i++) { pos[i] = m.bpos[3 * b + i]; ax[i] = m.axis[3 * b + i]; }
#pragma unroll
    for (int i = 0; i < 4; i++) bq[i] = m.bquat[4 * b + i];
  }
  // One quaternion product for every lane: bq (x) [c, s axis] for a hinge, bq (x) 1 for a body without a joint, and for the
  // root q_root (x) [c, s w/|w|] (STEP) or q_root (x) 1 -- products with the identity are exact.
  const bool is_free = jtype == GMR_JNT_FREE;
  const double a2 = STEP ? wx * wx + wy * wy + wz * wz : 0.0, ang = STEP ? fast_sqrt(a2) : 0.0;  // wave-uniform
  double half = jtype == GMR_JNT_HINGE ? 0.5 * q[qadr] : 0.0;
  if (is_free) {
    bq[0] = q[3]; bq[1] = q[4]; bq[2] = q[5]; bq[3] = q[6];
    pos[0] = q[0]; pos[1] = q[1]; pos[2] = q[2];
    if constexpr (STEP) {
      const double ia = a2 > 0 ? fast_rcp(ang) : 0.0;
      half = 0.5 * ang;
      ax[0] = wx * ia; ax[1] = wy * ia; ax[2] = wz * ia;
    }
  }
  double s, c, ql[4];
  sincos_fk(half, &s, &c);
  const double jq[4] = {c, s * ax[0], s * ax[1], s * ax[2]};
  qmul(bq, jq, ql);
  if (is_free) {
    qnormalize(ql);  // mj_integratePos normalises the advanced quaternion, mj_kinematics the root's (a second pass changes <= 1 ulp)
    if (STEP && a2 > 0) { q[3] = ql[0]; q[4] = ql[1]; q[5] = ql[2]; q[6] = ql[3]; }
  }
  // lane = body, so "my ancestor's pose" is another lane's registers: fetched through the LDS crossbar (ds_bpermute, no LDS
  // memory, no bank conflicts, no write-then-read round trip per round); only the final poses are stored
  const int nr = nrounds;
  auto fetch = [](int addr4, double v) {
    const int lo = __builtin_amdgcn_ds_bpermute(addr4, __double2loint(v)), hi = __builtin_amdgcn_ds_bpermute(addr4, __double2hiint(v));
    return __hiloint2double(hi, lo);
  };
  auto fold = [&](const double qa[4], const double pa[3]) {  // my pose relative to the ancestor's frame -> relative to ITS reference
    double t[3], qo[4];
    qrot(qa, pos, t);
    qmul(qa, ql, qo);
    pos[0] = pa[0] + t[0]; pos[1] = pa[1] + t[1]; pos[2] = pa[2] + t[2];
    ql[0] = qo[0]; ql[1] = qo[1]; ql[2] = qo[2]; ql[3] = qo[3];
  };
  int r0 = 0;
  if constexpr (JUMP) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (r < nr) {  // wave-uniform
        const int a4 = jp->addr[r];
        const double qa[4] = {fetch(a4, ql[0]), fetch(a4, ql[1]), fetch(a4, ql[2]), fetch(a4, ql[3])};
        const double pa[3] = {fetch(a4, pos[0]), fetch(a4, pos[1]), fetch(a4, pos[2])};
        if (__builtin_amdgcn_inverse_ballot_w64(jp->act[r])) fold(qa, pa);
      }
    }
    r0 = 4;
  }
  for (int r = r0; r < nr; ++r) {
    const int a = (int)((ancs >> (8 * r)) & 0xff);
    const bool act = a != 0xff;
    const int a4 = (act ? a : 0) << 2;
    const double qa[4] = {fetch(a4, ql[0]), fetch(a4, ql[1]), fetch(a4, ql[2]), fetch(a4, ql[3])};
    const double pa[3] = {fetch(a4, pos[0]), fetch(a4, pos[1]), fetch(a4, pos[2])};
    if (act) fold(qa, pa);
  }
  qrenorm(ql);  // mj_kinematics normalises every body's quaternion; a product of a few unit quaternions needs one Newton step
  if (has) {
    xpos[3 * lane] = pos[0]; xpos[3 * lane + 1] = pos[1]; xpos[3 * lane + 2] = pos[2];
    xquat[4 * lane] = ql[0]; xquat[4 * lane + 1] = ql[1]; xquat[4 * lane + 2] = ql[2]; xquat[4 * lane + 3] = ql[3];
  }
  __syncthreads();
}

// ------------------------------------------------------------------ residual, lane = task
// e = Log(T_wb^-1 T_wt) in [v; w] order (mink FrameTask.compute_error, via motion_retarget.py:188-200).
// Returns this lane's |e|^2; e and the two scalars (kap, bet) of Jl^-1(e) stay in registers for the assembly.
// The half-angle sine/cosine are the relative quaternion's own |v| and |w|, so no trig beyond one atan2.
// rs / xb (optional): row `s` of R(xquat_b) and R' xpos_b, which task_block_quad needs again for the same pose.
__device__ __forceinline__ double task_residual(int body, int slot, const double *xpos, const double *xquat, const double *tp,
                                                const double *tq, double e[6], double &kap, double &bet, int s = 0,
                                                double *rs = nullptr, double *xb = nullptr) {
  const double qb[4] = {xquat[4 * body], xquat[4 * body + 1], xquat[4 * body + 2], xquat[4 * body + 3]};
  const double qt[4] = {tq[4 * slot], tq[4 * slot + 1], tq[4 * slot + 2], tq[4 * slot + 3]};
  const double qc[4] = {qb[0], -qb[1], -qb[2], -qb[3]};
  double qr[4], R[9], d[3], t[3];
  qmul(qc, qt, qr);
  q2mat(qb, R);
  const double xw[3] = {xpos[3 * body], xpos[3 * body + 1], xpos[3 * body + 2]};
  d[0] = tp[3 * slot] - xw[0];
  d[1] = tp[3 * slot + 1] - xw[1];
  d[2] = tp[3 * slot + 2] - xw[2];
  mtv(R, d, t);
  if (rs) {
    mtv(R, xw, xb);
    rs[0] = s == 0 ? R[0] : (s == 1 ? R[3] : R[6]);
    rs[1] = s == 0 ? R[1] : (s == 1 ? R[4] : R[7]);
    rs[2] = s == 0 ? R[2] : (s == 1 ? R[5] : R[8]);
  }
  // SO3 log, short side (mink.lie.so3.SO3.log).  n = sin and |w| = cos of half the rotation angle; the angle itself is
  // th = |f| n, which saves the square root of |om|^2.
  const double w = qr[0], n2 = qr[1] * qr[1] + qr[2] * qr[2] + qr[3] * qr[3];
  const double n = fast_sqrt(n2), aw = fabs(w);
  // one expression for mink's three cases (n^2 < eps: series; |w| < eps: pi / n; else 2 atan2(n, |w|) / n); the sign is mink's
  const double f = (aw < kLieEps ? (w > 0 ? 1.0 : -1.0) : (w < 0 ? -1.0 : 1.0)) * so3_log_factor(n, aw);
  double c2;
  const double om[3] = {f * qr[1], f * qr[2], f * qr[3]};
  const double th = fabs(f) * n, th2 = th * th;
  kap = 0.0; bet = 0.0;  // the two scalars of Jl^-1 the task block needs (below mink's threshold it uses the identity)
  if (th2 < kLieEps) c2 = 1.0 / 12.0;
  else {
    const double ith = fast_rcp(th), ith2 = ith * ith, in = fast_rcp(n), cot = aw * in;
    c2 = (1.0 - 0.5 * th * cot) * ith2;
    const double delta = 0.25 * th * in * in - 0.5 * cot;
    kap = c2;
    bet = (c2 - 0.5 * delta * ith) * ith2;
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
// LL = wp^2 U'U, LA = wp^2 U'V, AA = wp^2 V'V + wr^2 U'U, g = A_t' W^2 e  -> out[27].
__device__ __forceinline__ void task_block(int body, const double *xpos, const double *xquat, const double e[6], double kap,
                                             double bet, double wp, double wr, double *out) {
  const double *u = e, *ph = e + 3;
  const double th2 = ph[0] * ph[0] + ph[1] * ph[1] + ph[2] * ph[2];
  const double pu = ph[0] * u[0] + ph[1] * u[1] + ph[2] * u[2];
  const bool small = th2 < kLieEps;  // mink SE3.ljacinv returns the identity below this threshold (kap = bet = 0 from the residual)
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
  // symmetric 3x3 products
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = i; j < 3; j++) {
      const double uu = U[i] * U[j] + U[3 + i] * U[3 + j] + U[6 + i] * U[6 + j];
      const double vv = V[i] * V[j] + V[3 + i] * V[3 + j] + V[6 + i] * V[6 + j];
      out[bt_ll(i, j)] = wp2 * uu;
      out[bt_aa(i, j)] = wp2 * vv + wr2 * uu;
    }
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) out[bt_la(i, j)] = wp2 * (U[i] * V[j] + U[3 + i] * V[3 + j] + U[6 + i] * V[6 + j]);
  // g = A_t' W^2 e = -[U'(wp2 e_v) ; V'(wp2 e_v) + U'(wr2 e_w)]
  const double ev[3] = {wp2 * e[0], wp2 * e[1], wp2 * e[2]}, ew[3] = {wr2 * e[3], wr2 * e[4], wr2 * e[5]};
#pragma unroll
  for (int j = 0; j < 3; j++) {
    out[21 + j] = -(U[j] * ev[0] + U[3 + j] * ev[1] + U[6 + j] * ev[2]);
    out[24 + j] = -(V[j] * ev[0] + V[3 + j] * ev[1] + V[6 + j] * ev[2] + U[j] * ew[0] + U[3 + j] * ew[1] + U[6 + j] * ew[2]);
  }
}

// y = B [m; a] for a 6x6 symmetric block [[LL, LA],[LA', AA]] in the layout of kBT
template <class T>
__device__ __forceinline__ void sym6_mul(const T *B, const T m[3], const T a[3], T fl[3], T fa[3]) {
#pragma unroll
  for (int i = 0; i < 3; i++) {
    fl[i] = B[bt_ll(i, 0)] * m[0] + B[bt_ll(i, 1)] * m[1] + B[bt_ll(i, 2)] * m[2] + B[bt_la(i, 0)] * a[0] + B[bt_la(i, 1)] * a[1] + B[bt_la(i, 2)] * a[2];
    fa[i] = B[bt_la(0, i)] * m[0] + B[bt_la(1, i)] * m[1] + B[bt_la(2, i)] * m[2] + B[bt_aa(i, 0)] * a[0] + B[bt_aa(i, 1)] * a[1] + B[bt_aa(i, 2)] * a[2];
  }
}

// ------------------------------------------------------------------ task block on three lanes per task
// Same block as task_block, produced by lanes 4t + s, s = 0..2 (lane 4t + 3 idles): the part that depends on the error only
// (A, Bo) is computed by all three, then lane s builds column s of U = A R' and of V = (Bo - A [R'x]x) R' (R'[x]x = [R'x]x R'
// for a rotation, which removes the cross-column term), fetches the two other columns from its quad neighbours with DPP
// quad_perm, and forms the 9 products that make element 3k + s of the block.  ~250 instructions instead of ~460.
template <int CTRL>
__device__ __forceinline__ double quad_get(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ float quad_get(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ double pick3(double a, double b, double c, int s) { return s == 0 ? a : (s == 1 ? b : c); }
template <class T>
__device__ __forceinline__ void cross_t(const T a[3], const T b[3], T o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
// T = double: the block in float64 (T = float was round 2's mixed-precision experiment).
template <class T>
__device__ __forceinline__ void task_block_quad(int s, const double e64[6], double kap64, double bet64, const double rs64[3],
                                                const double xb64[3], double wp64, double wr64, T *out) {
  const T e[6] = {(T)e64[0], (T)e64[1], (T)e64[2], (T)e64[3], (T)e64[4], (T)e64[5]};
  const T kap = (T)kap64, bet = (T)bet64, wp = (T)wp64, wr = (T)wr64;
  const T rs[3] = {(T)rs64[0], (T)rs64[1], (T)rs64[2]}, xb[3] = {(T)xb64[0], (T)xb64[1], (T)xb64[2]};
  const T *u = e, *ph = e + 3;
  const T th2 = ph[0] * ph[0] + ph[1] * ph[1] + ph[2] * ph[2];
  const T pu = ph[0] * u[0] + ph[1] * u[1] + ph[2] * u[2];
  // mink SE3.ljacinv returns the identity below its threshold: kap = bet = 0 from the residual, and the two skew terms go too
  const T hs = e64[3] * e64[3] + e64[4] * e64[4] + e64[5] * e64[5] < kLieEps ? (T)0 : (T)0.5;
  // A = I - 1/2 [ph]x + kap (ph ph' - th2 I);  Bo = -1/2 [u]x + kap (ph u' + u ph' - 2 pu I) + k2 (ph ph' - th2 I), k2 = -2 bet pu.
  // Only column s of U = A R' and of V = (Bo - A [xb]x) R' is needed here, i.e. A r, Bo r and A (xb x r) with r = row s of R
  // (xb = R' x, the body-frame position of the body origin, and r come from the residual of the same pose): the matrices are
  // applied in vector form, never built.
  const T k2 = (T)-2 * bet * pu, a0 = (T)1 - kap * th2;
  auto dot = [](const T a[3], const T b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
  auto applyA = [&](const T v[3], T pv, T o[3]) {  // A v, pv = ph . v
    T c[3];
    cross_t(ph, v, c);
    const T bv = kap * pv;
#pragma unroll
    for (int i = 0; i < 3; i++) o[i] = a0 * v[i] - hs * c[i] + bv * ph[i];
  };
  const T pr = dot(ph, rs), ur = dot(u, rs);
  T uc[3], vc[3], w[3], aw[3], cu[3];
  applyA(rs, pr, uc);
  cross_t(xb, rs, w);
  applyA(w, dot(ph, w), aw);
  cross_t(u, rs, cu);
  const T c_ph = kap * ur + k2 * pr, c_u = kap * pr, c_r = (T)-2 * kap * pu - k2 * th2;
#pragma unroll
  for (int i = 0; i < 3; i++) vc[i] = c_ph * ph[i] + c_u * u[i] + c_r * rs[i] - hs * cu[i] - aw[i];
  // columns s+1 and s+2 from the quad neighbours (quad_perm [1,2,0,3] = 0xC9 and [2,0,1,3] = 0xD2)
  T u1[3], v1[3], v2[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    u1[i] = quad_get<0xC9>(uc[i]); v1[i] = quad_get<0xC9>(vc[i]);
    v2[i] = quad_get<0xD2>(vc[i]);
  }
  const T wp2 = wp * wp, wr2 = wr * wr;
  const T uu = dot(uc, uc), uu1 = dot(uc, u1);
  const T ev[3] = {wp2 * e[0], wp2 * e[1], wp2 * e[2]}, ew[3] = {wr2 * e[3], wr2 * e[4], wr2 * e[5]};
  if (s < 3) {
    out[s] = wp2 * uu;                                   // LL(s,s)
    out[3 + s] = wp2 * uu1;                              // LL(s,s+1)
    out[6 + s] = wp2 * dot(uc, vc);                      // LA(s,s)
    out[9 + s] = wp2 * dot(uc, v1);                      // LA(s,s+1)
    out[12 + s] = wp2 * dot(uc, v2);                     // LA(s,s+2)
    out[15 + s] = wp2 * dot(vc, vc) + wr2 * uu;          // AA(s,s)
    out[18 + s] = wp2 * dot(vc, v1) + wr2 * uu1;         // AA(s,s+1)
    out[21 + s] = -dot(uc, ev);                          // g = A_t' W^2 e = -[U'(wp2 e_v) ; V'(wp2 e_v) + U'(wr2 e_w)]
    out[24 + s] = -(dot(vc, ev) + dot(uc, ew));
  }
}

// ------------------------------------------------------------------ exact box QP, lane = dof (row of H)
// min 1/2 x'Hx + c'x, lo <= x <= hi.  H (dense, symmetric, NVP x NVP, column-major Hm[j*NVP+i]) sits in
// LDS; each lane pulls its row into VGPRs, the wave factors it in registers (right-looking Cholesky kept
// symmetric so that after step k lane k's registers j>k hold L[j][k], which makes BOTH triangular solves
// broadcast-style), and a primal active-set loop pins/releases bounds until the KKT conditions hold --
// the unique optimum DAQP returns for mink.solve_ik's QP.  `status` (0 free, 1 at lo, 2 at hi, 3 padding)
// persists across calls as the warm-started working set.  Returns the iteration count, negative if capped.
template <int NVP>
__device__ __forceinline__ int box_qp(int lane, int n_act, const double *Hm, double *Lb, double ci, double lo, double hi, int &status,
                                      double &x_out) {
  const bool real_row = lane < n_act;
  const int li = lane < NVP ? lane : 0;  // lanes beyond the matrix shadow row 0: they only ever receive broadcasts
  const int lw = lane < NVP ? lane : NVP;  // ... and write their (unused) column entries to a dummy slot
  double *Yb = Lb + NVP + 2;
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
    lane = launder(lane);  // keeps the 100+ lane-vs-constant masks of the unrolled loops from being hoisted into (spilled) SGPRs
    const u64 fixed = __ballot(status != 0);
    const u64 fixed_real = fixed & real_mask;
    const bool mine_fixed = status != 0;
    // right-hand side: free rows -c_i - sum_{j fixed} H_ij x_j ; fixed rows x_i
    double b = -ci;
    for (u64 mm = fixed_real; mm; mm &= mm - 1) {
      const int j = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(mm));
      b -= Hm[j * NVP + li] * rdlane(x, j);
    }
    b = mine_fixed ? x : b;
    // this lane's row of H, rows/columns of the working set replaced by the identity (straight-line: selects only)
    double R[NVP];
#pragma unroll
    for (int j = 0; j < NVP; j++) R[j] = Hm[j * NVP + li];  // padding rows are already identity in LDS
    if (fixed_real) {  // wave-uniform: only then rows/columns of the working set have to become the identity
#pragma unroll
      for (int j = 0; j < NVP; j++) {
        const bool fj = (fixed >> j) & 1ull;
        R[j] = (fj || mine_fixed) ? (j == lane ? 1.0 : 0.0) : R[j];
      }
    }
    double myinv = 1.0;
    // Right-looking Cholesky kept symmetric.  Column k of L and y_k = (L^-1 b)_k reach every lane through two
    // 64-entry LDS rows (one unconditional ds_write_b64 each per lane, then wave-uniform ds_read_b128 pairs), so the
    // VALU only issues the FMAs.  The forward solve rides along as an extra column (b_i -= l_i y_k).  Lane k uses
    // l_k = sqrt(d) - 1, which turns its (symmetric) trailing row sqrt(d) L[j][k] into L[j][k]: after step k lane
    // k's registers j > k hold column k of L, i.e. row k of L', and the backward solve is broadcast-style too.
#pragma unroll
    for (int k = 0; k < NVP; k++) {
      const double dkk = rdlane(R[k], k);
      const double inv = fast_rsqrt(dkk);
      const double lk = R[k] * inv;
      const double l = lane == k ? dkk * inv - 1.0 : (lane < k ? 0.0 : lk);
      myinv = lane == k ? inv : myinv;
      Lb[lw] = l;
      Yb[lw] = b * inv;
      __syncthreads();
      const double yk = Yb[k];
#pragma unroll
      for (int jj = (k + 1) & ~1; jj < NVP; jj += 2) {
        const double2 v = *reinterpret_cast<const double2 *>(Lb + jj);
        if (jj > k) R[jj] -= l * v.x;
        if (jj + 1 < NVP) R[jj + 1] -= l * v.y;
        if ((jj / 2) % GMR_QP_GROUP == GMR_QP_GROUP - 1) __builtin_amdgcn_sched_barrier(0);  // bound the broadcast values in flight
      }
      b = lane == k ? yk : b - l * yk;
      R[k] = lane > k ? lk : R[k];
    }
    // backward L' z = y (rows above k use L[k][i] = R_i[k])
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
    if (!mine_fixed) {
      const double d = z - x;
      if (z > hi + 1e-14 && d > 0) { a = (hi - x) / d; side = 2; }
      else if (z < lo - 1e-14 && d < 0) { a = (lo - x) / d; side = 1; }
    }
    const double amin = wave_min(a);
    if (amin <= 1.0) {
      const u64 who = __ballot(a == amin);
      const int blk = (int)__builtin_ctzll(who);
      const double al = fmax(amin, 0.0);
      if (!mine_fixed) x += al * (z - x);
      if (lane == blk) { x = side == 2 ? hi : lo; status = side; }
      continue;
    }
    if (!mine_fixed) x = z;
    if (!fixed_real) { ++it; break; }
    // multipliers of the working set: g_j = c_j + sum_i H_ji x_i
    double worst = gtol;
    int rel = -1;
    for (u64 mm = fixed_real; mm; mm &= mm - 1) {
      const int j = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(mm));
      const double hx = lane < NVP ? Hm[j * NVP + li] * x : 0.0;
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

// ------------------------------------------------------------------ structured exact box QP ("core + limbs")
// Same problem and same active-set logic as box_qp, for models whose dofs split into an upward-closed core (root + trunk,
// nc <= 10 dofs) and limbs that only couple through the core (every humanoid in the registry).  Lanes form four groups of 16:
// local rows 0 .. nl-1 (nl = 16 - nc) are the dofs of the limbs binned into the group, rows nl .. 15 a copy of the core.
// Each lane holds its row of the group's local 16 x 16 matrix [limb x limb, limb x core; core x limb, core x core]
// (core x core and the core rhs only in group 0, zeros in the copies).  Eliminating local pivots 0 .. nl-1 in all four groups
// at once leaves, in every copy, minus that group's Schur contribution; one cross-group sum makes every copy the full core
// system, which all groups then factor redundantly, so back-substitution needs no further exchange.  The 16-lane groups are
// DPP rows: pivot diagonal, rhs and column reach the rows below through v_mov_b64_dpp row_newbcast, the cross-group sum is
// two gfx950 permlane swaps -- no LDS and no barrier inside the factorisation.  LDL' without square roots; the pivot lane
// keeps its raw row (what the back-substitution consumes), rows below it clear their column entry so that the
// back-substitution needs no triangle mask.  16 registers per row instead of NVP.
// acc -= bcast_K(src) * u in ONE instruction: gfx950's v_fmac_f64 takes a DPP row_newbcast on its first operand (the only
// float64 ALU op that does; checked on the chip, tools/dpp_fmac_probe.hip).  The source lane must be enabled in EXEC (a disabled
// lane reads as 0).  The compiler does not see a DPP instruction inside inline asm, so the two hazards are covered by hand:
// NOP2 = the broadcast operand was written by the VALU instruction right before (2 wait states).
// The hazard cover is hand-placed around code the compiler schedules, so it is tied to the toolchain it was validated on (parity
// tests + a soak of thousands of bitwise-identical launches, tools/soak.py): ROCm 7.2's clang 22 targeting gfx950.  Another
// compiler has to repeat that validation; -DGMR_ALLOW_UNVALIDATED_TOOLCHAIN acknowledges it.
#if !defined(GMR_ALLOW_UNVALIDATED_TOOLCHAIN) && defined(__clang_major__) && (__clang_major__ != 22 || HIP_VERSION_MAJOR != 7)
#error "fmac_bcast_neg: the DPP hazard cover was validated with ROCm 7.x / clang 22 only (see the comment above); re-run tools/soak.py and tests -m gpu, then build with -DGMR_ALLOW_UNVALIDATED_TOOLCHAIN"
#endif
template <int K, bool NOP2 = false>
__device__ __forceinline__ void fmac_bcast_neg(double &acc, double src, double u) {
  // (volatile: the statement must stay inside the EXEC region it was written in -- the compiler does not know it is a VALU op)
  if constexpr (NOP2) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(u), "n"(K));
  else asm volatile("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(u), "n"(K));
}
// group_bcast whose source may have been written by one of the asm statements above a single instruction earlier
template <int K>
__device__ __forceinline__ double group_bcast_after_asm(double v) {
  double r;
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(v), "n"(K));
  return r;
}
template <int K>
__device__ __forceinline__ double group_bcast(double v) {  // value of lane (lane & 48) + K: one v_mov_b64_dpp row_newbcast, no LDS
  return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + K, 0xf, 0xf, true);  // every lane has a source: no 'old' value to keep
}
// v summed over the four 16-lane groups (same local lane), result in every group: gfx950 v_permlane{16,32}_swap, no LDS.
// permlane16_swap(x, x) leaves {rows 0,0,2,2} in the first operand and {rows 1,1,3,3} in the second; permlane32_swap the halves.
__device__ __forceinline__ double group_sum4(double v) {
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
// Four values at once, each summed over the four 16-lane groups and returned to every group, with the same pairing of the
// addends as group_sum4 (bitwise the same sums).  Swapping two DIFFERENT registers needs no copies, so the reduction runs as a
// reduce-scatter (after two levels row g holds the total of value g) followed by an all-gather: 12 swaps + 6 moves + 3 adds for
// four values instead of 16 swaps + 32 moves + 8 adds.
__device__ __forceinline__ void swap16(double &a, double &b) {  // odd rows of a <-> even rows of b
  const auto rl = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const auto rh = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  a = __hiloint2double((int)rh[0], (int)rl[0]);
  b = __hiloint2double((int)rh[1], (int)rl[1]);
}
__device__ __forceinline__ void swap32(double &a, double &b) {  // upper half of a <-> lower half of b
  const auto rl = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const auto rh = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  a = __hiloint2double((int)rh[0], (int)rl[0]);
  b = __hiloint2double((int)rh[1], (int)rl[1]);
}
__device__ __forceinline__ void group_sum4x4(double &x, double &y, double &p, double &q) {
  swap16(x, y);                // x = rows {x0 y0 x2 y2}, y = {x1 y1 x3 y3}
  swap16(p, q);
  double z = x + y, w = p + q;  // z = {x01 y01 x23 y23}, w = {p01 q01 p23 q23}
  swap32(z, w);                // z = {x01 y01 p01 q01}, w = {x23 y23 p23 q23}
  double t = z + w;            // row g holds the total of value g
  double t2 = t;
  swap32(t, t2);               // t = {X Y X Y}, t2 = {P Q P Q}
  x = t; y = t;
  swap16(x, y);                // x = {X X X X}, y = {Y Y Y Y}
  p = t2; q = t2;
  swap16(p, q);
}
// Two values (same pairing of the addends, so every group ends up with bitwise the same sums): 6 swaps, 4 copies, 2 adds.
__device__ __forceinline__ void group_sum2x4(double &x, double &y) {
  swap16(x, y);      // x = rows {x0 y0 x2 y2}, y = {x1 y1 x3 y3}
  double z = x + y;  // {x01 y01 x23 y23}
  double w = z;
  swap32(z, w);      // z = {x01 y01 x01 y01}, w = {x23 y23 x23 y23}
  const double t = z + w;  // {X Y X Y}
  x = t; y = t;
  swap16(x, y);      // x = {X X X X}, y = {Y Y Y Y}
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {  // f(std::integral_constant<int, I>) for I in [I, N)
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for_down(F &&f) {  // I = N-1 .. I
  if constexpr (I < N) {
    f(std::integral_constant<int, N - 1>{});
    static_for_down<I, N - 1>(f);
  }
}

__device__ __forceinline__ int box_qp_struct(int lane, int nl, bool owner, bool pad, const double *Hs, double ci, double lo,
                                             double hi, int &status, double &x_out) {
  const int a = lane & 15;           // local row
  const int gb = lane & 48;          // first lane of my group
  const bool core_row = a >= nl;
  const bool shadow = !owner && !pad;  // copy of a core dof in groups 1..3: mirrors the owner in group 0 (lane a)
  double x = 0.0;
  if (owner) {
    if (status == 1) x = lo;
    else if (status == 2) x = hi;
    else if (lo > 0.0) { x = lo; status = 1; }
    else if (hi < 0.0) { x = hi; status = 2; }
  }
  constexpr int kMaxIt = 6 * 64 + 16;
  int it = 0;
  for (; it < kMaxIt; ++it) {
    lane = launder(lane);
    double R[16];
#pragma unroll
    for (int b = 0; b < 16; b++) R[b] = lds1(Hs + b * 64 + lane);
    double bb = (owner ? -ci : 0.0);
    if (__ballot(owner && status != 0)) {  // wave-uniform, ~4 % of the solves: a working set exists
      // effective status / value of every structured row (copies follow their owner, padding is pinned at 0); rows / columns
      // of the working set become the identity, their values move to the rhs
      const int st_src = __shfl(status, a);      // group 0's lane with my local index
      const double x_src = __shfl(x, a);
      const int st = pad ? 3 : (shadow ? st_src : status);
      const double xe = pad ? 0.0 : (shadow ? x_src : x);
      const bool fixed_me = st != 0;
      const u64 fixed = __ballot(fixed_me && !pad);
      const unsigned fg = (unsigned)(fixed >> gb) & 0xffffu;  // my group's fixed columns
      static_for<0, 16>([&](auto B) {
        constexpr int b = B;
        const double xb = group_bcast<b>(xe);
        const bool fb = (fg >> b) & 1u;
        if (fb && !fixed_me) bb -= R[b] * xb;
        R[b] = (fb || fixed_me) ? (b == a ? 1.0 : 0.0) : R[b];
      });
      if (fixed_me) bb = xe;
    }
    // ---- elimination of local pivot k in every group; the limb pivots first, then (after the cross-group sum) the core ----
    double myinvd = 1.0;
    // Pivot k of every group at once: d_k and b_k come from the pivot lane, column k (= row k by symmetry) entry by entry from
    // the lanes below it, all by row_newbcast -- no LDS, no barrier.  The steps nest: step k runs with the rows a >= k enabled
    // and narrows EXEC to a > k for its updates and for everything after it, so a step costs one compare instead of two
    // masked regions; rows above the pivot simply stay switched off until the back-substitution.  `myinvd` is overwritten by
    // every step a row still takes part in, the last of which is its own.
    auto elim = [&](auto &&self, auto K) -> void {
      constexpr int k = K;
      if constexpr (k >= 6 && k <= 10) {
        if (k == launder_uniform(nl)) {  // first core pivot: every copy of the core block / rhs <- sum over the four groups.  Only core rows
                                         // (a >= nl) are enabled here, in every group alike; limb columns are already zero in them
          {  // R[k..15] and bb (columns left of the first core pivot are limb columns, zero in the core rows), four or two at a time
            double none = 0.0;
            if constexpr (k == 6) { group_sum4x4(R[6], R[7], R[8], R[9]); group_sum4x4(R[10], R[11], R[12], R[13]); group_sum4x4(R[14], R[15], bb, none); }
            if constexpr (k == 7) { group_sum4x4(R[7], R[8], R[9], R[10]); group_sum4x4(R[11], R[12], R[13], R[14]); group_sum2x4(R[15], bb); }
            if constexpr (k == 8) { group_sum4x4(R[8], R[9], R[10], R[11]); group_sum4x4(R[12], R[13], R[14], R[15]); group_sum2x4(bb, none); }
            if constexpr (k == 9) { group_sum4x4(R[9], R[10], R[11], R[12]); group_sum4x4(R[13], R[14], R[15], bb); }
            if constexpr (k == 10) { group_sum4x4(R[10], R[11], R[12], R[13]); group_sum4x4(R[14], R[15], bb, none); }
          }
        }
      }
      const double ck = R[k];  // this row's entry in the pivot column
      const double invd = fast_rcp(group_bcast_after_asm<k>(ck));
      const double bk = group_bcast<k>(bb);
      myinvd = invd;
      const double u = ck * invd;
      R[k] = 0.0;  // dead from here (also the pivot's own diagonal): the back-substitution needs no triangle mask
      if constexpr (k < 15) {
        // rows below the pivot (their lanes are exactly the sources of the broadcasts inside): local row a > k in each of the
        // four 16-lane groups is a compile-time lane mask, so narrowing EXEC costs one scalar instruction and no compare
        constexpr unsigned below16 = (0xffffu << (k + 1)) & 0xffffu;
        if (__builtin_amdgcn_inverse_ballot_w64(kmask<below16 | (below16 << 16)>())) {
          static_for<k + 1, 16>([&](auto J) {
            constexpr int jj = J;
            fmac_bcast_neg<jj>(R[jj], ck, u);  // R[jj] -= u * (column-k entry of row jj)
          });
          bb -= u * bk;
          self(self, std::integral_constant<int, k + 1>{});
        }
      }
    };
    elim(elim, std::integral_constant<int, 0>{});
    // ---- back-substitution, k = 15 .. 0: x_k = (y_k - sum_{b>k} R_k[b] x_b) / d_k ----
    static_for_down<0, 16>([&](auto K) {  // R[k] is 0 on and left of the diagonal (cleared by the elimination)
      constexpr int k = K;
      fmac_bcast_neg<k, true>(bb, bb * myinvd, R[k]);  // bb -= R[k] * x_k
    });
    const double z = bb * myinvd;
    // ratio test along x -> z over the free variables (owner lanes only).  Usually nothing blocks: decide that with compares
    // and one ballot, and only then pay for the divisions and the wave minimum.
    int side = 0;
    const bool freev = owner && status == 0;
    if (freev) {
      if (z > hi + 1e-14 && z - x > 0) side = 2;
      else if (z < lo - 1e-14 && z - x < 0) side = 1;
    }
    if (__ballot(side != 0)) {  // wave-uniform
      const double al = side ? ((side == 2 ? hi : lo) - x) / (z - x) : 2.0;
      const double amin = wave_min(al);
      const u64 who = __ballot(al == amin);
      const int blk = (int)__builtin_ctzll(who);
      const double ac = fmax(amin, 0.0);
      if (freev) x += ac * (z - x);
      if (lane == blk) { x = side == 2 ? hi : lo; status = side; }
      continue;
    }
    if (freev) x = z;
    const u64 fixed_own = __ballot(owner && status != 0);
    if (!fixed_own) { ++it; break; }
    // multipliers of the working set: g_j = c_j + (H x)_j.  (H x) per structured row with the converged x of every column,
    // then the four copies of a core row are summed (limb rows are complete inside their group).
    double hx = 0.0;
    {
      const double x_src2 = __shfl(x, a);
      const double xe2 = pad ? 0.0 : (shadow ? x_src2 : x);
      static_for<0, 16>([&](auto B) {
        constexpr int b = B;
        hx += Hs[b * 64 + lane] * group_bcast<b>(xe2);
      });
      const double t = group_sum4(hx);
      hx = core_row ? t : hx;
    }
    const double g = ci + hx;
    const double viol = (owner && status != 0) ? (status == 1 ? -g : g) : -INFINITY;
    const double worst = wave_max(viol);
    const double gtol = 1e-10 * (1.0 + wave_max(owner ? fabs(ci) : 0.0));  // only solves with a working set get here (~4 %)
    if (!(worst > gtol)) { ++it; break; }
    const u64 whr = __ballot(viol == worst);
    if (lane == (int)__builtin_ctzll(whr)) status = 0;
  }
  x_out = x;
  return it < kMaxIt ? it : -it;
}

// The launch arguments are re-read where they are used (a scalar load per use, once per frame or per solve) instead of living
// in SGPRs for the whole work item: by-value kernel arguments are loaded once in the prologue, and the ~40 SGPRs they occupy
// across the QP were a third of this kernel's SGPR spills.  The pointer is laundered (opaque to the optimiser) at every use,
// so the loads stay where they are written.  It points into the kernarg segment (ik_kernel) or at the item's entry of a group
// launch in device memory (ik_group_kernel); both are read-only for the launch, hence the constant address space.
using IkLaunchK = const IkLaunch __attribute__((address_space(4)));
__device__ __forceinline__ IkLaunchK *ik_args(IkLaunchK *p) {
  asm volatile("" : "+s"(p));
  return p;
}

// ------------------------------------------------------------------ the kernel body: one work item on one wavefront
// A live session as ONE resident wavefront (SURVEY f-4: "a persistent-kernel single-sequence mode"): instead of a launch and a
// stream synchronisation per frame, the host posts frames in a pinned mailbox and the wavefront -- ik_body<.., LIVE = true>,
// whose frame loop then waits for the mailbox instead of walking an array -- solves them as they arrive, its tree, plans and
// configuration staying in LDS / registers between frames.
//   host:   inputs -> pinned buffer, then box.seq = 2 n + offset_to_ground (release)                   ... spins on box.ack == seq
//   device: spins on box.seq != last (acquire, system scope: also drops stale L1 lines), solves the frame in the pinned buffer,
//           results -> pinned buffer, box.ack = n (release)
// Every wait is bounded: the wavefront leaves -- storing its state for the next one and writing its generation to box.exited_gen,
// so that the host launches a fresh one with its next frame (launches are numbered; "alive" = the latest generation has not
// reported leaving) -- after idle_ticks of the 100 MHz constant clock without a frame, after max_polls polls (a second bound
// that needs no clock), after max_frames frames, or when the host raises box.stop.
struct IkSessionBox {
  unsigned seq, pad3, stop, pad0;           // written by the host
  unsigned ack, exited_gen, pad1, pad2;     // written by the device: last frame answered; generation of the last wavefront that left
};
struct IkLive {
  IkSessionBox *box;
  unsigned long long idle_ticks;
  unsigned max_polls, max_frames;
};

template <int NVP, bool SQ, bool LIVE = false, bool PROBE = false>
__device__ __forceinline__ void ik_body(DevModelG &m, IkLaunchK *Lk, const LdsLayout lay, const int item, const IkLive live = IkLive{}) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x;
  double *q = lds + lay.q, *xpos = lds + lay.xpos, *xquat = lds + lay.xquat, *tp = lds + lay.tp, *tq = lds + lay.tq;
  double *Bt = lds + lay.B, *S = lds + lay.S, *F = lds + lay.F, *Hm = lds + lay.H;
  double *bodyc = lds + (GMR_IK_STAGE_TREE ? lay.bodyc : 0);
  if constexpr (GMR_IK_STAGE_TREE != 0) stage_tree(m, lane, bodyc);
  // Once per wavefront: the zero block (absent sources of the composite plan) and the two phase plans, from the L2-resident
  // model into LDS, where a plan entry costs one short-latency read per pass instead of an L2 round trip.
  if (lane < kBT) lds[lay.zero + lane] = 0.0;
  {
    uint2 *hp = reinterpret_cast<uint2 *>(lds + lay.hplan);
    if (!(SQ && m.npairp <= 64 * kHPlanRegsSQ))  // (a plan that fits the registers has no LDS copy)
      for (int i = lane; i < m.npairp; i += 64) hp[i] = ld_plain(&m.hplan[i]);
    uint4 *cp = reinterpret_cast<uint4 *>(lds + lay.cplan);
    const int n0 = 4 * m.ncpass[0], n1 = 4 * m.ncpass[1];
    for (int i = lane; i < n0; i += 64) cp[i] = ld_plain(&m.comp_plan[i]);
    for (int i = lane; i < n1; i += 64) cp[n0 + i] = ld_plain(&m.comp_plan[4 * kMaxCompPass + i]);
  }
  const gmr_work_item w = Lk->items[item];
  constexpr int kHPlanRegs = SQ ? kHPlanRegsSQ : 0, kCompRegs = SQ ? kCompRegsSQ : 0;
  uint2 hreg[kHPlanRegs > 0 ? kHPlanRegs : 1];  // this lane's entries of the first rounds of the H pair plan
#pragma unroll
  for (int r = 0; r < kHPlanRegs; ++r) hreg[r] = 64 * r < m.npairp ? ld_plain(&m.hplan[64 * r + lane]) : uint2{0, 0};
  FkJump fkj;  // (lane = body; bodies beyond the tree and finished chains fetch from themselves and do not fold)
  {
    const u64 an = lane < m.nbody ? m.fkanc[lane] : ~0ull;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int a = (int)((an >> (8 * r)) & 0xff);
      fkj.addr[r] = (a != 0xff ? a : lane) << 2;
      fkj.act[r] = __ballot(a != 0xff);
    }
  }
  GMR_STAMP_DECL();
  const int nq = m.nq, n_act = m.n_act, nslot = m.nslot, root_slot = m.root_slot, npairp = m.npairp, nbody = m.nbody, fkrounds = m.fkrounds;
  // active-dof constants of this lane (row of the QP); the rest is re-read where it is used
  const bool real_row = lane < n_act;
  const int arow = real_row ? lane : 0;
  const int a_body = m.abody[arow], a_kind = real_row ? m.akind[arow] : -1, a_qadr = m.aqadr[arow], a_lim = real_row ? m.alimited[arow] : 0;
  const bool is_slot = lane < nslot;
  const int s_col = Lk->slot_col[is_slot ? lane : 0], root_col = Lk->slot_col[root_slot];

  // structured QP: this lane's row in the 4 x 16 layout
  const int sq_g = SQ ? (int)m.sq_gdof[lane] : -1;
  const bool sq_own = SQ ? m.sq_owner[lane] != 0 : false, sq_pad = sq_g < 0;
  const int sq_mydiag = SQ && real_row ? m.sq_diag[lane] : 0;
  const int sq_owner_lane = SQ && real_row ? m.sq_lane_of_dof[lane] : 0;  // the QP lane that owns this lane's dof
  // joint range and qpos address of the dof this lane owns in the structured QP layout, kept in registers: the limits are
  // rebuilt every solve and must not wait for L2.  Unlimited dofs get an infinite range.
  int sq_qadr = 0;
  double sq_rlo = -INFINITY, sq_rhi = INFINITY;
  if (SQ && sq_own) {
    sq_qadr = m.aqadr[sq_g];
    if (m.alimited[sq_g]) { sq_rlo = m.arange[2 * sq_g]; sq_rhi = m.arange[2 * sq_g + 1]; }
  }
  int sq_status = 0;
  for (int i = lane; i < nq; i += 64) q[i] = w.init_row >= 0 ? Lk->qinit[(size_t)w.init_row * nq + i] : m.qpos0[i];
  int status = real_row ? 0 : 3;
  __syncthreads();
  const double hscale = w.height_scale != 0.0 ? w.height_scale : 1.0;  // per-clip human height factor (gmr_blob.h)

  const int nfr = LIVE ? (int)live.max_frames : w.n_burn + w.n_out;
  unsigned live_last = 0, live_seq = 0;
  bool live_otg = false;
  if constexpr (LIVE) live_last = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&live.box->ack, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM));
  bool poses_valid = false;
  // Verification walk (check_stride > 0, gmr_blob.h): the item runs down a clip whose chunks were already solved
  // speculatively.  At every chunk boundary the state is compared with the state B that chunk started its output from: equal
  // -> the chunk's stored frames are what a sequential run would produce, adopt its stored final state F and skip it;
  // different -> solve the chunk here, from the true state.  kc = chunk index, left = frames left in the chunk being solved.
  int out_done = 0, kc = 0, left = 0;
  int cost_acc = 0;  // PROBE: solves spent on this item
  for (int kf = 0; kf < nfr; ++kf) {
    if constexpr (LIVE) {  // wait for the host's next frame (every value made wave-uniform: the loop must not diverge)
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      unsigned polls = 0;
      bool leave = false;
      for (;;) {
        live_seq = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&live.box->seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM));
        if (live_seq != live_last) break;
        const int stop = __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&live.box->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
        if (stop != 0 || ++polls >= live.max_polls || __builtin_amdgcn_s_memrealtime() - t0 > live.idle_ticks) { leave = true; break; }
        __builtin_amdgcn_s_sleep(2);
      }
      if (leave) break;
      live_otg = (live_seq & 1u) != 0;  // bit 0 of the posted word: this frame's offset_to_ground (one PCIe read less than a second field)
    }
    if (w.check_stride > 0 && left == 0) {
      double *B = ik_args(Lk)->qfinal + (size_t)(w.burn_row + kc) * nq;
      // The floating base's quaternion and its negative are one rotation.  The sequential run carries its sign along from qpos0; a
      // speculative chunk took the sign of the target it was started on (key-points from files come in either), so the states are
      // compared up to that sign, and a chunk adopted with the other sign has its stored quaternions turned to the sequence's.
      double dq = 0.0;
      for (int i = lane; i < nq; i += 64) dq += (i >= 3 && i < 7) ? q[i] * B[i] : 0.0;
      const double sgn = wave_sum(dq) < 0.0 ? -1.0 : 1.0;
      double d = 0.0;
      for (int i = lane; i < nq; i += 64) d = fmax(d, fabs(q[i] - ((i >= 3 && i < 7) ? sgn * B[i] : B[i])));
      const int len = min(w.check_stride, nfr - kf);
      if (wave_max(d) < ik_args(Lk)->prm.check_tol) {  // wave-uniform
        const double *Fk = ik_args(Lk)->qfinal + (size_t)(w.final_row + kc) * nq;
        __syncthreads();
        for (int i = lane; i < nq; i += 64) q[i] = (i >= 3 && i < 7) ? sgn * Fk[i] : Fk[i];
        if (sgn < 0.0) {  // (rare: wave-uniform)
          double *qo = ik_args(Lk)->qout + (size_t)(w.frame_begin + kf) * nq;
          for (int e = lane; e < 4 * len; e += 64) { double *p = qo + (size_t)(e >> 2) * nq + 3 + (e & 3); *p = -*p; }
          if (lane >= 3 && lane < 7) B[lane] = -B[lane];  // the state this chunk's output starts from, in the sequence's sign (a changed B row = rows to re-send, distributed.py)
        }
        __syncthreads();
        poses_valid = false;
        ++kc;
        kf += len - 1;
        continue;
      }
      for (int i = lane; i < nq; i += 64) B[i] = q[i];  // this chunk now starts from the state found here
      left = len;
    }
    const int64_t f = LIVE ? (int64_t)0 : w.frame_begin + kf;  // a live frame always sits in row 0 of the pinned buffers
    if (w.check_stride == 0 && kf == w.n_burn && w.burn_row >= 0) {  // state the first output frame starts from
      double *qfin = ik_args(Lk)->qfinal;
      if (qfin)
        for (int i = lane; i < nq; i += 64) qfin[(size_t)w.burn_row * nq + i] = q[i];
    }
    GMR_STAMP(10);
    // ---- target preparation (update_targets: scale_human_data + offset_human_data, table-1 offsets) ----
    {
      double hp[3] = {0, 0, 0}, hq[4] = {1, 0, 0, 0}, rp[3];
      IkLaunchK *La = ik_args(Lk);
      const int64_t base = f * La->n_cols;
      if (La->in_f64) {
        const double *P = (const double *)La->hpos, *Q = (const double *)La->hquat;
#pragma unroll
        for (int i = 0; i < 3; i++) rp[i] = P[(base + root_col) * 3 + i];
        if (is_slot) {
#pragma unroll
          for (int i = 0; i < 3; i++) hp[i] = P[(base + s_col) * 3 + i];
#pragma unroll
          for (int i = 0; i < 4; i++) hq[i] = Q[(base + s_col) * 4 + i];
        }
      } else {
        const float *P = (const float *)La->hpos, *Q = (const float *)La->hquat;
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
        const double s_scale = hscale * m.sscale[lane], root_scale = hscale * m.sscale[root_slot];
        const double s_poff[3] = {m.spoff[3 * lane], m.spoff[3 * lane + 1], m.spoff[3 * lane + 2]};
        const double s_roff[4] = {m.sroff[4 * lane], m.sroff[4 * lane + 1], m.sroff[4 * lane + 2], m.sroff[4 * lane + 3]};
#pragma unroll
        for (int i = 0; i < 3; i++)
          p[i] = (lane == root_slot) ? root_scale * rp[i] : (hp[i] - rp[i]) * s_scale + root_scale * rp[i];
        qnormalize(hq);
        qmul(hq, s_roff, qo);
        qrenorm(qo);
        q2mat(qo, R);
        mv(R, s_poff, g);
#pragma unroll
        for (int i = 0; i < 3; i++) p[i] += g[i];
        if (m.sfoot[lane]) pz = p[2];
      }
      if (LIVE ? live_otg : La->prm.offset_to_ground != 0) {
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
    if (kf == 0 && w.init_row == GMR_INIT_ROOT_TARGET && m.root_tslot >= 0) {  // wave-uniform: speculative chunk start (gmr_blob.h)
      const int rts = m.root_tslot;
      if (m.root_planar) {  // a planar base takes the target's place and heading only
        const double tw = tq[4 * rts], tx = tq[4 * rts + 1], ty = tq[4 * rts + 2], tz = tq[4 * rts + 3];
        // heading (ZYX yaw) of the target as a half-angle pair, without trigonometry: (cy, sy) = (C, S) / |(C, S)|
        const double S = 2.0 * (tw * tz + tx * ty), C = 1.0 - 2.0 * (ty * ty + tz * tz), n2 = S * S + C * C;
        double ch = 1.0, sh = 0.0;
        if (n2 > 1e-24) {
          const double rn = fast_rsqrt(n2), cy = C * rn, sy = S * rn;
          if (cy >= 0.0) { ch = fast_sqrt(0.5 * (1.0 + cy)); sh = fast_div(0.5 * sy, ch); }
          else { sh = fast_sqrt(0.5 * (1.0 - cy)); sh = sy < 0.0 ? -sh : sh; ch = fast_div(0.5 * sy, sh); }
        }
        const int ln = launder(lane);  // (cold path: keep its lane predicates out of the frame loop's hoisted masks)
        if (ln < 2) q[ln] = tp[3 * rts + ln];
        if (ln >= 3 && ln < 7) q[ln] = ln == 3 ? ch : ln == 6 ? sh : 0.0;
      } else {
        const int ln = launder(lane);
        if (ln < 7) q[ln] = ln < 3 ? tp[3 * rts + ln] : tq[4 * rts + ln - 3];
      }
      __syncthreads();
    }
    GMR_STAMP(0);

    int solves = 0, qpflag = 0;
    // residual state of the task lanes: e, the two scalars of Jl^-1, row ts of R and R' x of the task body.  It outlives a stage:
    // when both tables map the same bodies to the same targets (they differ in weights only), stage 2 starts from exactly the
    // residual stage 1 ended with -- same poses, same targets -- and only the weighted sum is formed anew.
    double e[6] = {0, 0, 0, 0, 0, 0}, jl_kap = 0.0, jl_bet = 0.0, t_rs[3] = {0, 0, 0}, t_xb[3] = {0, 0, 0}, sum_r2 = 0.0;
    for (int tab = 0; tab < 2; ++tab) {
      if (!m.use_table[tab]) continue;
      const int nt = m.ntask[tab];
      // task lanes: up to 16 tasks get a quad each (lanes 4t .. 4t+2 share the task block, task_block_quad); more than 16 one lane
      const bool quad = nt <= 16;
      const int tl = quad ? lane >> 2 : lane, ts = quad ? lane & 3 : 0;
      const bool is_task = tl < nt;
      const int trow = tab * GMR_MAX_TASKS + (is_task ? tl : 0);
      const int t_body = m.tbody[trow], t_slot = m.tslot[trow];
      const double t_wp = m.twp[trow], t_wr = m.twr[trow];
      const int a_comp = m.acomp[tab * 64 + arow];
      // composite plan of this table: entry of quarter-wave lane >> 4 per pass = four source block offsets + destination
      // (LDS bytes); this lane moves the 16 bytes at el16 of each block.  The first kCompRegs passes are resolved to addresses here.
      const int np = m.ncpass[tab];
      const unsigned el16 = 16u * (lane & 15);
      const uint4 *cplan_tab = reinterpret_cast<const uint4 *>(lds + lay.cplan) + (tab ? 4 * m.ncpass[0] : 0) + (lane >> 4);
      unsigned cadr[kCompRegs > 0 ? kCompRegs : 1][5];
#pragma unroll
      for (int p = 0; p < kCompRegs; ++p) {
        const uint4 e = p < np ? cplan_tab[4 * p] : uint4{0, 0, 0, 0};
        cadr[p][0] = (e.x & 0xffffu) + el16; cadr[p][1] = (e.x >> 16) + el16;
        cadr[p][2] = (e.y & 0xffffu) + el16; cadr[p][3] = (e.y >> 16) + el16;
        cadr[p][4] = e.z + el16;
      }

      // q has not moved since the FK that closed the previous solve (previous stage or previous frame): the poses in LDS
      // are still those of q, so only the very first stage of a work item evaluates FK at entry.
      if (!poses_valid) fk_phase<GMR_IK_STAGE_TREE != 0>(m, bodyc, nbody, fkrounds, lane, q, xpos, xquat);
      poses_valid = true;
      GMR_STAMP(1);
      // |e|^2 (convergence test) and |W e|^2 (LM damping) of all tasks after one residual evaluation
      auto residual_sums = [&](double &sum_r2, double &sum_mu) {
        double r2 = 0.0, mu = 0.0;
        if (is_task) {
          r2 = task_residual(t_body, t_slot, xpos, xquat, tp, tq, e, jl_kap, jl_bet, ts, t_rs, t_xb);
          mu = t_wp * t_wp * (e[0] * e[0] + e[1] * e[1] + e[2] * e[2]) + t_wr * t_wr * (e[3] * e[3] + e[4] * e[4] + e[5] * e[5]);
        }
        if (quad) quad_sum2(ts == 3 ? r2 : (ts == 2 ? mu : 0.0), sum_r2, sum_mu);
        else { sum_r2 = wave_sum(r2); sum_mu = wave_sum(mu); }
      };
      double sum_mu;
      if (tab == 1 && m.same_tasks) {  // wave-uniform: e, sum_r2 carried over from stage 1; |W e|^2 with this table's weights
        double mu = 0.0, unused;
        if (is_task) mu = t_wp * t_wp * (e[0] * e[0] + e[1] * e[1] + e[2] * e[2]) + t_wr * t_wr * (e[3] * e[3] + e[4] * e[4] + e[5] * e[5]);
        if (quad) quad_sum2(ts == 2 ? mu : 0.0, unused, sum_mu);
        else sum_mu = wave_sum(mu);
      } else {
        residual_sums(sum_r2, sum_mu);
      }
      double curr = fast_sqrt(sum_r2);
      GMR_STAMP(2);
      int num_iter = 0;
      bool first = true;
      for (;;) {
        // ---- per-task 6x6 blocks ----
        GMR_DUP(3) if (is_task) {
          if (quad) task_block_quad<blk_t>(ts, e, jl_kap, jl_bet, t_rs, t_xb, t_wp, t_wr, reinterpret_cast<blk_t *>(Bt + kBT * tl));
          else {
            task_block(t_body, xpos, xquat, e, jl_kap, jl_bet, t_wp, t_wr, Bt + kBT * tl);
          }
        }
        const double diag = ik_args(Lk)->prm.damping + ik_args(Lk)->prm.lm_damping * sum_mu;
        const double lgain_generic = SQ ? 0.0 : ik_args(Lk)->prm.limit_gain;  // (read here, in uniform control flow)
        GMR_STAMP(3);
        // ---- screws S_i (world frame, about the origin) ----
        double Si[6] = {0, 0, 0, 0, 0, 0};
        GMR_DUP(4) if (real_row) {
          // every dof as an axis in its body's frame: the hinge axis, or e_k for the six root dofs (translations are world
          // aligned and carry no moment) -- one code path, selects only at the two ends
          const double qb[4] = {xquat[4 * a_body], xquat[4 * a_body + 1], xquat[4 * a_body + 2], xquat[4 * a_body + 3]};
          const double xb[3] = {xpos[3 * a_body], xpos[3 * a_body + 1], xpos[3 * a_body + 2]};
          double a_axis[3];
          if constexpr (GMR_IK_STAGE_TREE != 0) {  // the staged joint tree carries the axis: no L2 read inside the solve
            const double *bc = bodyc + kBodyC * a_body + 7;
            a_axis[0] = bc[0]; a_axis[1] = bc[1]; a_axis[2] = bc[2];
          } else {
            const int ab = launder(a_body);
            a_axis[0] = m.axis[3 * ab]; a_axis[1] = m.axis[3 * ab + 1]; a_axis[2] = m.axis[3 * ab + 2];
          }
          const bool is_root = a_kind < 6, is_trans = a_kind < 3;
          const int kk = is_trans ? a_kind : a_kind - 3;
#pragma unroll
          for (int i = 0; i < 3; i++) a_axis[i] = is_root ? (kk == i ? 1.0 : 0.0) : a_axis[i];
          double ax[3], mo[3];
          qrot(qb, a_axis, ax);  // R(q) a without forming R
          cross(xb, ax, mo);
#pragma unroll
          for (int i = 0; i < 3; i++) { Si[i] = is_trans ? a_axis[i] : mo[i]; Si[3 + i] = is_trans ? 0.0 : ax[i]; }
          {
            blk_t *So = reinterpret_cast<blk_t *>(S + 6 * lane);  // (mixed: the first half of the dof's 48-byte slot)
#pragma unroll
            for (int k = 0; k < 6; k++) So[k] = (blk_t)Si[k];
          }
        }
        __syncthreads();
        GMR_STAMP(4);
        // ---- composites: Bc[c] = sum of task blocks below the joint ----
        GMR_DUP(5) {  // Composites are built children-first by a host-made plan: each pass, every quarter of the wave (16 lanes, 14 used,
                      // two block elements per lane) sums up to four blocks (task blocks or finished composites) into one
                      // composite.  A quarter's plan entry is four source block offsets (absent ones point at the zero block) and a
                      // destination; no masks, no selects.  Same wave, so LDS program order makes a pass see the previous one's
                      // writes without a barrier.
          if (__builtin_amdgcn_inverse_ballot_w64(kmask<0x3fff3fffu>())) {  // (lane & 15) < kBTLanes
            char *lb = reinterpret_cast<char *>(lds);
            auto pass = [&](unsigned s0, unsigned s1, unsigned s2, unsigned s3, unsigned dst) {
              const double2 v0 = *reinterpret_cast<const double2 *>(lb + s0), v1 = *reinterpret_cast<const double2 *>(lb + s1);
              const double2 v2 = *reinterpret_cast<const double2 *>(lb + s2), v3 = *reinterpret_cast<const double2 *>(lb + s3);
              double2 sum;
              sum.x = (v0.x + v1.x) + (v2.x + v3.x);
              sum.y = (v0.y + v1.y) + (v2.y + v3.y);
              *reinterpret_cast<double2 *>(lb + dst) = sum;
            };
            // the first kCompRegs passes run from addresses resolved at the stage's entry (no plan read, no unpacking)
#pragma unroll
            for (int p = 0; p < kCompRegs; ++p)
              if (p < np) pass(cadr[p][0], cadr[p][1], cadr[p][2], cadr[p][3], cadr[p][4]);
            for (int p = kCompRegs; p < np; ++p) {
              const uint4 cur = cplan_tab[4 * p];
              pass((cur.x & 0xffffu) + el16, (cur.x >> 16) + el16, (cur.y & 0xffffu) + el16, (cur.y >> 16) + el16, (cur.z | cur.w) + el16);
            }
          }
        }
        __syncthreads();
        GMR_STAMP(5);
        double ci = 0.0, lo = -1e30, hi = 1e30, hdiag = 0.0;  // hdiag: S_i . F_i, the undamped diagonal of H
        GMR_DUP(6) if (real_row) {
          double B[kBTLanes * 2];  // the composite block, seven b128 reads per half
#pragma unroll
          for (int c = 0; c < kBTLanes; c++) {
            const double2 v = lds2(Bt + kBT * a_comp + 2 * c);  // a_comp: block index from Bt on (a one-task composite IS the task block)
            B[2 * c] = v.x; B[2 * c + 1] = v.y;
          }
          double Fi[6];
          sym6_mul<double>(B, Si, Si + 3, Fi, Fi + 3);
          hdiag = Si[0] * Fi[0] + Si[1] * Fi[1] + Si[2] * Fi[2] + Si[3] * Fi[3] + Si[4] * Fi[4] + Si[5] * Fi[5];
#pragma unroll
          for (int k = 0; k < 6; k++) F[6 * lane + k] = Fi[k];
          ci = Si[0] * B[21] + Si[1] * B[22] + Si[2] * B[23] + Si[3] * B[24] + Si[4] * B[25] + Si[5] * B[26];
          if (!SQ && a_lim) {  // mink ConfigurationLimit: -gain (q - lower) <= dq <= gain (upper - q)
            const double qv = q[a_qadr];
            const int ar = launder(2 * lane);
            lo = -lgain_generic * (qv - m.arange[ar]);
            hi = lgain_generic * (m.arange[ar + 1] - qv);
          }
        }
        __syncthreads();  // Bc / poses are dead from here: H overwrites them
        GMR_STAMP(6);
        // ---- H into LDS: zero fill, then the structurally non-zero pairs spread over all lanes ----
        // every structurally non-zero off-diagonal pair (i below j): H[i][j] = H[j][i] = S_j . F_i, spread over all lanes by a
        // host-made plan of LDS byte offsets (padding entries land in the dummy slots)
        auto h_pairs = [&]() {
          // Two rounds of 64 pairs at a time (the plan is padded to a multiple of 128): their twelve gathers are in flight together
          // and the two dot-product chains interleave -- per-round branches used to serialise read -> wait -> chain -> write five
          // times over.  Plans of up to kHPlanRegs rounds (G1: 5) sit in registers for the whole work item; longer ones are read
          // from their LDS copy, one iteration ahead.
          char *lb = reinterpret_cast<char *>(lds);
          auto two_rounds = [&](const uint2 ca, const uint2 cb) {
            const double2 *Sa = reinterpret_cast<const double2 *>(lb + (ca.x & 0xffffu)), *Fa = reinterpret_cast<const double2 *>(lb + (ca.x >> 16));
            const double2 *Sb = reinterpret_cast<const double2 *>(lb + (cb.x & 0xffffu)), *Fb = reinterpret_cast<const double2 *>(lb + (cb.x >> 16));
            const double2 a0 = Sa[0], a1 = Sa[1], a2 = Sa[2], f0 = Fa[0], f1 = Fa[1], f2 = Fa[2];
            const double2 b0 = Sb[0], b1 = Sb[1], b2 = Sb[2], g0 = Fb[0], g1 = Fb[1], g2 = Fb[2];
            const double da = a0.x * f0.x + a0.y * f0.y + a1.x * f1.x + a1.y * f1.y + a2.x * f2.x + a2.y * f2.y;
            const double db = b0.x * g0.x + b0.y * g0.y + b1.x * g1.x + b1.y * g1.y + b2.x * g2.x + b2.y * g2.y;
            *reinterpret_cast<double *>(lb + (ca.y & 0xffffu)) = da;
            *reinterpret_cast<double *>(lb + (ca.y >> 16)) = da;
            *reinterpret_cast<double *>(lb + (cb.y & 0xffffu)) = db;
            *reinterpret_cast<double *>(lb + (cb.y >> 16)) = db;
          };
          if (kHPlanRegs > 0 && npairp <= 64 * kHPlanRegs) {  // wave-uniform
#pragma unroll
            for (int it = 0; it < kHPlanRegs / 2; ++it)
              if (128 * it < npairp) two_rounds(hreg[2 * it], hreg[2 * it + 1]);
          } else {
            const uint2 *hp = reinterpret_cast<const uint2 *>(lds + lay.hplan) + lane;
            const int n2 = npairp >> 7;
            uint2 pa = hp[0], pb = hp[64];
            for (int it = 0; it < n2; ++it) {
              const uint2 ca = pa, cb = pb;
              if (it + 1 < n2) { pa = hp[128 * (it + 1)]; pb = hp[128 * (it + 1) + 64]; }
              two_rounds(ca, cb);
            }
          }
        };
        double dq;
        int qit;
        if constexpr (SQ) {
          // structured layout Hs[col * 64 + lane]: every pair lands in the (at most two) rows that carry it
#pragma unroll
          for (int i = 0; i < 8; i++) reinterpret_cast<double2 *>(Hm)[i * 64 + lane] = double2{0.0, 0.0};  // same wave: LDS keeps program order
          GMR_DUP(7) h_pairs();
          if (real_row) Hm[sq_mydiag] = hdiag + diag;
          if (sq_pad) Hm[(lane & 15) * 64 + lane] = 1.0;
          __syncthreads();
          GMR_STAMP(7);
          // c from the dof-indexed lane to the lane that owns the dof in the QP layout (and dq back below): a lane permute,
          // no LDS memory and no barrier
          const double c_in = __shfl(ci, sq_own ? sq_g : 0);
          const double s_ci = sq_own ? c_in : 0.0;
          // mink ConfigurationLimit, evaluated by the lane that owns the dof in the QP layout: -gain (q - lower) <= dq <= gain (upper - q)
          const double qv = q[sq_qadr];
          const double lgain = ik_args(Lk)->prm.limit_gain;
          const double s_lo = fmax(-lgain * (qv - sq_rlo), -1e30), s_hi = fmin(lgain * (sq_rhi - qv), 1e30);
          double xs;
          GMR_DUP_QP_TWICE();
          qit = box_qp_struct(lane, m.sq_nlimb, sq_own, sq_pad, Hm, s_ci, s_lo, s_hi, sq_status, xs);
          const double x_back = __shfl(xs, sq_owner_lane);
          dq = real_row ? x_back : 0.0;
        } else {
#pragma unroll
          for (int i = 0; i < (NVP * NVP + 63) / 64; i++)
            if (i * 64 + lane < NVP * NVP) Hm[i * 64 + lane] = 0.0;  // same wave: LDS keeps program order, no barrier needed
          h_pairs();
          if (lane < NVP) Hm[lane * NVP + lane] = real_row ? hdiag + diag : 1.0;
          __syncthreads();
          GMR_STAMP(7);
          qit = box_qp<NVP>(lane, n_act, Hm, lds + lay.Lb, ci, lo, hi, status, dq);
        }
        if (qit < 0) qpflag = 1;
        GMR_STAMP(8);
        GMR_STAMP_QP();
        // ---- integrate (mj_integratePos): translations and hinges here, the root rotation inside the FK that follows ----
        const double wx = rdlane(dq, 3), wy = rdlane(dq, 4), wz = rdlane(dq, 5);
        if (real_row) {
          if (a_kind < 3) q[a_kind] += dq;
          else if (a_kind == 6) q[a_qadr] += dq;
        }
        __syncthreads();
        GMR_STAMP(9);
        ++solves;
        fk_phase<GMR_IK_STAGE_TREE != 0, true, true>(m, bodyc, nbody, fkrounds, lane, q, xpos, xquat, wx, wy, wz, &fkj);
        GMR_DUP_FK_TWICE();
        GMR_STAMP(1);
        double next = 0.0;
        GMR_DUP(2) {
          residual_sums(sum_r2, sum_mu);
          next = fast_sqrt(sum_r2);
        }
        GMR_STAMP(2);
        if (!first) ++num_iter;
        first = false;
        if (!(curr - next > ik_args(Lk)->prm.tol && num_iter < ik_args(Lk)->prm.max_iter)) break;
        curr = next;
      }
    }
    if constexpr (PROBE) cost_acc += solves;
    if (kf >= w.n_burn) {
      IkLaunchK *Lo = ik_args(Lk);
      double *qout = Lo->qout;
      int *itp = Lo->iters;
      bool bad = false;  // a non-finite coordinate (x - x is 0 exactly for every finite x): reported in bit 31 of the frame's count
      for (int i = lane; i < nq; i += 64) { const double v = q[i]; qout[(size_t)f * nq + i] = v; bad |= !(v - v == 0.0); }
      if (itp) {  // (wave-uniform)
        const int nonfinite = __builtin_amdgcn_ballot_w64(bad) != 0ull;
        if (lane == 0) itp[f] = solves | (qpflag << 30) | (nonfinite << 31);
      }
      ++out_done;
      if (w.check_stride > 0 && --left == 0) {  // a chunk solved here: its final state
        double *Fk = ik_args(Lk)->qfinal + (size_t)(w.final_row + kc) * nq;
        for (int i = lane; i < nq; i += 64) Fk[i] = q[i];
        ++kc;
      }
    }
    __syncthreads();
    if constexpr (LIVE) {  // the result stores of every lane are issued (barrier above): order them before the acknowledgement
      __threadfence_system();
      if (lane == 0) __hip_atomic_store(&live.box->ack, live_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      live_last = live_seq;
    }
  }
  {
    IkLaunchK *Le = ik_args(Lk);
    int *fdone = Le->frames_done;
    if (fdone && lane == 0) fdone[Le->order ? Le->order[item] : item] = out_done;
    if constexpr (PROBE) {
      int *cost = Le->cost;
      if (cost && lane == 0) cost[Le->order ? Le->order[item] : item] = cost_acc;
    }
    double *qfin = Le->qfinal;
    if (w.check_stride == 0 && w.final_row >= 0 && qfin)
      for (int i = lane; i < nq; i += 64) qfin[(size_t)w.final_row * nq + i] = q[i];
  }
  GMR_STAMP_FLUSH();
}

// One model per launch: the launch arguments are the kernel's own (IkLaunch is the second kernel argument, offset 8 behind the
// model pointer in the kernarg segment).
template <int NVP, bool SQ>
__global__ void __launch_bounds__(64, SQ ? GMR_IK_WAVES_PER_SIMD : 1) ik_kernel(const DevModel *__restrict__ mp, IkLaunch L, LdsLayout lay) {
  IkLaunchK *Lk = (IkLaunchK *)((const char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr() + 8);
  const int *perm = Lk->perm;  // a device-made launch order (gmr_ik_solve_ordered), else items run in array order
  const int item = perm ? __builtin_amdgcn_readfirstlane(perm[blockIdx.x]) : (int)blockIdx.x;
  if ((unsigned)item >= (unsigned)Lk->n_items) return;  // (an order that is not a permutation must not reach outside the item array)
  ik_body<NVP, SQ>(*(DevModelG *)mp, Lk, lay, item);
}

// The probe in front of an ordered launch (gmr_ik_plan_order): the first frames of every item, solved for their cost only --
// nothing is written but cost[item], the number of solves they took.
template <int NVP, bool SQ>
__global__ void __launch_bounds__(64, SQ ? GMR_IK_WAVES_PER_SIMD : 1) ik_probe_kernel(const DevModel *__restrict__ mp, IkLaunch L, LdsLayout lay) {
  IkLaunchK *Lk = (IkLaunchK *)((const char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr() + 8);
  ik_body<NVP, SQ, false, true>(*(DevModelG *)mp, Lk, lay, (int)blockIdx.x);
}

// Launch order from the probe: items by predicted cost (probe solves per frame x frames), most expensive first.  One workgroup;
// a counting sort over 4096 cost buckets -- the order inside a bucket is whatever the atomics make it, which is all a
// longest-expected-first launch needs.  frames[i] / probed[i]: frames of item i in total / in the probe.
__global__ void __launch_bounds__(1024) plan_order_kernel(const int *__restrict__ cost, const int *__restrict__ frames, const int *__restrict__ probed,
                                                          int n, int *__restrict__ order_out) {
  __shared__ unsigned hist[4096];
  __shared__ float smax;
  const int t = threadIdx.x;
  auto key = [&](int i) { return probed[i] > 0 ? (float)cost[i] * (float)frames[i] / (float)probed[i] : 0.0f; };
  float mx = 0.0f;
  for (int i = t; i < n; i += 1024) mx = fmaxf(mx, key(i));
  for (int i = t; i < 4096; i += 1024) hist[i] = 0u;
  if (t == 0) smax = 0.0f;
  __syncthreads();
  atomicMax(reinterpret_cast<unsigned *>(&smax), __float_as_uint(mx));  // keys are non-negative: their bit patterns order like the values
  __syncthreads();
  const float scale = smax > 0.0f ? 4095.0f / smax : 0.0f;
  auto bucket = [&](int i) { return 4095 - min(4095, (int)(key(i) * scale)); };  // bucket 0 = the most expensive
  for (int i = t; i < n; i += 1024) atomicAdd(&hist[bucket(i)], 1u);
  __syncthreads();
  if (t == 0) {  // exclusive scan (4096 entries: not worth a parallel scan next to the launches it orders)
    unsigned acc = 0;
    for (int b = 0; b < 4096; ++b) { const unsigned c = hist[b]; hist[b] = acc; acc += c; }
  }
  __syncthreads();
  for (int i = t; i < n; i += 1024) order_out[atomicAdd(&hist[bucket(i)], 1u)] = i;
}

// Several models in ONE launch (BASELINE config 4: heterogeneous trees): every workgroup looks up the entry its work item belongs
// to -- model, launch arguments (its own input / output arrays and columns) and LDS layout -- and runs the same body.  All
// members are built for one kernel variant (the host forces a common NVP; gmr_group_create).
struct IkGroupEntry {
  const DevModel *m;
  IkLaunch L;
  LdsLayout lay;
  int item_base, pad;  // first workgroup of this entry
};
template <int NVP, bool SQ>
__global__ void __launch_bounds__(64, SQ ? GMR_IK_WAVES_PER_SIMD : 1) ik_group_kernel(const IkGroupEntry *__restrict__ entries,
                                                                             const int *__restrict__ block_entry) {
#if defined(__HIP_DEVICE_COMPILE__)  // (the host pass only needs the kernel's stub; address-space-qualified copies do not parse there)
  // (readfirstlane: tell the compiler these are wave-uniform, so that everything derived from them stays in SGPRs)
  const int e = __builtin_amdgcn_readfirstlane(block_entry[blockIdx.x]);
  const uintptr_t ea = (uintptr_t)(entries + e);
  const uintptr_t eu = (uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ea) |
                       ((uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ea >> 32)) << 32);
  const auto *E = (const IkGroupEntry __attribute__((address_space(4))) *)eu;
  LdsLayout lay;  // (field by field: every member a scalar load into SGPRs)
#define GMR_LAY(f) lay.f = E->lay.f;
  GMR_LAY(zero) GMR_LAY(hplan) GMR_LAY(cplan) GMR_LAY(q) GMR_LAY(tp) GMR_LAY(tq) GMR_LAY(S) GMR_LAY(F) GMR_LAY(Lb) GMR_LAY(bodyc)
  GMR_LAY(xpos) GMR_LAY(xquat) GMR_LAY(B) GMR_LAY(Bc) GMR_LAY(H) GMR_LAY(total_doubles)
#undef GMR_LAY
  ik_body<NVP, SQ>(*(DevModelG *)E->m, &E->L, lay, (int)blockIdx.x - E->item_base);
#endif
}

// The resident wavefront of a persistent session (see IkSessionBox above): entries[0] is the session's one-frame work item.
template <int NVP, bool SQ>
__global__ void __launch_bounds__(64, 1) ik_session_kernel(const IkGroupEntry *__restrict__ entries, IkSessionBox *box,
                                                                               unsigned long long idle_ticks, unsigned max_polls,
                                                                               unsigned max_frames, unsigned gen) {
#if defined(__HIP_DEVICE_COMPILE__)
  const auto *E = (const IkGroupEntry __attribute__((address_space(4))) *)(uintptr_t)entries;
  LdsLayout lay;
#define GMR_LAY(f) lay.f = E->lay.f;
  GMR_LAY(zero) GMR_LAY(hplan) GMR_LAY(cplan) GMR_LAY(q) GMR_LAY(tp) GMR_LAY(tq) GMR_LAY(S) GMR_LAY(F) GMR_LAY(Lb) GMR_LAY(bodyc)
  GMR_LAY(xpos) GMR_LAY(xquat) GMR_LAY(B) GMR_LAY(Bc) GMR_LAY(H) GMR_LAY(total_doubles)
#undef GMR_LAY
  ik_body<NVP, SQ, true>(*(DevModelG *)E->m, &E->L, lay, 0, IkLive{box, idle_ticks, max_polls, max_frames});
  __syncthreads();
  __threadfence_system();
  if (threadIdx.x == 0) __hip_atomic_store(&box->exited_gen, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
}

// ------------------------------------------------------------------ evaluation kernel (no solve)
// One wavefront per frame: FK in the MuJoCo convention for a given qpos, and the stage errors against the frame's
// prepared targets -- what `error1()` / `error2()` (motion_retarget.py:188-200) and `configuration.data.xpos` expose
// in the reference.  err_out [N][2] (0 for a table that is not used), xpos_out [N][nbody][3], xquat_out [N][nbody][4]
// wxyz; any output may be NULL.
struct EvalLaunch {
  const void *hpos, *hquat;  // may be NULL when err_out is NULL
  const int *slot_col;
  const double *qpos;        // [N][nq]
  double *err_out, *xpos_out, *xquat_out;
  int in_f64, n_cols, offset_to_ground, pad;
  long long n_frames;
  const double *hscale;      // [N] per-frame factor on the human scale table, or NULL (= 1.0)
  double *task_err_out;      // [N][ntask[0] + ntask[1]][6] per-task Log(T_body^-1 T_target) = [v; w], or NULL
};

__global__ void __launch_bounds__(64) eval_kernel(const DevModel *__restrict__ mp, EvalLaunch L, LdsLayout lay) {
  DevModelG &m = *(DevModelG *)mp;
  extern __shared__ double lds[];
  const int lane = threadIdx.x;
  const long long f = blockIdx.x;
  double *q = lds + lay.q, *xpos = lds + lay.xpos, *xquat = lds + lay.xquat, *tp = lds + lay.tp, *tq = lds + lay.tq;
  const int nq = m.nq, nbody = m.nbody, nslot = m.nslot, root_slot = m.root_slot;
  for (int i = lane; i < nq; i += 64) q[i] = L.qpos[(size_t)f * nq + i];
  __syncthreads();
  fk_phase<false>(m, nullptr, nbody, m.fkrounds, lane, q, xpos, xquat);
  if (L.xpos_out)
    for (int i = lane; i < 3 * nbody; i += 64) L.xpos_out[(size_t)f * 3 * nbody + i] = xpos[i];
  if (L.xquat_out)
    for (int i = lane; i < 4 * nbody; i += 64) L.xquat_out[(size_t)f * 4 * nbody + i] = xquat[i];
  if (!L.err_out && !L.task_err_out) return;
  {  // target preparation, as in ik_kernel
    const bool is_slot = lane < nslot;
    const int s_col = L.slot_col[is_slot ? lane : 0], root_col = L.slot_col[root_slot];
    double hp[3] = {0, 0, 0}, hq[4] = {1, 0, 0, 0}, rp[3];
    const long long base = f * L.n_cols;
    if (L.in_f64) {
      const double *P = (const double *)L.hpos, *Q = (const double *)L.hquat;
      for (int i = 0; i < 3; i++) { rp[i] = P[(base + root_col) * 3 + i]; hp[i] = P[(base + s_col) * 3 + i]; }
      for (int i = 0; i < 4; i++) hq[i] = Q[(base + s_col) * 4 + i];
    } else {
      const float *P = (const float *)L.hpos, *Q = (const float *)L.hquat;
      for (int i = 0; i < 3; i++) { rp[i] = (double)P[(base + root_col) * 3 + i]; hp[i] = (double)P[(base + s_col) * 3 + i]; }
      for (int i = 0; i < 4; i++) hq[i] = (double)Q[(base + s_col) * 4 + i];
    }
    double pz = INFINITY, p[3] = {0, 0, 0}, qo[4] = {1, 0, 0, 0}, R[9], g[3];
    if (is_slot) {
      const double hs = L.hscale ? L.hscale[f] : 1.0;
      const double s_scale = hs * m.sscale[lane], root_scale = hs * m.sscale[root_slot];
      const double s_poff[3] = {m.spoff[3 * lane], m.spoff[3 * lane + 1], m.spoff[3 * lane + 2]};
      const double s_roff[4] = {m.sroff[4 * lane], m.sroff[4 * lane + 1], m.sroff[4 * lane + 2], m.sroff[4 * lane + 3]};
      for (int i = 0; i < 3; i++) p[i] = (lane == root_slot) ? root_scale * rp[i] : (hp[i] - rp[i]) * s_scale + root_scale * rp[i];
      qnormalize(hq);
      qmul(hq, s_roff, qo);
      qrenorm(qo);
      q2mat(qo, R);
      mv(R, s_poff, g);
      for (int i = 0; i < 3; i++) p[i] += g[i];
      if (m.sfoot[lane]) pz = p[2];
    }
    if (L.offset_to_ground) p[2] = p[2] - wave_min(pz) + 0.1;
    if (is_slot) {
      for (int i = 0; i < 3; i++) tp[3 * lane + i] = p[i];
      for (int i = 0; i < 4; i++) tq[4 * lane + i] = qo[i];
    }
  }
  __syncthreads();
  for (int tab = 0; tab < 2; ++tab) {
    double err = 0.0;
    if (m.use_table[tab]) {
      const bool is_task = lane < m.ntask[tab];
      const int trow = tab * GMR_MAX_TASKS + (is_task ? lane : 0);
      double e[6], kap, bet;  // (kap, bet: unused here)
      err = fast_sqrt(wave_sum(is_task ? task_residual(m.tbody[trow], m.tslot[trow], xpos, xquat, tp, tq, e, kap, bet) : 0.0));
      if (L.task_err_out && is_task) {
        double *o = L.task_err_out + ((size_t)f * (m.ntask[0] + m.ntask[1]) + (tab ? m.ntask[0] : 0) + lane) * 6;
        for (int i = 0; i < 6; i++) o[i] = e[i];
      }
    }
    if (lane == 0 && L.err_out) L.err_out[(size_t)f * 2 + tab] = err;
  }
}

}  // namespace gmr
