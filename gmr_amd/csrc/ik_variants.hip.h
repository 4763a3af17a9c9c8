// ik_variants.hip.h -- measurement instruments of ik_kernel.hip.h, compiled into VARIANT builds only (tools/build_variant.sh passes
// -DGMR_IK_VARIANTS); the shipped library never includes this file and its ik_body carries empty macros in their place.
//
//   -DGMR_IK_STAMPS      per-phase cycle shares: s_memtime stamps summed per wavefront, read with gmr_debug_read_stamps (tools/ik_stamps.py)
//   -DGMR_IK_MARKS       phase boundaries as comments in the ISA (tools/isa_regions.py --marks, tools/spill_report.py)
//   -DGMR_DUP_PHASE=p    phase p of every solve runs twice (all phases are idempotent): the launch-time difference to the normal build
//                        is what that phase costs at full occupancy (tools/gpu_lds_by_phase.sh, profiles/r01_v4_phase_throughput.txt)
//
// Closed experiments that used to sit in the kernel body behind more switches -- the mixed-precision assembly (GMR_IK_MIXED), the
// SGPR-reload cost probe (GMR_EXP_READLANE), the register-broadcast Cholesky (GMR_QP_LDS_BCAST=0) -- are in the history up to commit
// d74eaa7; their results are in profiles/experiment_log_r01_r02.md.
#pragma once

#ifdef GMR_IK_STAMPS
#define GMR_STAMP(i) do { const u64 t_ = __builtin_readcyclecounter(); stamp_acc[i] += t_ - stamp_last; stamp_last = t_; } while (0)
#define GMR_STAMP_DECL() u64 stamp_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; u64 stamp_last = __builtin_readcyclecounter()
// slot 15: QP iterations (not cycles); slot 14: solves that end with a non-empty working set
#define GMR_STAMP_QP() do { stamp_acc[15] += (u64)(qit < 0 ? -qit : qit); if constexpr (SQ) stamp_acc[14] += __ballot(sq_own && sq_status != 0) ? 1 : 0; } while (0)
#define GMR_STAMP_FLUSH() do { if (lane == 0 && Lk->dbg) for (int i = 0; i < 16; i++) atomicAdd(Lk->dbg + i, stamp_acc[i]); } while (0)
#else
#ifdef GMR_IK_MARKS
#define GMR_STAMP(i) asm volatile("; gmr-mark " #i)
#else
#define GMR_STAMP(i) do { } while (0)
#endif
#define GMR_STAMP_DECL() do { } while (0)
#define GMR_STAMP_QP() do { } while (0)
#define GMR_STAMP_FLUSH() do { } while (0)
#endif

#ifdef GMR_DUP_PHASE
#define GMR_DUP(p) for (int rep_ = 0, nrep_ = launder(GMR_DUP_PHASE == (p) ? 2 : 1); rep_ < nrep_; ++rep_)
// phase 8, the structured QP (it returns through references: run it on copies); phase 1, the FK that closes a solve
#define GMR_DUP_QP_TWICE() do { if (GMR_DUP_PHASE == 8) { int st2 = sq_status; double x2; (void)box_qp_struct(lane, m.sq_nlimb, sq_own, sq_pad, Hm, s_ci, s_lo, s_hi, st2, x2); asm volatile("" :: "v"(x2)); } } while (0)
#define GMR_DUP_FK_TWICE() do { if (launder(GMR_DUP_PHASE == 1 ? 1 : 0)) fk_phase<GMR_IK_STAGE_TREE != 0>(m, bodyc, nbody, fkrounds, lane, q, xpos, xquat); } while (0)
#else
#define GMR_DUP(p)
#define GMR_DUP_QP_TWICE()
#define GMR_DUP_FK_TWICE()
#endif
