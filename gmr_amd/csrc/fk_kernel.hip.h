// fk_kernel.hip.h -- batched forward kinematics in the KinematicsModel convention (float32, xyzw).
//
// Replaces KinematicsModel.forward_kinematics (reference kinematics_model.py:213-246) with its helpers
// torch_utils.quat_mul :117-138, quat_rotate :65-75, axis_angle_to_quat :353-359, as called by the
// dataset scripts after IK (scripts/smplx_to_robot_dataset.py:106-123, bvh_to_robot_dataset.py:108-125).
//
// One frame per lane.  The joint tree is wave-uniform, so its constants come in through scalar loads
// (SGPRs); the chain itself runs in VGPRs.  Bodies are in depth-first order, so a body's parent is either
// the previous body (pose still in registers) or an earlier branch point whose pose was parked in a
// per-lane LDS slot (slot lifetimes planned on the host).
// The poses go out through an LDS stage, kFkGroup bodies at a time, as runs of kFkGroup x 12 (16) contiguous
// bytes per frame written by consecutive lanes.  (With one lane storing its own frame directly, every store
// instruction touched 64 different cache lines and the kernel sat at the L2 request rate, 9 % of the HBM
// roofline.)  The dof row of a lane's frame is read by the lane itself, four consecutive angles per 16-byte
// load into a register window that slides down the row one group ahead of its use (hinges are numbered in
// body order, so the chain consumes the row front to back): 8 L2 requests per frame instead of 29, and no
// LDS -- the dof tile of round 1 (116 B per lane for G1) was what held the kernel at 1.5 waves per SIMD, where
// the dependent chain of every body is exposed latency.  LDS per lane is now the branch slots (2 for G1: 56 B)
// and the output stage (100 B): 8 workgroups = 16 waves per CU.
// Algorithmic HBM traffic per frame: (3+4+ndof) x 4 B in, nbody x 12 B out (+ nbody x 16 B if rotations
// are requested).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace gmr {

constexpr int kFkThreads = 128;
#ifndef GMR_FK_GROUP
#define GMR_FK_GROUP 8
#endif
constexpr int kFkGroup = GMR_FK_GROUP;             // bodies per output flush
constexpr int kFkPosStride = 3 * kFkGroup + 1;    // LDS words per lane of the position stage (odd: conflict-free)
constexpr int kFkRotStride = 4 * kFkGroup + 1;
constexpr int kFkMaxSlots = 12;

// Everything the chain needs about one body, in one 80-byte record: read with two scalar loads (s_load_dwordx16 + x4) and
// fetched one body ahead of its use, instead of a dozen dependent scalar loads per body in front of the wave-uniform branches.
struct FkBody {
  int src_slot, dofidx, save_slot, pad0;  // src_slot: -1 = the previous body is the parent; dofidx: -1 = no hinge
  float lpos[3], pad1;                    // local translation
  float lrot[4];                          // local rotation xyzw, raw XML values (not normalised)
  double axis[3], pad2;                   // unit hinge axis in float64 (torch promotes the hinge quaternion to float64)
};
static_assert(sizeof(FkBody) == 80, "FkBody layout");

struct FkTree {     // device arrays, [nbody]
  const int *parent, *dofidx, *src_slot, *save_slot;  // dofidx: -1 if no hinge; src_slot: -1 = previous body
  const float *lpos;   // [nb][3] local translation
  const float *lrot;   // [nb][4] local rotation xyzw, raw XML values (not normalised)
  const float *jaxis;  // [nb][3] hinge axis, unit (double normalised, then rounded)
  const double *jaxis64;  // [nb][3] unit axis in float64 (torch promotes the hinge quaternion to float64)
  const FkBody *body;     // [nb] the same, one record per body (what the kernels read)
  int nbody, ndof, nslots, dof_in_order;  // dof_in_order: dofidx never decreases along the bodies (the register window needs it)
};

__device__ __forceinline__ void fk_quat_mul(const float a[4], const float b[4], float o[4]) {
#pragma clang fp contract(off)
  // torch_utils.quat_mul (xyzw), same operation order
  const float x1 = a[0], y1 = a[1], z1 = a[2], w1 = a[3], x2 = b[0], y2 = b[1], z2 = b[2], w2 = b[3];
  const float ww = (z1 + x1) * (x2 + y2), yy = (w1 - y1) * (w2 + z2), zz = (w1 + y1) * (w2 - z2);
  const float xx = ww + yy + zz, qq = 0.5f * (xx + (z1 - x1) * (x2 - y2));
  o[3] = qq - ww + (z1 - y1) * (y2 - z2);
  o[0] = qq - xx + (x1 + w1) * (x2 + w2);
  o[1] = qq - yy + (w1 - x1) * (y2 + z2);
  o[2] = qq - zz + (z1 + y1) * (w2 - x2);
}
__device__ __forceinline__ void fk_quat_rotate(const float q[4], const float v[3], float o[3]) {
#pragma clang fp contract(off)
  // torch_utils.quat_rotate: v (2w^2-1) + 2w (q x v) + 2 q (q.v)
  const float w = q[3], k = 2.0f * w * w - 1.0f;
  const float cx = q[1] * v[2] - q[2] * v[1], cy = q[2] * v[0] - q[0] * v[2], cz = q[0] * v[1] - q[1] * v[0];
  const float d = q[0] * v[0] + q[1] * v[1] + q[2] * v[2];
  o[0] = v[0] * k + cx * w * 2.0f + q[0] * d * 2.0f;
  o[1] = v[1] * k + cy * w * 2.0f + q[1] * d * 2.0f;
  o[2] = v[2] * k + cz * w * 2.0f + q[2] * d * 2.0f;
}

// MODE 0: write body_pos (and body_rot if non-null).  MODE 1: per-clip min of z (atomics on an ordered-int key).
// Dynamic LDS (floats): [nslots][7][kFkThreads] branch slots |
//                       [kFkThreads][kFkPosStride] position stage | [kFkThreads][kFkRotStride] rotation stage (MODE 0)
// The tree arrays are read-only for the whole launch, but the kernel also stores to global memory, so the compiler will not
// prove them invariant by itself and reads them with per-lane VMEM loads (a full memory round trip in front of every
// wave-uniform branch of the chain).  Reading them through the constant address space makes them scalar loads: SGPR
// results, scalar branches, the scalar cache.
template <class T>
__device__ __forceinline__ T fk_const(const T *p, int i) {
  return (reinterpret_cast<const T __attribute__((address_space(4))) *>(reinterpret_cast<uintptr_t>(p)))[i];
}

// axis_angle_to_quat (torch_utils.py:353-359 through kinematics_model.py:21-36): sin/cos in float32, the product with the
// float64 axis and the renormalisation in float64, the result rounded to float32.
__device__ __forceinline__ void fk_hinge_quat(const double ax[3], float ang, float jq[4]) {
#pragma clang fp contract(off)
  const float th = ang / 2.0f;
  float sf, cf;
  sincosf(th, &sf, &cf);  // one range reduction for both; same values as sinf / cosf
  const double s = (double)sf, c = (double)cf;
  const double qx = ax[0] * s, qy = ax[1] * s, qz = ax[2] * s;
  // / max(|q|, 1e-9) in float64.  |q|^2 = 1 + e with |e| ~ 1e-7 (unit axis, float32 sin^2 + cos^2), so the clamp never binds
  // and 1 / sqrt(1 + e) = 1 - e / 2 + 3 e^2 / 8 - ...: the first Newton step from 1, r = 1.5 - 0.5 |q|^2, is exact to
  // 3 e^2 / 8 < 1e-13 -- far below the float32 rounding of the result.  Anything else takes the general path.
  const double n2 = qx * qx + qy * qy + qz * qz + c * c;
  double r = 1.5 - 0.5 * n2;
  if (__builtin_expect(__any(fabs(n2 - 1.0) > 1e-5), 0)) {
    double rr = __builtin_amdgcn_rsq(n2);
    rr = rr * (1.5 - 0.5 * n2 * rr * rr);
    rr = rr * (1.5 - 0.5 * n2 * rr * rr);
    r = n2 < 1e-18 ? 1e9 : rr;
  }
  jq[0] = (float)(qx * r); jq[1] = (float)(qy * r); jq[2] = (float)(qz * r); jq[3] = (float)(c * r);
}

// One body record through the constant address space (scalar loads).
__device__ __forceinline__ FkBody fk_body(const FkTree &t, int j) {
#if defined(__HIP_DEVICE_COMPILE__)
  const auto *p = reinterpret_cast<const FkBody __attribute__((address_space(4))) *>(reinterpret_cast<uintptr_t>(t.body)) + j;
  FkBody r;
  r.src_slot = p->src_slot; r.dofidx = p->dofidx; r.save_slot = p->save_slot; r.pad0 = 0; r.pad1 = 0.f; r.pad2 = 0.0;
#pragma unroll
  for (int i = 0; i < 3; i++) { r.lpos[i] = p->lpos[i]; r.axis[i] = p->axis[i]; }
#pragma unroll
  for (int i = 0; i < 4; i++) r.lrot[i] = p->lrot[i];
  return r;
#else
  return t.body[j];
#endif
}

template <int MODE>
__global__ void __launch_bounds__(kFkThreads) fk_kernel(FkTree t, const float *__restrict__ root_pos,
                                                        const float *__restrict__ root_rot, const float *__restrict__ dof,
                                                        int64_t n_frames, float *__restrict__ body_pos,
                                                        float *__restrict__ body_rot, const int64_t *__restrict__ seq_offsets,
                                                        int n_seq, int *__restrict__ min_key) {
#pragma clang fp contract(off)  // torch does not fuse; also keeps MODE 0 and MODE 1 bit-identical
  extern __shared__ float fk_lds[];
  const int tid = threadIdx.x;
  const int nbody = t.nbody, ndof = t.ndof;
  const int64_t f0 = (int64_t)blockIdx.x * kFkThreads;
  const int64_t f = f0 + tid;
  const int nfb = (int)(n_frames - f0 < kFkThreads ? n_frames - f0 : kFkThreads);  // frames of this workgroup
  const bool live = tid < nfb;
  const int64_t fc = live ? f : n_frames - 1;  // clamp: dead lanes recompute the last frame, never store
  float *slots = fk_lds;
  float *pstage = slots + (size_t)t.nslots * 7 * kFkThreads;
  float *rstage = pstage + kFkThreads * kFkPosStride;
  // dof row of this lane's frame: a window of four consecutive angles in registers, the next window already in flight
  struct D4 { float v[4]; };  // 16 bytes, 4-byte aligned (rows start on dword boundaries only)
  const float *myrow = dof + fc * ndof;
  const bool windowed = ndof >= 4 && t.dof_in_order;
  const int last_start = ndof - 4;
  D4 dwin{{0.f, 0.f, 0.f, 0.f}}, dnext{{0.f, 0.f, 0.f, 0.f}};
  int win_group = -1;  // group (di >> 2) held by dwin; dnext holds win_group + 1
  if (windowed) dnext = *reinterpret_cast<const D4 *>(myrow);
  float cp[3], cr[4];
#pragma unroll
  for (int i = 0; i < 3; i++) cp[i] = root_pos[fc * 3 + i];
#pragma unroll
  for (int i = 0; i < 4; i++) cr[i] = root_rot[fc * 4 + i];
  float zmin = cp[2];
  if (fk_const(t.save_slot, 0) >= 0) {
    float *s = slots + (size_t)fk_const(t.save_slot, 0) * 7 * kFkThreads + tid;
#pragma unroll
    for (int i = 0; i < 3; i++) s[i * kFkThreads] = cp[i];
#pragma unroll
    for (int i = 0; i < 4; i++) s[(3 + i) * kFkThreads] = cr[i];
  }
  const bool want_rot = MODE == 0 && body_rot != nullptr;
  // flush the stage: bodies j0 .. j0+cnt-1 of every frame of the workgroup, consecutive lanes -> consecutive words
  struct F3 { float x, y, z; };  // 12 bytes, 4-byte aligned: one global_store_dwordx3 per body
  auto flush_n = [&](int j0, const auto cnt) {  // cnt: int, or an integral constant so that the index split is a shift
    __syncthreads();
    {
      const int n = nfb * cnt;  // one (frame, body) item per lane and store
      for (int i = tid; i < n; i += kFkThreads) {
        const int fr = i / cnt, b = i - fr * cnt;
        const float *sp = pstage + fr * kFkPosStride + 3 * b;
        *reinterpret_cast<F3 *>(body_pos + ((f0 + fr) * nbody + j0 + b) * 3) = F3{sp[0], sp[1], sp[2]};
      }
    }
    if (want_rot) {
      const int n = nfb * cnt;
      for (int i = tid; i < n; i += kFkThreads) {
        const int fr = i / cnt, b = i - fr * cnt;
        const float *sp = rstage + fr * kFkRotStride + 4 * b;
        *reinterpret_cast<float4 *>(body_rot + ((f0 + fr) * nbody + j0 + b) * 4) = make_float4(sp[0], sp[1], sp[2], sp[3]);
      }
    }
    __syncthreads();
  };
  auto stage = [&](int k) {  // pose of the current body into stage column k
#pragma unroll
    for (int i = 0; i < 3; i++) pstage[tid * kFkPosStride + 3 * k + i] = cp[i];
    if (want_rot) {
#pragma unroll
      for (int i = 0; i < 4; i++) rstage[tid * kFkRotStride + 4 * k + i] = cr[i];
    }
  };
  if (MODE == 0) stage(0);
  FkBody nxt = fk_body(t, nbody > 1 ? 1 : 0);
  for (int j = 1; j < nbody; ++j) {
    const FkBody rec = nxt;
    nxt = fk_body(t, j + 1 < nbody ? j + 1 : j);  // one body ahead
    if (MODE == 0 && (j % kFkGroup) == 0) flush_n(j - kFkGroup, std::integral_constant<int, kFkGroup>{});
    float pp[3], pr[4];
    const int src = rec.src_slot;
    if (src < 0) {
#pragma unroll
      for (int i = 0; i < 3; i++) pp[i] = cp[i];
#pragma unroll
      for (int i = 0; i < 4; i++) pr[i] = cr[i];
    } else {
      const float *s = slots + (size_t)src * 7 * kFkThreads + tid;
#pragma unroll
      for (int i = 0; i < 3; i++) pp[i] = s[i * kFkThreads];
#pragma unroll
      for (int i = 0; i < 4; i++) pr[i] = s[(3 + i) * kFkThreads];
    }
    float jq[4] = {0.f, 0.f, 0.f, 1.f};
    const int di = rec.dofidx;
    if (di >= 0) {
      float ang;
      if (windowed) {  // (all conditions wave-uniform: the tree is)
        const int g = di >> 2;
        while (win_group < g) {  // slide: hinges come in row order, so this advances by one group at a time
          dwin = dnext;
          ++win_group;
          const int nxt = 4 * (win_group + 1);
          if (nxt < ndof) dnext = *reinterpret_cast<const D4 *>(myrow + (nxt < last_start ? nxt : last_start));
        }
        const int start = 4 * g < last_start ? 4 * g : last_start;
        const int k = di - start;
        ang = k == 0 ? dwin.v[0] : k == 1 ? dwin.v[1] : k == 2 ? dwin.v[2] : dwin.v[3];
      } else {
        ang = myrow[di];
      }
      fk_hinge_quat(rec.axis, ang, jq);
    }
    const float lt[3] = {rec.lpos[0], rec.lpos[1], rec.lpos[2]};
    const float lr[4] = {rec.lrot[0], rec.lrot[1], rec.lrot[2], rec.lrot[3]};
    float wt[3], tmp[4];
    fk_quat_rotate(pr, lt, wt);
#pragma unroll
    for (int i = 0; i < 3; i++) cp[i] = pp[i] + wt[i];
    fk_quat_mul(lr, jq, tmp);
    fk_quat_mul(pr, tmp, cr);
    if (MODE == 0) stage(j % kFkGroup);
    else zmin = fminf(zmin, cp[2]);
    const int sv = rec.save_slot;
    if (sv >= 0) {
      float *s = slots + (size_t)sv * 7 * kFkThreads + tid;
#pragma unroll
      for (int i = 0; i < 3; i++) s[i * kFkThreads] = cp[i];
#pragma unroll
      for (int i = 0; i < 4; i++) s[(3 + i) * kFkThreads] = cr[i];
    }
  }
  if (MODE == 0) {
    const int j0 = (nbody - 1) / kFkGroup * kFkGroup;  // the last, possibly partial, group
    flush_n(j0, nbody - j0);
  }
  if (MODE == 1) {  // every lane takes part: dead lanes carry a copy of the last frame, which cannot change its clip's minimum
    // clip of this frame: binary search in seq_offsets (wave-divergent, tiny)
    int lo = 0, hi = n_seq;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (seq_offsets[mid] <= fc) lo = mid; else hi = mid;
    }
    // order-preserving int key of a float: flip the magnitude bits of negatives
    int k = __float_as_int(zmin);
    k = k >= 0 ? k : (k ^ 0x7fffffff);
    // clips are thousands of frames long: almost every wavefront sits inside one clip, so reduce in the wave and issue one
    // atomic instead of 64 contending ones; wavefronts that straddle a boundary fall back to one atomic per lane
    const int lo0 = __builtin_amdgcn_readfirstlane(lo);
    if (__all(lo == lo0)) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) k = min(k, __shfl_xor(k, off));
      if ((tid & 63) == 0) atomicMin(min_key + lo0, k);
    } else {
      atomicMin(min_key + lo, k);
    }
  }
}

// ------------------------------------------------------------------ positions only: one wavefront per workgroup, no barriers
// gmr_fk without rotations (what the dataset scripts call, scripts/smplx_to_robot_dataset.py:110-112) is bound by how the
// poses leave the chip: flushed eight bodies at a time (fk_kernel above) every store covers 96-byte runs 456 bytes apart, i.e.
// partial cache lines whose other parts arrive several flushes later -- by then the 4 MB L2 of the XCD (32 CUs x 8 workgroups x
// 58 KB of output tiles in flight) has written them back partially.  Here a wavefront keeps the complete image of its output
// tile (64 frames x nbody x 3 floats, contiguous in memory) in LDS and writes it once, linearly, 16 bytes per lane and store:
// full lines only.  PARTS = 2 halves the LDS image (bodies in two passes) for twice the waves per CU at the price of one
// shared line per frame between the two passes.  Same chain arithmetic as fk_kernel (identical results).
constexpr int kFkWave = 64;

struct FkD4 { float v[4]; };  // 16 bytes, 4-byte aligned

template <int PARTS>
__global__ void __launch_bounds__(kFkWave) fk_pos_kernel(FkTree t, const float *__restrict__ root_pos, const float *__restrict__ root_rot,
                                                         const float *__restrict__ dof, int64_t n_frames, float *__restrict__ body_pos) {
#pragma clang fp contract(off)
  extern __shared__ float fk_lds[];
  const int lane = threadIdx.x;
  const int nbody = t.nbody, ndof = t.ndof, row = 3 * nbody;
  const int64_t f0 = (int64_t)blockIdx.x * kFkWave;
  const int nfb = (int)(n_frames - f0 < kFkWave ? n_frames - f0 : kFkWave);
  const int64_t fc = lane < nfb ? f0 + lane : n_frames - 1;  // dead lanes recompute the last frame; their rows are not flushed
  float *slots = fk_lds;
  float *img = slots + (size_t)t.nslots * 7 * kFkWave;
  const int hb = (nbody + PARTS - 1) / PARTS;  // bodies per part
  const int prow = PARTS == 1 ? row : 3 * hb;   // words per frame in the LDS image
  const float *myrow = dof + fc * ndof;
  const bool windowed = ndof >= 4 && t.dof_in_order;
  const int last_start = ndof - 4;
  FkD4 dwin{{0.f, 0.f, 0.f, 0.f}}, dnext{{0.f, 0.f, 0.f, 0.f}};
  int win_group = -1;
  if (windowed) dnext = *reinterpret_cast<const FkD4 *>(myrow);
  float cp[3], cr[4];
#pragma unroll
  for (int i = 0; i < 3; i++) cp[i] = root_pos[fc * 3 + i];
#pragma unroll
  for (int i = 0; i < 4; i++) cr[i] = root_rot[fc * 4 + i];
  if (fk_const(t.save_slot, 0) >= 0) {
    float *s = slots + (size_t)fk_const(t.save_slot, 0) * 7 * kFkWave + lane;
#pragma unroll
    for (int i = 0; i < 3; i++) s[i * kFkWave] = cp[i];
#pragma unroll
    for (int i = 0; i < 4; i++) s[(3 + i) * kFkWave] = cr[i];
  }
  auto flush_part = [&](int j0, int nb_part) {  // bodies j0 .. j0+nb_part-1 of the tile's nfb frames
    __syncthreads();  // (one wavefront: orders the LDS writes of the other lanes before the reads below)
    if (PARTS == 1) {
      const int nwords = nfb * row, n4 = nwords >> 2;
      float *dst = body_pos + f0 * row;  // 16-byte aligned: 64 rows of 12 nbody bytes per tile
      for (int i = lane; i < n4; i += kFkWave)
        *reinterpret_cast<float4 *>(dst + 4 * i) = *reinterpret_cast<const float4 *>(img + 4 * i);
      for (int i = 4 * n4 + lane; i < nwords; i += kFkWave) dst[i] = img[i];
    } else {
      const int w = 3 * nb_part;
      for (int fr = 0; fr < nfb; ++fr) {
        float *dst = body_pos + (f0 + fr) * row + 3 * j0;
        for (int o = lane; o < w; o += kFkWave) dst[o] = img[fr * prow + o];
      }
    }
    __syncthreads();
  };
#pragma unroll
  for (int i = 0; i < 3; i++) img[lane * prow + i] = cp[i];
  FkBody nxt = fk_body(t, nbody > 1 ? 1 : 0);
  for (int j = 1; j < nbody; ++j) {
    const FkBody rec = nxt;
    nxt = fk_body(t, j + 1 < nbody ? j + 1 : j);  // one body ahead
    if (PARTS > 1 && j % hb == 0) flush_part(j - hb, hb);
    float pp[3], pr[4];
    const int src = rec.src_slot;
    if (src < 0) {
#pragma unroll
      for (int i = 0; i < 3; i++) pp[i] = cp[i];
#pragma unroll
      for (int i = 0; i < 4; i++) pr[i] = cr[i];
    } else {
      const float *s = slots + (size_t)src * 7 * kFkWave + lane;
#pragma unroll
      for (int i = 0; i < 3; i++) pp[i] = s[i * kFkWave];
#pragma unroll
      for (int i = 0; i < 4; i++) pr[i] = s[(3 + i) * kFkWave];
    }
    float jq[4] = {0.f, 0.f, 0.f, 1.f};
    const int di = rec.dofidx;
    if (di >= 0) {
      float ang;
      if (windowed) {
        const int g = di >> 2;
        while (win_group < g) {
          dwin = dnext;
          ++win_group;
          const int nxt = 4 * (win_group + 1);
          if (nxt < ndof) dnext = *reinterpret_cast<const FkD4 *>(myrow + (nxt < last_start ? nxt : last_start));
        }
        const int start = 4 * g < last_start ? 4 * g : last_start;
        const int k = di - start;
        ang = k == 0 ? dwin.v[0] : k == 1 ? dwin.v[1] : k == 2 ? dwin.v[2] : dwin.v[3];
      } else {
        ang = myrow[di];
      }
      fk_hinge_quat(rec.axis, ang, jq);
    }
    const float lt[3] = {rec.lpos[0], rec.lpos[1], rec.lpos[2]};
    const float lr[4] = {rec.lrot[0], rec.lrot[1], rec.lrot[2], rec.lrot[3]};
    float wt[3], tmp[4];
    fk_quat_rotate(pr, lt, wt);
#pragma unroll
    for (int i = 0; i < 3; i++) cp[i] = pp[i] + wt[i];
    fk_quat_mul(lr, jq, tmp);
    fk_quat_mul(pr, tmp, cr);
    {
      const int jj = PARTS == 1 ? j : j % hb;
#pragma unroll
      for (int i = 0; i < 3; i++) img[lane * prow + 3 * jj + i] = cp[i];
    }
    const int sv = rec.save_slot;
    if (sv >= 0) {
      float *s = slots + (size_t)sv * 7 * kFkWave + lane;
#pragma unroll
      for (int i = 0; i < 3; i++) s[i * kFkWave] = cp[i];
#pragma unroll
      for (int i = 0; i < 4; i++) s[(3 + i) * kFkWave] = cr[i];
    }
  }
  {
    const int j0 = PARTS == 1 ? 0 : (nbody - 1) / hb * hb;
    flush_part(j0, nbody - j0);
  }
}

__global__ void fk_minkey_init(int *keys, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) keys[i] = 0x7f800000;  // +inf
}
__global__ void fk_minkey_decode(const int *keys, float *out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    int k = keys[i];
    k = k >= 0 ? k : (k ^ 0x7fffffff);
    out[i] = __int_as_float(k);
  }
}

}  // namespace gmr
