// kin_ops_kernel.hip.h -- the rest of the KinematicsModel operator surface (float32, xyzw), next to forward_kinematics.
//
// Replaces, for batches resident in HBM (reference general_motion_retargeting/kinematics_model.py):
//   KinematicsModel.dof_to_rot :172-182 (Joint.dof_to_rot :21-36, torch_utils.axis_angle_to_quat :353-359)      dof_to_rot_kernel
//   KinematicsModel.rot_to_dof :184-197 (Joint.rot_to_dof :38-53, torch_utils.quat_to_axis_angle :320-341)      rot_to_dof_kernel
//   KinematicsModel.convert_local_rot_to_global :199-211 (torch_utils.quat_mul :117-138)                        local_to_global_kernel
//   forward_kinematics(..., fitted_shape=) :225                                                                fk_scale_bodies_kernel
// The reference loops over the joints in Python with one [T, .] tensor op per joint and step.  All three are byte movers with a
// little arithmetic per element: one element per lane, consecutive lanes on consecutive 16-byte (4-byte) elements of the contiguous
// [T, joints, 4] arrays, so every load and store instruction of a wavefront covers whole cache lines.  The model's small per-joint
// tables are staged in LDS once per workgroup.
// Algorithmic HBM bytes per frame: dof_to_rot 4 ndof + 16 (nb - 1); rot_to_dof 16 (nb - 1) + 4 ndof; local -> global 32 nb.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fk_kernel.hip.h"
#include "tree_chain.hip.h"

namespace gmr {

#ifndef GMR_KIN_CHAIN_PASSES
#define GMR_KIN_CHAIN_PASSES 8  // 64-element passes per batch of local_to_global_kernel (16 KB of LDS images per wavefront)
#endif
constexpr int kKinThreads = 256;
constexpr int kKinMaxBodies = 64;  // GMR_MAX_BODIES

struct KinTables {            // device arrays
  const int *dof_body;        // [ndof] body that owns hinge d
  const float *lim_lo, *lim_hi;  // [ndof] float32 of the XML range (KinematicsModel._dof_lower_limits / _dof_upper_limits)
  const uint8_t *depth;       // [nb] tree level of body j (0 for the root)
  const uint8_t *order;       // [nb] bodies sorted by level (stable)
  int max_depth;
};

// ---------------------------------------------------------------------------------------------------------------- dof_to_rot
// Item i = (frame f, joint j = 1 + i % (nb - 1)): out[i] = hinge quaternion of the joint's angle, or the identity.
__global__ void __launch_bounds__(kKinThreads) dof_to_rot_kernel(FkTree t, const float *__restrict__ dof, int64_t n_frames, float *__restrict__ out) {
  __shared__ double s_axis[kKinMaxBodies * 3];
  __shared__ int s_dof[kKinMaxBodies];
  const int nb = t.nbody, nj = nb - 1, ndof = t.ndof;
  for (int b = threadIdx.x; b < nb; b += kKinThreads) {
    s_dof[b] = t.dofidx[b];
    for (int i = 0; i < 3; ++i) s_axis[3 * b + i] = t.jaxis64[3 * b + i];
  }
  __syncthreads();
  // (frame, joint) of this thread's items by increments: one 64-bit division per thread instead of one per item
  const int64_t n = n_frames * nj, stride = (int64_t)gridDim.x * kKinThreads, i0 = (int64_t)blockIdx.x * kKinThreads + threadIdx.x;
  const int64_t sf = stride / nj;
  const int sj = (int)(stride - sf * nj);
  int64_t f = i0 / nj;
  int jj = (int)(i0 - f * nj);
  for (int64_t i = i0; i < n; i += stride, f += sf, jj += sj) {
    if (jj >= nj) { jj -= nj; ++f; }
    const int j = 1 + jj;
    const int d = s_dof[j];
    float q[4] = {0.f, 0.f, 0.f, 1.f};
    if (d >= 0) {
      const double ax[3] = {s_axis[3 * j], s_axis[3 * j + 1], s_axis[3 * j + 2]};
      fk_hinge_quat(ax, dof[f * ndof + d], q);
    }
    reinterpret_cast<float4 *>(out)[i] = make_float4(q[0], q[1], q[2], q[3]);
  }
}

// ---------------------------------------------------------------------------------------------------------------- rot_to_dof
// Item i = (frame f, hinge d): quat_pos, angle = 2 atan2(|xyz|, w), axis = xyz / |xyz| (default z and angle 0 at |xyz| <= 1e-5),
// the angle's sign from the float64 dot product with the joint axis, clamped to the joint's range.
__global__ void __launch_bounds__(kKinThreads) rot_to_dof_kernel(FkTree t, KinTables k, const float *__restrict__ rot, int64_t n_frames,
                                                                 float *__restrict__ out) {
#pragma clang fp contract(off)
  __shared__ double s_axis[kKinMaxBodies * 3];
  __shared__ int s_body[kKinMaxBodies];
  __shared__ float s_lo[kKinMaxBodies], s_hi[kKinMaxBodies];
  const int nb = t.nbody, nj = nb - 1, ndof = t.ndof;
  for (int d = threadIdx.x; d < ndof; d += kKinThreads) {
    const int b = k.dof_body[d];
    s_body[d] = b; s_lo[d] = k.lim_lo[d]; s_hi[d] = k.lim_hi[d];
    for (int i = 0; i < 3; ++i) s_axis[3 * d + i] = t.jaxis64[3 * b + i];
  }
  __syncthreads();
  const int64_t n = n_frames * ndof, stride = (int64_t)gridDim.x * kKinThreads, i0 = (int64_t)blockIdx.x * kKinThreads + threadIdx.x;
  const int64_t sf = stride / ndof;
  const int sd = (int)(stride - sf * ndof);
  int64_t f = i0 / ndof;
  int d = (int)(i0 - f * ndof);
  for (int64_t i = i0; i < n; i += stride, f += sf, d += sd) {
    if (d >= ndof) { d -= ndof; ++f; }
    const float4 q0 = reinterpret_cast<const float4 *>(rot)[f * nj + (s_body[d] - 1)];
    const float sg = 1.0f - 2.0f * (q0.w < 0.0f ? 1.0f : 0.0f);
    const float x = sg * q0.x, y = sg * q0.y, z = sg * q0.z, w = sg * q0.w;
    const float len = sqrtf(x * x + y * y + z * z);
    float ang = 2.0f * atan2f(len, w);
    float ax = x / len, ay = y / len, az = z / len;
    if (!(len > 1e-5f)) { ang = 0.0f; ax = 0.0f; ay = 0.0f; az = 1.0f; }
    const double dot = (double)ax * s_axis[3 * d] + (double)ay * s_axis[3 * d + 1] + (double)az * s_axis[3 * d + 2];
    if (dot < 0.0) ang = -ang;
    ang = ang < s_lo[d] ? s_lo[d] : ang;
    ang = ang > s_hi[d] ? s_hi[d] : ang;
    out[i] = ang;
  }
}

// ---------------------------------------------------------------------------------------------------------- local -> global
// global[j] = global[parent(j)] (x) local[j], global[0] = local[0]: one product per element, in the reference's order, so the results
// are bit-identical to its sequential float32 loop.  A workgroup is one wavefront; it takes batches of F = 64 P / nb whole frames: a
// batch's input is one contiguous run of F nb quaternions, read with dense 16-byte loads (the next batch's loads are issued before this
// batch's arithmetic) and laid down in LDS as it is.  Within a batch the elements are dealt to the lanes DEPTH-MAJOR (all frames'
// roots first, the deepest bodies last), in P passes of 64: a pass holds bodies of one or two tree levels for all F frames, every
// lane busy, and its parents' results are already in the LDS image of the output from the passes before (a pass that spans several
// levels takes one round per level).  A batch therefore costs about P + depth rounds of one LDS read, one product and one LDS write --
// against nb - 1 dependent steps per frame in the reference's loop -- and the results leave in the input's order as dense 16-byte stores.
typedef float KinV4 __attribute__((ext_vector_type(4)));  // a native 16-byte vector: arrays of it stay in registers
template <int P>
__global__ void __launch_bounds__(64) local_to_global_kernel(FkTree t, KinTables kt, const float *__restrict__ local, int64_t n_frames,
                                                            float *__restrict__ global_) {
  extern __shared__ __align__(16) unsigned char kin_smem[];
  KinV4 *ibuf = reinterpret_cast<KinV4 *>(kin_smem);  // [64 P] the batch's local rotations, input order
  KinV4 *obuf = ibuf + 64 * P;                          // [64 P] its global rotations
  const int lane = threadIdx.x, nb = t.nbody;
  const int F = 64 * P / nb;  // frames per batch (nb <= 64: at least P)
  // this lane's elements of a batch: k = 64 p + lane -> (rank k / F in the depth order, frame k % F)
  int fbase[P], depth[P], self[P], par[P], dlo[P], dhi[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int k = 64 * p + lane, rank = k / F, f = k - rank * F;
    const bool ok = rank < nb;
    const int j = ok ? kt.order[rank] : 0;
    fbase[p] = ok ? f * nb : 0x7fffffff;  // beyond every batch: the lane idles in this pass
    self[p] = j;
    par[p] = ok && j > 0 ? t.parent[j] : 0;
    depth[p] = ok ? kt.depth[j] : 0x7fffffff;
    // the levels of this pass (wave-uniform): ranks are sorted by depth, so lane 0 holds the shallowest and the last element the deepest
    const int last = min(63, F * nb - 1 - 64 * p);
    dlo[p] = __builtin_amdgcn_readlane(depth[p], 0);
    dhi[p] = last >= 0 ? __builtin_amdgcn_readlane(depth[p], last < 0 ? 0 : last) : -1;
  }
  const int64_t n_batches = (n_frames + F - 1) / F;
  const KinV4 *src = reinterpret_cast<const KinV4 *>(local);
  KinV4 *dst = reinterpret_cast<KinV4 *>(global_);
  // the prefetch registers: loads are unconditional (indices clamped to the array's last element) so that they stay registers
  KinV4 nxt[P];
  const int64_t e_last = n_frames * nb - 1;
  int64_t b = blockIdx.x;
  if (b >= n_batches) return;
#pragma unroll
  for (int p = 0; p < P; ++p) nxt[p] = src[min(b * F * nb + 64 * p + lane, e_last)];
  for (; b < n_batches; b += gridDim.x) {
    const int fb = (int)min((int64_t)F, n_frames - b * F), ne = fb * nb;
#pragma unroll
    for (int p = 0; p < P; ++p)
      if (64 * p + lane < ne) ibuf[64 * p + lane] = nxt[p];
    wave_lds_sync();
    {
      const int64_t bn = min(b + (int64_t)gridDim.x, n_batches - 1);  // (the last iteration re-reads a batch it does not use)
#pragma unroll
      for (int p = 0; p < P; ++p) nxt[p] = src[min(bn * F * nb + 64 * p + lane, e_last)];
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const bool live = fbase[p] < ne;  // a whole frame of this batch
      float l[4] = {0.f, 0.f, 0.f, 1.f};
      if (live) {
        const KinV4 r0 = ibuf[fbase[p] + self[p]];
        l[0] = r0[0]; l[1] = r0[1]; l[2] = r0[2]; l[3] = r0[3];
        if (depth[p] == 0) obuf[fbase[p] + self[p]] = r0;  // the root row is copied
      }
      if (dlo[p] == 0) wave_lds_sync();  // the copies are read by level 1, in this pass or the next
      for (int d = max(dlo[p], 1); d <= dhi[p]; ++d) {
        if (live && depth[p] == d) {
          const KinV4 g4 = obuf[fbase[p] + par[p]];
          const float g[4] = {g4[0], g4[1], g4[2], g4[3]};
          float o[4];
          fk_quat_mul(g, l, o);
          obuf[fbase[p] + self[p]] = KinV4{o[0], o[1], o[2], o[3]};
        }
        wave_lds_sync();
      }
    }
    const int64_t e0 = b * F * nb;
#pragma unroll
    for (int p = 0; p < P; ++p)
      if (64 * p + lane < ne) dst[e0 + 64 * p + lane] = obuf[64 * p + lane];
    // (the next batch's obuf writes come after the wave_lds_sync that follows its ibuf writes)
  }
}

// ------------------------------------------------------------------------------------------------------------ fitted_shape
// forward_kinematics(..., fitted_shape) scales every body's local translation, in float32, before the chain (kinematics_model.py:225).
// The FK kernels read one FkBody record per body: this writes the scaled copy of the record table that one call's launch reads.
// `shape` is [nb][width], width 1 (a scalar per body) or 3.
__global__ void fk_scale_bodies_kernel(const FkBody *__restrict__ body, const float *__restrict__ shape, int width, int nb, FkBody *__restrict__ out) {
#pragma clang fp contract(off)
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nb) return;
  FkBody r = body[j];
  for (int i = 0; i < 3; ++i) r.lpos[i] = r.lpos[i] * shape[j * width + (width == 3 ? i : 0)];
  out[j] = r;
}

}  // namespace gmr
