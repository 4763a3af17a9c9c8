// bvh_kernel.hip.h -- BVH skeleton FK for the LAFAN1-style input adapter (float64).
//
// Replaces the numeric part of load_lafan1_file (reference general_motion_retargeting/utils/lafan1.py:8-40):
// utils.euler_to_quat (lafan_vendor/utils.py:56-75), utils.quat_fk (:88-103), the Y-up -> Z-up rotation
// (lafan1.py:20-21,31-32), centimetres -> metres (:32) and the synthesised LeftFootMod / RightFootMod entries
// (foot position + toe orientation, :36-39) -- and, in the row layouts, the slicing of read_bvh's motion rows into local
// positions and Euler angles (lafan_vendor/extract.py:140-166) and the degrees -> radians step of load_lafan1_file.
//
// One wavefront per run of frames, lane = joint (tree_chain.hip.h): a frame's motion row is read once, densely, straight from
// the parsed text rows; the chain runs in registers + the LDS exchange buffer by pointer jumping; Z-up, scale and the extra
// entries are applied in the same pass; every output row is written once.  Optionally only the columns the IK config consumes
// are emitted (`out_col`), so the IK kernel reads a dense [T, 14, 7].
// remove_quat_discontinuities (extract.py:164) only flips quaternion signs along time and is not applied: every
// consumer of the orientations (scipy Rotation in update_targets, the SE3 log) is sign-insensitive.
#pragma once

#include "tree_chain.hip.h"

namespace gmr {

constexpr int kBvhMaxJoints = 192;  // three joints per lane
constexpr int kBvhMaxExtra = 8;

// Input layouts.  SPLIT: the arrays of gmr_bvh_fk (local positions + Euler angles in radians, [T][J][3] each).  ROWS3/6/9: the
// file's own motion rows [T][ncol] in degrees -- 3 root position values + 3 angles per joint; (position, angles) per joint;
// 3 root position values + (position, angles, scale) per non-root joint (extract.py:145-156) -- with the joints' constant
// offsets from the HIERARCHY section beside them.
enum { BVH_SPLIT = 0, BVH_ROWS3 = 3, BVH_ROWS6 = 6, BVH_ROWS9 = 9 };

struct BvhSkeleton {
  int n_joints, n_extra, order[3], layout;
  int n_out;        // columns of the output arrays
  int pad;
  short parent[kBvhMaxJoints];
  short out_col[kBvhMaxJoints];      // output column of joint j, -1 = not emitted
  short extra_pos_src[kBvhMaxExtra], extra_rot_src[kBvhMaxExtra], extra_col[kBvhMaxExtra];  // extra_col -1 = not emitted
};

// euler_to_quat (lafan_vendor/utils.py:56-75): q(e0, axis o0) (x) (q(e1, o1) (x) q(e2, o2)), wxyz.  When the three axes are distinct
// (every BVH file names each rotation channel once) the two quaternion products collapse to eight triple products of the half-angle
// sines and cosines: with eps = +1 for an even order (xyz, yzx, zxy) and -1 for an odd one,
//   w = c0 c1 c2 - eps s0 s1 s2,  [o0] = s0 c1 c2 + eps c0 s1 s2,  [o1] = c0 s1 c2 - eps s0 c1 s2,  [o2] = c0 c1 s2 + eps s0 s1 c2.
// Repeated axes take the general products.
__device__ __forceinline__ void bvh_euler_quat(const double (&s)[3], const double (&c)[3], int o0, int o1, int o2, double q[4]) {
  if (o0 != o1 && o1 != o2 && o0 != o2) {  // wave-uniform (kernel arguments)
    const double eps = ((o1 - o0 + 3) % 3 == 1) ? 1.0 : -1.0;
    const double cc = c[1] * c[2], ss = eps * (s[1] * s[2]), sc = s[1] * c[2], cs = eps * (c[1] * s[2]);
    const double w = c[0] * cc - s[0] * ss;
    const double va = s[0] * cc + c[0] * ss, vb = c[0] * sc - s[0] * cs, vc = c[0] * (c[1] * s[2]) + s[0] * (eps * sc);
    q[0] = w;
    q[1] = o0 == 0 ? va : (o1 == 0 ? vb : vc);
    q[2] = o0 == 1 ? va : (o1 == 1 ? vb : vc);
    q[3] = o0 == 2 ? va : (o1 == 2 ? vb : vc);
  } else {
    double q0[4] = {c[0], o0 == 0 ? s[0] : 0.0, o0 == 1 ? s[0] : 0.0, o0 == 2 ? s[0] : 0.0};
    double q1[4] = {c[1], o1 == 0 ? s[1] : 0.0, o1 == 1 ? s[1] : 0.0, o1 == 2 ? s[1] : 0.0};
    double q2[4] = {c[2], o2 == 0 ? s[2] : 0.0, o2 == 1 ? s[2] : 0.0, o2 == 2 ? s[2] : 0.0};
    double t[4];
    qmul(q1, q2, t);
    qmul(q0, t, q);
  }
}

// Frames of input a wavefront stages in LDS per batch and the doubles one stage buffer takes; the launch's dynamic LDS holds TWO
// such buffers (the batch being evaluated and the next one, already on its way): enough frames that the one full drain of the
// memory counter per batch is amortised, few enough that five wavefronts per SIMD still fit beside the exchange buffer.
struct BvhBatch { int frames, doubles; };
__host__ __device__ inline BvhBatch bvh_batch(int per_frame, int groups) {
  int n = 2304 / 8 / (per_frame > 0 ? per_frame : 1);  // ~2 KB of rows per buffer (two buffers: bvh_stage_bytes)
  if (n > 8) n = 8;
  if (n < groups) n = groups < 8 ? groups : 8;
  if (n < 1) n = 1;
  if (n >= groups) n = n / groups * groups;
  BvhBatch b;
  b.frames = n;
  b.doubles = (n * per_frame + 3) & ~1;  // (+ room to start the angle rows of the split layout on a 16-byte boundary)
  return b;
}

// pos_out [T][n_out][3] metres, Z-up; quat_out [T][n_out][4] wxyz.  rot = [[1,0,0],[0,0,-1],[0,1,0]] (a +90 deg turn about x).
// pbase / rbase: position and angle sources (see the layouts), `offsets` [J][3] the constant local positions (row layouts).
// A wavefront handles frames [blockIdx.x * chunk, ... + chunk) in batches: the batch's input rows -- one dense run of the motion
// array -- are copied into LDS with linear 16-byte loads, then `groups` frames per iteration are evaluated from there.  On gfx9-family
// hardware a wait for a load also drains the stores issued before it (one counter, out-of-order return), so input is fetched once
// per batch, not once per frame, and the output stores of a whole batch are in flight behind the arithmetic.
// S9: the 9-channel layout; K: joints per lane.
template <int K, bool S9>
__global__ void __launch_bounds__(64) bvh_fk_kernel(BvhSkeleton sk, const double *__restrict__ pbase, const double *__restrict__ rbase,
                                                    const double *__restrict__ offsets, int pstride, int rstride, double ang_scale,
                                                    int64_t T, int chunk, double scale, double *__restrict__ pos_out,
                                                    double *__restrict__ quat_out) {
  extern __shared__ __attribute__((aligned(16))) double stage_all[];  // two buffers of input rows (bvh_batch)
  constexpr int NX = 512 * K;  // exchange buffer (7 x (64 K + 1) doubles) and, after the chain, the iteration's output rows
  constexpr int XS = 64 * K + 1;  // slots per component: the items' and one that holds the identity pose
  __shared__ __attribute__((aligned(16))) double xbuf[NX];
  __shared__ int xi[64 * K];
  double (*xb)[XS] = reinterpret_cast<double (*)[XS]>(xbuf);
  const int lane = threadIdx.x;
  const int J = sk.n_joints, NO = sk.n_out;
  const ChainGeom geo = chain_geom(J);
  const int jp = geo.jp, G = geo.groups;
  const bool split = rbase != pbase;           // two arrays (gmr_bvh_fk) or the file's own rows
  const int per_frame = split ? pstride + rstride : pstride;
  const BvhBatch bat = bvh_batch(per_frame, G);
  const int nbatch = bat.frames;
  const int rsec = split ? (nbatch * pstride + 1) & ~1 : 0;  // where the angle rows start in the stage (16-byte aligned)
  // output rows of one iteration (G frames x NO columns) gathered in LDS and written as dense 16-byte-per-lane runs -- whole cache
  // lines instead of 24- and 32-byte pieces at a stride -- when they fit the exchange buffer; else straight from the lanes
  const int qsec = (G * NO * 3 + 1) & ~1;
  const bool staged_out = qsec + G * NO * 4 <= NX;

  // ---- per item, once per wavefront: joint, frame group, sources, output columns, pointer-jumping plan
  int jn[K], grp[K], poff[K], roff[K], soff[K], ocol[K], pslot[K];
  double off[K][3];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int s = 64 * k + lane;
    const int j = G > 1 ? (s & (jp - 1)) : s;
    grp[k] = G > 1 ? s / jp : 0;
    const bool has = j < J;
    jn[k] = has ? j : -1;
    const int par = has ? (int)sk.parent[j] : -1;
    pslot[k] = par >= 0 ? s - j + par : -1;
    ocol[k] = has ? (int)sk.out_col[j] : -1;
    poff[k] = roff[k] = soff[k] = -1;
    off[k][0] = off[k][1] = off[k][2] = 0.0;
    if (has) {
      switch (sk.layout) {
        case BVH_SPLIT: poff[k] = 3 * j; roff[k] = 3 * j; break;
        case BVH_ROWS3: poff[k] = j == 0 ? 0 : -1; roff[k] = 3 + 3 * j; break;
        case BVH_ROWS6: poff[k] = 6 * j; roff[k] = 6 * j + 3; break;
        default:  // BVH_ROWS9: the root keeps a zero rotation (extract.py:152-156)
          poff[k] = j == 0 ? 0 : 3 + 9 * (j - 1);
          roff[k] = j == 0 ? -1 : 6 + 9 * (j - 1);
          soff[k] = j == 0 ? -1 : 9 + 9 * (j - 1);
      }
      if (sk.layout != BVH_SPLIT && sk.layout != BVH_ROWS6 && j > 0) {
        off[k][0] = offsets[3 * j]; off[k][1] = offsets[3 * j + 1]; off[k][2] = offsets[3 * j + 2];
      }
    }
  }
  // extra entries: per item the column that copies its position / its orientation (the first such entry; a joint that feeds
  // several entries of one kind -- no skeleton here does -- sends the wavefront through the entry loop instead)
  int epc[K], erc[K];
  bool extra_loop = false;
#pragma unroll
  for (int k = 0; k < K; ++k) { epc[k] = -1; erc[k] = -1; }
  for (int e = 0; e < sk.n_extra; ++e) {
    const int c = sk.extra_col[e];
    if (c < 0) continue;
    for (int e2 = 0; e2 < e; ++e2)
      if (sk.extra_col[e2] >= 0 && (sk.extra_pos_src[e2] == sk.extra_pos_src[e] || sk.extra_rot_src[e2] == sk.extra_rot_src[e])) extra_loop = true;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (sk.extra_pos_src[e] == jn[k] && epc[k] < 0) epc[k] = c;
      if (sk.extra_rot_src[e] == jn[k] && erc[k] < 0) erc[k] = c;
    }
  }
  unsigned long long plan[K];
  const int rounds = chain_plan<K>(pslot, lane, xi, plan);
  const int o0 = sk.order[0], o1 = sk.order[1], o2 = sk.order[2];

  const int64_t f_begin = (int64_t)blockIdx.x * chunk;
  const int64_t f_end = f_begin + chunk < T ? f_begin + chunk : T;

  // one dense run of `n` doubles from global memory into the stage at `dst`: 16 bytes per lane and load, whole cache lines but
  // for the run's two ends (the rows are 8-byte aligned, so the 16-byte loads are not: the memory pipeline takes that)
  // one dense run of `n` doubles from global memory into a stage buffer at `dst`, written by the memory pipeline itself
  // (global_load_lds_dwordx4: 16 bytes per lane, LDS address = M0 + 16 lane; no registers, no LDS store instructions).  The rows are
  // 8-byte aligned, so the run starts on an 8-byte boundary; an odd last double travels as two 4-byte pieces.
  auto dma_in = [&](const double *src, double *buf, int dst, int n) {
    const int n2 = n >> 1;
    for (int base = 0; base < n2; base += 64) {  // wave-uniform
      if (base + lane < n2)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 2 * (base + lane)),
                                         (__attribute__((address_space(3))) void *)(buf + dst + 2 * base), 16, 0, 0);
    }
    if ((n & 1) && lane < 2)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const char *)(src + n - 1) + 4 * lane),
                                       (__attribute__((address_space(3))) void *)(buf + dst + n - 1), 4, 0, 0);
  };
  auto fetch = [&](int64_t fb0, double *buf) {  // the rows of the batch that starts at frame fb0 (nothing past the wavefront's run)
    if (fb0 >= f_end) return;
    const int nb = (int)(f_end - fb0 < nbatch ? f_end - fb0 : nbatch);
    dma_in(pbase + fb0 * pstride, buf, 0, nb * pstride);
    if (split) dma_in(rbase + fb0 * rstride, buf, rsec, nb * rstride);
  };

  // one dense run of `n` doubles from xbuf[src...] to global memory, 16 bytes per lane and store
  auto copy_out = [&](double *dst, int src, int n) {
    constexpr int U = (NX / 2 + 63) / 64;
    struct __attribute__((aligned(8))) Pair { double a, b; };
    const int n2 = n >> 1;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (64 * u >= n2) break;  // wave-uniform
      if (lane + 64 * u < n2) {
        const double2 v = *reinterpret_cast<const double2 *>(&xbuf[src + 2 * (lane + 64 * u)]);
        Pair w; w.a = v.x; w.b = v.y;
        *reinterpret_cast<Pair *>(dst + 2 * (lane + 64 * u)) = w;
      }
    }
    if ((n & 1) && lane == 0) dst[n - 1] = xbuf[src + n - 1];
  };

  // Batches.  Buffer b & 1 holds batch b; batch b + 1 is requested before batch b is evaluated.  gfx9-family hardware counts
  // loads and stores in ONE counter and returns them out of order, so waiting for a load means draining every store issued before
  // the wait: that single wait per batch sits in FRONT of the last iteration's stores, where everything still in flight is at least
  // an iteration old, and the request for the batch after next goes out right behind it.
  fetch(f_begin, stage_all);
  fetch(f_begin + nbatch, stage_all + bat.doubles);
  __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0) in the gfx9 encoding (expcnt left at 7)
  int bi = 0;
  for (int64_t fb0 = f_begin; fb0 < f_end; fb0 += nbatch, bi ^= 1) {
    const int nb = (int)(f_end - fb0 < nbatch ? f_end - fb0 : nbatch);
    double *stage = stage_all + bi * bat.doubles;
    wave_lds_sync();

    for (int fi = 0; fi < nb; fi += G) {
      double q[K][4], p[K][3];
      if (lane < 7) xb[lane][64 * K] = lane == 0 ? 1.0 : 0.0;  // the identity pose finished chains fold (the buffer also carries output rows)
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int row = fi + grp[k] < nb ? fi + grp[k] : 0;  // (lanes of a frame past the end redo the first one; nothing of theirs is stored)
        const double *rr = &stage[rsec + row * rstride + (roff[k] >= 0 ? roff[k] : 0)];
        const double *pr = &stage[row * pstride + (poff[k] >= 0 ? poff[k] : 0)];
        const bool hr = roff[k] >= 0;  // no angle source: zero rotation
        const double half[3] = {0.5 * ((hr ? rr[0] : 0.0) * ang_scale), 0.5 * ((hr ? rr[1] : 0.0) * ang_scale), 0.5 * ((hr ? rr[2] : 0.0) * ang_scale)};
        if (S9) {  // 9-channel rows: offset + position * scale (two roundings, as numpy does it)
          const double *sc = &stage[row * pstride + (soff[k] >= 0 ? soff[k] : 0)];
#pragma unroll
          for (int c = 0; c < 3; ++c) p[k][c] = poff[k] < 0 ? off[k][c] : (soff[k] < 0 ? pr[c] : __dadd_rn(off[k][c], __dmul_rn(pr[c], sc[c])));
        } else {
#pragma unroll
          for (int c = 0; c < 3; ++c) p[k][c] = poff[k] < 0 ? off[k][c] : pr[c];
        }
        double sn[3], cs[3];
        sincos_n<3>(half, sn, cs);
        bvh_euler_quat(sn, cs, o0, o1, o2, q[k]);
      }
      // quat_fk by pointer jumping: pose relative to the ancestor of the round, composed with that ancestor's own; a chain that
      // has reached the root folds the identity slot (exact: 1 q = q, 0 + R(1) p = p), so no lane is masked and nothing is selected
#pragma unroll
      for (int r = 0; r < kChainMaxRounds; ++r) {
        if (r >= rounds) break;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const int s = 64 * k + lane;
          xb[0][s] = q[k][0]; xb[1][s] = q[k][1]; xb[2][s] = q[k][2]; xb[3][s] = q[k][3];
          xb[4][s] = p[k][0]; xb[5][s] = p[k][1]; xb[6][s] = p[k][2];
        }
        wave_lds_sync();
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const unsigned an = chain_anc(plan[k], r);
          const unsigned a = an == kNoAnc ? 64u * K : an;
          const double aq[4] = {xb[0][a], xb[1][a], xb[2][a], xb[3][a]};
          const double ap[3] = {xb[4][a], xb[5][a], xb[6][a]};
          double o[4], v[3];
          qmul(aq, q[k], o);
          qrot(aq, p[k], v);  // quat_mul_vec
          q[k][0] = o[0]; q[k][1] = o[1]; q[k][2] = o[2]; q[k][3] = o[3];
          p[k][0] = ap[0] + v[0]; p[k][1] = ap[1] + v[1]; p[k][2] = ap[2] + v[2];
        }
        wave_lds_sync();
      }
      // Z-up, scale, output (joint columns, then the extra entries: position of one joint, orientation of another)
      const int gvalid = nb - fi < G ? nb - fi : G;  // frames of this iteration
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int row = fi + grp[k];
        if (jn[k] < 0 || row >= nb) continue;
        const int64_t f = fb0 + row;
        const double h = 0.70710678118654757;  // rq (x) q with rq = [h, h, 0, 0]: the +90 degree turn about x
        const double o[4] = {h * q[k][0] - h * q[k][1], h * q[k][1] + h * q[k][0], h * q[k][2] - h * q[k][3], h * q[k][3] + h * q[k][2]};
        const double x = p[k][0] * scale, y = -p[k][2] * scale, z = p[k][1] * scale;  // p @ rot.T / 100
        auto put_pos = [&](int c) {
          if (staged_out) { double *gp = &xbuf[(grp[k] * NO + c) * 3]; gp[0] = x; gp[1] = y; gp[2] = z; }
          else { double *gp = pos_out + (f * NO + c) * 3; gp[0] = x; gp[1] = y; gp[2] = z; }
        };
        auto put_quat = [&](int c) {
          if (staged_out) { double *gq = &xbuf[qsec + (grp[k] * NO + c) * 4]; gq[0] = o[0]; gq[1] = o[1]; gq[2] = o[2]; gq[3] = o[3]; }
          else { double *gq = quat_out + (f * NO + c) * 4; gq[0] = o[0]; gq[1] = o[1]; gq[2] = o[2]; gq[3] = o[3]; }
        };
        if (ocol[k] >= 0) { put_pos(ocol[k]); put_quat(ocol[k]); }
        if (!extra_loop) {
          if (epc[k] >= 0) put_pos(epc[k]);
          if (erc[k] >= 0) put_quat(erc[k]);
        } else {
          for (int e = 0; e < sk.n_extra; ++e) {
            const int c = sk.extra_col[e];
            if (c < 0) continue;
            if (sk.extra_pos_src[e] == jn[k]) put_pos(c);
            if (sk.extra_rot_src[e] == jn[k]) put_quat(c);
          }
        }
      }
      const bool last = fi + G >= nb;
      if (last) {  // this buffer is consumed: the other one must have landed, then this one is refilled with the batch after next
        wave_lds_sync();
        __builtin_amdgcn_s_waitcnt(0x0070);
        fetch(fb0 + 2 * (int64_t)nbatch, stage);
      }
      if (staged_out) {
        wave_lds_sync();
        copy_out(pos_out + (fb0 + fi) * NO * 3, 0, gvalid * NO * 3);
        copy_out(quat_out + (fb0 + fi) * NO * 4, qsec, gvalid * NO * 4);
        wave_lds_sync();
      }
    }
  }
}

}  // namespace gmr
