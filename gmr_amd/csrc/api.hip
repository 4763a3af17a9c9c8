// api.hip -- C ABI of libgmr_amd.so (include/gmr_amd.h): model handle, scheduling, kernel launches.
//
// Host side is plain C++: blob validation, the derived index structures the kernels want (tree depth
// levels, active-dof list, ancestor masks, composite nodes of the task tree, FK branch-slot plan), one
// device allocation per model, a grow-only scheduling workspace, and shape checks in front of every
// launch.  No torch, no CPU fallback: every failure is reported, nothing is silently computed elsewhere.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <map>
#include <mutex>
#include <numeric>
#include <string>
#include <vector>

#include "fk_kernel.hip.h"
#include "ik_kernel.hip.h"
#include "bvh_kernel.hip.h"
#include "smplx_kernel.hip.h"
#include "bvh_parse_kernel.hip.h"
#include "kin_ops_kernel.hip.h"
#include "bvh_text.h"

using gmr::u64;

struct gmr_model {
  int hplan_extra_cycles = 0;  // modelled LDS bank-conflict cycles of the H pair plan, per solve (after ordering)
  int device = -1;
  gmr_blob_header h{};
  std::vector<uint8_t> blob;
  std::string err;
  // device storage
  void *dev = nullptr;
  size_t dev_bytes = 0;
  gmr::DevModel dm{};                      // host copy of the device model (scalars are read back by the host API)
  const gmr::DevModel *dm_dev = nullptr;  // the struct in device memory: the one model argument of the IK kernels
  const gmr::DevModel *dm_eval_dev = nullptr;  // the same with the full body tree (gmr_evaluate)
  gmr::LdsLayout lay_eval{};
  int lds_bytes_eval = 0;
  gmr::FkTree fk{};
  gmr::KinTables kin{};                    // dof_to_rot / rot_to_dof / local -> global (kin_ops_kernel.hip.h)
  gmr::LdsLayout lay{};
  int nvp = 0, n_act = 0, lds_bytes = 0, fk_lds_bytes = 0, fk_lds_bytes_min = 0;  // _min: the min-height mode has no output stage
  unsigned long long *dbg = nullptr;  // diagnostic builds (GMR_IK_STAMPS) only
  int min_nvp = 0;                    // group members are built for a common kernel variant (gmr_group_create)
  hipMemPool_t pool = nullptr;        // the library's scratch pool of this device (scratch_pool)
  int fk_pos_parts = 1;               // gmr_fk without rotations: fk_pos_kernel<parts> (GMR_AMD_FK_PARTS=0 falls back to fk_kernel<0>)
  bool force_generic = false;         // GMR_AMD_GENERIC_QP=1: use the dense generic QP even where the structured one applies
};

// State of a single-sequence session (gmr_session_*): one frame per call, warm start carried on the device.
struct gmr_session {
  gmr_model *m = nullptr;
  hipStream_t st = nullptr;
  uint8_t *host = nullptr;   // pinned, device-visible: [pos | quat | qpos_out(nq) | solves]
  uint8_t *dev = nullptr;    // [qpos state (nq doubles) | work item | slot_col]
  size_t pos_bytes = 0, quat_bytes = 0, quat_off = 0, out_off = 0;
  gmr::IkLaunch L{};
  // persistent mode (gmr_session_set_persistent): a resident wavefront fed through a pinned mailbox
  gmr::IkSessionBox *box = nullptr;        // inside `host`
  gmr::IkSessionBox *box_dev = nullptr;    // its device address
  gmr::IkGroupEntry *entries = nullptr;    // device: the one-frame work item with offset_to_ground = 0 / 1
  int idle_ms = 0;                         // 0 = one launch per frame
  unsigned seq = 0, gen = 0;               // frames posted; resident wavefronts launched (alive: box->exited_gen != gen)
};

// Several models built for one kernel variant (gmr_group_*).
struct gmr_group {
  std::vector<gmr_model *> models;
  int device = -1, nvp = 0;
  bool sq = false;
  std::string err;
};


namespace {

void set_err(gmr_model *m, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (m) m->err = buf;
}

#define HIP_TRY(m, expr)                                                              \
  do {                                                                                \
    hipError_t e_ = (expr);                                                           \
    if (e_ != hipSuccess) {                                                           \
      set_err(m, "%s failed: %s", #expr, hipGetErrorString(e_));                      \
      return GMR_EDEVICE;                                                             \
    }                                                                                 \
  } while (0)

template <class T>
const T *blob_ptr(const std::vector<uint8_t> &b, uint32_t off) {
  return reinterpret_cast<const T *>(b.data() + off);
}

bool range_ok(const gmr_blob_header &h, uint32_t off, size_t bytes) {
  return off >= sizeof(gmr_blob_header) && (off & 7u) == 0 && (size_t)off + bytes <= h.total_bytes;
}

// Packs host arrays into one device image, remembering where each one went.
struct Packer {
  std::vector<uint8_t> buf;
  template <class T>
  size_t add(const std::vector<T> &v) {
    size_t off = (buf.size() + 15) & ~size_t(15);
    buf.resize(off + std::max<size_t>(v.size(), 1) * sizeof(T));
    if (!v.empty()) memcpy(buf.data() + off, v.data(), v.size() * sizeof(T));
    return off;
  }
};

// Per-call scheduling data (work items, slot columns, clip offsets) lives in stream-ordered memory: allocated on the call's
// stream and released on it right behind the launch that reads it.  Calls on the same handle from different streams or host
// threads therefore never share (or pull away) each other's launch metadata; the pool keeps released blocks for reuse
// (release threshold set in gmr_model_create), so a call costs no device synchronisation.
struct CallScratch {
  void *p = nullptr;
  hipStream_t st = nullptr;
  ~CallScratch() { if (p) (void)hipFreeAsync(p, st); }
};
// The library's own pool, one per device, created on first use and kept for the life of the process: the few KB of per-call
// scheduling data never touch the device's default pool, whose settings belong to the host application.  Released
// blocks stay cached up to 64 MiB, so a call costs no device synchronisation.
hipMemPool_t scratch_pool(int device) {
  static std::mutex mu;
  static std::map<int, hipMemPool_t> pools;
  std::lock_guard<std::mutex> lock(mu);
  auto it = pools.find(device);
  if (it != pools.end()) return it->second;
  hipMemPoolProps props{};
  props.allocType = hipMemAllocationTypePinned;
  props.handleTypes = hipMemHandleTypeNone;
  props.location.type = hipMemLocationTypeDevice;
  props.location.id = device;
  hipMemPool_t pool = nullptr;
  if (hipMemPoolCreate(&pool, &props) != hipSuccess) { (void)hipGetLastError(); pool = nullptr; }
  if (pool) {
    uint64_t keep = 64ull << 20;
    (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
  }
  pools[device] = pool;
  return pool;
}
int scratch_alloc(gmr_model *m, CallScratch &sc, size_t bytes, hipStream_t st) {
  sc.st = st;
  if (m->pool) HIP_TRY(m, hipMallocFromPoolAsync(&sc.p, std::max<size_t>(bytes, 256), m->pool, st));
  else HIP_TRY(m, hipMallocAsync(&sc.p, std::max<size_t>(bytes, 256), st));
  return GMR_OK;
}

// Kernel variants by padded system size.  GMR_IK_DEV_ONLY36 (experiments only, never the shipped library) builds just
// ik_kernel<36, true> to cut compile time.
#ifdef GMR_IK_DEV_ONLY36
#define GMR_FOR_EACH_NVP(X) X(36)
#else
#define GMR_FOR_EACH_NVP(X) X(32) X(36) X(40) X(48) X(64)
#endif

int pick_nvp(int n_act) {
#define GMR_X(v) if (n_act <= v) return v;
  GMR_FOR_EACH_NVP(GMR_X)
#undef GMR_X
  return -1;
}

template <int NVP>
void launch_ik(const gmr_model *m, const gmr::IkLaunch &L, hipStream_t st, bool probe) {
  const bool sq = m->dm.sq_ok && !m->force_generic;
  if (probe) {
    if (sq) hipLaunchKernelGGL((gmr::ik_probe_kernel<NVP, true>), dim3(L.n_items), dim3(64), m->lds_bytes, st, m->dm_dev, L, m->lay);
#ifndef GMR_IK_DEV_ONLY36
    else hipLaunchKernelGGL((gmr::ik_probe_kernel<NVP, false>), dim3(L.n_items), dim3(64), m->lds_bytes, st, m->dm_dev, L, m->lay);
#endif
    return;
  }
  if (sq)
    hipLaunchKernelGGL((gmr::ik_kernel<NVP, true>), dim3(L.n_items), dim3(64), m->lds_bytes, st, m->dm_dev, L, m->lay);
#ifndef GMR_IK_DEV_ONLY36
  else
    hipLaunchKernelGGL((gmr::ik_kernel<NVP, false>), dim3(L.n_items), dim3(64), m->lds_bytes, st, m->dm_dev, L, m->lay);
#endif
}

int launch_ik_variant(gmr_model *m, const gmr::IkLaunch &L, hipStream_t st, bool probe = false) {
  switch (m->nvp) {
#define GMR_X(v) case v: launch_ik<v>(m, L, st, probe); break;
    GMR_FOR_EACH_NVP(GMR_X)
#undef GMR_X
    default: set_err(m, "internal: no kernel variant for nvp=%d", m->nvp); return GMR_EUNSUPPORTED;
  }
  HIP_TRY(m, hipGetLastError());
  return GMR_OK;
}

template <int NVP>
void launch_ik_group(const gmr_group *g, const gmr::IkGroupEntry *entries, const int *block_entry, int n_blocks, int lds_bytes, hipStream_t st) {
  if (g->sq)
    hipLaunchKernelGGL((gmr::ik_group_kernel<NVP, true>), dim3(n_blocks), dim3(64), lds_bytes, st, entries, block_entry);
#ifndef GMR_IK_DEV_ONLY36
  else
    hipLaunchKernelGGL((gmr::ik_group_kernel<NVP, false>), dim3(n_blocks), dim3(64), lds_bytes, st, entries, block_entry);
#endif
}


template <int NVP>
void launch_ik_session(const gmr_model *m, const gmr::IkGroupEntry *entries, gmr::IkSessionBox *box, unsigned long long idle_ticks,
                       unsigned max_polls, unsigned max_frames, unsigned gen, hipStream_t st) {
  if (m->dm.sq_ok && !m->force_generic)
    hipLaunchKernelGGL((gmr::ik_session_kernel<NVP, true>), dim3(1), dim3(64), m->lds_bytes, st, entries, box, idle_ticks, max_polls, max_frames, gen);
#ifndef GMR_IK_DEV_ONLY36
  else
    hipLaunchKernelGGL((gmr::ik_session_kernel<NVP, false>), dim3(1), dim3(64), m->lds_bytes, st, entries, box, idle_ticks, max_polls, max_frames, gen);
#endif
}

int build_device_model(gmr_model *m) {
  const gmr_blob_header &h = m->h;
  const auto &B = m->blob;
  const int nb = h.nbody;
  const int32_t *parent = blob_ptr<int32_t>(B, h.off_parent), *jtype = blob_ptr<int32_t>(B, h.off_jnt_type);
  const int32_t *qadr = blob_ptr<int32_t>(B, h.off_qpos_adr), *dadr = blob_ptr<int32_t>(B, h.off_dof_adr);
  const int32_t *limited = blob_ptr<int32_t>(B, h.off_jnt_limited);
  const double *bpos = blob_ptr<double>(B, h.off_body_pos), *bquat = blob_ptr<double>(B, h.off_body_quat);
  const double *bquat_raw = blob_ptr<double>(B, h.off_body_quat_raw), *axis = blob_ptr<double>(B, h.off_jnt_axis);
  const double *range = blob_ptr<double>(B, h.off_jnt_range), *qpos0 = blob_ptr<double>(B, h.off_qpos0);

  // ---- tree sanity: depth-first order, free root, hinges elsewhere ----
  if (parent[0] != -1 || jtype[0] != GMR_JNT_FREE || qadr[0] != 0 || dadr[0] != 0) {
    set_err(m, "body 0 must be the free-joint root");
    return GMR_EINVAL;
  }
  std::vector<int> depth(nb, 0);
  int maxdepth = 0, nq = 7, nv = 6;
  for (int b = 1; b < nb; ++b) {
    if (parent[b] < 0 || parent[b] >= b) { set_err(m, "bodies are not in depth-first order"); return GMR_EINVAL; }
    depth[b] = depth[parent[b]] + 1;
    maxdepth = std::max(maxdepth, depth[b]);
    if (jtype[b] == GMR_JNT_HINGE) {
      if (qadr[b] != nq || dadr[b] != nv) { set_err(m, "joint addresses out of order at body %d", b); return GMR_EINVAL; }
      ++nq; ++nv;
    } else if (jtype[b] != GMR_JNT_NONE) { set_err(m, "unsupported joint type at body %d", b); return GMR_EUNSUPPORTED; }
  }
  if (nq != h.nq || nv != h.nv) { set_err(m, "nq/nv do not match the joint list"); return GMR_EINVAL; }

  // ---- tasks ----
  std::vector<int> tbody(2 * GMR_MAX_TASKS, 0), tslot(2 * GMR_MAX_TASKS, 0);
  std::vector<double> twp(2 * GMR_MAX_TASKS, 0.0), twr(2 * GMR_MAX_TASKS, 0.0);
  for (int k = 0; k < 2; ++k) {
    const int32_t *tb = blob_ptr<int32_t>(B, h.off_task_body[k]), *ts = blob_ptr<int32_t>(B, h.off_task_slot[k]);
    const double *wp = blob_ptr<double>(B, h.off_task_wp[k]), *wr = blob_ptr<double>(B, h.off_task_wr[k]);
    for (int t = 0; t < h.ntask[k]; ++t) {
      if (tb[t] < 0 || tb[t] >= nb || ts[t] < 0 || ts[t] >= h.nslot) { set_err(m, "task %d of table %d out of range", t, k + 1); return GMR_EINVAL; }
      tbody[k * GMR_MAX_TASKS + t] = tb[t]; tslot[k * GMR_MAX_TASKS + t] = ts[t];
      twp[k * GMR_MAX_TASKS + t] = wp[t]; twr[k * GMR_MAX_TASKS + t] = wr[t];
    }
  }
  // body a is an ancestor-or-self of body b ?
  auto above = [&](int a, int b) {
    while (b > a) b = parent[b];
    return b == a;
  };
  // ---- bodies the IK needs: the task bodies and their ancestors.  Everything else (the 14 finger links of
  //      unitree_g1_with_hands, head / hand cosmetics) never enters a residual or a screw, so the IK kernel works on this
  //      pruned tree (fewer lanes, less LDS per wavefront); gmr_evaluate keeps the full tree for configuration.data.xpos. ----
  std::vector<int> keep, bmap(nb, -1);
  {
    std::vector<char> needed(nb, 0);
    needed[0] = 1;
    for (int k = 0; k < 2; ++k)
      for (int t = 0; t < h.ntask[k]; ++t)
        for (int b = tbody[k * GMR_MAX_TASKS + t]; b >= 0 && !needed[b]; b = parent[b]) needed[b] = 1;
    for (int b = 0; b < nb; ++b)
      if (needed[b]) { bmap[b] = (int)keep.size(); keep.push_back(b); }
  }
  const int nb_ik = (int)keep.size();
  // ---- active dofs: root 6 + hinges with at least one task (either table) at or below them ----
  std::vector<int> abody, akind, aqadr, alim;
  std::vector<double> arange;
  // the root's six rows.  A planar base (gmr_blob.h root_dof_mask 0x23) keeps all six in the QP, with z / roll / pitch as *null*
  // dofs (kind 7): the kernel's hinge path gives them the root body's hinge axis -- zero -- hence a zero screw, a zero gradient,
  // a diagonal of damping alone and dq = 0 exactly: the reference's 27-dof problem solved inside the 30-row one, with no
  // branch added to the kernel (rows 3, 4, 5 stay the root's rotation step: 0, 0, yaw).  They keep their place in the
  // elimination order but are left out of the ancestor relation (no H pairs) and of the structured QP's core.
  const int root_mask = h.root_dof_mask ? h.root_dof_mask : 0x3F;
  if (root_mask != 0x3F && root_mask != 0x23) { set_err(m, "root_dof_mask 0x%x: only a free joint (0x3f) or a planar base (0x23) is supported", root_mask); return GMR_EUNSUPPORTED; }
  if (axis[0] != 0.0 || axis[1] != 0.0 || axis[2] != 0.0) { set_err(m, "the root body must not carry a hinge axis"); return GMR_EINVAL; }
  for (int k = 0; k < 6; ++k) {
    abody.push_back(0); akind.push_back(((root_mask >> k) & 1) ? k : 7); aqadr.push_back(k < 3 ? k : 3); alim.push_back(0); arange.push_back(0); arange.push_back(0);
  }
  const int n_root = 6;
  for (int b = 1; b < nb; ++b) {
    if (jtype[b] != GMR_JNT_HINGE) continue;
    bool used = false;
    for (int k = 0; k < 2 && !used; ++k)
      for (int t = 0; t < h.ntask[k] && !used; ++t) used = above(b, tbody[k * GMR_MAX_TASKS + t]);
    if (!used) continue;
    abody.push_back(b); akind.push_back(6); aqadr.push_back(qadr[b]); alim.push_back(limited[b] ? 1 : 0);
    arange.push_back(range[2 * b]); arange.push_back(range[2 * b + 1]);
  }
  const int n_act = (int)abody.size();
  const int nvp = pick_nvp(std::max(n_act, m->min_nvp));
  if (nvp < 0) { set_err(m, "%d active dofs exceed the kernel limit of 64", n_act); return GMR_EUNSUPPORTED; }
  // ancestor relation among active dofs in depth-first order (j above i implies j < i)
  auto dfs_anc = [&](int j, int i) {
    if (j >= i) return false;
    if (akind[i] != 6) return true;      // root dofs (null ones included) form a chain 0 <- 1 <- ... <- 5
    if (akind[j] != 6) return true;      // every hinge hangs below the root's six dofs
    return abody[j] != abody[i] && above(abody[j], abody[i]);
  };
  // ---- elimination order of the QP: sort dofs by height (longest chain of dofs below), tallest first.  Dofs of equal
  //      height are never above one another, so a kinematic tree eliminates leaves-first without fill-in and all dofs of
  //      one height can be eliminated in the same step (level-scheduled sparse U D U' in ik_kernel's box_qp). ----
  std::vector<int> height(n_act, 0), perm(n_act);
  for (int i = n_act - 1; i >= 0; --i)
    for (int j = 0; j < i; ++j)
      if (dfs_anc(j, i)) height[j] = std::max(height[j], height[i] + 1);
  std::vector<u64> anc_dfs(n_act, 0);  // relation in depth-first indices, taken before the arrays are permuted
  for (int i = 0; i < n_act; ++i)
    for (int j = 0; j < i; ++j)
      if (dfs_anc(j, i) && akind[i] != 7 && akind[j] != 7) anc_dfs[i] |= 1ull << j;  // a null dof keeps its place in the order but couples with nothing
  std::iota(perm.begin(), perm.end(), 0);
  std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return height[a] > height[b]; });
  for (int k = 0; k < n_root; ++k)
    if (perm[k] != k) { set_err(m, "internal: root dofs must lead the elimination order"); return GMR_EINVAL; }
  {
    std::vector<int> b2(n_act), k2(n_act), q2(n_act), l2(n_act), h2(n_act);
    std::vector<double> r2(2 * n_act);
    for (int a = 0; a < n_act; ++a) {
      const int o = perm[a];
      b2[a] = abody[o]; k2[a] = akind[o]; q2[a] = aqadr[o]; l2[a] = alim[o]; h2[a] = height[o];
      r2[2 * a] = arange[2 * o]; r2[2 * a + 1] = arange[2 * o + 1];
    }
    abody = b2; akind = k2; aqadr = q2; alim = l2; arange = r2; height = h2;
  }
  std::vector<u64> aanc(64, 0);
  for (int a = 0; a < n_act; ++a)
    for (int b = 0; b < n_act; ++b)
      if ((anc_dfs[perm[a]] >> perm[b]) & 1ull) {
        if (b >= a) { set_err(m, "internal: elimination order breaks ancestor-first indexing"); return GMR_EINVAL; }
        aanc[a] |= 1ull << b;
      }
  // ---- FK pointer-jumping plan: ancestor folded in each round, one byte per round (for the pruned IK tree and the full tree) ----
  auto fk_plan = [&](const std::vector<int> &par, std::vector<u64> &plan, int &rounds) -> bool {
    const int n = (int)par.size();
    int maxd = 0;
    std::vector<int> dep(n, 0);
    for (int b2 = 1; b2 < n; ++b2) { dep[b2] = dep[par[b2]] + 1; maxd = std::max(maxd, dep[b2]); }
    rounds = 0;
    while ((1 << rounds) <= maxd) ++rounds;
    if (rounds > 8) return false;
    plan.assign(n, ~0ull);
    std::vector<int> anc(par);
    for (int r = 0; r < rounds; ++r) {
      std::vector<int> nxt(n, -1);
      for (int b2 = 0; b2 < n; ++b2) {
        const u64 byte = anc[b2] < 0 ? 0xffull : (u64)anc[b2];
        plan[b2] = (plan[b2] & ~(0xffull << (8 * r))) | (byte << (8 * r));
        nxt[b2] = anc[b2] < 0 ? -1 : anc[anc[b2]];
      }
      anc = nxt;
    }
    for (int b2 = 0; b2 < n; ++b2) if (anc[b2] >= 0) return false;
    return true;
  };
  std::vector<int> par_full(parent, parent + nb), par_ik(nb_ik);
  for (int i = 0; i < nb_ik; ++i) par_ik[i] = keep[i] == 0 ? -1 : bmap[parent[keep[i]]];
  std::vector<u64> fkanc, fkanc_full;
  int fkrounds = 0, fkrounds_full = 0;
  if (!fk_plan(par_ik, fkanc, fkrounds) || !fk_plan(par_full, fkanc_full, fkrounds_full)) { set_err(m, "tree too deep"); return GMR_EUNSUPPORTED; }
  (void)maxdepth;
  // ---- structurally non-zero off-diagonal pairs of H: (i, j) with dof j strictly above dof i ----
  std::vector<unsigned short> hpair;
  for (int i = 0; i < n_act; ++i)
    for (int j = 0; j < i; ++j)
      if ((aanc[i] >> j) & 1ull) hpair.push_back((unsigned short)((i << 8) | j));
  // ---- structured QP layout ("core + limbs"): the active dofs are split into an upward-closed core C (root + trunk) and the
  //      connected pieces left when C is removed (limbs), which never couple with one another.  Four 16-lane groups each hold
  //      one bin of limbs (rows 0 .. 15-|C|) and a copy of the core (rows 16-|C| .. 15); see box_qp_struct in ik_kernel.hip.h.
  std::vector<signed char> sq_gdof(64, -1), sq_owner(64, 0);
  std::vector<int> sq_lane_of_dof(64, 0), sq_diag(64, 0);
  std::vector<unsigned> sq_dst;
  int sq_ok = 0, sq_nlimb = 0;
  {
    std::vector<int> dparent(n_act, -1), nanc(n_act, 0);
    for (int i = 0; i < n_act; ++i) nanc[i] = __builtin_popcountll(aanc[i]);
    for (int i = 0; i < n_act; ++i)
      for (int j = 0; j < i; ++j)
        if (((aanc[i] >> j) & 1ull) && (dparent[i] < 0 || nanc[j] > nanc[dparent[i]])) dparent[i] = j;
    std::vector<char> in_core(n_act, 0);
    for (int k = 0; k < n_root; ++k) in_core[k] = akind[k] != 7;  // null dofs are one-row "limbs": they need no core column
    std::vector<std::vector<int>> bins;
    for (int guard = 0; guard < 64 && !sq_ok; ++guard) {
      const int nc = (int)std::count(in_core.begin(), in_core.end(), (char)1);
      if (nc > 10) break;
      // components of the non-core dofs (dofs are indexed ancestors-first, so a parent is labelled before its children)
      std::vector<int> comp(n_act, -1), csize, ctop;
      for (int i = 0; i < n_act; ++i) {
        if (in_core[i]) continue;
        if (dparent[i] < 0 || in_core[dparent[i]]) { comp[i] = (int)csize.size(); csize.push_back(1); ctop.push_back(i); }
        else { comp[i] = comp[dparent[i]]; csize[comp[i]]++; }
      }
      const int cap = 16 - nc;
      int worst = -1;
      for (size_t c = 0; c < csize.size(); ++c)
        if (csize[c] > cap && (worst < 0 || csize[c] > csize[worst])) worst = (int)c;
      bool packed = worst < 0;
      if (packed) {  // first-fit decreasing into <= 4 bins
        std::vector<int> order(csize.size());
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return csize[a] > csize[b]; });
        bins.assign(4, {});
        std::vector<int> fill(4, 0);
        for (int c : order) {
          int g = 0;
          while (g < 4 && fill[g] + csize[c] > cap) ++g;
          if (g == 4) { packed = false; break; }
          fill[g] += csize[c];
          for (int i = 0; i < n_act; ++i) if (comp[i] == c) bins[g].push_back(i);
        }
        if (!packed) for (size_t c = 0; c < csize.size(); ++c) if (worst < 0 || csize[c] > csize[worst]) worst = (int)c;
      }
      if (packed) { sq_ok = 1; sq_nlimb = cap; break; }
      in_core[ctop[worst]] = 1;  // grow the core by the top dof of the largest piece and try again
    }
    if (sq_ok) {
      const int nl = sq_nlimb;
      std::vector<int> core;
      for (int i = 0; i < n_act; ++i) if (in_core[i]) core.push_back(i);
      std::vector<int> grp(n_act, 0), loc(n_act, 0);
      for (int g = 0; g < 4; ++g) {
        std::sort(bins[g].begin(), bins[g].end());
        for (size_t a = 0; a < bins[g].size(); ++a) {
          const int i = bins[g][a], L = 16 * g + (int)a;
          grp[i] = g; loc[i] = (int)a; sq_gdof[L] = (signed char)i; sq_owner[L] = 1; sq_lane_of_dof[i] = L;
        }
        for (size_t c = 0; c < core.size(); ++c) {
          const int L = 16 * g + nl + (int)c;
          sq_gdof[L] = (signed char)core[c]; sq_owner[L] = g == 0;
          if (g == 0) { sq_lane_of_dof[core[c]] = L; loc[core[c]] = nl + (int)c; grp[core[c]] = 0; }
        }
      }
      for (int i = 0; i < n_act; ++i) sq_diag[i] = loc[i] * 64 + sq_lane_of_dof[i];
      auto hs = [](int lane, int col) { return (unsigned)(col * 64 + lane); };
      for (unsigned short pr : hpair) {
        const int i = pr >> 8, j = pr & 0xff;  // j above i
        unsigned a, b;
        if (in_core[i]) { a = hs(sq_lane_of_dof[i], loc[j]); b = hs(sq_lane_of_dof[j], loc[i]); }  // both core: group 0
        else {
          const int g = grp[i], Li = sq_lane_of_dof[i];
          const int Lj = in_core[j] ? 16 * g + loc[j] : sq_lane_of_dof[j];  // the copy of the core dof inside i's group
          if (!in_core[j] && grp[j] != g) { set_err(m, "internal: structured QP partition"); return GMR_EINVAL; }
          a = hs(Li, loc[j]); b = hs(Lj, loc[i]);
        }
        sq_dst.push_back(a | (b << 16));
      }
    }
  }
  // ---- composites per table: dofs sharing the same set of tasks below them share one 6x6 block ----
  std::vector<int> acomp(2 * 64, 0);
  std::vector<unsigned> compmask(2 * 2 * GMR_MAX_TASKS, 0u), comp_own(64, 0u), comp_kids(64, 0u);
  int ncomp[2] = {0, 0};
  for (int k = 0; k < 2; ++k) {
    std::map<unsigned, int> ids;
    for (int i = 0; i < n_act; ++i) {
      unsigned mk = 0;
      for (int t = 0; t < h.ntask[k]; ++t)
        if (akind[i] != 6 || above(abody[i], tbody[k * GMR_MAX_TASKS + t])) mk |= 1u << t;
      auto it = ids.find(mk);
      if (it == ids.end()) {
        it = ids.emplace(mk, ncomp[k]).first;
        compmask[k * 2 * GMR_MAX_TASKS + ncomp[k]] = mk;
        ++ncomp[k];
      }
      acomp[k * 64 + i] = it->second;
    }
    if (ncomp[k] > 2 * GMR_MAX_TASKS) { set_err(m, "too many composite nodes"); return GMR_EUNSUPPORTED; }
    // renumber the composites so that a composite's children (maximal strict subsets) always have lower ids, and split
    // each into "own tasks" + "child composites": Bc[c] = sum_{t in own} Bt[t] + sum_{d in kids} Bc[d]  (any number of either:
    // the pass plan below splits a long sum into entries of four sources)
    const int nc = ncomp[k];
    if (nc > 32) { set_err(m, "more than 32 composite nodes"); return GMR_EUNSUPPORTED; }
    std::vector<int> order(nc), newid(nc);
    std::iota(order.begin(), order.end(), 0);
    auto mask_of = [&](int c) { return compmask[k * 2 * GMR_MAX_TASKS + c]; };
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return __builtin_popcount(mask_of(a)) < __builtin_popcount(mask_of(b)); });
    std::vector<unsigned> sorted_mask(nc);
    for (int o = 0; o < nc; ++o) { newid[order[o]] = o; sorted_mask[o] = mask_of(order[o]); }
    for (int i = 0; i < n_act; ++i) acomp[k * 64 + i] = newid[acomp[k * 64 + i]];
    for (int o = 0; o < nc; ++o) {
      compmask[k * 2 * GMR_MAX_TASKS + o] = sorted_mask[o];
      unsigned own = sorted_mask[o], kids = 0, covered = 0;
      for (int d = o - 1; d >= 0; --d) {
        const unsigned md = sorted_mask[d];
        if (md == 0 || md == sorted_mask[o] || (md & ~sorted_mask[o]) != 0 || (md & covered) != 0) continue;
        kids |= 1u << d; covered |= md;
      }
      own &= ~covered;
      comp_own[k * 32 + o] = own; comp_kids[k * 32 + o] = kids;
    }
  }
  const int ns = h.nslot;
  // ---- LDS layout of the IK kernel (doubles).  Lifetimes inside one solve:
  //   poses (FK .. screws) | Bt (task blocks .. composites) | Bc (composites .. F)   -> Bc overwrites the dead poses
  //   S, F (screws / F .. H assembly)                                                 -> the factorisation's broadcast rows Lb overwrite S
  //   H (H assembly .. QP) overwrites [Bt | poses/Bc]
  const int ntmax = std::max(h.ntask[0], h.ntask[1]), ncmax = std::max(ncomp[0], ncomp[1]);
  const bool sq = sq_ok && !m->force_generic;
  gmr::LdsLayout &L = m->lay;  // (named L here; the composite plan below needs the byte offsets)
  auto even = [](int x) { return (x + 1) & ~1; };
  int o = 0;
  L.zero = o; o += gmr::kBT;  // a block of zeros: the absent sources of the composite plan (never aliased)
  const int npairp_host = ((int)hpair.size() + 127) / 128 * 128;
  // H pair plan, 8 bytes per entry, staged once per wavefront -- unless the structured kernel keeps all of it in registers
  const bool hplan_in_regs = sq && npairp_host <= 64 * gmr::kHPlanRegsSQ;
  L.hplan = o; o += hplan_in_regs ? 0 : npairp_host;
  L.q = o; o += even(nq);
  L.tp = o; o += even(3 * ns);
  L.tq = o; o += 4 * ns;
  L.bodyc = GMR_IK_STAGE_TREE ? o : -1; o += GMR_IK_STAGE_TREE ? even(gmr::kBodyC * nb_ik) : 0;
  L.S = o; L.Lb = o; o += std::max(6 * nvp, sq ? 128 : 2 * (nvp + 2));
  L.F = o; o += 6 * nvp;
  L.B = o; L.H = o;
  const int bt = even(gmr::kBT * ntmax), px = std::max(even(7 * nb_ik), even(gmr::kBT * ncmax));
  L.xpos = o + bt; L.xquat = L.xpos + even(3 * nb_ik); L.Bc = o + bt;
  o += std::max(bt + px, (sq ? 1024 : nvp * nvp) + 2);  // + a dummy slot for the unused lanes of the pair rounds
  L.cplan = o;  // composite plan, 16 bytes per (table, pass, quarter-wave); sized once the passes are scheduled (below)
  L.total_doubles = o;
  m->lds_bytes = o * (int)sizeof(double);
  m->nvp = nvp;
  m->n_act = n_act;
  // ---- composite plan: passes of <= 4 entries, an entry = one composite summed from <= 4 blocks, children before parents ----
  if (m->lds_bytes > 65535) { set_err(m, "model needs %d bytes of LDS per wavefront (plan offsets are 16 bit)", m->lds_bytes); return GMR_EUNSUPPORTED; }
  std::vector<uint32_t> comp_plan((size_t)2 * gmr::kMaxCompPass * 4 * 4, 0u);
  int ncpass[2] = {0, 0};
  for (int k = 0; k < 2; ++k) {
    struct Entry { int dst; std::vector<int> src; std::vector<int> deps; int done_pass = -1; int height = 0; };
    // src / dst are block ids: task t -> t, composite c -> 64 + c;  deps: entries that must have run in an earlier pass
    std::vector<Entry> ent;
    std::vector<int> last_entry_of_comp(ncomp[k], -1);
    // Shallow plan: a composite of ONE task is that task's block (no entry, the dof lanes read the task block itself); a
    // composite of up to four tasks is summed from the task blocks directly (no dependency on its children); only larger ones
    // build on child composites.  G1: 3 dependent passes instead of 5.
    std::vector<int> alias_task(ncomp[k], -1);
    for (int c = 0; c < ncomp[k]; ++c) {
      const unsigned mk = compmask[k * 2 * GMR_MAX_TASKS + c];
      if (__builtin_popcount(mk) == 1) alias_task[c] = __builtin_ctz(mk);
    }
    for (int i = 0; i < n_act; ++i) {
      const int c = acomp[k * 64 + i];
      acomp[k * 64 + i] = alias_task[c] >= 0 ? alias_task[c] : ntmax + c;  // block index from Bt on (Bc = Bt + kBT ntmax)
    }
    for (int c = 0; c < ncomp[k]; ++c) {
      if (alias_task[c] >= 0) continue;
      const unsigned mk = compmask[k * 2 * GMR_MAX_TASKS + c];
      std::vector<int> srcs;
      if (__builtin_popcount(mk) <= 4) {
        for (unsigned t = mk; t; t &= t - 1) srcs.push_back(__builtin_ctz(t));
      } else {
        for (unsigned own = comp_own[k * 32 + c]; own; own &= own - 1) srcs.push_back(__builtin_ctz(own));
        for (unsigned kids = comp_kids[k * 32 + c]; kids; kids &= kids - 1) {
          const int d = __builtin_ctz(kids);
          srcs.push_back(alias_task[d] >= 0 ? alias_task[d] : 64 + d);
        }
      }
      // child composites first: the entries that wait for them should be few
      std::stable_sort(srcs.begin(), srcs.end(), [](int a, int b) { return (a >= 64) > (b >= 64); });
      size_t i = 0;
      bool first = true;
      while (i < srcs.size() || first) {
        Entry e;
        e.dst = 64 + c;
        if (!first) { e.src.push_back(64 + c); e.deps.push_back(last_entry_of_comp[c]); }
        while (i < srcs.size() && e.src.size() < 4) {
          const int sidx = srcs[i++];
          e.src.push_back(sidx);
          if (sidx >= 64) e.deps.push_back(last_entry_of_comp[sidx - 64]);
        }
        last_entry_of_comp[c] = (int)ent.size();
        ent.push_back(std::move(e));
        first = false;
      }
    }
    const int ne = (int)ent.size();
    // height = longest chain of dependants above an entry (critical-path priority)
    for (int i = ne - 1; i >= 0; --i)
      for (int d : ent[i].deps) ent[d].height = std::max(ent[d].height, ent[i].height + 1);
    int done = 0, pass = 0;
    const int boff = m->lay.B * 8, coff = m->lay.Bc * 8, zoff = m->lay.zero * 8, scratch = m->lay.F * 8;  // F is dead during this phase
    auto block_off = [&](int id) { return id >= 64 ? coff + gmr::kBT * 8 * (id - 64) : boff + gmr::kBT * 8 * id; };
    while (done < ne) {
      if (pass >= gmr::kMaxCompPass) { set_err(m, "composite plan of table %d needs more than %d passes", k + 1, gmr::kMaxCompPass); return GMR_EUNSUPPORTED; }
      int pick[4] = {-1, -1, -1, -1};
      for (int slot = 0; slot < 4; ++slot) {
        int best = -1;
        for (int i = 0; i < ne; ++i) {
          if (ent[i].done_pass >= 0 || i == pick[0] || i == pick[1] || i == pick[2]) continue;
          bool ready = true;
          for (int d : ent[i].deps) ready = ready && ent[d].done_pass >= 0 && ent[d].done_pass < pass;
          if (ready && (best < 0 || ent[i].height > ent[best].height)) best = i;
        }
        pick[slot] = best;
      }
      if (pick[0] < 0) { set_err(m, "internal: composite plan stalled"); return GMR_EINVAL; }
      for (int quarter = 0; quarter < 4; ++quarter) {
        uint32_t *row = comp_plan.data() + ((size_t)(k * gmr::kMaxCompPass + pass) * 4 + quarter) * 4;
        const int ei = pick[quarter];
        uint32_t so[4] = {(uint32_t)zoff, (uint32_t)zoff, (uint32_t)zoff, (uint32_t)zoff}, dst = (uint32_t)scratch;
        if (ei >= 0) {
          for (size_t j = 0; j < ent[ei].src.size(); ++j) so[j] = (uint32_t)block_off(ent[ei].src[j]);
          dst = (uint32_t)block_off(ent[ei].dst);
        }
        row[0] = so[0] | (so[1] << 16); row[1] = so[2] | (so[3] << 16); row[2] = dst; row[3] = 0;
      }
      for (int slot = 0; slot < 4; ++slot)
        if (pick[slot] >= 0) { ent[pick[slot]].done_pass = pass; ++done; }
      ++pass;
    }
    ncpass[k] = pass;
  }
  if (getenv("GMR_DEBUG_PLAN")) fprintf(stderr, "gmr: composite plan: %d / %d composites, %d / %d passes\n", ncomp[0], ncomp[1], ncpass[0], ncpass[1]);
  m->lay.total_doubles += 8 * (ncpass[0] + ncpass[1]);
  m->lds_bytes = m->lay.total_doubles * (int)sizeof(double);
  if (m->lds_bytes > 65535) { set_err(m, "model needs %d bytes of LDS per wavefront (plan offsets are 16 bit)", m->lds_bytes); return GMR_EUNSUPPORTED; }
  // ---- H assembly plan: per pair the LDS byte offsets of S_j, F_i and of the two entries of H it fills ----
  std::vector<uint32_t> hplan;
  {
    const gmr::LdsLayout &Ly = m->lay;
    const int hsize = sq ? 1024 : nvp * nvp;  // the two doubles after H are the dummy slots of the padding entries
    for (size_t p = 0; p < hpair.size(); ++p) {
      const int i = hpair[p] >> 8, j = hpair[p] & 0xff;
      const unsigned d0 = sq ? (sq_dst[p] & 0xffffu) : (unsigned)(j * nvp + i), d1 = sq ? (sq_dst[p] >> 16) : (unsigned)(i * nvp + j);
      hplan.push_back((uint32_t)((Ly.S + 6 * j) * 8) | ((uint32_t)((Ly.F + 6 * i) * 8) << 16));
      hplan.push_back((uint32_t)((Ly.H + (int)d0) * 8) | ((uint32_t)((Ly.H + (int)d1) * 8) << 16));
    }
    while ((hplan.size() / 2) % 128 != 0) {
      hplan.push_back((uint32_t)(Ly.S * 8) | ((uint32_t)(Ly.F * 8) << 16));
      hplan.push_back((uint32_t)((Ly.H + hsize) * 8) | ((uint32_t)((Ly.H + hsize + 1) * 8) << 16));
    }
    // Order of the entries = (round, lane) that executes them.  The gathers of S_j / F_i are ds_read_b128 (serviced in four
    // fixed 16-lane groups, a lane occupying the 16-byte slot (addr / 16) mod 16; equal addresses broadcast) and the two
    // scatters are ds_write_b64 (16 contiguous lanes per group, unit (addr / 8) mod 16): entries are swapped between positions
    // while that lowers the number of extra LDS cycles (hill climbing with a fixed seed -- the plan is a pure function of the
    // model).  Measured on G1: SQ_LDS_BANK_CONFLICT of the H phase (profiles/experiment_log_r01_r02.md, v11).
    {
      const int ne = (int)hplan.size() / 2, nr = ne / 64;
      static const int rgroup[64] = {0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1,
                                     2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 3, 2, 2, 2, 2, 3, 3, 3, 3, 2, 2, 2, 2, 2, 2, 2, 2, 3, 3, 3, 3};
      std::vector<int> at(ne);  // position -> entry
      for (int i = 0; i < ne; ++i) at[i] = i;
      // extra cycles of one 16-lane group: max over slots of the number of distinct addresses on it, minus one
      auto extra = [&](const int *pos, int shift, int word, int half) {
        unsigned addr[16]; int slot[16];
        for (int l = 0; l < 16; ++l) {
          const uint32_t wv = hplan[2 * at[pos[l]] + word];
          addr[l] = half ? (wv >> 16) : (wv & 0xffffu);
          slot[l] = (int)(addr[l] >> shift) & 15;
        }
        int worst = 1;
        for (int sl = 0; sl < 16; ++sl) {
          int nd = 0; unsigned seen[16];
          for (int l = 0; l < 16; ++l) {
            if (slot[l] != sl) continue;
            bool dup = false;
            for (int t = 0; t < nd; ++t) dup = dup || seen[t] == addr[l];
            if (!dup) seen[nd++] = addr[l];
          }
          worst = std::max(worst, nd);
        }
        return worst - 1;
      };
      auto round_cost = [&](int r) {
        int c = 0;
        for (int g = 0; g < 4; ++g) {
          int rp[16], wp[16], n = 0;
          for (int l = 0; l < 64; ++l) if (rgroup[l] == g) rp[n++] = 64 * r + l;
          for (int l = 0; l < 16; ++l) wp[l] = 64 * r + 16 * g + l;
          c += 3 * (extra(rp, 4, 0, 0) + extra(rp, 4, 0, 1));  // three b128 reads each of S_j and F_i
          c += extra(wp, 3, 1, 0) + extra(wp, 3, 1, 1);
        }
        return c;
      };
      std::vector<int> rc(nr);
      for (int r = 0; r < nr; ++r) rc[r] = round_cost(r);
      int total0 = 0;
      for (int r = 0; r < nr; ++r) total0 += rc[r];
      uint64_t rng = 0x9E3779B97F4A7C15ull;
      auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (uint32_t)(rng >> 32); };
      for (int iter = 0; iter < 40000 && nr > 0; ++iter) {
        const int a = (int)(next() % (uint32_t)ne), b = (int)(next() % (uint32_t)ne);
        if (a == b) continue;
        const int ra = a / 64, rb = b / 64;
        std::swap(at[a], at[b]);
        const int na = round_cost(ra), nb2 = ra == rb ? na : round_cost(rb);
        const int before = rc[ra] + (ra == rb ? 0 : rc[rb]), after = na + (ra == rb ? 0 : nb2);
        if (after <= before) { rc[ra] = na; rc[rb] = nb2; }
        else std::swap(at[a], at[b]);
      }
      std::vector<uint32_t> ordered(hplan.size());
      for (int i = 0; i < ne; ++i) { ordered[2 * i] = hplan[2 * at[i]]; ordered[2 * i + 1] = hplan[2 * at[i] + 1]; }
      int total = 0;
      for (int r = 0; r < nr; ++r) total += rc[r];
      m->hplan_extra_cycles = total;
      if (getenv("GMR_DEBUG_PLAN")) fprintf(stderr, "gmr: H pair plan %d entries, modelled conflict cycles per solve %d -> %d\n", ne, total0, total);
      hplan.swap(ordered);
    }
  }
  // ---- FK (KinematicsModel convention) tables and branch-slot plan ----
  std::vector<int> dofidx(nb, -1), src_slot(nb, -1), save_slot(nb, -1), last_child(nb, -1), nchild_other(nb, 0);
  std::vector<float> lpos(3 * nb), lrot(4 * nb), jaxis(3 * nb);
  std::vector<double> jaxis64(3 * nb);
  for (int b = 0; b < nb; ++b) {
    if (jtype[b] == GMR_JNT_HINGE) dofidx[b] = qadr[b] - 7;
    for (int i = 0; i < 3; ++i) { lpos[3 * b + i] = (float)bpos[3 * b + i]; jaxis[3 * b + i] = (float)axis[3 * b + i]; jaxis64[3 * b + i] = axis[3 * b + i]; }
    lrot[4 * b + 0] = (float)bquat_raw[4 * b + 1]; lrot[4 * b + 1] = (float)bquat_raw[4 * b + 2];
    lrot[4 * b + 2] = (float)bquat_raw[4 * b + 3]; lrot[4 * b + 3] = (float)bquat_raw[4 * b + 0];
    if (b > 0) { last_child[parent[b]] = b; if (parent[b] != b - 1) nchild_other[parent[b]]++; }
  }
  std::vector<int> free_slots;
  int nslots = 0;
  for (int b = 0; b < nb; ++b) {
    if (b > 0 && parent[b] != b - 1) {
      src_slot[b] = save_slot[parent[b]];
      if (src_slot[b] < 0) { set_err(m, "internal: FK slot plan"); return GMR_EINVAL; }
    }
    if (b > 0 && last_child[parent[b]] == b && save_slot[parent[b]] >= 0) free_slots.push_back(save_slot[parent[b]]);
    if (nchild_other[b] > 0) {
      if (!free_slots.empty()) { save_slot[b] = free_slots.back(); free_slots.pop_back(); }
      else save_slot[b] = nslots++;
    }
  }
  if (nslots > gmr::kFkMaxSlots) { set_err(m, "tree too bushy for the FK kernel (%d branch slots)", nslots); return GMR_EUNSUPPORTED; }

  // ---- pack + upload ----
  std::vector<int> v_parent(parent, parent + nb), v_jtype(jtype, jtype + nb), v_qadr(qadr, qadr + nb);
  std::vector<double> v_bpos(bpos, bpos + 3 * nb), v_bquat(bquat, bquat + 4 * nb), v_axis(axis, axis + 3 * nb), v_qpos0(qpos0, qpos0 + nq);
  std::vector<int> p_jtype(nb_ik), p_qadr(nb_ik);
  std::vector<double> p_bpos(3 * nb_ik), p_bquat(4 * nb_ik), p_axis(3 * nb_ik);
  for (int i = 0; i < nb_ik; ++i) {
    const int b = keep[i];
    p_jtype[i] = jtype[b]; p_qadr[i] = qadr[b];
    for (int c = 0; c < 3; ++c) { p_bpos[3 * i + c] = bpos[3 * b + c]; p_axis[3 * i + c] = axis[3 * b + c]; }
    for (int c = 0; c < 4; ++c) p_bquat[4 * i + c] = bquat[4 * b + c];
  }
  std::vector<int> tbody_ik(tbody), abody_ik(abody);
  for (int &b : tbody_ik) b = bmap[b] < 0 ? 0 : bmap[b];
  for (int &b : abody_ik) b = bmap[b];
  std::vector<double> v_sscale(blob_ptr<double>(B, h.off_slot_scale), blob_ptr<double>(B, h.off_slot_scale) + ns);
  std::vector<double> v_spoff(blob_ptr<double>(B, h.off_slot_pos_off), blob_ptr<double>(B, h.off_slot_pos_off) + 3 * ns);
  std::vector<double> v_sroff(blob_ptr<double>(B, h.off_slot_rot_off), blob_ptr<double>(B, h.off_slot_rot_off) + 4 * ns);
  std::vector<int> v_sfoot(blob_ptr<int32_t>(B, h.off_slot_is_foot), blob_ptr<int32_t>(B, h.off_slot_is_foot) + ns);
  abody_ik.resize(64, 0); abody.resize(64, 0); akind.resize(64, 0); aqadr.resize(64, 0); alim.resize(64, 0); arange.resize(128, 0.0);

  gmr::DevModel &dm = m->dm;
  dm = gmr::DevModel{};
  dm.nbody = nb_ik; dm.nq = nq; dm.nv = nv; dm.nslot = ns; dm.root_slot = h.root_slot; dm.n_act = n_act;
  {  // both tables used and task t of either table ties the same robot body to the same human slot: stage 2 inherits the residual
    bool same = h.use_table[0] && h.use_table[1] && h.ntask[0] == h.ntask[1] && h.ntask[0] > 0;
    for (int t = 0; same && t < h.ntask[0]; ++t)
      same = tbody_ik[t] == tbody_ik[GMR_MAX_TASKS + t] && tslot[t] == tslot[GMR_MAX_TASKS + t];
    dm.same_tasks = same ? 1 : 0;
  }
  dm.root_planar = root_mask == 0x23;
  dm.root_tslot = -1;  // GMR_INIT_ROOT_TARGET: the slot whose prepared target the floating base (body 0) is asked to track
  for (int k = 0; k < 2 && dm.root_tslot < 0; ++k)
    for (int t = 0; h.use_table[k] && t < h.ntask[k] && dm.root_tslot < 0; ++t)
      if (tbody[k * GMR_MAX_TASKS + t] == 0) dm.root_tslot = tslot[k * GMR_MAX_TASKS + t];
  for (int k = 0; k < 2; ++k) { dm.ntask[k] = h.ntask[k]; dm.use_table[k] = h.use_table[k] && h.ntask[k] > 0; dm.ncomp[k] = ncomp[k]; dm.ncpass[k] = ncpass[k]; }
  dm.npairp = (int)hplan.size() / 2; dm.fkrounds = fkrounds; dm.sq_ok = sq_ok; dm.sq_nlimb = sq_nlimb;
  bool fits = true;
  auto put = [&](auto &dst, const auto &src) {  // vector -> fixed-capacity array of the device struct
    using D = std::remove_reference_t<decltype(dst[0])>;
    static_assert(sizeof(D) % sizeof(src[0]) == 0, "element size mismatch");
    if (src.size() * sizeof(src[0]) > sizeof(dst)) { fits = false; return; }
    if (!src.empty()) memcpy(&dst[0], src.data(), src.size() * sizeof(src[0]));
  };
  put(dm.jtype, p_jtype); put(dm.qadr, p_qadr);
  put(dm.bpos, p_bpos); put(dm.bquat, p_bquat); put(dm.axis, p_axis); put(dm.qpos0, v_qpos0);
  put(dm.sscale, v_sscale); put(dm.spoff, v_spoff); put(dm.sroff, v_sroff); put(dm.sfoot, v_sfoot);
  put(dm.tbody, tbody_ik); put(dm.tslot, tslot); put(dm.twp, twp); put(dm.twr, twr);
  put(dm.abody, abody_ik); put(dm.akind, akind); put(dm.aqadr, aqadr); put(dm.alimited, alim);
  put(dm.arange, arange); put(dm.acomp, acomp);
  put(dm.hplan, hplan); put(dm.fkanc, fkanc); put(dm.comp_plan, comp_plan);
  put(dm.sq_gdof, sq_gdof); put(dm.sq_owner, sq_owner); put(dm.sq_lane_of_dof, sq_lane_of_dof); put(dm.sq_diag, sq_diag);
  // the evaluation model: same tables, full body tree (gmr_evaluate returns xpos / xquat of every body)
  std::vector<gmr::DevModel> dm_eval_v(1, dm);
  {
    gmr::DevModel &de = dm_eval_v[0];
    de.nbody = nb; de.fkrounds = fkrounds_full;
    put(de.jtype, v_jtype); put(de.qadr, v_qadr);
    put(de.bpos, v_bpos); put(de.bquat, v_bquat); put(de.axis, v_axis);
    put(de.tbody, tbody); put(de.abody, abody); put(de.fkanc, fkanc_full);
  }
  if (!fits) { set_err(m, "internal: a model table exceeds its fixed capacity"); return GMR_EUNSUPPORTED; }
  {  // LDS layout of eval_kernel: state, targets, poses of the full tree
    gmr::LdsLayout &E = m->lay_eval;
    E = gmr::LdsLayout{};
    int oe = 0;
    E.q = oe; oe += even(nq);
    E.tp = oe; oe += even(3 * ns);
    E.tq = oe; oe += 4 * ns;
    E.xpos = oe; oe += even(3 * nb);
    E.xquat = oe; oe += 4 * nb;
    E.total_doubles = oe;
    m->lds_bytes_eval = oe * (int)sizeof(double);
  }

  Packer P;
  const size_t o_dm = P.add(std::vector<gmr::DevModel>(1));
  const size_t o_dm_eval = P.add(dm_eval_v);
  const size_t o_parent = P.add(v_parent);
  const size_t o_dofidx = P.add(dofidx), o_src = P.add(src_slot), o_save = P.add(save_slot);
  const size_t o_lpos = P.add(lpos), o_lrot = P.add(lrot), o_jaxis = P.add(jaxis), o_jaxis64 = P.add(jaxis64);
  std::vector<gmr::FkBody> fkbody(nb);
  for (int b = 0; b < nb; ++b) {
    gmr::FkBody &r = fkbody[b];
    r = gmr::FkBody{};
    r.src_slot = src_slot[b]; r.dofidx = dofidx[b]; r.save_slot = save_slot[b];
    for (int i = 0; i < 3; ++i) { r.lpos[i] = lpos[3 * b + i]; r.axis[i] = jaxis64[3 * b + i]; }
    for (int i = 0; i < 4; ++i) r.lrot[i] = lrot[4 * b + i];
  }
  const size_t o_fkbody = P.add(fkbody);
  // kin_ops tables: hinge -> body, float32 limits, tree level of every body, bodies by level
  const int ndof_k = nq - 7;
  std::vector<int> k_dof_body(ndof_k, 0);
  std::vector<float> k_lo(ndof_k, 0.f), k_hi(ndof_k, 0.f);
  for (int b = 0; b < nb; ++b)
    if (dofidx[b] >= 0 && dofidx[b] < ndof_k) { k_dof_body[dofidx[b]] = b; k_lo[dofidx[b]] = (float)range[2 * b]; k_hi[dofidx[b]] = (float)range[2 * b + 1]; }
  std::vector<uint8_t> k_depth(nb, 0), k_order(nb, 0);
  int k_maxd = 0;
  for (int b = 1; b < nb; ++b) { k_depth[b] = (uint8_t)(k_depth[parent[b]] + 1); k_maxd = std::max<int>(k_maxd, k_depth[b]); }
  std::iota(k_order.begin(), k_order.end(), (uint8_t)0);
  std::stable_sort(k_order.begin(), k_order.end(), [&](uint8_t a, uint8_t b) { return k_depth[a] < k_depth[b]; });
  const size_t o_kdb = P.add(k_dof_body), o_klo = P.add(k_lo), o_khi = P.add(k_hi), o_kdepth = P.add(k_depth),
               o_korder = P.add(k_order);

  HIP_TRY(m, hipMalloc(&m->dev, P.buf.size()));
  m->dev_bytes = P.buf.size();
  const uint8_t *D = static_cast<const uint8_t *>(m->dev);
#define DP(T, off) reinterpret_cast<const T *>(D + (off))
  m->dm_dev = DP(gmr::DevModel, o_dm);
  m->dm_eval_dev = DP(gmr::DevModel, o_dm_eval);
  gmr::FkTree &fk = m->fk;
  fk.parent = DP(int, o_parent); fk.dofidx = DP(int, o_dofidx); fk.src_slot = DP(int, o_src); fk.save_slot = DP(int, o_save);
  fk.lpos = DP(float, o_lpos); fk.lrot = DP(float, o_lrot); fk.jaxis = DP(float, o_jaxis); fk.jaxis64 = DP(double, o_jaxis64); fk.body = DP(gmr::FkBody, o_fkbody);
  fk.nbody = nb; fk.ndof = nq - 7; fk.nslots = nslots;
  m->kin.dof_body = DP(int, o_kdb); m->kin.lim_lo = DP(float, o_klo); m->kin.lim_hi = DP(float, o_khi);
  m->kin.depth = DP(uint8_t, o_kdepth); m->kin.order = DP(uint8_t, o_korder);
  m->kin.max_depth = k_maxd;
  fk.dof_in_order = 1;
  for (int b = 0, prev = -1; b < nb; ++b)
    if (dofidx[b] >= 0) { if (dofidx[b] < prev) fk.dof_in_order = 0; prev = dofidx[b]; }
#undef DP
  memcpy(P.buf.data() + o_dm, &dm, sizeof(dm));
  HIP_TRY(m, hipMemcpy(m->dev, P.buf.data(), P.buf.size(), hipMemcpyHostToDevice));
  // branch slots + dof tile + position / rotation stages (fk_kernel.hip.h)
  m->fk_lds_bytes = (std::max(1, nslots) * 7 + gmr::kFkPosStride + gmr::kFkRotStride) * gmr::kFkThreads * (int)sizeof(float);
  m->fk_lds_bytes_min = std::max(1, nslots) * 7 * gmr::kFkThreads * (int)sizeof(float);

  if (m->lds_bytes > 160 * 1024) { set_err(m, "model needs %d bytes of LDS per wavefront", m->lds_bytes); return GMR_EUNSUPPORTED; }
  // opt in to > 64 KiB of dynamic LDS where a variant needs it
#define GMR_LDS_OPT_IN(k) hipFuncSetAttribute(reinterpret_cast<const void *>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#ifdef GMR_IK_DEV_ONLY36
#define GMR_X(v) GMR_LDS_OPT_IN((gmr::ik_kernel<v, true>)) GMR_LDS_OPT_IN((gmr::ik_probe_kernel<v, true>)) GMR_LDS_OPT_IN((gmr::ik_group_kernel<v, true>)) GMR_LDS_OPT_IN((gmr::ik_session_kernel<v, true>))
#else
#define GMR_X(v)                                                                                                                         \
  GMR_LDS_OPT_IN((gmr::ik_kernel<v, false>)) GMR_LDS_OPT_IN((gmr::ik_kernel<v, true>)) GMR_LDS_OPT_IN((gmr::ik_probe_kernel<v, false>))       \
  GMR_LDS_OPT_IN((gmr::ik_probe_kernel<v, true>)) GMR_LDS_OPT_IN((gmr::ik_group_kernel<v, false>)) GMR_LDS_OPT_IN((gmr::ik_group_kernel<v, true>)) \
  GMR_LDS_OPT_IN((gmr::ik_session_kernel<v, false>)) GMR_LDS_OPT_IN((gmr::ik_session_kernel<v, true>))
#endif
  GMR_FOR_EACH_NVP(GMR_X)
#undef GMR_X
  GMR_LDS_OPT_IN(gmr::eval_kernel)
  GMR_LDS_OPT_IN(gmr::fk_pos_kernel<1>) GMR_LDS_OPT_IN(gmr::fk_pos_kernel<2>)
  GMR_LDS_OPT_IN(gmr::fk_kernel<0>) GMR_LDS_OPT_IN(gmr::fk_kernel<1>)
#undef GMR_LDS_OPT_IN
  (void)hipGetLastError();
  return GMR_OK;
}

}  // namespace

extern "C" {

int gmr_abi_version(void) { return GMR_ABI_VERSION; }

static gmr_model *model_create_impl(const void *blob, size_t blob_bytes, int device, int min_nvp, int force_generic, char *err, size_t err_len) {
  auto fail = [&](gmr_model *m, const char *msg) -> gmr_model * {
    if (err && err_len) snprintf(err, err_len, "%s", m && !m->err.empty() ? m->err.c_str() : msg);
    if (m) gmr_model_destroy(m);
    return nullptr;
  };
  if (!blob || blob_bytes < sizeof(gmr_blob_header)) return fail(nullptr, "blob too small");
  gmr_blob_header h;
  memcpy(&h, blob, sizeof(h));
  if (h.magic != GMR_BLOB_MAGIC || h.version != GMR_BLOB_VERSION) return fail(nullptr, "bad blob magic/version");
  if (h.total_bytes != blob_bytes) return fail(nullptr, "blob size mismatch");
  if (h.nbody < 1 || h.nbody > GMR_MAX_BODIES) return fail(nullptr, "nbody outside [1, 64]");
  if (h.nq != h.nv + 1 || h.nv < 6) return fail(nullptr, "bad nq/nv");
  if (h.nslot < 0 || h.nslot > GMR_MAX_SLOTS || h.ntask[0] < 0 || h.ntask[0] > GMR_MAX_TASKS || h.ntask[1] < 0 || h.ntask[1] > GMR_MAX_TASKS)
    return fail(nullptr, "slot/task counts out of range");
  if (h.nslot > 0 && (h.root_slot < 0 || h.root_slot >= h.nslot)) return fail(nullptr, "root_slot out of range");
  const size_t nb = h.nbody, ns = h.nslot;
  bool ok = range_ok(h, h.off_parent, nb * 4) && range_ok(h, h.off_jnt_type, nb * 4) && range_ok(h, h.off_qpos_adr, nb * 4) &&
            range_ok(h, h.off_dof_adr, nb * 4) && range_ok(h, h.off_jnt_limited, nb * 4) && range_ok(h, h.off_body_pos, nb * 24) &&
            range_ok(h, h.off_body_quat, nb * 32) && range_ok(h, h.off_body_quat_raw, nb * 32) && range_ok(h, h.off_jnt_axis, nb * 24) &&
            range_ok(h, h.off_jnt_range, nb * 16) && range_ok(h, h.off_qpos0, (size_t)h.nq * 8) && range_ok(h, h.off_slot_scale, ns * 8) &&
            range_ok(h, h.off_slot_pos_off, ns * 24) && range_ok(h, h.off_slot_rot_off, ns * 32) && range_ok(h, h.off_slot_is_foot, ns * 4);
  for (int k = 0; k < 2; ++k)
    ok = ok && range_ok(h, h.off_task_body[k], (size_t)h.ntask[k] * 4) && range_ok(h, h.off_task_slot[k], (size_t)h.ntask[k] * 4) &&
         range_ok(h, h.off_task_wp[k], (size_t)h.ntask[k] * 8) && range_ok(h, h.off_task_wr[k], (size_t)h.ntask[k] * 8);
  if (!ok) return fail(nullptr, "blob array offsets out of range");

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, "no HIP device available (libgmr_amd has no CPU path)");
  if (device < 0 || device >= ndev) return fail(nullptr, "device ordinal out of range");
  gmr_model *m = new gmr_model();
  m->device = device;
  m->h = h;
  m->blob.assign(static_cast<const uint8_t *>(blob), static_cast<const uint8_t *>(blob) + blob_bytes);
  if (hipSetDevice(device) != hipSuccess) return fail(m, "hipSetDevice failed");
  m->pool = scratch_pool(device);  // (nullptr: no pool support on this runtime -- per-call scratch then comes from the device's default pool as it is)
  if (const char *e = getenv("GMR_AMD_GENERIC_QP")) m->force_generic = e[0] == '1';
  if (force_generic) m->force_generic = true;
  m->min_nvp = min_nvp;
  if (const char *e = getenv("GMR_AMD_FK_PARTS")) m->fk_pos_parts = e[0] == '0' ? 0 : e[0] == '2' ? 2 : 1;
  if (build_device_model(m) != GMR_OK) return fail(m, "model build failed");
  return m;
}

gmr_model *gmr_model_create(const void *blob, size_t blob_bytes, int device, char *err, size_t err_len) {
  return model_create_impl(blob, blob_bytes, device, 0, 0, err, err_len);
}

void gmr_model_destroy(gmr_model *m) {
  if (!m) return;
  if (m->device >= 0) (void)hipSetDevice(m->device);
  if (m->dev) (void)hipFree(m->dev);
  if (m->dbg) (void)hipFree(m->dbg);
  delete m;
}

const char *gmr_last_error(const gmr_model *m) { return m ? m->err.c_str() : "null model"; }

int gmr_model_info_get(const gmr_model *m, gmr_model_info *out) {
  if (!m || !out) return GMR_EINVAL;
  memset(out, 0, sizeof(*out));
  out->nbody = m->h.nbody; out->nq = m->h.nq; out->nv = m->h.nv; out->nslot = m->h.nslot;
  out->ntask[0] = m->h.ntask[0]; out->ntask[1] = m->h.ntask[1];
  out->n_active_dof = m->n_act; out->nv_padded = m->nvp; out->lds_bytes = m->lds_bytes; out->device = m->device;
  out->reserved[0] = m->dm.sq_ok && !m->force_generic ? 16 - m->dm.sq_nlimb : 0;  // core size of the structured QP, 0 = generic QP
  return GMR_OK;
}

// Validate one model's batch, put its scheduling data (length-sorted work items, the caller's index of each, the slot columns)
// into stream-ordered scratch and fill the launch arguments.  `order_host` receives the caller's index of every sorted item.
static int prepare_ik_launch(gmr_model *m, const void *human_pos, const void *human_quat, int in_dtype, int n_cols, const int32_t *slot_col,
                             int64_t n_frames, const gmr_work_item *items, int n_items, const gmr_ik_params *params,
                             const double *qpos_init, double *qpos_final, double *qpos_out, int32_t *iters_out, int32_t *frames_done,
                             gmr_ik_stats *stats, hipStream_t st, CallScratch &sc, gmr::IkLaunch &L, std::vector<gmr_work_item> &sorted,
                             bool keep_order = false) {
  if (m->h.nslot == 0 || (m->h.ntask[0] == 0 && m->h.ntask[1] == 0)) { set_err(m, "model has no IK config"); return GMR_ENOCONFIG; }
  if (!human_pos || !human_quat || !slot_col || !params || !qpos_out || (!items && n_items > 0)) { set_err(m, "null argument"); return GMR_EINVAL; }
  if (in_dtype != GMR_DTYPE_F32 && in_dtype != GMR_DTYPE_F64) { set_err(m, "in_dtype must be f32 or f64"); return GMR_EINVAL; }
  if (n_cols <= 0 || n_frames < 0 || n_items < 0) { set_err(m, "negative size"); return GMR_EINVAL; }
  if (params->max_iter < 0 || !(params->damping > 0.0)) { set_err(m, "damping must be > 0 (H must be positive definite) and max_iter >= 0"); return GMR_EINVAL; }
  for (int s = 0; s < m->h.nslot; ++s)
    if (slot_col[s] < 0 || slot_col[s] >= n_cols) { set_err(m, "slot_col[%d]=%d outside [0,%d)", s, slot_col[s], n_cols); return GMR_EINVAL; }
  int64_t tot = 0, out = 0;
  bool need_init = false, need_final = false;
  for (int i = 0; i < n_items; ++i) {
    const gmr_work_item &w = items[i];
    if (w.n_burn < 0 || w.n_out < 0 || w.frame_begin < 0 || w.frame_begin + w.n_burn + w.n_out > n_frames) {
      set_err(m, "work item %d covers frames outside [0,%lld)", i, (long long)n_frames);
      return GMR_EINVAL;
    }
    if (w.check_stride < 0 || (w.check_stride > 0 && (w.burn_row < 0 || w.final_row < 0 || w.n_burn != 0))) {
      set_err(m, "work item %d: a verification walk (check_stride) needs burn_row, final_row and n_burn = 0", i);
      return GMR_EINVAL;
    }
    if (w.init_row < GMR_INIT_ROOT_TARGET || !(w.height_scale >= 0.0) || !std::isfinite(w.height_scale)) {
      set_err(m, "work item %d: init_row must be a row, GMR_INIT_QPOS0 or GMR_INIT_ROOT_TARGET, and height_scale finite and >= 0", i);
      return GMR_EINVAL;
    }
    tot += w.n_burn + w.n_out; out += w.n_out;
    need_init |= w.init_row >= 0; need_final |= w.final_row >= 0 || w.burn_row >= 0;
  }
  if (need_init && !qpos_init) { set_err(m, "items reference qpos_init but it is NULL"); return GMR_EINVAL; }
  if (need_final && !qpos_final) { set_err(m, "items reference qpos_final but it is NULL"); return GMR_EINVAL; }
  if (stats) { memset(stats, 0, sizeof(*stats)); stats->n_items = n_items; stats->n_frames_total = tot; stats->n_frames_out = out; }
  L = gmr::IkLaunch{};
  L.n_items = n_items;
  sorted.clear();
  if (n_items == 0) return GMR_OK;

  HIP_TRY(m, hipSetDevice(m->device));
  // longest item first so that the tail of the grid is made of short ones
  std::vector<int> order(n_items);
  std::iota(order.begin(), order.end(), 0);
  if (!keep_order)  // (an ordered launch brings its own order: gmr_ik_solve_ordered)
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return items[a].n_burn + items[a].n_out > items[b].n_burn + items[b].n_out; });
  sorted.resize(n_items);
  for (int i = 0; i < n_items; ++i) sorted[i] = items[order[i]];
  const size_t items_bytes = sizeof(gmr_work_item) * (size_t)n_items, col_bytes = sizeof(int32_t) * (size_t)m->h.nslot;
  const size_t order_off = (items_bytes + 15) & ~size_t(15), order_bytes = sizeof(int) * (size_t)n_items;
  const size_t col_off = (order_off + order_bytes + 15) & ~size_t(15);
  int rc = scratch_alloc(m, sc, col_off + col_bytes, st);
  if (rc != GMR_OK) return rc;
  uint8_t *ws = static_cast<uint8_t *>(sc.p);
  // pageable-host copies are staged by the runtime before returning, so the vectors may die after the call
  HIP_TRY(m, hipMemcpyAsync(ws, sorted.data(), items_bytes, hipMemcpyHostToDevice, st));
  HIP_TRY(m, hipMemcpyAsync(ws + order_off, order.data(), order_bytes, hipMemcpyHostToDevice, st));
  HIP_TRY(m, hipMemcpyAsync(ws + col_off, slot_col, col_bytes, hipMemcpyHostToDevice, st));

  L.hpos = human_pos; L.hquat = human_quat; L.slot_col = reinterpret_cast<const int *>(ws + col_off);
  L.items = reinterpret_cast<const gmr_work_item *>(ws);
  L.order = reinterpret_cast<const int *>(ws + order_off); L.frames_done = frames_done;
  L.qinit = qpos_init; L.qfinal = qpos_final; L.qout = qpos_out; L.iters = iters_out;
  L.in_f64 = in_dtype == GMR_DTYPE_F64; L.n_cols = n_cols; L.n_items = n_items; L.prm = *params;
#ifdef GMR_IK_STAMPS
  if (!m->dbg) { HIP_TRY(m, hipMalloc(&m->dbg, 16 * sizeof(unsigned long long))); HIP_TRY(m, hipMemset(m->dbg, 0, 16 * sizeof(unsigned long long))); }
#endif
  L.dbg = m->dbg;
  return GMR_OK;
}

int gmr_ik_solve(gmr_model *m, const void *human_pos, const void *human_quat, int in_dtype, int n_cols, const int32_t *slot_col,
                 int64_t n_frames, const gmr_work_item *items, int n_items, const gmr_ik_params *params, const double *qpos_init,
                 double *qpos_final, double *qpos_out, int32_t *iters_out, int32_t *frames_done, gmr_ik_stats *stats, void *stream) {
  if (!m) return GMR_EINVAL;
  m->err.clear();
  hipStream_t st = static_cast<hipStream_t>(stream);
  CallScratch sc;
  gmr::IkLaunch L{};
  std::vector<gmr_work_item> sorted;
  int rc = prepare_ik_launch(m, human_pos, human_quat, in_dtype, n_cols, slot_col, n_frames, items, n_items, params, qpos_init, qpos_final,
                             qpos_out, iters_out, frames_done, stats, st, sc, L, sorted);
  if (rc != GMR_OK || n_items == 0) return rc;
  return launch_ik_variant(m, L, st);
}

// ---- launch order by predicted cost.  Items of equal length still differ in cost (solves per frame: 1.15 max / mean on the
// bench mix), and with only a few items per wavefront slot the order they start in decides how long the last ones run alone:
// 8192 x 3000 frames take 608 ms in array order and 549 ms most-expensive-first.  Lengths being equal, the only predictor is the
// clip itself: gmr_ik_plan_order solves the first probe_frames of every item for their cost alone (nothing else is written)
// and sorts on the device; gmr_ik_solve_ordered then runs the full items in that order.  Everything stays on the stream.
int gmr_ik_plan_order(gmr_model *m, const void *human_pos, const void *human_quat, int in_dtype, int n_cols, const int32_t *slot_col,
                      int64_t n_frames, const gmr_work_item *items, int n_items, const gmr_ik_params *params, const double *qpos_init,
                      int probe_frames, int32_t *order_out, void *stream) {
  if (!m) return GMR_EINVAL;
  m->err.clear();
  if (!order_out || probe_frames < 1) { set_err(m, "order_out is NULL or probe_frames < 1"); return GMR_EINVAL; }
  if (n_items < 0 || (!items && n_items > 0)) { set_err(m, "null argument"); return GMR_EINVAL; }
  hipStream_t st = static_cast<hipStream_t>(stream);
  std::vector<gmr_work_item> probe(items, items + n_items);
  std::vector<int> meta(3 * (size_t)std::max(n_items, 1));  // [cost | frames | probed]
  for (int i = 0; i < n_items; ++i) {
    gmr_work_item &w = probe[i];
    if (w.check_stride != 0) { set_err(m, "work item %d: verification walks cannot be probed", i); return GMR_EINVAL; }
    if (w.n_burn < 0 || w.n_out < 0) { set_err(m, "work item %d has a negative frame count", i); return GMR_EINVAL; }
    const int total = w.n_burn + w.n_out, p = std::min(total, probe_frames);
    meta[n_items + i] = total; meta[2 * n_items + i] = p;
    w.n_burn = p; w.n_out = 0; w.final_row = -1; w.burn_row = -1;  // frames solved, nothing written
  }
  if (n_items == 0) return GMR_OK;
  HIP_TRY(m, hipSetDevice(m->device));
  CallScratch sc, ms;
  gmr::IkLaunch L{};
  std::vector<gmr_work_item> sorted;
  double dummy_out = 0.0;  // (prepare_ik_launch insists on an output array; a probe item has no output frame to write)
  int rc = prepare_ik_launch(m, human_pos, human_quat, in_dtype, n_cols, slot_col, n_frames, probe.data(), n_items, params, qpos_init, nullptr,
                             &dummy_out, nullptr, nullptr, nullptr, st, sc, L, sorted);
  if (rc != GMR_OK) return rc;
  rc = scratch_alloc(m, ms, sizeof(int) * meta.size(), st);
  if (rc != GMR_OK) return rc;
  int *meta_dev = static_cast<int *>(ms.p);
  HIP_TRY(m, hipMemcpyAsync(meta_dev, meta.data(), sizeof(int) * meta.size(), hipMemcpyHostToDevice, st));
  L.qout = nullptr;
  L.cost = meta_dev;
  rc = launch_ik_variant(m, L, st, /*probe=*/true);
  if (rc != GMR_OK) return rc;
  hipLaunchKernelGGL(gmr::plan_order_kernel, dim3(1), dim3(1024), 0, st, meta_dev, meta_dev + n_items, meta_dev + 2 * n_items, n_items, order_out);
  HIP_TRY(m, hipGetLastError());
  return GMR_OK;
}

int gmr_ik_solve_ordered(gmr_model *m, const void *human_pos, const void *human_quat, int in_dtype, int n_cols, const int32_t *slot_col,
                         int64_t n_frames, const gmr_work_item *items, int n_items, const gmr_ik_params *params, const double *qpos_init,
                         double *qpos_final, double *qpos_out, int32_t *iters_out, int32_t *frames_done, gmr_ik_stats *stats,
                         const int32_t *launch_order, void *stream) {
  if (!m) return GMR_EINVAL;
  m->err.clear();
  if (!launch_order && n_items > 0) { set_err(m, "launch_order is NULL"); return GMR_EINVAL; }
  hipStream_t st = static_cast<hipStream_t>(stream);
  CallScratch sc;
  gmr::IkLaunch L{};
  std::vector<gmr_work_item> sorted;
  int rc = prepare_ik_launch(m, human_pos, human_quat, in_dtype, n_cols, slot_col, n_frames, items, n_items, params, qpos_init, qpos_final,
                             qpos_out, iters_out, frames_done, stats, st, sc, L, sorted, /*keep_order=*/true);
  if (rc != GMR_OK || n_items == 0) return rc;
  L.perm = launch_order;
  return launch_ik_variant(m, L, st);
}

// ------------------------------------------------------------------ several models in one launch (BASELINE config 4)
gmr_group *gmr_group_create(const void *const *blobs, const size_t *blob_bytes, int n_models, int device, char *err, size_t err_len) {
  auto fail = [&](gmr_group *g, const char *msg) -> gmr_group * {
    if (err && err_len && msg) snprintf(err, err_len, "%s", msg);
    if (g) gmr_group_destroy(g);
    return nullptr;
  };
  if (!blobs || !blob_bytes || n_models < 1 || n_models > 64) return fail(nullptr, "a group holds 1 .. 64 models");
  gmr_group *g = new gmr_group();
  g->device = device;
  // pass 1: every model as it would be built alone -> the common kernel variant; pass 2: rebuild the members that differ
  int nvp = 0;
  bool all_sq = true;
  for (int i = 0; i < n_models; ++i) {
    gmr_model *m = model_create_impl(blobs[i], blob_bytes[i], device, 0, 0, err, err_len);
    if (!m) return fail(g, nullptr);
    g->models.push_back(m);
    if (m->h.nslot == 0 || (m->h.ntask[0] == 0 && m->h.ntask[1] == 0)) return fail(g, "every group member needs an IK config");
    nvp = std::max(nvp, m->nvp);
    all_sq = all_sq && m->dm.sq_ok && !m->force_generic;
  }
  for (int i = 0; i < n_models; ++i) {
    gmr_model *m = g->models[i];
    const bool sq = m->dm.sq_ok && !m->force_generic;
    if (m->nvp == nvp && sq == all_sq) continue;
    gmr_model_destroy(m);
    g->models[i] = model_create_impl(blobs[i], blob_bytes[i], device, nvp, all_sq ? 0 : 1, err, err_len);
    if (!g->models[i]) return fail(g, nullptr);
    if (g->models[i]->nvp != nvp) return fail(g, "internal: group member did not take the common kernel variant");
  }
  g->nvp = nvp;
  g->sq = all_sq;
  return g;
}

void gmr_group_destroy(gmr_group *g) {
  if (!g) return;
  for (gmr_model *m : g->models) gmr_model_destroy(m);
  delete g;
}

int gmr_group_size(const gmr_group *g) { return g ? (int)g->models.size() : 0; }
gmr_model *gmr_group_model(gmr_group *g, int i) { return g && i >= 0 && i < (int)g->models.size() ? g->models[i] : nullptr; }
const char *gmr_group_last_error(const gmr_group *g) { return g ? g->err.c_str() : "null group"; }

int gmr_group_ik_solve(gmr_group *g, const gmr_group_input *inputs, const gmr_ik_params *params, void *stream) {
  if (!g || !inputs || !params) return GMR_EINVAL;
  g->err.clear();
  const int n = (int)g->models.size();
  hipStream_t st = static_cast<hipStream_t>(stream);
  std::vector<CallScratch> scratch(n);
  std::vector<gmr::IkGroupEntry> entries(n);
  std::vector<std::vector<gmr_work_item>> sorted(n);
  int total = 0, lds_bytes = 0;
  for (int i = 0; i < n; ++i) {
    gmr_model *m = g->models[i];
    m->err.clear();
    const gmr_group_input &in = inputs[i];
    gmr::IkLaunch L{};
    entries[i] = gmr::IkGroupEntry{};
    if (in.n_items == 0) continue;  // no work for this member
    int rc = prepare_ik_launch(m, in.human_pos, in.human_quat, in.in_dtype, in.n_cols, in.slot_col, in.n_frames, in.items, in.n_items, params,
                               in.qpos_init, in.qpos_final, in.qpos_out, in.iters_out, in.frames_done, nullptr, st, scratch[i], L, sorted[i]);
    if (rc != GMR_OK) { g->err = "member " + std::to_string(i) + ": " + m->err; return rc; }
    entries[i].m = m->dm_dev; entries[i].L = L; entries[i].lay = m->lay; entries[i].item_base = 0; entries[i].pad = 0;
    total += in.n_items;
    if (in.n_items > 0) lds_bytes = std::max(lds_bytes, m->lds_bytes);
  }
  if (total == 0) return GMR_OK;
  // one grid over all members' items: a block finds its entry in block_entry[] and its item as blockIdx - item_base, so the
  // blocks of an entry are contiguous (its items longest first); the entry with the longest items goes first
  std::vector<int> block_entry(total);
  std::vector<int> eorder(n);
  std::iota(eorder.begin(), eorder.end(), 0);
  auto longest = [&](int e) { return sorted[e].empty() ? -1 : sorted[e][0].n_burn + sorted[e][0].n_out; };
  std::stable_sort(eorder.begin(), eorder.end(), [&](int a, int b) { return longest(a) > longest(b); });
  int blk = 0;
  for (int e : eorder) {
    entries[e].item_base = blk;
    for (size_t k = 0; k < sorted[e].size(); ++k) block_entry[blk++] = e;
  }
  gmr_model *m0 = g->models[0];
  if (hipSetDevice(g->device) != hipSuccess) { g->err = "hipSetDevice failed"; return GMR_EDEVICE; }
  CallScratch gs;
  const size_t ent_bytes = sizeof(gmr::IkGroupEntry) * (size_t)n, be_off = (ent_bytes + 15) & ~size_t(15);
  int rc = scratch_alloc(m0, gs, be_off + sizeof(int) * (size_t)total, st);
  if (rc != GMR_OK) { g->err = m0->err; return rc; }
  uint8_t *ws = static_cast<uint8_t *>(gs.p);
  if (hipMemcpyAsync(ws, entries.data(), ent_bytes, hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemcpyAsync(ws + be_off, block_entry.data(), sizeof(int) * (size_t)total, hipMemcpyHostToDevice, st) != hipSuccess) {
    g->err = "hipMemcpyAsync failed";
    return GMR_EDEVICE;
  }
  const auto *d_entries = reinterpret_cast<const gmr::IkGroupEntry *>(ws);
  const int *d_be = reinterpret_cast<const int *>(ws + be_off);
  switch (g->nvp) {
#define GMR_X(v) case v: launch_ik_group<v>(g, d_entries, d_be, total, lds_bytes, st); break;
    GMR_FOR_EACH_NVP(GMR_X)
#undef GMR_X
    default: g->err = "internal: no kernel variant"; return GMR_EUNSUPPORTED;
  }
  if (hipGetLastError() != hipSuccess) { g->err = "kernel launch failed"; return GMR_EDEVICE; }
  return GMR_OK;
}

// ------------------------------------------------------------------ single-sequence sessions (teleop)

gmr_session *gmr_session_create(gmr_model *m, int in_dtype, int n_cols, const int32_t *slot_col, const gmr_ik_params *params) {
  if (!m) return nullptr;
  m->err.clear();
  if (m->h.nslot == 0 || (m->h.ntask[0] == 0 && m->h.ntask[1] == 0)) { set_err(m, "model has no IK config"); return nullptr; }
  if (!slot_col || !params || n_cols <= 0 || (in_dtype != GMR_DTYPE_F32 && in_dtype != GMR_DTYPE_F64)) { set_err(m, "bad session argument"); return nullptr; }
  if (params->max_iter < 0 || !(params->damping > 0.0)) { set_err(m, "damping must be > 0 and max_iter >= 0"); return nullptr; }
  for (int s = 0; s < m->h.nslot; ++s)
    if (slot_col[s] < 0 || slot_col[s] >= n_cols) { set_err(m, "slot_col[%d]=%d outside [0,%d)", s, slot_col[s], n_cols); return nullptr; }
  gmr_session *s = new gmr_session();
  s->m = m;
  auto fail = [&](const char *what, hipError_t e) -> gmr_session * {
    set_err(m, "%s failed: %s", what, hipGetErrorString(e));
    gmr_session_destroy(s);
    return nullptr;
  };
  hipError_t e;
  if ((e = hipSetDevice(m->device)) != hipSuccess) return fail("hipSetDevice", e);
  if ((e = hipStreamCreateWithFlags(&s->st, hipStreamNonBlocking)) != hipSuccess) return fail("hipStreamCreate", e);
  const size_t elt = in_dtype == GMR_DTYPE_F64 ? 8 : 4, nq = (size_t)m->h.nq;
  s->pos_bytes = (size_t)n_cols * 3 * elt; s->quat_bytes = (size_t)n_cols * 4 * elt;
  s->quat_off = (s->pos_bytes + 15) & ~size_t(15);
  s->out_off = (s->quat_off + s->quat_bytes + 15) & ~size_t(15);
  const size_t box_off = (s->out_off + (nq + 1) * 8 + 63) & ~size_t(63);
  const size_t host_bytes = box_off + sizeof(gmr::IkSessionBox);
  if ((e = hipHostMalloc(reinterpret_cast<void **>(&s->host), host_bytes, hipHostMallocMapped | hipHostMallocCoherent)) != hipSuccess) return fail("hipHostMalloc", e);
  memset(s->host, 0, host_bytes);
  void *host_dev = nullptr;
  if ((e = hipHostGetDevicePointer(&host_dev, s->host, 0)) != hipSuccess) return fail("hipHostGetDevicePointer", e);
  const size_t item_off = (nq * 8 + 15) & ~size_t(15), col_off = item_off + sizeof(gmr_work_item);
  const size_t ent_off = (col_off + sizeof(int32_t) * (size_t)m->h.nslot + 15) & ~size_t(15);
  if ((e = hipMalloc(reinterpret_cast<void **>(&s->dev), ent_off + 2 * sizeof(gmr::IkGroupEntry))) != hipSuccess) return fail("hipMalloc", e);
  gmr_work_item w{};
  w.frame_begin = 0; w.n_burn = 0; w.n_out = 1; w.init_row = 0; w.final_row = 0; w.burn_row = -1; w.height_scale = 1.0;
  if ((e = hipMemcpy(s->dev + item_off, &w, sizeof(w), hipMemcpyHostToDevice)) != hipSuccess) return fail("hipMemcpy", e);
  if ((e = hipMemcpy(s->dev + col_off, slot_col, sizeof(int32_t) * (size_t)m->h.nslot, hipMemcpyHostToDevice)) != hipSuccess) return fail("hipMemcpy", e);
  uint8_t *hd = static_cast<uint8_t *>(host_dev);
  gmr::IkLaunch &L = s->L;
  L.hpos = hd; L.hquat = hd + s->quat_off;
  L.slot_col = reinterpret_cast<const int *>(s->dev + col_off);
  L.items = reinterpret_cast<const gmr_work_item *>(s->dev + item_off);
  L.qinit = reinterpret_cast<const double *>(s->dev); L.qfinal = reinterpret_cast<double *>(s->dev);
  L.qout = reinterpret_cast<double *>(hd + s->out_off); L.iters = reinterpret_cast<int *>(hd + s->out_off + nq * 8);
  L.in_f64 = in_dtype == GMR_DTYPE_F64; L.n_cols = n_cols; L.n_items = 1; L.prm = *params;
  L.dbg = nullptr;
  s->box = reinterpret_cast<gmr::IkSessionBox *>(s->host + box_off);
  s->box_dev = reinterpret_cast<gmr::IkSessionBox *>(hd + box_off);
  s->entries = reinterpret_cast<gmr::IkGroupEntry *>(s->dev + ent_off);
  gmr::IkGroupEntry ent[2];
  for (int k = 0; k < 2; ++k) {
    ent[k] = gmr::IkGroupEntry{};
    ent[k].m = m->dm_dev; ent[k].L = L; ent[k].L.prm.offset_to_ground = k; ent[k].lay = m->lay;
  }
  if ((e = hipMemcpy(s->entries, ent, sizeof(ent), hipMemcpyHostToDevice)) != hipSuccess) return fail("hipMemcpy", e);
  if (gmr_session_reset(s, nullptr) != GMR_OK) { gmr_session_destroy(s); return nullptr; }
  return s;
}

// Persistent mode: ask the resident wavefront to leave and wait until it has (it also leaves by itself when idle).
static void session_park(gmr_session *s) {
  if (!s->box || !s->st) return;
  __atomic_store_n(&s->box->stop, 1u, __ATOMIC_RELEASE);
  (void)hipStreamSynchronize(s->st);
  __atomic_store_n(&s->box->stop, 0u, __ATOMIC_RELEASE);
  __atomic_store_n(&s->box->exited_gen, s->gen, __ATOMIC_RELEASE);  // (what the wavefront wrote itself, unless it faulted)
}

void gmr_session_destroy(gmr_session *s) {
  if (!s) return;
  if (s->m && s->m->device >= 0) (void)hipSetDevice(s->m->device);
  session_park(s);
  if (s->st) { (void)hipStreamSynchronize(s->st); (void)hipStreamDestroy(s->st); }
  if (s->host) (void)hipHostFree(s->host);
  if (s->dev) (void)hipFree(s->dev);
  delete s;
}

int gmr_session_reset(gmr_session *s, const double *qpos) {
  if (!s) return GMR_EINVAL;
  gmr_model *m = s->m;
  const double *src = qpos ? qpos : blob_ptr<double>(m->blob, m->h.off_qpos0);
  HIP_TRY(m, hipSetDevice(m->device));
  session_park(s);
  HIP_TRY(m, hipStreamSynchronize(s->st));
  HIP_TRY(m, hipMemcpy(s->dev, src, sizeof(double) * (size_t)m->h.nq, hipMemcpyHostToDevice));
  return GMR_OK;
}

int gmr_session_set_persistent(gmr_session *s, int idle_ms) {
  if (!s || idle_ms < 0 || idle_ms > 10000) return GMR_EINVAL;
  gmr_model *m = s->m;
  HIP_TRY(m, hipSetDevice(m->device));
  session_park(s);
  s->idle_ms = idle_ms;
  return GMR_OK;
}

// One frame through the resident wavefront.  (Re)launches it when none is alive: the first frame, or after it left for idleness.
static int session_step_persistent(gmr_session *s, int offset_to_ground) {
  gmr_model *m = s->m;
  gmr::IkSessionBox *box = s->box;
  auto alive = [&]() { return __atomic_load_n(&box->exited_gen, __ATOMIC_ACQUIRE) != s->gen; };
  auto launch = [&]() -> int {
    const unsigned gen = ++s->gen;
    const unsigned long long ticks = 100000ull * (unsigned long long)s->idle_ms;  // s_memrealtime: 100 MHz
    const unsigned max_polls = (unsigned)std::min<unsigned long long>(4000ull * (unsigned long long)s->idle_ms, 0x7fffffffull);
    switch (m->nvp) {
#define GMR_X(v) case v: launch_ik_session<v>(m, s->entries, s->box_dev, ticks, max_polls, 1u << 20, gen, s->st); break;
      GMR_FOR_EACH_NVP(GMR_X)
#undef GMR_X
      default: set_err(m, "internal: no kernel variant for nvp=%d", m->nvp); return GMR_EUNSUPPORTED;
    }
    HIP_TRY(m, hipGetLastError());
    return GMR_OK;
  };
  const unsigned seq = (++s->seq << 1) | (offset_to_ground ? 1u : 0u);  // never equal to the previous word: the frame counter moved
  __atomic_store_n(&box->seq, seq, __ATOMIC_RELEASE);
  if (!alive()) {
    int rc = launch();
    if (rc != GMR_OK) return rc;
  }
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0;; ++spins) {
    if (__atomic_load_n(&box->ack, __ATOMIC_ACQUIRE) == seq) return GMR_OK;
    if (!alive() && __atomic_load_n(&box->ack, __ATOMIC_ACQUIRE) != seq) {
      int rc = launch();  // it left between our post and its last look at the mailbox
      if (rc != GMR_OK) return rc;
    }
    if ((spins & 1023u) == 1023u) {
      if (hipStreamQuery(s->st) != hipErrorNotReady && __atomic_load_n(&box->ack, __ATOMIC_ACQUIRE) != seq && alive()) {  // the kernel is gone without saying so: a fault
        set_err(m, "persistent session kernel ended unexpectedly: %s", hipGetErrorString(hipGetLastError()));
        __atomic_store_n(&box->exited_gen, s->gen, __ATOMIC_RELEASE);
        return GMR_EDEVICE;
      }
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(10)) { set_err(m, "persistent session: no answer within 10 s"); return GMR_EDEVICE; }
    }
  }
}

int gmr_session_step(gmr_session *s, const void *human_pos, const void *human_quat, int offset_to_ground, double *qpos_out,
                     int32_t *solves_out) {
  if (!s) return GMR_EINVAL;
  gmr_model *m = s->m;
  m->err.clear();
  if (!human_pos || !human_quat || !qpos_out) { set_err(m, "null argument"); return GMR_EINVAL; }
  memcpy(s->host, human_pos, s->pos_bytes);
  memcpy(s->host + s->quat_off, human_quat, s->quat_bytes);
  HIP_TRY(m, hipSetDevice(m->device));
  if (s->idle_ms > 0) {
    int rc = session_step_persistent(s, offset_to_ground);
    if (rc != GMR_OK) return rc;
  } else {
    gmr::IkLaunch L = s->L;
    L.prm.offset_to_ground = offset_to_ground ? 1 : 0;
    int rc = launch_ik_variant(m, L, s->st);
    if (rc != GMR_OK) return rc;
    HIP_TRY(m, hipStreamSynchronize(s->st));
  }
  const size_t nq = (size_t)m->h.nq;
  memcpy(qpos_out, s->host + s->out_off, nq * 8);
  if (solves_out) memcpy(solves_out, s->host + s->out_off + nq * 8, sizeof(int32_t));
  return GMR_OK;
}

int gmr_session_state(gmr_session *s, double *qpos_out) {
  if (!s || !qpos_out) return GMR_EINVAL;
  gmr_model *m = s->m;
  HIP_TRY(m, hipSetDevice(m->device));
  session_park(s);
  HIP_TRY(m, hipStreamSynchronize(s->st));
  HIP_TRY(m, hipMemcpy(qpos_out, s->dev, sizeof(double) * (size_t)m->h.nq, hipMemcpyDeviceToHost));
  return GMR_OK;
}

int gmr_evaluate(gmr_model *m, const double *qpos, int64_t n_frames, const void *human_pos, const void *human_quat, int in_dtype,
                 int n_cols, const int32_t *slot_col, int offset_to_ground, const double *height_scale, double *err_out,
                 double *task_err_out, double *xpos_out, double *xquat_out, void *stream) {
  if (!m) return GMR_EINVAL;
  m->err.clear();
  if (!qpos || n_frames < 0) { set_err(m, "null argument / negative size"); return GMR_EINVAL; }
  if (n_frames > 0x7fffffff) { set_err(m, "too many frames for one launch"); return GMR_EINVAL; }
  if (err_out || task_err_out) {
    if (m->h.nslot == 0) { set_err(m, "model has no IK config"); return GMR_ENOCONFIG; }
    if (!human_pos || !human_quat || !slot_col || n_cols <= 0 || (in_dtype != GMR_DTYPE_F32 && in_dtype != GMR_DTYPE_F64)) {
      set_err(m, "err_out needs the human key-points, their dtype and slot_col");
      return GMR_EINVAL;
    }
    for (int s = 0; s < m->h.nslot; ++s)
      if (slot_col[s] < 0 || slot_col[s] >= n_cols) { set_err(m, "slot_col[%d]=%d outside [0,%d)", s, slot_col[s], n_cols); return GMR_EINVAL; }
  }
  if (n_frames == 0) return GMR_OK;
  HIP_TRY(m, hipSetDevice(m->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  gmr::EvalLaunch L{};
  L.qpos = qpos; L.err_out = err_out; L.xpos_out = xpos_out; L.xquat_out = xquat_out; L.n_frames = n_frames;
  L.offset_to_ground = offset_to_ground; L.hscale = height_scale; L.task_err_out = task_err_out;
  CallScratch sc;
  if (err_out || task_err_out) {
    const size_t col_bytes = sizeof(int32_t) * (size_t)m->h.nslot;
    int rc = scratch_alloc(m, sc, col_bytes, st);
    if (rc != GMR_OK) return rc;
    HIP_TRY(m, hipMemcpyAsync(sc.p, slot_col, col_bytes, hipMemcpyHostToDevice, st));
    L.hpos = human_pos; L.hquat = human_quat; L.slot_col = static_cast<const int *>(sc.p);
    L.in_f64 = in_dtype == GMR_DTYPE_F64; L.n_cols = n_cols;
  }
  hipLaunchKernelGGL(gmr::eval_kernel, dim3((unsigned)n_frames), dim3(64), m->lds_bytes_eval, st, m->dm_eval_dev, L, m->lay_eval);
  HIP_TRY(m, hipGetLastError());
  return GMR_OK;
}

/* One FK launch over `fk` (the model's tree, or a copy whose record table is a call's scaled one). */
static int fk_launch(gmr_model *m, const gmr::FkTree &fk, const float *root_pos, const float *root_rot_xyzw, const float *dof, int64_t n_frames,
                     float *body_pos_out, float *body_rot_out, void *stream) {
  const int64_t nblk = (n_frames + gmr::kFkThreads - 1) / gmr::kFkThreads;
  if (nblk > 0x7fffffff) { set_err(m, "too many frames for one launch"); return GMR_EINVAL; }
  // the rotation stage is the last LDS region: a positions-only call does not allocate it (more workgroups per CU)
  if (!body_rot_out && m->fk_pos_parts > 0) {  // positions only: one wavefront per tile, the whole tile image in LDS (fk_pos_kernel)
    const int64_t nw = (n_frames + gmr::kFkWave - 1) / gmr::kFkWave;
    if (nw > 0x7fffffff) { set_err(m, "too many frames for one launch"); return GMR_EINVAL; }
    const int nb = fk.nbody;
    auto image = [&](int parts) { return (std::max(1, fk.nslots) * 7 + (parts == 1 ? 3 * nb : 3 * ((nb + parts - 1) / parts))) * gmr::kFkWave * (int)sizeof(float); };
    int parts = m->fk_pos_parts;
    if (parts == 1 && image(1) > 160 * 1024) parts = 2;  // a tile image beyond the CU's LDS: two half images, else the grouped flush below
    const int lds = image(parts);
    if (lds > 160 * 1024) parts = 0;
    if (parts == 0) goto grouped;
    if (parts == 1)
      hipLaunchKernelGGL((gmr::fk_pos_kernel<1>), dim3((unsigned)nw), dim3(gmr::kFkWave), lds, static_cast<hipStream_t>(stream), fk, root_pos,
                         root_rot_xyzw, dof, n_frames, body_pos_out);
    else
      hipLaunchKernelGGL((gmr::fk_pos_kernel<2>), dim3((unsigned)nw), dim3(gmr::kFkWave), lds, static_cast<hipStream_t>(stream), fk, root_pos,
                         root_rot_xyzw, dof, n_frames, body_pos_out);
    HIP_TRY(m, hipGetLastError());
    return GMR_OK;
  }
grouped : {
  const int fk_lds = body_rot_out ? m->fk_lds_bytes : m->fk_lds_bytes - gmr::kFkRotStride * gmr::kFkThreads * (int)sizeof(float);
  hipLaunchKernelGGL((gmr::fk_kernel<0>), dim3((unsigned)nblk), dim3(gmr::kFkThreads), fk_lds, static_cast<hipStream_t>(stream), fk,
                     root_pos, root_rot_xyzw, dof, n_frames, body_pos_out, body_rot_out, (const int64_t *)nullptr, 0, (int *)nullptr);
  HIP_TRY(m, hipGetLastError());
  return GMR_OK;
}
}

int gmr_fk(gmr_model *m, const float *root_pos, const float *root_rot_xyzw, const float *dof, int64_t n_frames, float *body_pos_out,
           float *body_rot_out, void *stream) {
  return gmr_fk_shape(m, root_pos, root_rot_xyzw, dof, nullptr, 0, n_frames, body_pos_out, body_rot_out, stream);
}

int gmr_fk_shape(gmr_model *m, const float *root_pos, const float *root_rot_xyzw, const float *dof, const float *fitted_shape, int shape_width,
                 int64_t n_frames, float *body_pos_out, float *body_rot_out, void *stream) {
  if (!m) return GMR_EINVAL;
  m->err.clear();
  if (!root_pos || !root_rot_xyzw || !body_pos_out || (!dof && m->h.nq > 7) || n_frames < 0) { set_err(m, "null argument / negative size"); return GMR_EINVAL; }
  if (fitted_shape && shape_width != 1 && shape_width != 3) { set_err(m, "fitted_shape is [nbody] (width 1) or [nbody][3] (width 3)"); return GMR_EINVAL; }
  if (n_frames == 0) return GMR_OK;
  HIP_TRY(m, hipSetDevice(m->device));
  if (!fitted_shape) return fk_launch(m, m->fk, root_pos, root_rot_xyzw, dof, n_frames, body_pos_out, body_rot_out, stream);
  // the call's own scaled copy of the body records, in stream-ordered scratch released behind the launch that reads it
  hipStream_t st = static_cast<hipStream_t>(stream);
  CallScratch sc;
  const int nb = m->fk.nbody;
  int rc = scratch_alloc(m, sc, sizeof(gmr::FkBody) * (size_t)nb, st);
  if (rc != GMR_OK) return rc;
  hipLaunchKernelGGL(gmr::fk_scale_bodies_kernel, dim3((nb + 63) / 64), dim3(64), 0, st, m->fk.body, fitted_shape, shape_width, nb, static_cast<gmr::FkBody *>(sc.p));
  gmr::FkTree fk = m->fk;
  fk.body = static_cast<const gmr::FkBody *>(sc.p);
  return fk_launch(m, fk, root_pos, root_rot_xyzw, dof, n_frames, body_pos_out, body_rot_out, stream);
}

static unsigned kin_grid(int64_t items) {  // grid-stride kernels: enough workgroups of 256 to fill the chip several times over, never more than the work
  const int64_t want = (items + gmr::kKinThreads - 1) / gmr::kKinThreads;
  return (unsigned)std::max<int64_t>(1, std::min<int64_t>(want, 256 * 32));
}

int gmr_dof_to_rot(gmr_model *m, const float *dof, int64_t n_frames, float *joint_rot_out, void *stream) {
  if (!m) return GMR_EINVAL;
  m->err.clear();
  if (n_frames == 0 || (n_frames > 0 && m->fk.nbody < 2)) return GMR_OK;
  if ((!dof && m->fk.ndof > 0) || !joint_rot_out || n_frames < 0) { set_err(m, "null argument / negative size"); return GMR_EINVAL; }
  HIP_TRY(m, hipSetDevice(m->device));
  hipLaunchKernelGGL(gmr::dof_to_rot_kernel, dim3(kin_grid(n_frames * (m->fk.nbody - 1))), dim3(gmr::kKinThreads), 0, static_cast<hipStream_t>(stream), m->fk,
                     dof, n_frames, joint_rot_out);
  HIP_TRY(m, hipGetLastError());
  return GMR_OK;
}

int gmr_rot_to_dof(gmr_model *m, const float *joint_rot, int64_t n_frames, float *dof_out, void *stream) {
  if (!m) return GMR_EINVAL;
  m->err.clear();
  if (n_frames == 0 || (n_frames > 0 && m->fk.ndof == 0)) return GMR_OK;
  if (!joint_rot || !dof_out || n_frames < 0) { set_err(m, "null argument / negative size"); return GMR_EINVAL; }
  HIP_TRY(m, hipSetDevice(m->device));
  hipLaunchKernelGGL(gmr::rot_to_dof_kernel, dim3(kin_grid(n_frames * m->fk.ndof)), dim3(gmr::kKinThreads), 0, static_cast<hipStream_t>(stream), m->fk, m->kin,
                     joint_rot, n_frames, dof_out);
  HIP_TRY(m, hipGetLastError());
  return GMR_OK;
}

int gmr_local_rot_to_global(gmr_model *m, const float *local_rot, int64_t n_frames, float *global_rot_out, void *stream) {
  if (!m) return GMR_EINVAL;
  m->err.clear();
  if (n_frames == 0) return GMR_OK;
  if (!local_rot || !global_rot_out || n_frames < 0) { set_err(m, "null argument / negative size"); return GMR_EINVAL; }
  if (local_rot == global_rot_out) { set_err(m, "local_rot and global_rot_out must not alias"); return GMR_EINVAL; }
  HIP_TRY(m, hipSetDevice(m->device));
  constexpr int P = GMR_KIN_CHAIN_PASSES;
  const int nb = m->fk.nbody, F = 64 * P / nb;
  const int64_t n_batches = (n_frames + F - 1) / F;
  const int lds = 2 * 64 * P * 16;
  const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(n_batches, 256 * 64));
  hipLaunchKernelGGL((gmr::local_to_global_kernel<P>), dim3(grid), dim3(64), lds, static_cast<hipStream_t>(stream), m->fk, m->kin, local_rot, n_frames,
                     global_rot_out);
  HIP_TRY(m, hipGetLastError());
  return GMR_OK;
}

int gmr_fk_min_height(gmr_model *m, const float *root_pos, const float *root_rot_xyzw, const float *dof, const int64_t *seq_offsets,
                      int n_seq, float *min_z_out, void *stream) {
  if (!m) return GMR_EINVAL;
  m->err.clear();
  if (!root_pos || !root_rot_xyzw || !seq_offsets || !min_z_out || n_seq < 0 || (!dof && m->h.nq > 7)) { set_err(m, "null argument / negative size"); return GMR_EINVAL; }
  if (n_seq == 0) return GMR_OK;
  for (int s = 0; s < n_seq; ++s)
    if (seq_offsets[s + 1] < seq_offsets[s] || seq_offsets[0] != 0) { set_err(m, "seq_offsets must start at 0 and be non-decreasing"); return GMR_EINVAL; }
  const int64_t n_frames = seq_offsets[n_seq];
  HIP_TRY(m, hipSetDevice(m->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t off_bytes = sizeof(int64_t) * (size_t)(n_seq + 1), key_off = (off_bytes + 15) & ~size_t(15);
  CallScratch sc;
  int rc = scratch_alloc(m, sc, key_off + sizeof(int) * (size_t)n_seq, st);
  if (rc != GMR_OK) return rc;
  int *keys = reinterpret_cast<int *>(static_cast<uint8_t *>(sc.p) + key_off);
  HIP_TRY(m, hipMemcpyAsync(sc.p, seq_offsets, off_bytes, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(gmr::fk_minkey_init, dim3((n_seq + 255) / 256), dim3(256), 0, st, keys, n_seq);
  if (n_frames > 0) {
    const int64_t nblk = (n_frames + gmr::kFkThreads - 1) / gmr::kFkThreads;
    if (nblk > 0x7fffffff) { set_err(m, "too many frames for one launch"); return GMR_EINVAL; }
    hipLaunchKernelGGL((gmr::fk_kernel<1>), dim3((unsigned)nblk), dim3(gmr::kFkThreads), m->fk_lds_bytes_min, st, m->fk, root_pos, root_rot_xyzw, dof,
                       n_frames, (float *)nullptr, (float *)nullptr, static_cast<const int64_t *>(sc.p), n_seq, keys);
  }
  hipLaunchKernelGGL(gmr::fk_minkey_decode, dim3((n_seq + 255) / 256), dim3(256), 0, st, keys, min_z_out, n_seq);
  HIP_TRY(m, hipGetLastError());
  return GMR_OK;
}

/* Frames per wavefront of the adapter kernels: runs long enough that a wavefront's setup (plan, tables) is amortised and its
 * consecutive output rows complete each other's cache lines, short enough that a small clip still fills the chip. */
static int adapter_chunk(int64_t n_frames, int groups) {
  int64_t c = (n_frames + 16383) / 16384;  /* ~16 k wavefronts = 16 per SIMD on 256 CUs when there is that much work */
  c = std::max<int64_t>(c, 8);
  c = std::min<int64_t>(c, 64);
  c = (c + groups - 1) / groups * groups;
  return (int)c;
}

int gmr_smplx_keypoints_cols(const int32_t *parents, int n_joints, int joints_stride, const double *global_orient, const double *full_pose,
                             const double *joints, int64_t n_frames, int64_t n_frames_out, int resample, const int32_t *out_cols, int n_out,
                             double *pos_out, double *quat_out, void *stream) {
  return gmr_smplx_keypoints_in(parents, n_joints, joints_stride, global_orient, full_pose, joints, GMR_DTYPE_F64, n_frames, n_frames_out, resample,
                                out_cols, n_out, pos_out, quat_out, stream);
}

int gmr_smplx_keypoints_in(const int32_t *parents, int n_joints, int joints_stride, const void *global_orient, const void *full_pose,
                           const void *joints, int in_dtype, int64_t n_frames, int64_t n_frames_out, int resample, const int32_t *out_cols, int n_out,
                           double *pos_out, double *quat_out, void *stream) {
  if (!parents || !global_orient || !full_pose || !joints || !pos_out || !quat_out || n_frames < 0 || n_frames_out < 0) return GMR_EINVAL;
  if (in_dtype != GMR_DTYPE_F32 && in_dtype != GMR_DTYPE_F64) return GMR_EINVAL;
  if (n_joints < 1 || n_joints > gmr::kSmplMaxJoints || joints_stride < n_joints) return GMR_EUNSUPPORTED;
  if (!resample && n_frames_out != n_frames) return GMR_EINVAL;
  if (n_frames_out > 0 && n_frames == 0) return GMR_EINVAL;
  gmr::SmplSkeleton sk{};
  sk.joints_stride = joints_stride; sk.pose_stride = n_joints; sk.resample = resample ? 1 : 0;
  if (parents[0] != -1) return GMR_EINVAL;
  for (int j = 1; j < n_joints; ++j)
    if (parents[j] < 0 || parents[j] >= j) return GMR_EINVAL;
  short col_of[gmr::kSmplMaxJoints];      /* output column of model joint j, -1 = not emitted */
  unsigned char live[gmr::kSmplMaxJoints]; /* emitted, or an ancestor of an emitted joint */
  if (out_cols) {  /* out_cols[c] = joint emitted as column c; ancestors are chained internally */
    if (n_out < 1 || n_out > n_joints) return GMR_EINVAL;
    for (int j = 0; j < n_joints; ++j) { col_of[j] = -1; live[j] = 0; }
    for (int c = 0; c < n_out; ++c) {
      const int j = out_cols[c];
      if (j < 0 || j >= n_joints || col_of[j] >= 0) return GMR_EINVAL;
      col_of[j] = (short)c;
      for (int a = j; a >= 0 && !live[a]; a = parents[a]) live[a] = 1;
    }
    sk.n_out = n_out;
  } else {
    for (int j = 0; j < n_joints; ++j) { col_of[j] = (short)j; live[j] = 1; }
    sk.n_out = n_joints;
  }
  /* the kernel's skeleton: the live joints only, renumbered in order (a parent precedes its children, the root stays first) */
  short cidx[gmr::kSmplMaxJoints];
  int nl = 0;
  for (int j = 0; j < n_joints; ++j) {
    cidx[j] = -1;
    if (!live[j]) continue;
    cidx[j] = (short)nl;
    sk.src[nl] = (short)j;
    sk.out_col[nl] = col_of[j];
    sk.parent[nl] = j == 0 ? (short)-1 : cidx[parents[j]];
    ++nl;
  }
  sk.n_joints = nl;
  if (n_frames_out == 0) return GMR_OK;
  const int chunk = adapter_chunk(n_frames_out, gmr::chain_geom(nl).groups);
  const int64_t nblk = (n_frames_out + chunk - 1) / chunk;
  if (nblk > 0x7fffffff) return GMR_EINVAL;
  if (in_dtype == GMR_DTYPE_F32)
    hipLaunchKernelGGL(gmr::smplx_keypoints_kernel<float>, dim3((unsigned)nblk), dim3(64), 0, static_cast<hipStream_t>(stream), sk,
                       static_cast<const float *>(global_orient), static_cast<const float *>(full_pose), static_cast<const float *>(joints), n_frames,
                       n_frames_out, chunk, pos_out, quat_out);
  else
    hipLaunchKernelGGL(gmr::smplx_keypoints_kernel<double>, dim3((unsigned)nblk), dim3(64), 0, static_cast<hipStream_t>(stream), sk,
                       static_cast<const double *>(global_orient), static_cast<const double *>(full_pose), static_cast<const double *>(joints), n_frames,
                       n_frames_out, chunk, pos_out, quat_out);
  return hipGetLastError() == hipSuccess ? GMR_OK : GMR_EDEVICE;
}

int gmr_smplx_keypoints(const int32_t *parents, int n_joints, int joints_stride, const double *global_orient, const double *full_pose,
                        const double *joints, int64_t n_frames, int64_t n_frames_out, int resample, double *pos_out, double *quat_out,
                        void *stream) {
  return gmr_smplx_keypoints_cols(parents, n_joints, joints_stride, global_orient, full_pose, joints, n_frames, n_frames_out, resample,
                                  nullptr, 0, pos_out, quat_out, stream);
}

int gmr_bvh_parse_header(const char *text, size_t len, int max_joints, char *names_out, size_t names_cap, int32_t *parents_out,
                         double *offsets_out, int32_t *channels_out, int32_t *order_out, int64_t *n_frames_out, double *frame_time_out,
                         size_t *motion_offset_out) {
  if (!text || max_joints <= 0 || !names_out || !parents_out || !offsets_out || !channels_out || !order_out || !n_frames_out ||
      !frame_time_out || !motion_offset_out)
    return -1;
  gmr_bvh::Cursor c{text, text + len};
  gmr_bvh::Header h{};
  h.max_joints = max_joints; h.names = names_out; h.names_cap = names_cap;
  h.parents = parents_out; h.offsets = offsets_out; h.channels = channels_out;
  if (!c.next() || !c.is("HIERARCHY") || !c.next() || !c.is("ROOT")) return -1;
  int rc = gmr_bvh::joint(c, h, -1, 0);
  if (rc) return rc;
  if (h.order[0] < 0) return -1;  // no joint names three rotation channels where the reference looks for the Euler order
  int64_t nf;
  double ft;
  if (!c.next() || !c.is("MOTION") || !c.next() || !c.is("Frames:") || !gmr_bvh::integer(c, nf) || !c.next() || !c.is("Frame") ||
      !c.next() || !c.is("Time:") || !gmr_bvh::number(c, ft))
    return -1;
  const char *p = c.p;  // the motion rows start behind the end of the Frame Time line
  while (p < text + len && *p != '\n') ++p;
  if (p < text + len) ++p;
  for (int i = 0; i < 3; ++i) order_out[i] = h.order[i];
  *n_frames_out = nf; *frame_time_out = ft; *motion_offset_out = (size_t)(p - text);
  return h.n;
}

int64_t gmr_bvh_parse_motion(const char *text, size_t len, int64_t max_lines, double *out, int64_t max_out, int64_t *n_lines,
                             int64_t *n_cols) {
  // Decimal -> double by Clinger's fast path: a mantissa below 2^53 and a power of ten up to 10^22 are both exact doubles, so
  // one multiply / divide gives the correctly rounded result (what Python's float() returns).  Anything else (more than 19
  // digits, huge exponents, inf / nan) goes through strtod.
  static const double p10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16,
                                 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
  if (!text || !out || max_lines < 0 || max_out < 0) return -1;
  const char *p = text, *end = text + len;
  int64_t count = 0, lines = 0, cols0 = -1;
  while (p < end && lines < max_lines) {
    int64_t cols = 0;
    for (;;) {  // one line
      while (p < end && (*p == ' ' || *p == '\t' || *p == '\r')) ++p;
      if (p >= end || *p == '\n') break;
      const char *tok = p;
      bool neg = false;
      if (*p == '-' || *p == '+') { neg = *p == '-'; ++p; }
      uint64_t mant = 0;
      int nd = 0, e10 = 0;
      bool digits = false, fast = true;
      while (p < end && *p >= '0' && *p <= '9') {
        digits = true;
        if (nd < 19) { mant = mant * 10 + (uint64_t)(*p - '0'); if (mant) ++nd; } else fast = false;
        ++p;
      }
      if (p < end && *p == '.') {
        ++p;
        while (p < end && *p >= '0' && *p <= '9') {
          digits = true;
          if (nd < 19) { mant = mant * 10 + (uint64_t)(*p - '0'); if (mant) ++nd; --e10; } else fast = false;  // further digits: below 1 ulp of 19 digits, but be exact
          ++p;
        }
      }
      if (digits && p < end && (*p == 'e' || *p == 'E')) {
        const char *q = p + 1;
        bool eneg = false;
        if (q < end && (*q == '-' || *q == '+')) { eneg = *q == '-'; ++q; }
        if (q < end && *q >= '0' && *q <= '9') {
          int ev = 0;
          while (q < end && *q >= '0' && *q <= '9') { if (ev < 10000) ev = ev * 10 + (*q - '0'); ++q; }
          e10 += eneg ? -ev : ev;
          p = q;
        }
      }
      const bool ends = p >= end || *p == ' ' || *p == '\t' || *p == '\r' || *p == '\n';
      double v;
      if (digits && ends && fast && mant < (1ull << 53) && e10 >= -22 && e10 <= 22) {
        v = (double)mant;
        v = e10 < 0 ? v / p10[-e10] : v * p10[e10];
        if (neg) v = -v;
      } else {  // slow path: let strtod decide (also rejects garbage)
        const char *t = tok;
        while (t < end && !(*t == ' ' || *t == '\t' || *t == '\r' || *t == '\n')) ++t;
        std::string buf(tok, t);
        char *stop = nullptr;
        if (!gmr_bvh::float_token(buf.data(), buf.size())) return -1;  // (what strtod would take but Python's float() does not: hex, nan(...))
        v = strtod(buf.c_str(), &stop);
        if (buf.empty() || stop != buf.c_str() + buf.size()) return -1;
        p = t;
      }
      if (count >= max_out) return -1;
      out[count++] = v;
      ++cols;
    }
    if (p < end && *p == '\n') ++p;
    if (cols == 0) continue;  // blank line
    if (cols0 < 0) cols0 = cols;
    else if (cols != cols0) return -1;
    ++lines;
  }
  if (n_lines) *n_lines = lines;
  if (n_cols) *n_cols = cols0 < 0 ? 0 : cols0;
  return count;
}

int gmr_bvh_parse_motion_device(const char *text, int64_t text_bytes, int n_files, const int64_t *seg_begin, const int64_t *seg_end,
                                const int64_t *n_lines, int64_t n_cols, const int64_t *row_begin, double *rows_out, int32_t *status_out,
                                int64_t *n_tokens_out, int64_t *slow_out, int64_t max_slow, int64_t *n_slow, void *stream) {
  if (!text || !seg_begin || !seg_end || !n_lines || !row_begin || !rows_out || !status_out || !n_slow || n_files < 0 || n_cols < 1 ||
      max_slow < 0 || (max_slow > 0 && !slow_out))
    return GMR_EINVAL;
  *n_slow = 0;
  if (n_files == 0) return GMR_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  std::vector<gmr::TxtFile> files((size_t)n_files);
  int64_t nchunk = 0;
  for (int f = 0; f < n_files; ++f) {
    if (seg_begin[f] < 0 || seg_end[f] < seg_begin[f] || seg_end[f] > text_bytes || n_lines[f] < 0 || row_begin[f] < 0) return GMR_EINVAL;
    gmr::TxtFile &t = files[(size_t)f];
    t.seg_begin = seg_begin[f]; t.seg_end = seg_end[f]; t.chunk0 = nchunk;
    t.tok_limit = n_lines[f] * n_cols; t.out_base = row_begin[f] * n_cols; t.n_tokens = 0; t.status = 0; t.pad = 0;
    nchunk += std::max<int64_t>(1, (seg_end[f] - seg_begin[f] + gmr::kTxtChunk - 1) / gmr::kTxtChunk);  // (an empty block still owns a chunk: chunk -> file stays a search)
  }
  if (nchunk > 0x7fffffff) return GMR_EUNSUPPORTED;
  // one stream-ordered scratch block: file table | per-chunk counts | per-chunk bases | slow list | slow counter
  const size_t o_files = 0, o_counts = (sizeof(gmr::TxtFile) * (size_t)n_files + 15) & ~(size_t)15;
  const size_t o_bases = (o_counts + sizeof(int) * (size_t)nchunk + 15) & ~(size_t)15;
  const size_t o_slow = o_bases + sizeof(int64_t) * (size_t)nchunk;
  const size_t o_cnt = o_slow + sizeof(int64_t) * 3 * (size_t)max_slow;
  const size_t total = o_cnt + 16;
  char *scr = nullptr;
  if (hipMallocAsync((void **)&scr, total, st) != hipSuccess) return GMR_EDEVICE;
  int rc = GMR_OK;
  unsigned long long cnt = 0;
  do {
    if (hipMemcpyAsync(scr + o_files, files.data(), sizeof(gmr::TxtFile) * (size_t)n_files, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemsetAsync(scr + o_cnt, 0, 16, st) != hipSuccess) { rc = GMR_EDEVICE; break; }
    gmr::TxtFile *dfiles = reinterpret_cast<gmr::TxtFile *>(scr + o_files);
    int *counts = reinterpret_cast<int *>(scr + o_counts);
    int64_t *bases = reinterpret_cast<int64_t *>(scr + o_bases);
    const unsigned char *utext = reinterpret_cast<const unsigned char *>(text);
    hipLaunchKernelGGL(gmr::bvh_txt_count_kernel, dim3((unsigned)nchunk), dim3(64), 0, st, utext, dfiles, n_files, counts);
    hipLaunchKernelGGL(gmr::bvh_txt_scan_kernel, dim3((unsigned)n_files), dim3(64), 0, st, dfiles, counts, bases);
    hipLaunchKernelGGL(gmr::bvh_txt_parse_kernel, dim3((unsigned)nchunk), dim3(64), 0, st, utext, dfiles, n_files, bases, n_cols, rows_out,
                       reinterpret_cast<int64_t *>(scr + o_slow), max_slow, reinterpret_cast<unsigned long long *>(scr + o_cnt));
    if (hipGetLastError() != hipSuccess) { rc = GMR_EDEVICE; break; }
    if (hipMemcpyAsync(files.data(), dfiles, sizeof(gmr::TxtFile) * (size_t)n_files, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipMemcpyAsync(&cnt, scr + o_cnt, sizeof(cnt), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) { rc = GMR_EDEVICE; break; }
    const int64_t keep = std::min<int64_t>((int64_t)cnt, max_slow);
    if (keep > 0 && (hipMemcpyAsync(slow_out, scr + o_slow, sizeof(int64_t) * 3 * (size_t)keep, hipMemcpyDeviceToHost, st) != hipSuccess ||
                     hipStreamSynchronize(st) != hipSuccess)) { rc = GMR_EDEVICE; break; }
    *n_slow = (int64_t)cnt;
    for (int f = 0; f < n_files; ++f) {
      const gmr::TxtFile &t = files[(size_t)f];
      status_out[f] = (t.status & 1) | (t.n_tokens < t.tok_limit ? 2 : 0);
      if (n_tokens_out) n_tokens_out[f] = t.n_tokens;
    }
  } while (0);
  (void)hipFreeAsync(scr, st);
  return rc;
}

static int bvh_fk_launch(const int32_t *parents, int n_joints, const int32_t *euler_order, const int32_t *extra_pos_src,
                         const int32_t *extra_rot_src, int n_extra, int layout, const double *pbase, const double *rbase,
                         const double *offsets, int64_t pstride, int64_t rstride, double ang_scale, int64_t n_frames, double scale,
                         const int32_t *out_cols, int n_out, double *pos_out, double *quat_out, void *stream) {
  if (!parents || !euler_order || !pbase || !rbase || !pos_out || !quat_out || n_frames < 0) return GMR_EINVAL;
  if (n_joints < 1 || n_joints > gmr::kBvhMaxJoints || n_extra < 0 || n_extra > gmr::kBvhMaxExtra) return GMR_EUNSUPPORTED;
  if (n_extra > 0 && (!extra_pos_src || !extra_rot_src)) return GMR_EINVAL;
  gmr::BvhSkeleton sk{};
  sk.n_joints = n_joints; sk.n_extra = n_extra; sk.layout = layout;
  for (int i = 0; i < 3; ++i) {
    if (euler_order[i] < 0 || euler_order[i] > 2) return GMR_EINVAL;
    sk.order[i] = euler_order[i];
  }
  if (parents[0] != -1) return GMR_EINVAL;
  for (int j = 0; j < n_joints; ++j) {
    if (j > 0 && (parents[j] < 0 || parents[j] >= j)) return GMR_EINVAL;  // hierarchy order
    sk.parent[j] = (short)parents[j];
  }
  for (int k = 0; k < n_extra; ++k) {
    if (extra_pos_src[k] < 0 || extra_pos_src[k] >= n_joints || extra_rot_src[k] < 0 || extra_rot_src[k] >= n_joints) return GMR_EINVAL;
    sk.extra_pos_src[k] = (short)extra_pos_src[k]; sk.extra_rot_src[k] = (short)extra_rot_src[k];
  }
  const int nb = n_joints + n_extra;
  if (out_cols) {  /* out_cols[c] = entry (joint, or n_joints + extra) emitted as column c */
    if (n_out < 1 || n_out > nb) return GMR_EINVAL;
    for (int j = 0; j < n_joints; ++j) sk.out_col[j] = -1;
    for (int k = 0; k < n_extra; ++k) sk.extra_col[k] = -1;
    for (int c = 0; c < n_out; ++c) {
      const int e = out_cols[c];
      if (e < 0 || e >= nb) return GMR_EINVAL;
      short &slot = e < n_joints ? sk.out_col[e] : sk.extra_col[e - n_joints];
      if (slot >= 0) return GMR_EINVAL;
      slot = (short)c;
    }
    sk.n_out = n_out;
  } else {
    for (int j = 0; j < n_joints; ++j) sk.out_col[j] = (short)j;
    for (int k = 0; k < n_extra; ++k) sk.extra_col[k] = (short)(n_joints + k);
    sk.n_out = nb;
  }
  if (n_frames == 0) return GMR_OK;
  const gmr::ChainGeom geo = gmr::chain_geom(n_joints);
  const int chunk = adapter_chunk(n_frames, geo.groups);
  const int64_t nblk = (n_frames + chunk - 1) / chunk;
  if (nblk > 0x7fffffff) return GMR_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int K = geo.jp <= 64 ? 1 : geo.jp / 64;
#define GMR_BVH_LAUNCH(KK, SS)                                                                                                              \
  hipLaunchKernelGGL((gmr::bvh_fk_kernel<KK, SS>), dim3((unsigned)nblk), dim3(64), lds_bytes, st, sk, pbase, rbase, offsets, (int)pstride, (int)rstride, ang_scale, \
                     n_frames, chunk, scale, pos_out, quat_out)
  const bool s9 = layout == gmr::BVH_ROWS9;
  const int per_frame = (int)(pbase != rbase ? pstride + rstride : pstride);
  const size_t lds_bytes = 2 * (size_t)gmr::bvh_batch(per_frame, geo.groups).doubles * sizeof(double);  // two stage buffers
  if (K == 1) { if (s9) GMR_BVH_LAUNCH(1, true); else GMR_BVH_LAUNCH(1, false); }
  else if (K == 2) { if (s9) GMR_BVH_LAUNCH(2, true); else GMR_BVH_LAUNCH(2, false); }
  else { if (s9) GMR_BVH_LAUNCH(3, true); else GMR_BVH_LAUNCH(3, false); }
#undef GMR_BVH_LAUNCH
  return hipGetLastError() == hipSuccess ? GMR_OK : GMR_EDEVICE;
}

int gmr_bvh_fk(const int32_t *parents, int n_joints, const int32_t *euler_order, const int32_t *extra_pos_src,
               const int32_t *extra_rot_src, int n_extra, const double *local_pos, const double *euler_rad, int64_t n_frames,
               double scale, double *pos_out, double *quat_out, void *stream) {
  return bvh_fk_launch(parents, n_joints, euler_order, extra_pos_src, extra_rot_src, n_extra, gmr::BVH_SPLIT, local_pos, euler_rad, nullptr,
                       3 * (int64_t)n_joints, 3 * (int64_t)n_joints, 1.0, n_frames, scale, nullptr, 0, pos_out, quat_out, stream);
}

int gmr_bvh_fk_rows(const int32_t *parents, int n_joints, const int32_t *euler_order, const int32_t *extra_pos_src,
                    const int32_t *extra_rot_src, int n_extra, int channels, const double *offsets, const double *rows, int64_t n_cols,
                    int64_t n_frames, double scale, const int32_t *out_cols, int n_out, double *pos_out, double *quat_out, void *stream) {
  if (!offsets || n_joints < 1) return GMR_EINVAL;
  int64_t want;
  if (channels == 3) want = 3 + 3 * (int64_t)n_joints;
  else if (channels == 6) want = 6 * (int64_t)n_joints;
  else if (channels == 9) want = 3 + 9 * (int64_t)(n_joints - 1);
  else return GMR_EUNSUPPORTED;
  if (n_cols != want) return GMR_EINVAL;
  return bvh_fk_launch(parents, n_joints, euler_order, extra_pos_src, extra_rot_src, n_extra, channels, rows, rows, offsets, n_cols, n_cols,
                       3.14159265358979323846 / 180.0, n_frames, scale, out_cols, n_out, pos_out, quat_out, stream);
}

#ifdef GMR_IK_STAMPS
/* diagnostic builds only: read and clear the per-phase cycle sums */
int gmr_debug_read_stamps(gmr_model *m, unsigned long long *out16) {
  if (!m || !m->dbg) return GMR_EINVAL;
  HIP_TRY(m, hipDeviceSynchronize());
  HIP_TRY(m, hipMemcpy(out16, m->dbg, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  HIP_TRY(m, hipMemset(m->dbg, 0, 16 * sizeof(unsigned long long)));
  return GMR_OK;
}
#endif

}  // extern "C"
