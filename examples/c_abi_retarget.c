/* c_abi_retarget.c -- the drop-in boundary used from plain C: no Python, no torch, no C++.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include examples/c_abi_retarget.c \
 *       -Lgmr_amd/lib -lgmr_amd -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/gmr_amd/lib -Wl,-rpath,/opt/rocm/lib -o c_abi_retarget
 *   ./c_abi_retarget DIR
 *
 * DIR holds raw little-endian files (written by tests/test_gpu_api.py::test_c_abi_from_plain_c, or any producer of the same
 * layouts): model.blob (include/gmr_blob.h), meta.txt ("n_frames n_cols nslot nq nbody n_clips"), slot_col.i32, seq_offsets.i64,
 * pos.f32 [N][B][3], quat.f32 [N][B][4].  Writes qpos.f64 [N][nq], iters.i32 [N], body_pos.f32 [N][nbody][3].
 * What it does is the caller loop of scripts/smplx_to_robot_dataset.py:84-112 on the C ABI: one work item per clip through
 * gmr_ik_solve (GeneralMotionRetargeting.retarget per frame), then gmr_fk (KinematicsModel.forward_kinematics) for
 * local_body_pos with zero root position and identity root rotation. */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gmr_amd.h"

static void *slurp(const char *dir, const char *name, size_t *bytes) {
  char path[1024];
  snprintf(path, sizeof path, "%s/%s", dir, name);
  FILE *f = fopen(path, "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  void *p = malloc((size_t)n + 1);
  if (fread(p, 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "short read on %s\n", path); exit(2); }
  fclose(f);
  if (bytes) *bytes = (size_t)n;
  return p;
}
static void spill(const char *dir, const char *name, const void *p, size_t bytes) {
  char path[1024];
  snprintf(path, sizeof path, "%s/%s", dir, name);
  FILE *f = fopen(path, "wb");
  if (!f || fwrite(p, 1, bytes, f) != bytes) { fprintf(stderr, "cannot write %s\n", path); exit(2); }
  fclose(f);
}
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 3; } } while (0)

int main(int argc, char **argv) {
  if (argc != 2) { fprintf(stderr, "usage: %s DIR\n", argv[0]); return 2; }
  const char *dir = argv[1];
  if (gmr_abi_version() != GMR_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 2; }
  size_t blob_bytes, meta_bytes;
  void *blob = slurp(dir, "model.blob", &blob_bytes);
  char *meta = (char *)slurp(dir, "meta.txt", &meta_bytes);
  meta[meta_bytes] = 0;
  long long N;
  int B, nslot, nq, nbody, n_clips;
  if (sscanf(meta, "%lld %d %d %d %d %d", &N, &B, &nslot, &nq, &nbody, &n_clips) != 6) { fprintf(stderr, "bad meta.txt\n"); return 2; }
  int32_t *slot_col = (int32_t *)slurp(dir, "slot_col.i32", NULL);
  int64_t *offs = (int64_t *)slurp(dir, "seq_offsets.i64", NULL);
  float *pos = (float *)slurp(dir, "pos.f32", NULL), *quat = (float *)slurp(dir, "quat.f32", NULL);

  char err[256];
  gmr_model *m = gmr_model_create(blob, blob_bytes, 0, err, sizeof err);
  if (!m) { fprintf(stderr, "gmr_model_create: %s\n", err); return 3; }
  gmr_model_info info;
  gmr_model_info_get(m, &info);
  if (info.nq != nq || info.nbody != nbody || info.nslot != nslot) { fprintf(stderr, "model / meta mismatch\n"); return 2; }

  /* one work item per clip: fresh state (qpos0), frames in order, warm start carried inside the clip */
  gmr_work_item *items = (gmr_work_item *)calloc((size_t)n_clips, sizeof *items);
  for (int c = 0; c < n_clips; ++c) {
    items[c].frame_begin = offs[c];
    items[c].n_out = (int32_t)(offs[c + 1] - offs[c]);
    items[c].init_row = GMR_INIT_QPOS0; items[c].final_row = -1; items[c].burn_row = -1;
  }
  gmr_ik_params prm = {0.5, 1e-3, 0.95, 1.0, 10, 0, 1e-7}; /* the reference's constants */

  void *d_pos, *d_quat; double *d_qpos; int32_t *d_iters; float *d_rp, *d_rr, *d_dof, *d_bp;
  HIP_OK(hipMalloc(&d_pos, (size_t)N * B * 3 * 4)); HIP_OK(hipMalloc(&d_quat, (size_t)N * B * 4 * 4));
  HIP_OK(hipMalloc((void **)&d_qpos, (size_t)N * nq * 8)); HIP_OK(hipMalloc((void **)&d_iters, (size_t)N * 4));
  HIP_OK(hipMemcpy(d_pos, pos, (size_t)N * B * 3 * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_quat, quat, (size_t)N * B * 4 * 4, hipMemcpyHostToDevice));
  gmr_ik_stats st;
  int rc = gmr_ik_solve(m, d_pos, d_quat, GMR_DTYPE_F32, B, slot_col, N, items, n_clips, &prm, NULL, NULL, d_qpos, d_iters, NULL, &st, NULL);
  if (rc != GMR_OK) { fprintf(stderr, "gmr_ik_solve: %d %s\n", rc, gmr_last_error(m)); return 3; }
  HIP_OK(hipDeviceSynchronize());
  double *qpos = (double *)malloc((size_t)N * nq * 8);
  int32_t *iters = (int32_t *)malloc((size_t)N * 4);
  HIP_OK(hipMemcpy(qpos, d_qpos, (size_t)N * nq * 8, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(iters, d_iters, (size_t)N * 4, hipMemcpyDeviceToHost));

  /* local_body_pos: FK with zero root position and identity root rotation (xyzw), float32 dofs */
  const int ndof = nq - 7;
  float *rp = (float *)calloc((size_t)N * 3, 4), *rr = (float *)calloc((size_t)N * 4, 4), *dof = (float *)malloc((size_t)N * ndof * 4);
  for (long long f = 0; f < N; ++f) {
    rr[4 * f + 3] = 1.0f;
    for (int j = 0; j < ndof; ++j) dof[f * ndof + j] = (float)qpos[f * nq + 7 + j];
  }
  HIP_OK(hipMalloc((void **)&d_rp, (size_t)N * 3 * 4)); HIP_OK(hipMalloc((void **)&d_rr, (size_t)N * 4 * 4));
  HIP_OK(hipMalloc((void **)&d_dof, (size_t)N * ndof * 4)); HIP_OK(hipMalloc((void **)&d_bp, (size_t)N * nbody * 3 * 4));
  HIP_OK(hipMemcpy(d_rp, rp, (size_t)N * 3 * 4, hipMemcpyHostToDevice)); HIP_OK(hipMemcpy(d_rr, rr, (size_t)N * 4 * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_dof, dof, (size_t)N * ndof * 4, hipMemcpyHostToDevice));
  rc = gmr_fk(m, d_rp, d_rr, d_dof, N, d_bp, NULL, NULL);
  if (rc != GMR_OK) { fprintf(stderr, "gmr_fk: %d %s\n", rc, gmr_last_error(m)); return 3; }
  HIP_OK(hipDeviceSynchronize());
  float *bp = (float *)malloc((size_t)N * nbody * 3 * 4);
  HIP_OK(hipMemcpy(bp, d_bp, (size_t)N * nbody * 3 * 4, hipMemcpyDeviceToHost));

  spill(dir, "qpos.f64", qpos, (size_t)N * nq * 8);
  spill(dir, "iters.i32", iters, (size_t)N * 4);
  spill(dir, "body_pos.f32", bp, (size_t)N * nbody * 3 * 4);
  printf("ok: %lld frames in %d clips, %lld frames solved\n", N, n_clips, (long long)st.n_frames_out);
  gmr_model_destroy(m);
  hipFree(d_pos); hipFree(d_quat); hipFree(d_qpos); hipFree(d_iters); hipFree(d_rp); hipFree(d_rr); hipFree(d_dof); hipFree(d_bp);
  return 0;
}
