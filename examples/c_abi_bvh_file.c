/* c_abi_bvh_file.c -- one BVH file to robot qpos from plain C (no Python, no torch, no C++): the file loop body of
 * scripts/bvh_to_robot_dataset.py:75-104 on the C ABI.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include examples/c_abi_bvh_file.c \
 *       -Lgmr_amd/lib -lgmr_amd -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/gmr_amd/lib -Wl,-rpath,/opt/rocm/lib -o c_abi_bvh_file
 *   ./c_abi_bvh_file FILE.bvh DIR
 *
 * DIR holds model.blob (include/gmr_blob.h; a bvh_to_<robot> model) and cols.txt: "nq n_out" on the first line, then one bone name
 * per line -- the bones the IK config consumes, in the solver's order (GeneralMotionRetargeting.ik_columns).  Writes qpos.f64 [T][nq],
 * iters.i32 [T], keypoints.f64 [T][n_out][7] (position + wxyz quaternion as the solver saw them).
 * Steps: gmr_bvh_parse_header (host; read_bvh's hierarchy loop, lafan_vendor/extract.py:60-139) -> the file's bytes to the device ->
 * gmr_bvh_parse_motion_device (the MOTION block, extract.py:140-156; tokens off its exact path are parsed here with
 * gmr_bvh_parse_motion and patched in) -> gmr_bvh_fk_rows (load_lafan1_file, utils/lafan1.py:8-40, only the columns asked for) ->
 * gmr_ik_solve (the retarget loop, one work item for the clip). */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gmr_amd.h"

#define MAXJ 256
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 3; } } while (0)

static void *slurp(const char *path, size_t *bytes) {
  FILE *f = fopen(path, "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  void *p = malloc((size_t)n + 1);
  if (fread(p, 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "short read on %s\n", path); exit(2); }
  fclose(f);
  ((char *)p)[n] = 0;
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
static int find(char names[][64], int n, const char *s) {
  for (int i = 0; i < n; ++i)
    if (strcmp(names[i], s) == 0) return i;
  return -1;
}

int main(int argc, char **argv) {
  if (argc != 3) { fprintf(stderr, "usage: %s FILE.bvh DIR\n", argv[0]); return 2; }
  const char *dir = argv[2];
  if (gmr_abi_version() != GMR_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 2; }
  char path[1024];
  size_t file_bytes, blob_bytes;
  char *text = (char *)slurp(argv[1], &file_bytes);
  snprintf(path, sizeof path, "%s/model.blob", dir);
  void *blob = slurp(path, &blob_bytes);
  snprintf(path, sizeof path, "%s/cols.txt", dir);
  char *cols_txt = (char *)slurp(path, NULL);

  /* ---- HIERARCHY + MOTION header (host) */
  static char names_buf[64 * MAXJ], bone[MAXJ + 8][64];
  static int32_t parents[MAXJ], channels[MAXJ];
  static double offsets[MAXJ][3];
  int32_t order[3];
  int64_t n_frames;
  double frame_time;
  size_t moff;
  int J = gmr_bvh_parse_header(text, file_bytes, MAXJ, names_buf, sizeof names_buf, parents, &offsets[0][0], channels, order, &n_frames,
                               &frame_time, &moff);
  if (J <= 0) { fprintf(stderr, "not a BVH file this loader understands (%d)\n", J); return 2; }
  { const char *p = names_buf; for (int j = 0; j < J; ++j) { strncpy(bone[j], p, 63); p += strlen(p) + 1; } }
  const int ch = channels[J - 1];  /* the LAST joint's count shapes the rows (extract.py:104-106) */
  const int64_t n_cols = ch == 3 ? 3 + 3 * (int64_t)J : ch == 6 ? 6 * (int64_t)J : 3 + 9 * (int64_t)(J - 1);
  /* LeftFootMod / RightFootMod = the foot's position with the toe's orientation (lafan1.py:36-39) */
  int32_t ep[2], er[2];
  int E = 0;
  const char *sides[2] = {"Left", "Right"};
  for (int s = 0; s < 2; ++s) {
    char a[64], b[64];
    snprintf(a, sizeof a, "%sFoot", sides[s]); snprintf(b, sizeof b, "%sToe", sides[s]);
    if (find(bone, J, a) >= 0 && find(bone, J, b) >= 0) {
      ep[E] = find(bone, J, a); er[E] = find(bone, J, b);
      snprintf(bone[J + E], 64, "%sFootMod", sides[s]);
      ++E;
    }
  }
  /* ---- the columns the model consumes */
  int nq = 0, n_out = 0;
  char *line = strtok(cols_txt, "\n");
  if (!line || sscanf(line, "%d %d", &nq, &n_out) != 2 || n_out < 1 || n_out > 64) { fprintf(stderr, "bad cols.txt\n"); return 2; }
  int32_t out_cols[64], slot_col[64];
  for (int c = 0; c < n_out; ++c) {
    line = strtok(NULL, "\n");
    const int e = line ? find(bone, J + E, line) : -1;
    if (e < 0) { fprintf(stderr, "bone %s is not in the file\n", line ? line : "(missing)"); return 2; }
    out_cols[c] = e;
    slot_col[c] = c;  /* the key-point arrays hold exactly the solver's columns, in its order */
  }

  /* ---- device: text -> rows -> key-points -> qpos */
  char err[256];
  gmr_model *m = gmr_model_create(blob, blob_bytes, 0, err, sizeof err);
  if (!m) { fprintf(stderr, "gmr_model_create: %s\n", err); return 3; }
  const int64_t T = n_frames;
  char *d_text;
  double *d_rows, *d_off, *d_pos, *d_quat, *d_qpos;
  int32_t *d_iters;
  HIP_OK(hipMalloc((void **)&d_text, file_bytes + 16));
  HIP_OK(hipMalloc((void **)&d_rows, sizeof(double) * (size_t)(T * n_cols + 1)));
  HIP_OK(hipMalloc((void **)&d_off, sizeof(double) * 3 * (size_t)J));
  HIP_OK(hipMalloc((void **)&d_pos, sizeof(double) * 3 * (size_t)(T * n_out + 1)));
  HIP_OK(hipMalloc((void **)&d_quat, sizeof(double) * 4 * (size_t)(T * n_out + 1)));
  HIP_OK(hipMalloc((void **)&d_qpos, sizeof(double) * (size_t)(T * nq + 1)));
  HIP_OK(hipMalloc((void **)&d_iters, sizeof(int32_t) * (size_t)(T + 1)));
  HIP_OK(hipMemcpy(d_text, text, file_bytes, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_off, offsets, sizeof(double) * 3 * (size_t)J, hipMemcpyHostToDevice));
  const int64_t seg_b = (int64_t)moff, seg_e = (int64_t)file_bytes, row0 = 0;
  int32_t status = 0;
  int64_t n_tok = 0, n_slow = 0;
  static int64_t slow[3 * 4096];
  int rc = gmr_bvh_parse_motion_device(d_text, (int64_t)file_bytes, 1, &seg_b, &seg_e, &T, n_cols, &row0, d_rows, &status, &n_tok, slow, 4096, &n_slow, NULL);
  if (rc != GMR_OK) { fprintf(stderr, "gmr_bvh_parse_motion_device: %d\n", rc); return 3; }
  if (status != 0 || n_slow > 4096) {  /* the host parser decides (and says what is wrong with a malformed file) */
    double *rows = (double *)malloc(sizeof(double) * (size_t)(T * n_cols + 1));
    int64_t nl = 0, nc = 0;
    if (gmr_bvh_parse_motion(text + moff, file_bytes - moff, T, rows, T * n_cols + 1, &nl, &nc) != T * n_cols || nl != T || nc != n_cols) {
      fprintf(stderr, "malformed motion block\n");
      return 2;
    }
    HIP_OK(hipMemcpy(d_rows, rows, sizeof(double) * (size_t)(T * n_cols), hipMemcpyHostToDevice));
    free(rows);
  } else {
    for (int64_t k = 0; k < n_slow; ++k) {  /* tokens off the exact fast path: strtod on the host, patched into the rows */
      const char *tok = text + slow[3 * k + 2];
      size_t len = 0;
      while (tok + len < text + file_bytes && !(tok[len] == ' ' || tok[len] == '\t' || tok[len] == '\r' || tok[len] == '\n')) ++len;
      double v;
      int64_t nl = 0, nc = 0;
      if (gmr_bvh_parse_motion(tok, len, 1, &v, 1, &nl, &nc) != 1) { fprintf(stderr, "bad number in the motion block\n"); return 2; }
      HIP_OK(hipMemcpy(d_rows + slow[3 * k + 1], &v, sizeof v, hipMemcpyHostToDevice));
    }
  }
  rc = gmr_bvh_fk_rows(parents, J, order, E ? ep : NULL, E ? er : NULL, E, ch, d_off, d_rows, n_cols, T, 0.01, out_cols, n_out, d_pos, d_quat, NULL);
  if (rc != GMR_OK) { fprintf(stderr, "gmr_bvh_fk_rows: %d\n", rc); return 3; }
  gmr_work_item item;
  memset(&item, 0, sizeof item);
  item.frame_begin = 0; item.n_burn = 0; item.n_out = (int32_t)T; item.init_row = GMR_INIT_QPOS0; item.final_row = -1; item.burn_row = -1;
  item.check_stride = 0; item.height_scale = 1.0;
  gmr_ik_params prm = {0.5, 1e-3, 0.95, 1.0, 10, 0, 1e-7}; /* the reference's constants */
  rc = gmr_ik_solve(m, d_pos, d_quat, GMR_DTYPE_F64, n_out, slot_col, T, &item, 1, &prm, NULL, NULL, d_qpos, d_iters, NULL, NULL, NULL);
  if (rc != GMR_OK) { fprintf(stderr, "gmr_ik_solve: %s\n", gmr_last_error(m)); return 3; }
  HIP_OK(hipDeviceSynchronize());

  double *qpos = (double *)malloc(sizeof(double) * (size_t)(T * nq + 1)), *kp = (double *)malloc(sizeof(double) * 7 * (size_t)(T * n_out + 1));
  double *hp = (double *)malloc(sizeof(double) * 3 * (size_t)(T * n_out + 1)), *hq = (double *)malloc(sizeof(double) * 4 * (size_t)(T * n_out + 1));
  int32_t *iters = (int32_t *)malloc(sizeof(int32_t) * (size_t)(T + 1));
  HIP_OK(hipMemcpy(qpos, d_qpos, sizeof(double) * (size_t)(T * nq), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(iters, d_iters, sizeof(int32_t) * (size_t)T, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(hp, d_pos, sizeof(double) * 3 * (size_t)(T * n_out), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(hq, d_quat, sizeof(double) * 4 * (size_t)(T * n_out), hipMemcpyDeviceToHost));
  for (int64_t i = 0; i < T * n_out; ++i) { memcpy(kp + 7 * i, hp + 3 * i, 24); memcpy(kp + 7 * i + 3, hq + 4 * i, 32); }
  spill(dir, "qpos.f64", qpos, sizeof(double) * (size_t)(T * nq));
  spill(dir, "iters.i32", iters, sizeof(int32_t) * (size_t)T);
  spill(dir, "keypoints.f64", kp, sizeof(double) * 7 * (size_t)(T * n_out));
  printf("ok: %lld frames, %d joints (%d-channel rows, %lld numbers, %lld off the exact path), %d key-point columns\n", (long long)T, J, ch,
         (long long)n_tok, (long long)n_slow, n_out);
  gmr_model_destroy(m);
  return 0;
}
