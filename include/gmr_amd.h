/* gmr_amd.h -- C ABI of libgmr_amd.so, the MI355X (gfx950) retarget engine.
 *
 * Plain C: opaque handle, raw pointers, sizes.  No torch / C++ types cross this
 * boundary.  Every pointer documented "device" is a HIP device pointer on the
 * GPU the handle was created for (e.g. torch.Tensor.data_ptr()); "host" pointers
 * are ordinary CPU memory.  The caller owns every buffer; the library owns only
 * the handle (which keeps a copy of the model and a small scheduling workspace).
 * Calls on one handle may be issued on different HIP streams (per-call scheduling
 * data is stream-ordered, nothing is shared between launches); gmr_last_error's
 * buffer is per handle, so concurrent host threads should use one handle each.  All calls return 0 on success and a negative GMR_E* code on
 * failure, with a message available from gmr_last_error().  There is no CPU
 * fallback: without a usable HIP device gmr_model_create fails.
 *
 * What each entry point replaces in the reference (Zudva/GMR):
 *   gmr_model_create   GeneralMotionRetargeting.__init__ + setup_retarget_configuration
 *                      (general_motion_retargeting/motion_retarget.py:13-114): mj.MjModel.from_xml_path,
 *                      mink.Configuration, the two lists of mink.FrameTask;  and
 *                      KinematicsModel.__init__ (kinematics_model.py:69-99)
 *   gmr_ik_solve       the caller loop `for frame in frames: qpos = retargeter.retarget(frame)`
 *                      (scripts/smplx_to_robot_dataset.py:84-89, scripts/bvh_to_robot_dataset.py:95-104)
 *                      i.e. GeneralMotionRetargeting.retarget / update_targets / error1 / error2
 *                      (motion_retarget.py:117-200) and, inside it, mink.solve_ik +
 *                      Configuration.integrate_inplace (call sites motion_retarget.py:147-150,156-159,
 *                      166-169,176-179)
 *   gmr_group_*        several robots' batches in one launch (one GeneralMotionRetargeting per robot in the reference)
 *   gmr_session_*      the live loop of scripts/optitrack_to_robot.py:37-46 / smplx_to_robot.py:103-126 / bvh_to_robot.py:
 *                      one frame in, one qpos out, state carried inside (`retargeter.retarget(frame)` called once per
 *                      captured frame): a latency path next to the throughput path of gmr_ik_solve
 *   gmr_evaluate       error1() / error2() (motion_retarget.py:188-200) and configuration.data.xpos / xquat
 *                      (mink.Configuration.update = mj_kinematics) at given qpos, without solving
 *   gmr_fk             KinematicsModel.forward_kinematics (kinematics_model.py:213-246)
 *   gmr_fk_shape       the same with `fitted_shape` (per-body scale of the local translations, kinematics_model.py:225)
 *   gmr_fk_min_height  the clip-global `torch.min(body_pos[..., 2])` of the height adjust
 *                      (scripts/smplx_to_robot_dataset.py:118-126)
 *   gmr_dof_to_rot     KinematicsModel.dof_to_rot (kinematics_model.py:172-182; Joint.dof_to_rot :21-36)
 *   gmr_rot_to_dof     KinematicsModel.rot_to_dof (kinematics_model.py:184-197; Joint.rot_to_dof :38-53), clamped to the joint limits
 *   gmr_local_rot_to_global  KinematicsModel.convert_local_rot_to_global (kinematics_model.py:199-211)
 *   gmr_smplx_keypoints, gmr_smplx_keypoints_cols, gmr_smplx_keypoints_in  the numeric part of get_smplx_data_offline_fast (general_motion_retargeting/utils/smpl.py:109-198)
 *                      after the SMPL-X body model: slerp/lerp to the target frame rate, orientation chaining
 *   gmr_bvh_parse_header the HIERARCHY section of read_bvh (general_motion_retargeting/utils/lafan_vendor/extract.py:60-139)
 *   gmr_bvh_parse_motion, gmr_bvh_parse_motion_device  the MOTION block of read_bvh (general_motion_retargeting/utils/lafan_vendor/extract.py:140-166): the
 *                      per-line regex + float() loop that dominates BVH loading in the reference
 *   gmr_bvh_fk, gmr_bvh_fk_rows  the numeric part of load_lafan1_file (general_motion_retargeting/utils/lafan1.py:8-40):
 *                      euler_to_quat + quat_fk (utils/lafan_vendor/utils.py:56-103), Y-up -> Z-up, cm -> m,
 *                      LeftFootMod / RightFootMod synthesis
 */
#ifndef GMR_AMD_H
#define GMR_AMD_H

#include <stddef.h>
#include <stdint.h>

#include "gmr_blob.h"

#ifdef __cplusplus
extern "C" {
#endif

#define GMR_ABI_VERSION 5

#define GMR_OK 0
#define GMR_EINVAL (-1)    /* bad argument / blob / shape                     */
#define GMR_EDEVICE (-2)   /* HIP runtime error (no device, launch failure)   */
#define GMR_EUNSUPPORTED (-3) /* model outside the kernel's limits            */
#define GMR_ENOCONFIG (-4) /* IK requested on a model compiled without tasks  */

#define GMR_DTYPE_F32 0
#define GMR_DTYPE_F64 1

typedef struct gmr_model gmr_model;

typedef struct gmr_model_info {
  int32_t nbody, nq, nv, nslot;
  int32_t ntask[2];
  int32_t n_active_dof; /* dofs with at least one task below them (the QP size)   */
  int32_t nv_padded;    /* compile-time system size of the kernel variant chosen  */
  int32_t lds_bytes;    /* dynamic LDS per sequence (one wavefront)               */
  int32_t device;
  int32_t reserved[6];
} gmr_model_info;

/* Per-call statistics written by gmr_ik_solve when stats != NULL (host memory). */
typedef struct gmr_ik_stats {
  int64_t n_items;        /* work items launched (one wavefront each)             */
  int64_t n_frames_total; /* frames processed, burn-in included                   */
  int64_t n_frames_out;   /* frames written                                       */
  int32_t reserved[4];
} gmr_ik_stats;

int gmr_abi_version(void);

/* blob: host pointer to a buffer in the layout of gmr_blob.h.  device: HIP device ordinal.
 * On failure returns NULL and, if err != NULL, writes a NUL-terminated message. */
gmr_model *gmr_model_create(const void *blob, size_t blob_bytes, int device, char *err, size_t err_len);
void gmr_model_destroy(gmr_model *m);
const char *gmr_last_error(const gmr_model *m);
int gmr_model_info_get(const gmr_model *m, gmr_model_info *out);

/* Batched two-stage IK over work items.
 *   human_pos  device, [n_frames][n_cols][3] in_dtype, metres
 *   human_quat device, [n_frames][n_cols][4] in_dtype, wxyz
 *   slot_col   host,   [nslot] column (0..n_cols-1) of each slot of the model
 *   items      host,   [n_items] runs of consecutive frames (gmr_work_item)
 *   qpos_init  device, [*][nq] f64 or NULL (rows referenced by items[].init_row)
 *   qpos_final device, [*][nq] f64 or NULL (rows referenced by items[].final_row)
 *   qpos_out   device, [n_frames][nq] f64; only frames covered by an item's n_out are written
 *   iters_out  device, [n_frames] int32 or NULL: solve_ik calls spent on the frame
 *              (bit 30 set if a QP hit its iteration cap -- never expected; bit 31 if the frame's qpos has a
 *              non-finite coordinate, e.g. from non-finite key-points: callers can check a batch without reading qpos)
 *   frames_done device, [n_items] int32 or NULL: output frames each item solved (n_out unless a
 *              check_stride item stopped early, see gmr_blob.h)
 *   stream     hipStream_t (as void*), NULL = default stream.  The call is
 *              asynchronous with respect to the host.                             */
int gmr_ik_solve(gmr_model *m, const void *human_pos, const void *human_quat, int in_dtype, int n_cols,
                 const int32_t *slot_col, int64_t n_frames, const gmr_work_item *items, int n_items,
                 const gmr_ik_params *params, const double *qpos_init, double *qpos_final, double *qpos_out,
                 int32_t *iters_out, int32_t *frames_done, gmr_ik_stats *stats, void *stream);

/* Launch order by predicted cost.  Work items start in array order (gmr_ik_solve puts longer items first); items of EQUAL
 * length still differ in cost -- solves per frame -- and with a few items per wavefront slot the start order decides how long the
 * last ones run alone (8192 clips x 3000 frames: 608 ms in array order, 549 ms most-expensive-first).
 *   gmr_ik_plan_order     solves the first probe_frames frames of every item for their cost only (nothing but order_out is
 *                         written) and orders the items by probe solves per frame x frames, most expensive first.
 *                         order_out device int32 [n_items]; other arguments as gmr_ik_solve; plain items only (no check_stride)
 *   gmr_ik_solve_ordered  gmr_ik_solve with workgroup b running item launch_order[b] (device int32 [n_items]; must be a
 *                         permutation of 0 .. n_items-1 -- entries outside that range are skipped, a repeated entry leaves another
 *                         item unsolved)
 * Both are asynchronous on `stream`; results are those of gmr_ik_solve bit for bit (the order only moves work in time).        */
int gmr_ik_plan_order(gmr_model *m, const void *human_pos, const void *human_quat, int in_dtype, int n_cols, const int32_t *slot_col,
                      int64_t n_frames, const gmr_work_item *items, int n_items, const gmr_ik_params *params, const double *qpos_init,
                      int probe_frames, int32_t *order_out, void *stream);
int gmr_ik_solve_ordered(gmr_model *m, const void *human_pos, const void *human_quat, int in_dtype, int n_cols, const int32_t *slot_col,
                         int64_t n_frames, const gmr_work_item *items, int n_items, const gmr_ik_params *params, const double *qpos_init,
                         double *qpos_final, double *qpos_out, int32_t *iters_out, int32_t *frames_done, gmr_ik_stats *stats,
                         const int32_t *launch_order, void *stream);

/* Several models in ONE launch (BASELINE config 4, "heterogeneous trees in one launch"): a group owns n models built for one
 * common kernel variant; gmr_group_ik_solve runs every member's work items in a single grid -- each wavefront looks up its
 * member's model, LDS layout and input / output arrays.  The reference analogue is one GeneralMotionRetargeting per robot in
 * the workers of one mp.Pool (scripts/smplx_to_robot_dataset.py:79-83, 241-242).
 *   gmr_group_create   blobs / blob_bytes host [n_models]: one packed model each (gmr_blob.h); NULL + message on failure
 *   gmr_group_model    borrowed handle of member i for everything else (gmr_fk, gmr_evaluate, sessions, gmr_ik_solve alone);
 *                      members are destroyed with the group
 *   gmr_group_ik_solve inputs host [n_models], member i's arguments exactly as gmr_ik_solve takes them (n_items = 0: no work
 *                      for that member); params are shared; asynchronous on `stream`                                        */
typedef struct gmr_group gmr_group;
typedef struct gmr_group_input {
  const void *human_pos, *human_quat; /* device */
  int32_t in_dtype, n_cols;
  const int32_t *slot_col;            /* host [nslot of member i] */
  int64_t n_frames;
  const gmr_work_item *items;         /* host [n_items] */
  int32_t n_items, reserved;
  const double *qpos_init;            /* device or NULL */
  double *qpos_final, *qpos_out;      /* device (qpos_final may be NULL) */
  int32_t *iters_out, *frames_done;   /* device or NULL */
} gmr_group_input;
gmr_group *gmr_group_create(const void *const *blobs, const size_t *blob_bytes, int n_models, int device, char *err, size_t err_len);
void gmr_group_destroy(gmr_group *g);
int gmr_group_size(const gmr_group *g);
gmr_model *gmr_group_model(gmr_group *g, int i);
const char *gmr_group_last_error(const gmr_group *g);
int gmr_group_ik_solve(gmr_group *g, const gmr_group_input *inputs, const gmr_ik_params *params, void *stream);

/* Single-sequence sessions ("teleop"): one frame per call, warm start carried in the session -- the semantics of calling
 * GeneralMotionRetargeting.retarget once per captured frame (motion_retarget.py:139-185).  Inputs and outputs are HOST
 * pointers: the session owns pinned, device-visible staging that the kernel reads and writes directly (no copy engines on
 * the path), one launch per frame on the session's own stream; gmr_session_step returns when qpos_out is filled.
 *   slot_col/in_dtype/n_cols/params as in gmr_ik_solve (params->offset_to_ground is overridden per step)
 *   human_pos host [n_cols][3], human_quat host [n_cols][4] wxyz, qpos_out host [nq] f64, solves_out host int32 or NULL
 *   gmr_session_reset: qpos host [nq] or NULL (= the model's qpos0, a fresh mink.Configuration)
 *   gmr_session_state: copies the current configuration to host [nq]
 * A session borrows its model: destroy sessions before the model.  Errors are reported on the model (gmr_last_error).     */
typedef struct gmr_session gmr_session;
gmr_session *gmr_session_create(gmr_model *m, int in_dtype, int n_cols, const int32_t *slot_col, const gmr_ik_params *params);
void gmr_session_destroy(gmr_session *s);
int gmr_session_reset(gmr_session *s, const double *qpos);
int gmr_session_step(gmr_session *s, const void *human_pos, const void *human_quat, int offset_to_ground, double *qpos_out,
                     int32_t *solves_out);
int gmr_session_state(gmr_session *s, double *qpos_out);
/* Persistent mode (idle_ms > 0): frames are handed to ONE resident wavefront through a pinned mailbox instead of a launch and a
 * stream synchronisation each -- the latency of a step drops from ~49 us to the wavefront's own solve time plus two PCIe hops.
 * The wavefront leaves by itself after idle_ms without a frame (and is relaunched by the next step), on reset / state / destroy,
 * and every wait inside it is bounded.  While it is resident, device-wide synchronisation (hipDeviceSynchronize, hipFree) waits
 * for it to idle out, which is why the mode is opt-in.  idle_ms = 0 returns to one launch per frame.  Results are identical. */
int gmr_session_set_persistent(gmr_session *s, int idle_ms);

/* Evaluate, per frame, the stage errors |concat_t Log(T_body^-1 T_target)| of both tables and/or the MuJoCo-convention FK.
 *   qpos device [n][nq] f64;  human_pos/human_quat/in_dtype/n_cols/slot_col as in gmr_ik_solve (needed only with err_out)
 *   height_scale device [n] f64 or NULL: per-frame factor on the human scale table (gmr_work_item.height_scale of the clip)
 *   err_out device [n][2] f64 or NULL;  xpos_out device [n][nbody][3] f64 or NULL;  xquat_out device [n][nbody][4] wxyz or NULL
 *   task_err_out device [n][ntask[0]+ntask[1]][6] f64 or NULL: FrameTask.compute_error of every task of table 1 then table 2
 *              (rows of an unused table are left untouched), Log(T_body^-1 T_target) as [v; w] */
int gmr_evaluate(gmr_model *m, const double *qpos, int64_t n_frames, const void *human_pos, const void *human_quat, int in_dtype,
                 int n_cols, const int32_t *slot_col, int offset_to_ground, const double *height_scale, double *err_out,
                 double *task_err_out, double *xpos_out, double *xquat_out, void *stream);

/* Batched FK in the KinematicsModel convention (float32, xyzw).
 *   root_pos device [n][3], root_rot_xyzw device [n][4], dof device [n][nq-7]
 *   body_pos_out device [n][nbody][3]; body_rot_out device [n][nbody][4] or NULL   */
int gmr_fk(gmr_model *m, const float *root_pos, const float *root_rot_xyzw, const float *dof, int64_t n_frames,
           float *body_pos_out, float *body_rot_out, void *stream);

/* The same with KinematicsModel.forward_kinematics' `fitted_shape`: every body's local translation is multiplied, in float32, by
 * its row of fitted_shape before the chain.
 *   fitted_shape device [nbody] (shape_width 1) or [nbody][3] (shape_width 3) float32, or NULL (= gmr_fk)
 * The scaled body table is the call's own (stream-ordered scratch): concurrent calls with different shapes do not interfere. */
int gmr_fk_shape(gmr_model *m, const float *root_pos, const float *root_rot_xyzw, const float *dof, const float *fitted_shape,
                 int shape_width, int64_t n_frames, float *body_pos_out, float *body_rot_out, void *stream);

/* The other KinematicsModel operators (float32, xyzw; hinge-or-fixed bodies as everywhere in this library; asynchronous on `stream`):
 *   gmr_dof_to_rot           dof device [n][nq-7]             -> joint_rot_out device [n][nbody-1][4]: the hinge quaternion of body
 *                            j+1's angle in row j, the identity for bodies without a hinge
 *   gmr_rot_to_dof           joint_rot device [n][nbody-1][4] -> dof_out device [n][nq-7]: angle about the joint axis of each hinge's
 *                            row (w made non-negative, 2 atan2(|xyz|, w), 0 below |xyz| = 1e-5, sign from the axis), clamped to the
 *                            joint's range
 *   gmr_local_rot_to_global  local_rot device [n][nbody][4]   -> global_rot_out device [n][nbody][4]: row 0 copied, every other row
 *                            global[parent] (x) local in the reference's operation order (results equal a sequential float32
 *                            evaluation bit for bit); the two arrays must not alias                                         */
int gmr_dof_to_rot(gmr_model *m, const float *dof, int64_t n_frames, float *joint_rot_out, void *stream);
int gmr_rot_to_dof(gmr_model *m, const float *joint_rot, int64_t n_frames, float *dof_out, void *stream);
int gmr_local_rot_to_global(gmr_model *m, const float *local_rot, int64_t n_frames, float *global_rot_out, void *stream);

/* Lowest body z per clip: min over frames [seq_offsets[s], seq_offsets[s+1]) and bodies of FK z.
 *   seq_offsets host [n_seq+1]; min_z_out device [n_seq] float32                    */
int gmr_fk_min_height(gmr_model *m, const float *root_pos, const float *root_rot_xyzw, const float *dof,
                      const int64_t *seq_offsets, int n_seq, float *min_z_out, void *stream);

/* SMPL-X key-points (stateless): axis-angle joint rotations + joint positions -> global orientations (wxyz) and positions,
 * optionally resampled to n_frames_out frames at times linspace(0, n_frames-1, n_frames_out).
 *   parents host [n_joints] (parents[0] = -1, parents[j] < j);  joints_stride: joints per frame in `joints` (>= n_joints)
 *   global_orient device [n_frames][3], full_pose device [n_frames][n_joints][3], joints device [n_frames][joints_stride][3]
 *   pos_out device [n_frames_out][n_joints][3], quat_out device [n_frames_out][n_joints][4]                                */
int gmr_smplx_keypoints(const int32_t *parents, int n_joints, int joints_stride, const double *global_orient, const double *full_pose,
                        const double *joints, int64_t n_frames, int64_t n_frames_out, int resample, double *pos_out, double *quat_out,
                        void *stream);

/* The same with a column selection: out_cols host [n_out] names the joints to emit, column c of the outputs = joint out_cols[c]
 * (each at most once); their ancestors are chained internally, joints that are neither emitted nor an ancestor of an emitted one are
 * not read.  With the 14 joints an smplx_to_*.json config consumes, gmr_ik_solve reads a dense [n_frames_out][14][7] instead of
 * picking 14 of 55 columns.  out_cols == NULL: all joints (n_out ignored).
 *   pos_out device [n_frames_out][n_out][3], quat_out device [n_frames_out][n_out][4]                                        */
int gmr_smplx_keypoints_cols(const int32_t *parents, int n_joints, int joints_stride, const double *global_orient, const double *full_pose,
                             const double *joints, int64_t n_frames, int64_t n_frames_out, int resample, const int32_t *out_cols, int n_out,
                             double *pos_out, double *quat_out, void *stream);

/* The same for input arrays of either element type: in_dtype GMR_DTYPE_F32 (what a body model emits: the arrays go in as they are,
 * each element promoted to float64 on load -- exactly what the reference's scipy / numpy calls do with float32 input -- at half the
 * bytes) or GMR_DTYPE_F64.  global_orient, full_pose, joints: device arrays of that type, shapes as above.                        */
int gmr_smplx_keypoints_in(const int32_t *parents, int n_joints, int joints_stride, const void *global_orient, const void *full_pose,
                           const void *joints, int in_dtype, int64_t n_frames, int64_t n_frames_out, int resample, const int32_t *out_cols,
                           int n_out, double *pos_out, double *quat_out, void *stream);

/* Host-side parse of a BVH file's HIERARCHY section and MOTION header (stateless, no device involved; grammar and the
 * reference semantics it keeps are documented in gmr_amd/csrc/bvh_text.h).  Replaces the hierarchy loop of read_bvh
 * (general_motion_retargeting/utils/lafan_vendor/extract.py:60-139).
 *   text/len        the file (or at least its header)
 *   names_out       host char[names_cap]: joint names, NUL-separated, in hierarchy order
 *   parents_out     host int32[max_joints] (-1 for the root); offsets_out host double[max_joints][3]
 *   channels_out    host int32[max_joints]: channel count of every joint
 *   order_out       host int32[3]: axes (0=x,1=y,2=z), in listed order, of the first joint whose rotation slice -- channels 0-2 of a
 *                   3-channel joint, 3-5 of any other -- is all rotations (a positions-only root of the 9-channel layout is skipped)
 *   n_frames_out, frame_time_out: the MOTION header; motion_offset_out: byte offset of the first motion row in text
 * Returns the number of joints, -1 on a malformed file, -2 if max_joints / names_cap are too small.                      */
int gmr_bvh_parse_header(const char *text, size_t len, int max_joints, char *names_out, size_t names_cap, int32_t *parents_out,
                         double *offsets_out, int32_t *channels_out, int32_t *order_out, int64_t *n_frames_out, double *frame_time_out,
                         size_t *motion_offset_out);

/* Host-side text parse of a BVH MOTION block (stateless, no device involved): the first max_lines non-empty lines of
 * text[0..len) are read as whitespace-separated decimal numbers into out (host, capacity max_out doubles), correctly
 * rounded like Python's float().  *n_lines = lines read, *n_cols = numbers on the first line.  Returns the count of numbers
 * written, or -1 on a malformed token, a line whose length differs from the first, or an overflow of max_out.            */
int64_t gmr_bvh_parse_motion(const char *text, size_t len, int64_t max_lines, double *out, int64_t max_out, int64_t *n_lines,
                             int64_t *n_cols);

/* The same parse on the device, for a batch of files whose text is already in device memory (one H2D copy of the files as they
 * are): identical values, bit for bit, for every token on the exact fast path (at most 19 significant digits, mantissa < 2^53, power
 * of ten within 10^+-22: one correctly rounded multiply / divide, what float() returns); every other token is REPORTED, not guessed:
 * the caller parses those with strtod (gmr_bvh_parse_motion's slow path) and patches rows_out.  Replaces the same reference lines
 * (general_motion_retargeting/utils/lafan_vendor/extract.py:140-156).
 *   text        device [text_bytes]   the files' bytes; seg_begin/seg_end host [n_files]: each file's MOTION block in it
 *   n_lines     host [n_files]  rows to read per file (the header's Frames:);  n_cols: numbers per row (one skeleton per batch)
 *   row_begin   host [n_files]  first row of each file in rows_out;  rows_out device [sum(n_lines)][n_cols] float64
 *   status_out  host [n_files]  0 = ok; bit 0: some row does not hold n_cols numbers, bit 1: fewer than n_lines rows -- parse that
 *               file with gmr_bvh_parse_motion, which reports what is wrong with it
 *   n_tokens_out host [n_files] or NULL: numbers found in the whole block
 *   slow_out    host [max_slow][3]  (file, index of the number in its file, byte offset in text) of the tokens off the fast path;
 *               *n_slow = how many there were (more than max_slow: treat the files as status != 0)
 * Runs on `stream` and synchronises it before returning.                                                                   */
int gmr_bvh_parse_motion_device(const char *text, int64_t text_bytes, int n_files, const int64_t *seg_begin, const int64_t *seg_end,
                                const int64_t *n_lines, int64_t n_cols, const int64_t *row_begin, double *rows_out, int32_t *status_out,
                                int64_t *n_tokens_out, int64_t *slow_out, int64_t max_slow, int64_t *n_slow, void *stream);

/* BVH skeleton FK (stateless).  Joints in hierarchy order (parents[0] = -1, parents[j] < j), one Euler triple per joint.
 *   parents, euler_order[3] (0=x,1=y,2=z, the order the channels are listed), extra_*_src[n_extra]: host
 *   local_pos  device [n_frames][n_joints][3]  local translations (file units)
 *   euler_rad  device [n_frames][n_joints][3]  channel angles in radians
 *   pos_out    device [n_frames][n_joints+n_extra][3]  = scale * (global position rotated to Z-up)
 *   quat_out   device [n_frames][n_joints+n_extra][4]  wxyz, rotated to Z-up
 * Extra entry k takes the position of joint extra_pos_src[k] and the orientation of joint extra_rot_src[k].        */
int gmr_bvh_fk(const int32_t *parents, int n_joints, const int32_t *euler_order, const int32_t *extra_pos_src,
               const int32_t *extra_rot_src, int n_extra, const double *local_pos, const double *euler_rad, int64_t n_frames,
               double scale, double *pos_out, double *quat_out, void *stream);

/* The same fed with the file's own motion rows, as gmr_bvh_parse_motion wrote them (degrees, file units): the slicing of a row into
 * root translation / per-joint channels (general_motion_retargeting/utils/lafan_vendor/extract.py:140-156) and the degrees -> radians
 * step of load_lafan1_file (utils/lafan1.py:13) happen in the kernel, so nothing is reshaped or copied on the host.
 *   channels   3: rows = 3 root position values + 3 angles per joint;  6: (position, angles) per joint;
 *              9: 3 root position values + (position, angles, scale) per non-root joint, local position = offset + position * scale,
 *                 zero root rotation
 *   offsets    device [n_joints][3]  the joints' OFFSET lines (local positions where the rows carry none)
 *   rows       device [n_frames][n_cols]
 *   out_cols   host [n_out] or NULL: entries (joint j, or n_joints + k for extra k) to emit, column c = entry out_cols[c];
 *              NULL = all n_joints + n_extra
 *   pos_out    device [n_frames][n_out][3], quat_out device [n_frames][n_out][4]                                                  */
int gmr_bvh_fk_rows(const int32_t *parents, int n_joints, const int32_t *euler_order, const int32_t *extra_pos_src,
                    const int32_t *extra_rot_src, int n_extra, int channels, const double *offsets, const double *rows, int64_t n_cols,
                    int64_t n_frames, double scale, const int32_t *out_cols, int n_out, double *pos_out, double *quat_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GMR_AMD_H */
