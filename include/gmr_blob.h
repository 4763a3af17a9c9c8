/* gmr_blob.h -- wire format of the packed robot + IK-config "model blob".
 *
 * One contiguous little-endian buffer: a fixed header followed by 8-byte
 * aligned arrays.  It is what the host packs from an MJCF model and an
 * ik_configs JSON file, what rank 0 broadcasts to the other GPUs, and what
 * both the HIP library (include/gmr_amd.h) and the CPU oracle (oracle/) parse.
 *
 * It carries exactly the data the reference keeps in
 *   - mujoco.MjModel (kinematic subset)      reference motion_retarget.py:27
 *   - KinematicsModel tensors                reference kinematics_model.py:76-99
 *   - the ik_config tables / offsets / scale reference motion_retarget.py:36-54,80-114
 */
#ifndef GMR_BLOB_H
#define GMR_BLOB_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GMR_BLOB_MAGIC 0x42524D47u /* "GMRB" */
#define GMR_BLOB_VERSION 1u

#define GMR_JNT_NONE 0
#define GMR_JNT_HINGE 1
#define GMR_JNT_FREE 2

#define GMR_MAX_BODIES 64 /* one body per lane in the IK kernel */
#define GMR_MAX_TASKS 32  /* per table */
#define GMR_MAX_SLOTS 32  /* distinct human bodies consumed */

typedef struct gmr_blob_header {
  uint32_t magic;
  uint32_t version;
  uint32_t total_bytes;
  uint32_t reserved0;

  int32_t nbody;      /* bodies, depth-first order, body 0 = free-joint root   */
  int32_t nq;         /* 7 + hinges                                            */
  int32_t nv;         /* 6 + hinges                                            */
  int32_t nslot;      /* human bodies kept by scale_human_data (scale table)   */
  int32_t ntask[2];   /* tasks with non-zero weight in table 1 / table 2       */
  int32_t use_table[2];
  int32_t root_slot;  /* slot of human_root_name                               */
  int32_t root_dof_mask; /* which of the root's free-joint dofs (x y z rx ry rz, bit k = dof k) the IK may move; 0 = all six.
                          * 0x23 = a planar base (slide x, slide y, hinge z on the root body; assets/galaxea_r1pro/r1_pro.xml:102-104):
                          * qpos keeps the free-joint layout, z / roll / pitch simply never change                                */
  int32_t reserved1[2];

  /* byte offsets from the start of the blob */
  uint32_t off_parent;        /* int32 [nbody]                                 */
  uint32_t off_jnt_type;      /* int32 [nbody]  GMR_JNT_*                      */
  uint32_t off_qpos_adr;      /* int32 [nbody]  -1 if none                     */
  uint32_t off_dof_adr;       /* int32 [nbody]  -1 if none                     */
  uint32_t off_jnt_limited;   /* int32 [nbody]                                 */
  uint32_t off_body_pos;      /* f64 [nbody][3]                                */
  uint32_t off_body_quat;     /* f64 [nbody][4] wxyz, unit (MuJoCo compile)    */
  uint32_t off_body_quat_raw; /* f64 [nbody][4] wxyz, as written in the XML    */
  uint32_t off_jnt_axis;      /* f64 [nbody][3] unit                           */
  uint32_t off_jnt_range;     /* f64 [nbody][2] radians                        */
  uint32_t off_qpos0;         /* f64 [nq]                                      */
  uint32_t off_slot_scale;    /* f64 [nslot]   human_scale_table * height ratio*/
  uint32_t off_slot_pos_off;  /* f64 [nslot][3] table-1 pos_offset - ground*z  */
  uint32_t off_slot_rot_off;  /* f64 [nslot][4] table-1 rot_offset wxyz, unit  */
  uint32_t off_slot_is_foot;  /* int32 [nslot] name contains "Foot"/"foot"     */
  uint32_t off_task_body[2];  /* int32 [ntask[k]] robot body index             */
  uint32_t off_task_slot[2];  /* int32 [ntask[k]] slot index                   */
  uint32_t off_task_wp[2];    /* f64 [ntask[k]] position_cost                  */
  uint32_t off_task_wr[2];    /* f64 [ntask[k]] orientation_cost               */
  uint32_t reserved2[3];
} gmr_blob_header;

/* One unit of IK work: a run of consecutive frames of one clip, processed in
 * time order with warm start.  The first n_burn frames only warm the state up
 * (no output); the following n_out frames write qpos_out[frame].  burn_row lets a
 * caller check a chunk's warm-up against its predecessor's final state.
 * check_stride > 0 makes the item a *verification walk* down a clip whose chunks
 * of check_stride frames were already solved speculatively (n_burn must be 0):
 * at the k-th chunk boundary (k = 0, 1, ...) the state is compared with
 * qpos_final[burn_row + k], the state that chunk started its own output from.
 * Equal to gmr_ik_params.check_tol (the floating base's quaternion up to its sign: q and
 * -q are one rotation, and a chunk started on a target carries the target's sign): the
 * chunk's stored frames stand -- their base quaternions, the B row and the adopted state are
 * turned to the sequence's sign where it differs -- the walk adopts
 * qpos_final[final_row + k] (its stored final state) and skips it.  Different: the
 * chunk is solved here from the true state, qpos_final[burn_row + k] and
 * qpos_final[final_row + k] are rewritten.  frames_done[item] = frames solved.
 * init_row >= 0 starts from qpos_init[init_row]; GMR_INIT_QPOS0 (-1) from qpos0
 * (reference: a fresh GeneralMotionRetargeting per clip, motion_retarget.py:75);
 * GMR_INIT_ROOT_TARGET (-2) from qpos0 with the floating base placed on the prepared
 * target of the root body's task in the first frame processed (table 1's, else table
 * 2's; plain qpos0 if neither table tracks the root body) -- the speculative start of
 * a mid-clip chunk: same heading and place as the human, so the burn-in converges into
 * the basin the sequential run is in instead of one found from the world origin.
 * height_scale multiplies every human_scale_table entry for this item (0 = 1.0): the
 * per-clip actual_human_height / the height the model was compiled with
 * (motion_retarget.py:36-43; scripts/smplx_to_robot_dataset.py:79-83 builds one
 * retargeter per file with that file's height). */
#define GMR_INIT_QPOS0 (-1)
#define GMR_INIT_ROOT_TARGET (-2)
typedef struct gmr_work_item {
  int64_t frame_begin; /* first frame processed (burn-in included)            */
  int32_t n_burn;
  int32_t n_out;
  int32_t init_row;    /* row of qpos_init, or -1                              */
  int32_t final_row;   /* row of qpos_final to receive the last state, or -1   */
  int32_t burn_row;    /* row of qpos_final to receive the state right before  */
                       /* the first output frame (after burn-in), or -1        */
  int32_t check_stride; /* 0: plain item                                        */
  double height_scale;  /* per-item factor on the human scale table, 0 = 1.0    */
} gmr_work_item;

/* Solver constants; defaults are the reference's hard-coded values. */
typedef struct gmr_ik_params {
  double damping;         /* 0.5    motion_retarget.py:19  (Tikhonov on dq)    */
  double tol;             /* 1e-3   motion_retarget.py:153,172                 */
  double limit_gain;      /* 0.95   mink ConfigurationLimit default            */
  double lm_damping;      /* 1.0    motion_retarget.py:88,105                  */
  int32_t max_iter;       /* 10     motion_retarget.py:56                      */
  int32_t offset_to_ground; /* 0    motion_retarget.py:122,252-270             */
  double check_tol;       /* 1e-7   max |dq| for a repair run to stop (check_stride) */
} gmr_ik_params;

#ifdef __cplusplus
}
#endif
#endif /* GMR_BLOB_H */
