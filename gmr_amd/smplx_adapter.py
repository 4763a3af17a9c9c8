"""SMPL-X key-point adapter on the fast path.

``get_smplx_data_offline_fast`` of the reference (general_motion_retargeting/utils/smpl.py:109-198) minus the Python:
the SMPL-X body model itself (licensed assets, smpl.py:12-34) stays with the caller and hands over
``global_orient [T,3]``, ``full_pose [T,J*3]`` (axis-angle), ``joints [T,>=J,3]`` and ``parents [J]``; this module
aligns them to the target frame rate (slerp per joint, lerp per coordinate) and chains the orientations down the
kinematic tree on the GPU (``gmr_smplx_keypoints``), returning the ``[T', J, 3]`` / ``[T', J, 4]`` tensors
``retarget_batch`` consumes, with the SMPL-X joint names as columns.

The reference module cannot be imported here (it needs the ``smplx`` package), so this row is "parity unpinned";
tests pin it against a scipy restatement of the cited lines.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _native

# smplx.joint_names.JOINT_NAMES[:55] (body, jaw, eyes, hands) -- the names the smplx_to_*.json configs refer to
SMPLX_JOINT_NAMES: List[str] = [
    "pelvis", "left_hip", "right_hip", "spine1", "left_knee", "right_knee", "spine2", "left_ankle", "right_ankle", "spine3",
    "left_foot", "right_foot", "neck", "left_collar", "right_collar", "head", "left_shoulder", "right_shoulder", "left_elbow",
    "right_elbow", "left_wrist", "right_wrist", "jaw", "left_eye_smplhf", "right_eye_smplhf",
    "left_index1", "left_index2", "left_index3", "left_middle1", "left_middle2", "left_middle3", "left_pinky1", "left_pinky2",
    "left_pinky3", "left_ring1", "left_ring2", "left_ring3", "left_thumb1", "left_thumb2", "left_thumb3",
    "right_index1", "right_index2", "right_index3", "right_middle1", "right_middle2", "right_middle3", "right_pinky1", "right_pinky2",
    "right_pinky3", "right_ring1", "right_ring2", "right_ring3", "right_thumb1", "right_thumb2", "right_thumb3",
]
# kinematic tree of the SMPL-X model (body_model.parents)
SMPLX_PARENTS: List[int] = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 15, 15, 15,
                            20, 25, 26, 20, 28, 29, 20, 31, 32, 20, 34, 35, 20, 37, 38,
                            21, 40, 41, 21, 43, 44, 21, 46, 47, 21, 49, 50, 21, 52, 53]


def get_smplx_data_offline_fast(global_orient, full_pose, joints, parents: Sequence[int] = SMPLX_PARENTS, src_fps: float = 30.0,
                                tgt_fps: float = 30.0, joint_names: Sequence[str] = SMPLX_JOINT_NAMES,
                                device: int = 0, columns: Optional[Sequence[str]] = None) -> Tuple[torch.Tensor, torch.Tensor, List[str], float]:
    """-> (pos [T',J,3], quat [T',J,4] wxyz, joint names, aligned_fps); float64 CUDA tensors.
    ``columns``: emit only these joints, in this order (e.g. ``ik_columns(config)``: the 14 an smplx_to_*.json config reads) --
    their ancestors are chained inside the kernel, the rest of the 55 is neither read nor written."""
    lib = _native.load()
    dev = torch.device("cuda", device)
    as_t = lambda a: (a if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a))).detach().to(dev, torch.float64)
    parents = np.ascontiguousarray(parents, dtype=np.int32)
    J = len(parents)
    go = as_t(global_orient).reshape(-1, 3).contiguous()
    T = int(go.shape[0])
    fp = as_t(full_pose).reshape(T, -1, 3)
    if fp.shape[1] < J:
        raise ValueError("full_pose has fewer joints than parents")
    fp = fp[:, :J].contiguous()
    jt = as_t(joints).reshape(T, -1, 3).contiguous()
    if jt.shape[1] < J or len(joint_names) < J:
        raise ValueError("joints / joint_names shorter than parents")
    frame_skip = int(src_fps / tgt_fps)  # smpl.py:119
    if tgt_fps < src_fps:
        T_out = T // frame_skip          # :127
        resample = 1
        aligned_fps = T_out / T * src_fps if T else tgt_fps  # :172
    else:
        T_out, resample, aligned_fps = T, 0, tgt_fps
    names = list(joint_names[:J])
    cols = None
    if columns is not None:
        sel = [str(c) for c in columns]
        missing = [c for c in sel if c not in names]
        if missing:
            raise KeyError(missing[0])
        cols = np.asarray([names.index(c) for c in sel], dtype=np.int32)
        names = sel
    B = len(names)
    pos = torch.empty((T_out, B, 3), dtype=torch.float64, device=dev)
    quat = torch.empty((T_out, B, 4), dtype=torch.float64, device=dev)
    vp = C.c_void_p
    if T_out > 0:
        rc = lib.gmr_smplx_keypoints_cols(parents.ctypes.data_as(vp), J, int(jt.shape[1]), vp(go.data_ptr()), vp(fp.data_ptr()), vp(jt.data_ptr()),
                                          T, T_out, resample, cols.ctypes.data_as(vp) if cols is not None else None, B,
                                          vp(pos.data_ptr()), vp(quat.data_ptr()), vp(torch.cuda.current_stream(dev).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"gmr_smplx_keypoints_cols failed with status {rc}")
    return pos, quat, names, float(aligned_fps)
