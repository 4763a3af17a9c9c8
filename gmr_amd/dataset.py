"""Dataset path: clips in -> the reference's motion dicts / pickles out, post-processing on the GPU.

Reproduces what ``process_file`` does after the retarget loop (reference
scripts/smplx_to_robot_dataset.py:93-146; the BVH variant scripts/bvh_to_robot_dataset.py:106-151 is the same
with both adjustments off):

* ``root_rot`` wxyz -> xyzw (:101-102), ``dof_pos = qpos[:, 7:]`` (:103)
* ``local_body_pos``: FK with zero root position and identity root rotation, float32 (:106-112)
* HEIGHT_ADJUST: FK with the real root, clip-global ``min z`` over all bodies, ``root_pos.z -= min`` (:118-126)
* ROOT_ORIGIN_OFFSET: subtract the first frame's root xy (:128-131)
* schema ``{fps, root_pos, root_rot, dof_pos, local_body_pos, link_body_list}`` (:134-141), read back by
  general_motion_retargeting/data_loader.py:4-18 and validated by scripts/smoke_test.py:19-72.

All clips of a batch go through two FK launches (``gmr_fk``, ``gmr_fk_min_height``); nothing is looped per clip
on the device.
"""
from __future__ import annotations

import os
import pickle
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .motion_retarget import GeneralMotionRetargeting


def motions_from_qpos(gmr: GeneralMotionRetargeting, qpos: torch.Tensor, seq_offsets: Sequence[int], fps,
                      height_adjust: bool = True, root_origin_offset: bool = True, ground_offset: float = 0.0) -> List[Dict]:
    """qpos ``[N, nq]`` float64 on the GPU (concatenated clips) -> one motion dict per clip."""
    if gmr.model.planar_base:
        # the dataset scripts read a free-joint root out of qpos (root_pos = qpos[:3], root_rot = qpos[3:7],
        # scripts/smplx_to_robot_dataset.py:97-103); the reference has no such path for galaxea_r1pro either
        raise NotImplementedError("the dataset post-processing assumes a free-joint root; use retarget_batch for a planar-base robot")
    eng = gmr._engine
    offs = np.asarray(seq_offsets, dtype=np.int64)
    N = int(qpos.shape[0])
    if offs[0] != 0 or offs[-1] != N:
        raise ValueError("seq_offsets must span [0, N]")
    fps_list = list(fps) if isinstance(fps, (list, tuple, np.ndarray)) else [fps] * (len(offs) - 1)
    root_pos = qpos[:, 0:3].clone()
    root_rot = qpos[:, [4, 5, 6, 3]].contiguous()          # wxyz -> xyzw
    dof_pos = qpos[:, 7:].contiguous()
    dof32 = dof_pos.to(torch.float32)
    zeros = torch.zeros((N, 3), dtype=torch.float32, device=qpos.device)
    ident = torch.zeros((N, 4), dtype=torch.float32, device=qpos.device)
    ident[:, 3] = 1.0
    local_body_pos, _ = eng.fk(zeros, ident, dof32, want_rot=False)
    if height_adjust and N > 0:
        lowest = eng.fk_min_height(root_pos.to(torch.float32), root_rot.to(torch.float32), dof32, offs).to(torch.float64)
        lens = torch.from_numpy(np.diff(offs)).to(qpos.device)
        root_pos[:, 2] = root_pos[:, 2] - torch.repeat_interleave(lowest, lens) + ground_offset
    if root_origin_offset and N > 0:
        nonempty = np.diff(offs) > 0
        first = torch.zeros((len(offs) - 1, 2), dtype=torch.float64, device=qpos.device)
        first[torch.from_numpy(nonempty).to(qpos.device)] = root_pos[torch.from_numpy(offs[:-1][nonempty]).to(qpos.device), :2]
        lens = torch.from_numpy(np.diff(offs)).to(qpos.device)
        root_pos[:, :2] = root_pos[:, :2] - torch.repeat_interleave(first, lens, dim=0)
    # to the host through pinned arrays, all four copies in flight together: a fresh pageable array costs more in first-touch
    # page faults than the copy itself, and torch's host allocator hands the pinned blocks back to the next batch once these
    # arrays are dropped (pickled and released)
    host = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in (root_pos, root_rot, dof_pos, local_body_pos)]
    for h, t in zip(host, (root_pos, root_rot, dof_pos, local_body_pos)):
        h.copy_(t, non_blocking=True)
    torch.cuda.current_stream(qpos.device).synchronize()
    rp, rr, dp, lb = (h.numpy() for h in host)
    names = list(gmr.model.body_names)
    out = []
    for s in range(len(offs) - 1):
        a, b = int(offs[s]), int(offs[s + 1])
        # row slices of the batch arrays (C-contiguous views): no second host copy; pickling a slice stores only the slice
        out.append({"fps": fps_list[s], "root_pos": rp[a:b], "root_rot": rr[a:b], "dof_pos": dp[a:b],
                    "local_body_pos": lb[a:b], "link_body_list": names})
    return out


def retarget_clips(gmr: GeneralMotionRetargeting, pos, quat, body_names: Sequence[str], seq_offsets: Sequence[int], fps=30,
                   height_adjust: bool = True, root_origin_offset: bool = True, chunk: int = 0, burn_in: int = 0,
                   human_heights: Optional[Sequence[float]] = None) -> List[Dict]:
    """The whole ``process_file`` compute path for a batch of clips: batched IK, FK, post-processing.  ``human_heights``:
    one ``actual_human_height`` per clip (the per-file ``GMR(..., actual_human_height=...)`` of
    scripts/smplx_to_robot_dataset.py:79-83)."""
    tpos = torch.from_numpy(np.ascontiguousarray(pos)) if isinstance(pos, np.ndarray) else pos
    tquat = torch.from_numpy(np.ascontiguousarray(quat)) if isinstance(quat, np.ndarray) else quat
    qpos = gmr.retarget_batch(tpos.to(gmr.device), tquat.to(gmr.device), body_names, seq_offsets=seq_offsets, chunk=chunk, burn_in=burn_in,
                              human_heights=human_heights)  # (raises on non-finite qpos / a capped QP)
    return motions_from_qpos(gmr, qpos, seq_offsets, fps, height_adjust=height_adjust, root_origin_offset=root_origin_offset)


def save_motion(path: str, motion: Dict, override: bool = False) -> bool:
    """Pickle one motion dict; like the scripts, skip files that already exist unless ``override`` (:219)."""
    if os.path.exists(path) and not override:
        return False
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    if str(path).endswith(".pt"):  # the torch twin of the schema (scripts/convert_motion_pkl_to_pt.py:40-48): arrays as tensors
        torch.save({k: torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v for k, v in motion.items()}, path)
        return True
    with open(path, "wb") as f:
        pickle.dump(motion, f)
    return True


def load_robot_motion(motion_file: str):
    """Reader with the contract of general_motion_retargeting/data_loader.py:4-18 (root_rot returned as wxyz)."""
    if str(motion_file).endswith(".pt"):  # (:50-58 of the converter: tensors back to arrays; files this package wrote)
        d = {k: v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v for k, v in torch.load(motion_file, map_location="cpu", weights_only=True).items()}
    else:
        with open(motion_file, "rb") as f:
            d = pickle.load(f)
    root_rot = d["root_rot"][:, [3, 0, 1, 2]]
    return d, d["fps"], d["root_pos"], root_rot, d["dof_pos"], d["local_body_pos"], d["link_body_list"]


def validate_motion(motion: Dict, nq: Optional[int] = None) -> None:
    """The structural checks of scripts/smoke_test.py:19-72."""
    for k in ("fps", "root_pos", "root_rot", "dof_pos"):
        if k not in motion:
            raise KeyError(k)
    T = motion["root_pos"].shape[0]
    if motion["root_pos"].shape != (T, 3) or motion["root_rot"].shape != (T, 4) or motion["dof_pos"].shape[0] != T:
        raise ValueError("bad motion shapes")
    if nq is not None and motion["dof_pos"].shape[1] != nq - 7:
        raise ValueError("dof count does not match the model")
    n = np.linalg.norm(motion["root_rot"], axis=1)
    if T and (n.min() < 0.5 or n.max() > 1.5):
        raise ValueError("root_rot is not a quaternion track")
