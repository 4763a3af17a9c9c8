"""``general_motion_retargeting.utils.smpl`` (utils/smpl.py) on this engine, behind the SMPL-X body model.

``load_smplx_file`` evaluates the licensed body model through the ``smplx`` package exactly as the reference does (:12-42) -- it needs that package
and the model files, neither of which this repository ships; ``get_smplx_data_offline_fast`` takes what it returns."""
from __future__ import annotations

import numpy as np


def load_smpl_file(smpl_file):
    """utils/smpl.py:8-10."""
    return np.load(smpl_file, allow_pickle=True)


def load_smplx_file(smplx_file, smplx_body_model_path):
    """utils/smpl.py:12-42: AMASS file -> (smplx_data, body_model, smplx_output, human_height).  Needs the ``smplx`` package and the licensed
    model files (SMPL-X body-model evaluation is outside this engine)."""
    try:
        import smplx
    except ImportError as ex:  # fail loudly: there is nothing to fall back to
        raise ImportError("load_smplx_file evaluates the SMPL-X body model and needs the `smplx` package (and its licensed model files); "
                          "dump its outputs once with gmr_amd.smplx_adapter.save_joint_file and use iter_joint_batches (INTEGRATION.md 1b)") from ex
    import torch
    from ..smplx_adapter import human_height_from_betas
    smplx_data = np.load(smplx_file, allow_pickle=True)
    body_model = smplx.create(smplx_body_model_path, "smplx", gender=str(smplx_data["gender"]), use_pca=False)
    n = smplx_data["pose_body"].shape[0]
    z = lambda k: torch.zeros(n, k).float()  # noqa: E731
    smplx_output = body_model(betas=torch.tensor(smplx_data["betas"]).float().view(1, -1), global_orient=torch.tensor(smplx_data["root_orient"]).float(),
                              body_pose=torch.tensor(smplx_data["pose_body"]).float(), transl=torch.tensor(smplx_data["trans"]).float(),
                              left_hand_pose=z(45), right_hand_pose=z(45), jaw_pose=z(3), leye_pose=z(3), reye_pose=z(3), return_full_pose=True)
    return smplx_data, body_model, smplx_output, human_height_from_betas(smplx_data["betas"])


def get_smplx_data_offline_fast(smplx_data, body_model, smplx_output, tgt_fps=30):
    """utils/smpl.py:109-198 -> (frames, aligned_fps): ``frames[t] = {joint: (position[3], quaternion wxyz[4])}`` for the model's joints, aligned to
    ``tgt_fps``; the slerp / lerp and the orientation chaining run in ``gmr_smplx_keypoints_in`` on the GPU (``gmr_amd.smplx_adapter`` keeps the
    result there for ``retarget_batch``; this wrapper brings it back as the reference's list of dicts)."""
    from ..smplx_adapter import SMPLX_JOINT_NAMES, get_smplx_data_offline_fast as _fast

    def arr(x):
        return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)
    src_fps = float(np.asarray(smplx_data["mocap_frame_rate"]).reshape(-1)[0])
    parents = [int(p) for p in arr(body_model.parents).reshape(-1)]
    T = int(smplx_data["pose_body"].shape[0])
    go = arr(smplx_output.global_orient).reshape(T, 3)
    fp = arr(smplx_output.full_pose).reshape(T, -1, 3)
    jt = arr(smplx_output.joints).reshape(T, -1, 3)
    names = SMPLX_JOINT_NAMES[: len(parents)]
    pos, quat, names, aligned_fps = _fast(go, fp, jt, parents, src_fps=src_fps, tgt_fps=tgt_fps, joint_names=names)
    p, q = pos.cpu().numpy(), quat.cpu().numpy()
    if not tgt_fps < src_fps:
        aligned_fps = tgt_fps  # (:175-176: the caller's own value, e.g. the int 30, when nothing was resampled)
    return [{n: (p[t, i], q[t, i]) for i, n in enumerate(names)} for t in range(p.shape[0])], aligned_fps
