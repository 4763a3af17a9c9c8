"""``KinematicsModel`` -- batched FK with the reference's interface, on the HIP kernel.

Mirror of general_motion_retargeting/kinematics_model.py:68-278 as the dataset scripts use it
(scripts/smplx_to_robot_dataset.py:93-123): constructor ``(file_path, device)``,
``forward_kinematics(root_pos, root_rot_xyzw, dof_pos) -> (body_pos, body_rot)`` in float32,
``body_names``, ``num_dof``, ``num_joint``, ``joint_dof_idx``, ``parent_indices``,
``get_dof_limits``.  ``file_path`` may be an MJCF ``.xml`` or a ``gmr_amd.robot.v1`` pack.
Unlike the reference, ``<include>`` files are resolved (engineai_pm01 loads) and any leading
batch shape is accepted.  ``rot_to_dof`` / ``convert_local_rot_to_global`` (unused by the
scripts) and ``fitted_shape`` are not provided.
"""
from __future__ import annotations

import torch

from .engine import Engine
from .mjcf import JNT_HINGE, load_robot
from .model import compile_model


def _device_index(device) -> int:
    d = torch.device(device)
    if d.type != "cuda":
        raise RuntimeError(f"KinematicsModel runs on a HIP device only (got {device!r}); there is no CPU path")
    return 0 if d.index is None else d.index


class KinematicsModel:
    def __init__(self, file_path, device="cuda:0"):
        self._file_path = str(file_path)
        self._robot = load_robot(self._file_path)
        self._engine = Engine(compile_model(self._robot, None), _device_index(device))
        self._device = self._engine.device
        rob = self._robot
        self._body_names = list(rob.body_names)
        self._parent_indices = torch.tensor(rob.parent.tolist(), dtype=torch.long, device=self._device)
        self._dof_idx = [int(rob.qpos_adr[b] - 7) if rob.jnt_type[b] == JNT_HINGE else -1 for b in range(rob.nbody)]
        lo, hi = rob.dof_limits()
        self._dof_lower_limits = torch.tensor(lo, dtype=torch.float, device=self._device)
        self._dof_upper_limits = torch.tensor(hi, dtype=torch.float, device=self._device)

    def forward_kinematics(self, root_pos, root_rot, dof_pos, fitted_shape=None):
        if fitted_shape is not None:
            raise NotImplementedError("fitted_shape is not supported")
        lead = root_pos.shape[:-1]
        rp = root_pos.reshape(-1, 3).to(self._device, torch.float32)
        rr = root_rot.reshape(-1, 4).to(self._device, torch.float32)
        dp = dof_pos.reshape(-1, self.num_dof).to(self._device, torch.float32)
        bp, br = self._engine.fk(rp, rr, dp, want_rot=True)
        return bp.reshape(*lead, self.num_joint, 3), br.reshape(*lead, self.num_joint, 4)

    def get_body_idx(self, body_name):
        return self._body_names.index(body_name)

    @property
    def body_names(self):
        return self._body_names

    @property
    def num_dof(self):
        return self._robot.nq - 7

    @property
    def num_joint(self):
        return self._robot.nbody

    @property
    def joint_dof_idx(self):
        return list(self._dof_idx)

    @property
    def parent_indices(self):
        return self._parent_indices

    def get_parent_idx(self, idx):
        return self._parent_indices[idx]

    def get_dof_limits(self):
        return self._dof_lower_limits, self._dof_upper_limits
