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

    def _f32(self, x, *tail):
        return torch.as_tensor(x).reshape(-1, *tail).to(self._device, torch.float32)

    def dof_to_rot(self, dof):
        """[..., num_dof] -> [..., num_joint - 1, 4] xyzw: each hinge's quaternion, the identity for bodies without one (:172-182)."""
        lead = dof.shape[:-1]
        return self._engine.dof_to_rot(self._f32(dof, self.num_dof)).reshape(*lead, self.num_joint - 1, 4)

    def rot_to_dof(self, rot):
        """[..., num_joint - 1, 4] -> [..., num_dof], clamped to the joint limits (:184-197)."""
        lead = rot.shape[:-2]
        return self._engine.rot_to_dof(self._f32(rot, self.num_joint - 1, 4)).reshape(*lead, self.num_dof)

    def convert_local_rot_to_global(self, local_rot):
        """[..., num_joint, 4] (row 0 = root rotation) -> global rotations, same shape (:199-211)."""
        lead = local_rot.shape[:-2]
        return self._engine.local_rot_to_global(self._f32(local_rot, self.num_joint, 4)).reshape(*lead, self.num_joint, 4)

    def forward_kinematics(self, root_pos, root_rot, dof_pos, fitted_shape=None):
        """fitted_shape: [num_joint] or [num_joint, 3], multiplied onto every body's local translation (:225)."""
        lead = root_pos.shape[:-1]
        rp = self._f32(root_pos, 3)
        rr = self._f32(root_rot, 4)
        dp = self._f32(dof_pos, self.num_dof)
        sh = None
        if fitted_shape is not None:
            sh = torch.as_tensor(fitted_shape).to(self._device, torch.float32)
            if sh.dim() == 2 and sh.shape[1] == 1:
                sh = sh[:, 0]
            if tuple(sh.shape) not in ((self.num_joint,), (self.num_joint, 3)):
                raise ValueError("fitted_shape must have one scalar or one 3-vector per body")
        bp, br = self._engine.fk(rp, rr, dp, want_rot=True, fitted_shape=sh)
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
