#!/usr/bin/env python3
"""More reference-generated FK golden vectors, for the inputs the first set does not reach (build container only):

    python tests/golden/make_golden_wide.py

``fk_<robot>_wide.npz``: joint angles far outside the limits (uniform in +-7 rad: the sin / cos range reduction of the kernels),
root rotations that are NOT unit quaternions (the reference multiplies them as they come, kinematics_model.py:222-241), root
positions of +-50 m, and exact zeros / +-pi; 96 frames, through the reference's own ``KinematicsModel.forward_kinematics``
(imported like in make_golden.py)."""
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("GMR_ROOT", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
pkg = types.ModuleType("general_motion_retargeting")
pkg.__path__ = [os.path.join(REF, "general_motion_retargeting")]
sys.modules["general_motion_retargeting"] = pkg
from general_motion_retargeting.kinematics_model import KinematicsModel  # noqa: E402

ROBOTS = {"unitree_g1": "unitree_g1/g1_mocap_29dof.xml", "unitree_g1_with_hands": "unitree_g1/g1_mocap_29dof_with_hands.xml",
          "booster_t1": "booster_t1/t1_mocap.xml"}


def main():
    T = 96
    for name, rel in ROBOTS.items():
        km = KinematicsModel(os.path.join(REF, "assets", rel), "cpu")
        rng = np.random.default_rng(7)
        dof = rng.uniform(-7.0, 7.0, (T, km.num_dof)).astype(np.float32)
        dof[0] = 0.0
        dof[1] = np.pi
        dof[2] = -np.pi
        dof[3] = 2 * np.pi
        root_pos = rng.uniform(-50.0, 50.0, (T, 3)).astype(np.float32)
        q = rng.normal(0, 1.0, (T, 4))
        q[: T // 2] /= np.linalg.norm(q[: T // 2], axis=1, keepdims=True)   # first half unit, second half as drawn (norms 0.3 .. 3)
        root_rot = q.astype(np.float32)
        bp, br = km.forward_kinematics(torch.from_numpy(root_pos), torch.from_numpy(root_rot), torch.from_numpy(dof))
        np.savez_compressed(os.path.join(HERE, f"fk_{name}_wide.npz"), root_pos=root_pos, root_rot=root_rot, dof_pos=dof,
                            body_pos=bp.numpy(), body_rot=br.numpy())
        print(name, bp.shape, float(np.abs(bp.numpy()).max()))


if __name__ == "__main__":
    main()
