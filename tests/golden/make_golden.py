#!/usr/bin/env python3
"""Generate golden vectors from the reference's own importable modules.

Run in the build container only (needs /root/reference; never on the GPU box):

    python tests/golden/make_golden.py

``general_motion_retargeting/__init__.py`` cannot be imported (it pulls in mink and
mujoco, which are absent), so a stub parent package with ``__path__`` set is
registered and the leaf modules that only need torch/numpy are imported below it:
``kinematics_model`` (+ ``torch_utils``) and ``rot_utils``.  Outputs (data only):

* ``fk_<robot>.npz``   inputs ``root_pos, root_rot (xyzw), dof_pos`` float32 [64,.] and
  the reference ``KinematicsModel.forward_kinematics`` outputs ``body_pos, body_rot``
* ``tree_<robot>.json`` ``body_names, parent_indices, joint_dof_idx, lower, upper``
* ``quat_mul_wxyz.npz`` random wxyz quaternion pairs and ``rot_utils.quat_mul_np`` products
"""
import json
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
from general_motion_retargeting import rot_utils  # noqa: E402
from general_motion_retargeting.kinematics_model import KinematicsModel  # noqa: E402

ROBOTS = {
    "unitree_g1": "unitree_g1/g1_mocap_29dof.xml",
    "unitree_g1_with_hands": "unitree_g1/g1_mocap_29dof_with_hands.xml",
    "booster_t1": "booster_t1/t1_mocap.xml",
    "stanford_toddy": "stanford_toddy/toddy_mocap.xml",
    "fourier_n1": "fourier_n1/n1_mocap.xml",
    # (added later; `python <this script> kuavo_s45 hightorque_hi booster_k1` writes only the robots named, leaving the older files untouched)
    "kuavo_s45": "kuavo_s45/biped_s45_collision.xml",
    "hightorque_hi": "hightorque_hi/hi_25dof.xml",
    "booster_k1": "booster_k1/K1_serial.xml",
}


def main():
    T = 64
    for name, rel in ROBOTS.items():
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue
        km = KinematicsModel(os.path.join(REF, "assets", rel), "cpu")
        lo, hi = km.get_dof_limits()
        rng = np.random.default_rng(0)
        dof = (lo.numpy() + rng.uniform(0, 1, (T, km.num_dof)) * (hi - lo).numpy()).astype(np.float32)
        root_pos = rng.normal(0, 1.0, (T, 3)).astype(np.float32)
        q = rng.normal(0, 1.0, (T, 4))
        root_rot = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
        bp, br = km.forward_kinematics(torch.from_numpy(root_pos), torch.from_numpy(root_rot), torch.from_numpy(dof))
        # the identity-root call of the dataset scripts (smplx_to_robot_dataset.py:106-112)
        bp0, _ = km.forward_kinematics(torch.zeros(T, 3), torch.tensor([[0.0, 0, 0, 1]]).repeat(T, 1), torch.from_numpy(dof))
        np.savez_compressed(
            os.path.join(HERE, f"fk_{name}.npz"), root_pos=root_pos, root_rot=root_rot, dof_pos=dof,
            body_pos=bp.numpy(), body_rot=br.numpy(), local_body_pos=bp0.numpy(),
        )
        with open(os.path.join(HERE, f"tree_{name}.json"), "w") as f:
            json.dump({
                "body_names": km.body_names,
                "parent_indices": km.parent_indices.tolist(),
                "joint_dof_idx": km.joint_dof_idx,
                "lower": [float(x) for x in lo], "upper": [float(x) for x in hi],
                "num_dof": km.num_dof,
            }, f)
        print(name, bp.shape, br.shape)
    if len(sys.argv) > 1:
        return
    rng = np.random.default_rng(1)
    a = rng.normal(size=(32, 4))
    b = rng.normal(size=(32, 4))
    np.savez_compressed(os.path.join(HERE, "quat_mul_wxyz.npz"), a=a, b=b, ab=rot_utils.quat_mul_np(a, b))


if __name__ == "__main__":
    main()
