#!/usr/bin/env python3
"""Golden vectors for the remaining KinematicsModel operators, from the reference's own importable module.

Run in the build container only (needs /root/reference; never on the GPU box):

    python tests/golden/make_golden_kin_ops.py

Same import recipe as make_golden.py (stub parent package; `kinematics_model` + `torch_utils` need only torch / numpy).
Outputs `kin_ops_<robot>.npz` (data only), all float32 unless noted:

* `dof_pos [T, ndof]`                    -> `joint_rot [T, nb-1, 4]` = KinematicsModel.dof_to_rot (kinematics_model.py:172-182)
* `rot_in [T, nb-1, 4]`                  -> `dof_back [T, ndof]`     = KinematicsModel.rot_to_dof (:184-197, clamped to the limits)
  rows: round trip of joint_rot | random unit quaternions | the same with w < 0 | rotations below the 1e-5 axis threshold | near pi
* `local_rot [T, nb, 4]`                 -> `global_rot [T, nb, 4]`  = convert_local_rot_to_global (:199-211)
* `shape1 [nb]`, `shape3 [nb, 3]`        -> `body_pos_shape1/3`, `body_rot_shape1/3` = forward_kinematics(..., fitted_shape=) (:213-246)
"""
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


def unit(rng, shape):
    q = rng.normal(0, 1.0, shape + (4,))
    return (q / np.linalg.norm(q, axis=-1, keepdims=True)).astype(np.float32)


def main():
    T = 40
    for name, rel in ROBOTS.items():
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue
        km = KinematicsModel(os.path.join(REF, "assets", rel), "cpu")
        nb, nd = km.num_joint, km.num_dof
        lo, hi = km.get_dof_limits()
        rng = np.random.default_rng(7)
        # beyond the limits on purpose: dof_to_rot does not clamp, rot_to_dof does
        dof = (lo.numpy() - 0.3 + rng.uniform(0, 1, (T, nd)) * ((hi - lo).numpy() + 0.6)).astype(np.float32)
        joint_rot = km.dof_to_rot(torch.from_numpy(dof))
        rot_in = np.empty((T, nb - 1, 4), np.float32)
        rot_in[:8] = joint_rot.numpy()[:8]
        rot_in[8:16] = unit(rng, (8, nb - 1))
        r = unit(rng, (8, nb - 1))
        rot_in[16:24] = np.where(r[..., 3:] > 0, -r, r)
        tiny = rng.normal(0, 2e-6, (8, nb - 1, 3))
        rot_in[24:32] = np.concatenate([tiny, np.sqrt(1 - (tiny ** 2).sum(-1, keepdims=True))], -1).astype(np.float32)
        ax = rng.normal(0, 1, (8, nb - 1, 3)); ax /= np.linalg.norm(ax, axis=-1, keepdims=True)
        ang = np.pi - rng.uniform(0, 1e-3, (8, nb - 1, 1))
        rot_in[32:40] = np.concatenate([ax * np.sin(ang / 2), np.cos(ang / 2)], -1).astype(np.float32)
        dof_back = km.rot_to_dof(torch.from_numpy(rot_in))
        local_rot = unit(rng, (T, nb))
        global_rot = km.convert_local_rot_to_global(torch.from_numpy(local_rot))
        root_pos = rng.normal(0, 1.0, (T, 3)).astype(np.float32)
        root_rot = unit(rng, (T,))
        shape1 = rng.uniform(0.8, 1.25, nb).astype(np.float32)
        shape3 = rng.uniform(0.8, 1.25, (nb, 3)).astype(np.float32)
        args = (torch.from_numpy(root_pos), torch.from_numpy(root_rot), torch.from_numpy(dof))
        bp1, br1 = km.forward_kinematics(*args, fitted_shape=torch.from_numpy(shape1))
        bp3, br3 = km.forward_kinematics(*args, fitted_shape=torch.from_numpy(shape3))
        np.savez_compressed(os.path.join(HERE, f"kin_ops_{name}.npz"), dof_pos=dof, joint_rot=joint_rot.numpy(), rot_in=rot_in,
                            dof_back=dof_back.numpy(), local_rot=local_rot, global_rot=global_rot.numpy(), root_pos=root_pos, root_rot=root_rot,
                            shape1=shape1, shape3=shape3, body_pos_shape1=bp1.numpy(), body_rot_shape1=br1.numpy(),
                            body_pos_shape3=bp3.numpy(), body_rot_shape3=br3.numpy())
        print(name, joint_rot.shape, dof_back.shape, global_rot.shape, bp1.dtype, dof_back.dtype)


if __name__ == "__main__":
    main()
