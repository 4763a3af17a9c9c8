#!/usr/bin/env python3
"""Differential run of the ORACLE's float32 KinematicsModel restatements (oracle/gmr_oracle.c: fk_kin, dof_to_rot, rot_to_dof,
local_rot_to_global, fitted_shape -- what the GPU kernels are compared with at sizes no golden file could hold) against the REFERENCE's
own class on random inputs.

Run in the build container only (needs /root/reference; never on the GPU box):

    python tests/golden/fuzz_fk_vs_reference.py [seconds] [seed]

Same import recipe as make_golden.py.  Robots: the eight of the registry's eleven the reference class can parse.  Inputs per draw: 1-4096 frames, joint angles up
to +-8 rad, root positions up to +-50 m, unit and non-unit root quaternions, random / tiny / near-pi joint rotations, body scales 0.5-2.
"""
import os
import sys
import time
import types

import numpy as np
import torch

REF = os.environ.get("GMR_ROOT", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pkg = types.ModuleType("general_motion_retargeting")
pkg.__path__ = [os.path.join(REF, "general_motion_retargeting")]
sys.modules["general_motion_retargeting"] = pkg
from general_motion_retargeting.kinematics_model import KinematicsModel  # noqa: E402

from oracle.oracle import Oracle  # noqa: E402
from tests.util import compiled  # noqa: E402

ROBOTS = {"unitree_g1": "unitree_g1/g1_mocap_29dof.xml", "unitree_g1_with_hands": "unitree_g1/g1_mocap_29dof_with_hands.xml", "booster_t1": "booster_t1/t1_mocap.xml",
          "stanford_toddy": "stanford_toddy/toddy_mocap.xml", "fourier_n1": "fourier_n1/n1_mocap.xml", "kuavo_s45": "kuavo_s45/biped_s45_collision.xml",
          "hightorque_hi": "hightorque_hi/hi_25dof.xml", "booster_k1": "booster_k1/K1_serial.xml"}


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    torch.set_num_threads(4)
    models = {}
    for name, rel in ROBOTS.items():
        models[name] = (KinematicsModel(os.path.join(REF, "assets", rel), "cpu"), Oracle(compiled("smplx", name).blob))
    worst = {"fk_pos": 0.0, "fk_rot": 0.0, "dof_to_rot": 0.0, "rot_to_dof": 0.0, "chain": 0.0, "fitted_shape": 0.0}
    t0, runs, frames = time.time(), 0, 0
    names = list(ROBOTS)
    while time.time() - t0 < seconds:
        name = names[int(rng.integers(len(names)))]
        km, orc = models[name]
        nb, nd = km.num_joint, km.num_dof
        n = int(rng.choice([1, 7, 64, 1000, int(rng.integers(1, 4097))]))
        amp = float(rng.choice([0.3, 1.5, 8.0]))
        dof = rng.uniform(-amp, amp, (n, nd)).astype(np.float32)
        rp = (rng.normal(size=(n, 3)) * float(rng.choice([1.0, 50.0]))).astype(np.float32)
        rr = rng.normal(size=(n, 4)).astype(np.float32)
        rr /= np.linalg.norm(rr, axis=1, keepdims=True)
        if rng.random() < 0.3:
            rr *= rng.uniform(0.5, 2.0, (n, 1)).astype(np.float32)
        T = torch.from_numpy
        bp_r, br_r = km.forward_kinematics(T(rp), T(rr), T(dof))
        bp, br = orc.fk_kin(rp, rr, dof)
        sp, sr = max(1.0, float(bp_r.abs().max())), max(1.0, float(br_r.abs().max()))
        worst["fk_pos"] = max(worst["fk_pos"], float(np.abs(bp - bp_r.numpy()).max()) / sp)
        worst["fk_rot"] = max(worst["fk_rot"], float(np.abs(br - br_r.numpy()).max()) / sr)
        worst["dof_to_rot"] = max(worst["dof_to_rot"], float(np.abs(orc.dof_to_rot(dof) - km.dof_to_rot(T(dof)).numpy()).max()))
        lr = rng.normal(size=(n, nb, 4)).astype(np.float32)
        lr /= np.linalg.norm(lr, axis=-1, keepdims=True)
        kind = rng.random()
        if kind < 0.3:
            lr[..., :3] *= np.float32(10.0 ** rng.uniform(-7, 0))
            lr[..., 3] = np.sqrt(np.maximum(0.0, 1.0 - (lr[..., :3].astype(np.float64) ** 2).sum(-1))).astype(np.float32) * np.where(rng.random((n, nb)) < 0.5, -1, 1)
        back_r = km.rot_to_dof(T(np.ascontiguousarray(lr[:, 1:]))).numpy()
        worst["rot_to_dof"] = max(worst["rot_to_dof"], float(np.abs(orc.rot_to_dof(lr[:, 1:]) - back_r).max()))
        worst["chain"] = max(worst["chain"], float(np.abs(orc.local_rot_to_global(lr) - km.convert_local_rot_to_global(T(lr)).numpy()).max()))
        shp = rng.uniform(0.5, 2.0, (nb, int(rng.choice([1, 3])))).astype(np.float32)
        sh_t = T(shp[:, 0].copy() if shp.shape[1] == 1 else shp)
        bps_r, _ = km.forward_kinematics(T(rp), T(rr), T(dof), fitted_shape=sh_t)
        bps, _ = orc.fk_kin(rp, rr, dof, fitted_shape=shp)
        worst["fitted_shape"] = max(worst["fitted_shape"], float(np.abs(bps - bps_r.numpy()).max()) / max(1.0, float(bps_r.abs().max())))
        runs += 1
        frames += n
    print(f"oracle float32 KinematicsModel restatements vs the reference's class: {runs} draws, {frames} frames, {len(ROBOTS)} robots, {time.time() - t0:.0f} s; worst difference "
          + ", ".join(f"{k} {v:.2e}" for k, v in worst.items()) + " (positions / rotations relative to the largest magnitude in play; bound 2e-6)")
    return 1 if max(worst.values()) > 2e-6 else 0


if __name__ == "__main__":
    sys.exit(main())
