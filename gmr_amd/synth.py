"""Synthetic "AMASS-shaped" workload generator.

No AMASS / SMPL-X / LAFAN1 data can exist here, so benchmark and parity inputs are
generated: a smooth in-limit robot trajectory -> robot FK (float64 numpy) -> exact
inverse of the reference's target preparation (motion_retarget.py:209-250) -> human
key-points laid out like the per-frame dicts of utils/smpl.py:185-196
(``pos[T, B, 3]`` metres, ``quat[T, B, 4]`` wxyz).  The *easy* variant is exactly
reachable; the *hard* variant adds 2 cm / 5 deg noise and stretches arm reach by
1.1x so joint limits become active.

This is workload generation only (numpy on the host); it is not a compute path of
the engine and nothing in the solver calls it.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np

from .mjcf import JNT_HINGE, RobotModel
from .model import CompiledModel


# ------------------------------------------------------------------ quaternion helpers (wxyz, vectorised)
def qmul(a, b):
    aw, ax, ay, az = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bw, bx, by, bz = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack(
        [aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
         aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw], axis=-1)


def qconj(a):
    return a * np.array([1.0, -1.0, -1.0, -1.0])


def qrot(q, v):
    w, u = q[..., :1], q[..., 1:]
    t = 2.0 * np.cross(u, v)
    return v + w * t + np.cross(u, t)


def qexp(rv):
    ang = np.linalg.norm(rv, axis=-1, keepdims=True)
    half = 0.5 * ang
    k = np.where(ang > 1e-12, np.sin(half) / np.maximum(ang, 1e-300), 0.5)
    return np.concatenate([np.cos(half), k * rv], axis=-1)


def fk_numpy(robot: RobotModel, qpos: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """MuJoCo-convention FK, float64, vectorised over frames: qpos [T,nq] -> xpos [T,nb,3], xquat [T,nb,4]."""
    T = qpos.shape[0]
    nb = robot.nbody
    xpos = np.zeros((T, nb, 3))
    xquat = np.zeros((T, nb, 4))
    xpos[:, 0] = qpos[:, 0:3]
    q0 = qpos[:, 3:7]
    xquat[:, 0] = q0 / np.linalg.norm(q0, axis=-1, keepdims=True)
    for b in range(1, nb):
        p = robot.parent[b]
        xpos[:, b] = xpos[:, p] + qrot(xquat[:, p], np.broadcast_to(robot.body_pos[b], (T, 3)))
        q = qmul(xquat[:, p], np.broadcast_to(robot.body_quat[b], (T, 4)))
        if robot.jnt_type[b] == JNT_HINGE:
            th = qpos[:, robot.qpos_adr[b]]
            jq = np.concatenate([np.cos(0.5 * th)[:, None], np.sin(0.5 * th)[:, None] * robot.jnt_axis[b]], axis=-1)
            q = qmul(q, jq)
        xquat[:, b] = q / np.linalg.norm(q, axis=-1, keepdims=True)
    return xpos, xquat


def _lowpass(x, k):
    k = max(1, min(k, x.shape[0]))
    ker = np.ones(k) / k
    return np.apply_along_axis(lambda v: np.convolve(v, ker, mode="same"), 0, x)


def synth_robot_trajectory(robot: RobotModel, T: int, rng: np.random.Generator, fps: float = 30.0, amp: float = 0.35,
                           yaw0: float = 1.0) -> np.ndarray:
    """Smooth in-limit qpos trajectory [T, nq].

    The initial heading is drawn from U(-yaw0, yaw0): a clip that starts facing away from the
    robot's qpos0 heading (|yaw| ~ pi) can park the reference algorithm in a far local minimum
    for hundreds of frames (LM damping ~ |We|^2), which is a property of the algorithm, not of
    the workload we want to time.
    """
    t = np.arange(T) / fps
    qpos = np.zeros((T, robot.nq))
    for b in robot.hinge_bodies():
        lo, hi = robot.jnt_range[b]
        if not robot.jnt_limited[b]:
            lo, hi = -1.0, 1.0
        mid, rg = 0.5 * (lo + hi), hi - lo
        a = rng.uniform(0.2, 1.0, 4)
        f = rng.uniform(0.1, 1.5, 4)
        ph = rng.uniform(0, 2 * np.pi, 4)
        s = (a[:, None] * np.sin(2 * np.pi * f[:, None] * t[None, :] + ph[:, None])).sum(0) / np.abs(a).sum()
        qpos[:, robot.qpos_adr[b]] = mid + amp * rg * s
    # root: low-passed planar random walk (<= 1.5 m/s), small height bob, yaw drift, small roll/pitch
    k = max(3, int(fps))
    vel = _lowpass(rng.normal(0, 1.0, (T, 2)), k)
    vel = vel / max(1e-9, np.abs(vel).max()) * 1.5 * rng.uniform(0.2, 1.0)
    xy = np.cumsum(vel, axis=0) / fps
    z = robot.body_pos[0, 2] + 0.05 * np.sin(2 * np.pi * rng.uniform(0.2, 1.0) * t + rng.uniform(0, 6.28))
    yaw_rate = _lowpass(rng.normal(0, 1.0, (T, 1)), k)[:, 0]
    yaw_rate = yaw_rate / max(1e-9, np.abs(yaw_rate).max()) * rng.uniform(0.2, 1.0)
    yaw = rng.uniform(-yaw0, yaw0) + np.cumsum(yaw_rate) / fps
    roll = 0.2 * np.sin(2 * np.pi * rng.uniform(0.1, 0.8) * t + rng.uniform(0, 6.28)) * rng.uniform(0, 1)
    pitch = 0.2 * np.sin(2 * np.pi * rng.uniform(0.1, 0.8) * t + rng.uniform(0, 6.28)) * rng.uniform(0, 1)
    zeros = np.zeros(T)
    qy = qexp(np.stack([zeros, zeros, yaw], -1))
    qp = qexp(np.stack([zeros, pitch, zeros], -1))
    qr = qexp(np.stack([roll, zeros, zeros], -1))
    qpos[:, 0:2] = xy
    qpos[:, 2] = z
    qpos[:, 3:7] = qmul(qmul(qy, qp), qr)
    return qpos


def human_from_robot(cm: CompiledModel, qpos: np.ndarray, rng: Optional[np.random.Generator] = None,
                     hard: bool = False, pad_to: int = 0) -> Tuple[np.ndarray, np.ndarray, List[str]]:
    """Invert target preparation so that the table-1 targets equal the robot FK poses.

    Returns (pos [T,B,3], quat [T,B,4] wxyz, body_names[B]) with B = nslot (+ padding).
    """
    robot = cm.robot
    T = qpos.shape[0]
    xpos, xquat = fk_numpy(robot, qpos)
    ns = cm.nslot
    slot_body = np.full(ns, -1, dtype=np.int64)
    for tb, ts in zip(cm.task_body[0], cm.task_slot[0]):
        slot_body[ts] = tb
    scale, poff, roff = cm.slot_scale, cm.slot_pos_off, cm.slot_rot_off
    tpos = xpos[:, slot_body]            # desired prepared targets [T,ns,3]
    tquat = xquat[:, slot_body]
    if hard:
        assert rng is not None
        tpos = tpos + rng.normal(0, 0.02, tpos.shape)
        noise = qexp(rng.normal(0, np.deg2rad(5.0), (T, ns, 3)))
        tquat = qmul(tquat, noise)
        root_t = tpos[:, cm.root_slot:cm.root_slot + 1]
        for s, n in enumerate(cm.slot_names):
            ln = n.lower()
            if any(k in ln for k in ("elbow", "wrist", "forearm", "hand")):
                tpos[:, s] = root_t[:, 0] + 1.1 * (tpos[:, s] - root_t[:, 0])
    # undo offset: q = q' (x) o^-1 ; p' = p'' - R(q') d
    hquat = qmul(tquat, np.broadcast_to(qconj(roff), (T, ns, 4)))
    p1 = tpos - qrot(tquat, np.broadcast_to(poff, (T, ns, 3)))
    # undo scaling about the root
    rs = cm.root_slot
    root_h = p1[:, rs] / scale[rs]
    hpos = (p1 - p1[:, rs:rs + 1]) / scale[None, :, None] + root_h[:, None, :]
    hpos[:, rs] = root_h
    names = list(cm.slot_names)
    if pad_to > ns:
        extra = pad_to - ns
        hpos = np.concatenate([hpos, np.repeat(hpos[:, rs:rs + 1], extra, axis=1)], axis=1)
        hquat = np.concatenate([hquat, np.repeat(hquat[:, rs:rs + 1], extra, axis=1)], axis=1)
        names += [f"_unused_{i}" for i in range(extra)]
    return hpos, hquat, names


def synth_clips(cm: CompiledModel, n_clips: int, T: int, seed: int = 0, hard: bool = False, pad_to: int = 0,
                dtype=np.float32, amp: float = 0.35):
    """n_clips clips of T frames, concatenated: (pos [N,B,3], quat [N,B,4], names, seq_offsets [n_clips+1], qpos_true [N,nq])."""
    rng = np.random.default_rng(seed)
    P, Q, G = [], [], []
    names: List[str] = []
    for _ in range(n_clips):
        qpos = synth_robot_trajectory(cm.robot, T, rng, amp=amp)
        hp, hq, names = human_from_robot(cm, qpos, rng, hard=hard, pad_to=pad_to)
        P.append(hp)
        Q.append(hq)
        G.append(qpos)
    pos = np.concatenate(P).astype(dtype)
    quat = np.concatenate(Q).astype(dtype)
    offs = np.arange(n_clips + 1, dtype=np.int64) * T
    return pos, quat, names, offs, np.concatenate(G)
