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
    if robot.planar_base:  # a mobile base drives in the plane: fixed height, heading only (same random draws as above)
        qpos[:, 2] = robot.body_pos[0, 2]
        qpos[:, 3:7] = qy
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


# ------------------------------------------------------------------ the same generator on a torch device
# bench.py's "unshaped" workload (every clip distinct, any initial heading, variable lengths) is 2.5e7 frames: minutes of
# numpy on the host, seconds as torch tensor ops on the GPU.  Same construction as above (trajectory -> FK -> inverse target
# preparation), drawn from a torch.Generator; workload generation only, not a compute path of the engine.
def _t_qmul(a, b):
    import torch
    aw, ax, ay, az = a.unbind(-1)
    bw, bx, by, bz = b.unbind(-1)
    return torch.stack([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw], dim=-1)


def _t_qrot(q, v):
    import torch
    w, u = q[..., :1], q[..., 1:]
    t = 2.0 * torch.cross(u, v.expand_as(u), dim=-1)
    return v + w * t + torch.cross(u, t, dim=-1)


def _t_qexp(rv):
    import torch
    ang = rv.norm(dim=-1, keepdim=True)
    half = 0.5 * ang
    k = torch.where(ang > 1e-12, torch.sin(half) / ang.clamp_min(1e-300), torch.full_like(ang, 0.5))
    return torch.cat([torch.cos(half), k * rv], dim=-1)


def _t_lowpass(x, k):
    """Centred moving average of width k along dim 1 of [S, T, C] (numpy's convolve(mode="same") with a box)."""
    import torch
    S, T, C = x.shape
    k = max(1, min(k, T))
    w = torch.ones((C, 1, k), dtype=x.dtype, device=x.device) / k
    lo = (k - 1) // 2 + ((k - 1) % 2)  # np.convolve 'same' centring for even k
    y = torch.nn.functional.pad(x.permute(0, 2, 1), (lo, k - 1 - lo))
    return torch.nn.functional.conv1d(y, w, groups=C).permute(0, 2, 1)


def synth_clips_torch(cm: CompiledModel, lengths, seed: int, device, hard=False, yaw0: float = np.pi, amp: float = 0.35,
                      fps: float = 30.0, dtype=None, clips_per_pass: int = 256):
    """Clips of the given lengths, all distinct, generated on ``device``: (pos [N,B,3], quat [N,B,4], names, seq_offsets).

    ``hard``: bool, or a bool per clip (2 cm / 5 deg noise + 1.1x arm reach, as in the numpy generator).  ``yaw0 = pi``: any
    initial heading, like real capture data (the numpy generator's default of 1 rad keeps a clip's first frames out of the
    reference algorithm's slow far-heading start-up)."""
    import torch
    dtype = dtype or torch.float32
    robot = cm.robot
    lengths = np.asarray(lengths, dtype=np.int64)
    S = len(lengths)
    hard = np.broadcast_to(np.asarray(hard, dtype=bool), (S,))
    offs = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    N, ns = int(offs[-1]), cm.nslot
    pos = torch.empty((N, ns, 3), dtype=dtype, device=device)
    quat = torch.empty((N, ns, 4), dtype=dtype, device=device)
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))
    f64 = torch.float64
    U = lambda shape, lo, hi: lo + (hi - lo) * torch.rand(shape, generator=gen, dtype=f64, device=device)  # noqa: E731
    Nrm = lambda shape: torch.randn(shape, generator=gen, dtype=f64, device=device)  # noqa: E731
    T_ = lambda a: torch.as_tensor(np.asarray(a), dtype=f64, device=device)  # noqa: E731
    slot_body = np.full(ns, -1, dtype=np.int64)
    for tb, ts in zip(cm.task_body[0], cm.task_slot[0]):
        slot_body[ts] = tb
    scale, poff, roff = T_(cm.slot_scale), T_(cm.slot_pos_off), T_(cm.slot_rot_off)
    roff_c = roff * T_([1.0, -1.0, -1.0, -1.0])
    stretch = np.array([any(k in n.lower() for k in ("elbow", "wrist", "forearm", "hand")) for n in cm.slot_names])
    rs = cm.root_slot
    order = np.argsort(-lengths, kind="stable")
    for p0 in range(0, S, clips_per_pass):
        ids = order[p0:p0 + clips_per_pass]
        s, Tm = len(ids), int(lengths[ids].max())
        t = torch.arange(Tm, dtype=f64, device=device) / fps
        qpos = torch.zeros((s, Tm, robot.nq), dtype=f64, device=device)
        for b in robot.hinge_bodies():
            lo, hi = robot.jnt_range[b] if robot.jnt_limited[b] else (-1.0, 1.0)
            a, f, ph = U((s, 4, 1), 0.2, 1.0), U((s, 4, 1), 0.1, 1.5), U((s, 4, 1), 0.0, 2 * np.pi)
            sig = (a * torch.sin(2 * np.pi * f * t + ph)).sum(1) / a.abs().sum(1)
            qpos[:, :, robot.qpos_adr[b]] = 0.5 * (lo + hi) + amp * (hi - lo) * sig
        k = max(3, int(fps))
        vel = _t_lowpass(Nrm((s, Tm, 2)), k)
        vel = vel / vel.abs().amax(dim=(1, 2), keepdim=True).clamp_min(1e-9) * 1.5 * U((s, 1, 1), 0.2, 1.0)
        qpos[:, :, 0:2] = torch.cumsum(vel, dim=1) / fps
        qpos[:, :, 2] = robot.body_pos[0, 2] + 0.05 * torch.sin(2 * np.pi * U((s, 1), 0.2, 1.0) * t + U((s, 1), 0.0, 6.28))
        yr = _t_lowpass(Nrm((s, Tm, 1)), k)[:, :, 0]
        yr = yr / yr.abs().amax(dim=1, keepdim=True).clamp_min(1e-9) * U((s, 1), 0.2, 1.0)
        yaw = U((s, 1), -yaw0, yaw0) + torch.cumsum(yr, dim=1) / fps
        roll = 0.2 * torch.sin(2 * np.pi * U((s, 1), 0.1, 0.8) * t + U((s, 1), 0.0, 6.28)) * U((s, 1), 0.0, 1.0)
        pitch = 0.2 * torch.sin(2 * np.pi * U((s, 1), 0.1, 0.8) * t + U((s, 1), 0.0, 6.28)) * U((s, 1), 0.0, 1.0)
        z = torch.zeros_like(yaw)
        qpos[:, :, 3:7] = _t_qmul(_t_qmul(_t_qexp(torch.stack([z, z, yaw], -1)), _t_qexp(torch.stack([z, pitch, z], -1))),
                                  _t_qexp(torch.stack([roll, z, z], -1)))
        if robot.planar_base:  # (as in synth_robot_trajectory: fixed height, heading only)
            qpos[:, :, 2] = float(robot.body_pos[0, 2])
            qpos[:, :, 3:7] = _t_qexp(torch.stack([z, z, yaw], -1))
        # FK (MuJoCo convention) for the bodies the tasks need
        q2 = qpos.reshape(s * Tm, robot.nq)
        need = np.zeros(robot.nbody, dtype=bool)
        for b in slot_body:
            while b >= 0 and not need[b]:
                need[b] = True
                b = robot.parent[b]
        xpos, xquat = {0: q2[:, 0:3]}, {0: q2[:, 3:7] / q2[:, 3:7].norm(dim=-1, keepdim=True)}
        for b in range(1, robot.nbody):
            if not need[b]:
                continue
            p = int(robot.parent[b])
            xpos[b] = xpos[p] + _t_qrot(xquat[p], T_(robot.body_pos[b]))
            q = _t_qmul(xquat[p], T_(robot.body_quat[b]).expand(s * Tm, 4))
            if robot.jnt_type[b] == JNT_HINGE:
                th = q2[:, robot.qpos_adr[b]]
                q = _t_qmul(q, torch.cat([torch.cos(0.5 * th)[:, None], torch.sin(0.5 * th)[:, None] * T_(robot.jnt_axis[b])], -1))
            xquat[b] = q / q.norm(dim=-1, keepdim=True)
        tpos = torch.stack([xpos[int(b)] for b in slot_body], 1)     # [s*Tm, ns, 3]
        tquat = torch.stack([xquat[int(b)] for b in slot_body], 1)
        del xpos, xquat
        hm = torch.as_tensor(hard[ids], device=device).repeat_interleave(Tm)
        if bool(hard[ids].any()):
            tpos = tpos + hm[:, None, None] * 0.02 * Nrm(tpos.shape)
            noise = _t_qexp(np.deg2rad(5.0) * Nrm(tpos.shape))
            tquat = torch.where(hm[:, None, None], _t_qmul(tquat, noise), tquat)
            root_t = tpos[:, rs:rs + 1]
            far = root_t + 1.1 * (tpos - root_t)
            sel = hm[:, None, None] & torch.as_tensor(stretch, device=device)[None, :, None]
            tpos = torch.where(sel, far, tpos)
        hquat = _t_qmul(tquat, roff_c.expand_as(tquat))
        p1 = tpos - _t_qrot(tquat, poff.expand_as(tpos))
        root_h = p1[:, rs] / scale[rs]
        hpos = (p1 - p1[:, rs:rs + 1]) / scale[None, :, None] + root_h[:, None, :]
        hpos[:, rs] = root_h
        hpos, hquat = hpos.reshape(s, Tm, ns, 3), hquat.reshape(s, Tm, ns, 4)
        for j, c in enumerate(ids):
            L = int(lengths[c])
            pos[offs[c]:offs[c] + L] = hpos[j, :L].to(dtype)
            quat[offs[c]:offs[c] + L] = hquat[j, :L].to(dtype)
        del hpos, hquat, tpos, tquat, qpos, q2
    return pos, quat, list(cm.slot_names), offs


# ------------------------------------------------------------------ adapter workloads (rows f-1, f-2): LAFAN1-shaped BVH rows, AMASS-shaped SMPL-X arrays
# The 22-bone LAFAN1 skeleton (bone names as bvh_to_g1.json reads them; offsets in cm, Y-up: a plausible human, not LAFAN1's data)
LAFAN1_BONES: List[Tuple[str, int, Tuple[float, float, float]]] = [
    ("Hips", -1, (0, 0, 0)), ("LeftUpLeg", 0, (10, -5, 0)), ("LeftLeg", 1, (0, -42, 0)), ("LeftFoot", 2, (0, -40, 0)), ("LeftToe", 3, (0, -6, 14)),
    ("RightUpLeg", 0, (-10, -5, 0)), ("RightLeg", 5, (0, -42, 0)), ("RightFoot", 6, (0, -40, 0)), ("RightToe", 7, (0, -6, 14)),
    ("Spine", 0, (0, 8, 0)), ("Spine1", 9, (0, 12, 0)), ("Spine2", 10, (0, 12, 0)), ("Neck", 11, (0, 22, 0)), ("Head", 12, (0, 10, 0)),
    ("LeftShoulder", 11, (4, 18, 0)), ("LeftArm", 14, (14, 0, 0)), ("LeftForeArm", 15, (28, 0, 0)), ("LeftHand", 16, (25, 0, 0)),
    ("RightShoulder", 11, (-4, 18, 0)), ("RightArm", 18, (-14, 0, 0)), ("RightForeArm", 19, (-28, 0, 0)), ("RightHand", 20, (-25, 0, 0))]


def lafan_rows_torch(n_frames: int, device, seed: int = 0):
    """3-channel motion rows [n_frames, 3 + 3 * 22] (root translation in cm, ZYX Euler angles in degrees) of a LAFAN1-shaped
    skeleton, generated on the device: band-limited joint angles, a wandering root.  Returns (rows, parents, offsets, order)."""
    import torch
    g = torch.Generator(device=device).manual_seed(seed)
    J = len(LAFAN1_BONES)
    t = torch.arange(n_frames, dtype=torch.float64, device=device)[:, None] / 30.0
    amp = torch.rand((1, 3 * J), generator=g, dtype=torch.float64, device=device) * 23.0 + 2.0
    frq = torch.rand((1, 3 * J), generator=g, dtype=torch.float64, device=device) * 1.1 + 0.1
    ph = torch.rand((1, 3 * J), generator=g, dtype=torch.float64, device=device) * 6.28
    rows = torch.empty((n_frames, 3 + 3 * J), dtype=torch.float64, device=device)
    rows[:, 3:] = amp * torch.sin(2 * np.pi * frq * t + ph)
    rows[:, 4] += torch.rand((n_frames,), generator=g, dtype=torch.float64, device=device) * 360.0 - 180.0   # heading: any
    rows[:, 0] = 100.0 * torch.sin(0.05 * t[:, 0])
    rows[:, 1] = 92.0 + 2.0 * torch.sin(t[:, 0])
    rows[:, 2] = 100.0 * torch.cos(0.031 * t[:, 0])
    parents = np.array([p for _, p, _ in LAFAN1_BONES], dtype=np.int32)
    offsets = np.array([o for _, _, o in LAFAN1_BONES], dtype=np.float64)
    return rows, parents, offsets, (2, 1, 0)


def smplx_arrays_torch(n_frames: int, device, n_joints: int = 55, joints_stride: int = 127, seed: int = 0):
    """AMASS-shaped SMPL-X body-model outputs on the device: global_orient [T,3], full_pose [T,J,3] (axis-angle, smooth in time:
    neighbouring mocap frames are a fraction of a degree apart), joints [T,joints_stride,3]."""
    import torch
    g = torch.Generator(device=device).manual_seed(seed)
    t = torch.arange(n_frames, dtype=torch.float64, device=device)[:, None, None] / 120.0
    base = torch.randn((1, n_joints, 3), generator=g, dtype=torch.float64, device=device) * 0.4
    frq = torch.rand((1, n_joints, 3), generator=g, dtype=torch.float64, device=device) * 1.4 + 0.1
    ph = torch.rand((1, n_joints, 3), generator=g, dtype=torch.float64, device=device) * 6.28
    full_pose = base + 0.5 * torch.sin(2 * np.pi * frq * t + ph)
    global_orient = full_pose[:, 0].contiguous()
    joints = torch.randn((1, joints_stride, 3), generator=g, dtype=torch.float64, device=device) * 0.5 \
        + 0.3 * torch.sin(2 * np.pi * 0.2 * t + torch.rand((1, joints_stride, 3), generator=g, dtype=torch.float64, device=device) * 6.28)
    return global_orient, full_pose.contiguous(), joints.contiguous()


def _bvh_header(names: List[str], parents, offsets, channels6: bool) -> str:
    """HIERARCHY text of a skeleton: 3 rotation channels per joint (ZYX) + 3 root translation channels, or 6 channels everywhere."""
    children = {i: [j for j, p in enumerate(parents) if p == i] for i in range(len(names))}
    out = ["HIERARCHY"]

    def emit(i, depth):
        ind = "\t" * depth
        out.append(f"{ind}{'ROOT' if parents[i] < 0 else 'JOINT'} {names[i]}")
        out.append(ind + "{")
        o = offsets[i]
        out.append(f"{ind}\tOFFSET {o[0]:.6f} {o[1]:.6f} {o[2]:.6f}")
        if channels6 or parents[i] < 0:
            out.append(f"{ind}\tCHANNELS 6 Xposition Yposition Zposition Zrotation Yrotation Xrotation")
        else:
            out.append(f"{ind}\tCHANNELS 3 Zrotation Yrotation Xrotation")
        if not children[i]:
            out.extend([f"{ind}\tEnd Site", ind + "\t{", f"{ind}\t\tOFFSET 0.000000 5.000000 0.000000", ind + "\t}"])
        for c in children[i]:
            emit(c, depth + 1)
        out.append(ind + "}")
    emit(0, 0)
    return "\n".join(out)


def _write_rows(path: str, header: str, rows: np.ndarray, frame_time: float = 1.0 / 30.0) -> None:
    body = "\n".join(" ".join(["%.6f"] * rows.shape[1]) % tuple(r) for r in rows)
    with open(path, "w") as fh:
        fh.write(f"{header}\nMOTION\nFrames: {rows.shape[0]}\nFrame Time: {frame_time:.6f}\n{body}\n")


def write_lafan_shaped_files(folder: str, n_files: int, frames: int, seed: int = 0) -> List[str]:
    """LAFAN1-shaped BVH files (22 bones, root translation + ZYX Euler angles, cm, Y-up) with band-limited random joint angles:
    the shape of the text a LAFAN1 folder holds.  The motion is NOT something a robot can follow (bone frames are arbitrary), so
    these files measure the loader, not the solver."""
    import os
    names = [n for n, _, _ in LAFAN1_BONES]
    parents = [p for _, p, _ in LAFAN1_BONES]
    offsets = [o for _, _, o in LAFAN1_BONES]
    header = _bvh_header(names, parents, offsets, channels6=False)
    rng = np.random.default_rng(seed)
    J, files = len(names), []
    t = np.arange(frames) / 30.0
    for k in range(n_files):
        a, f, ph = rng.uniform(2, 25, (1, 3 * J)), rng.uniform(0.1, 1.2, (1, 3 * J)), rng.uniform(0, 6.28, (1, 3 * J))
        ang = a * np.sin(2 * np.pi * f * t[:, None] + ph)
        ang[:, 1] += rng.uniform(-180, 180) + np.cumsum(rng.normal(0, 0.5, frames))  # heading (Y-up: yaw is the Y rotation)
        root = np.stack([np.cumsum(rng.normal(0, 1.0, frames)), 92 + 2 * np.sin(t), np.cumsum(rng.normal(0, 1.0, frames))], -1)
        p = os.path.join(folder, f"clip{k:03d}.bvh")
        _write_rows(p, header, np.concatenate([root, ang], axis=1))
        files.append(p)
    return files


def write_keypoint_files(folder: str, pos: np.ndarray, quat: np.ndarray, names: List[str], seq_offsets, prefix: str = "kp",
                         head_height: Optional[float] = None) -> List[str]:
    """Robot-consistent key-points (``synth_clips*`` output for a bvh_to_* config: metres, Z-up, wxyz) as BVH files the loader turns
    back into the same key-points: a flat hierarchy -- every bone a child of a root at the origin, 6 channels each, so a bone's
    channels are its global pose in the file's frame (cm, Y-up, ZYX Euler degrees) -- with ``<Side>FootMod`` stored the way LAFAN1
    implies it (lafan1.py:36-39): ``<Side>Foot`` carries its position, ``<Side>Toe`` its orientation.  ``head_height`` adds a
    ``Head`` bone that far above the lower foot, so that the loader's height estimate (lafan1.py:45-69) returns it."""
    import os
    from scipy.spatial.transform import Rotation as R
    bones, src_p, src_q = ["Root"], [-1], [-1]
    for i, n in enumerate(names):
        if n.endswith("FootMod"):
            side = n[: -len("FootMod")]
            bones += [side + "Foot", side + "Toe"]; src_p += [i, i]; src_q += [i, i]
        else:
            bones.append(n); src_p.append(i); src_q.append(i)
    feet = [i for i, n in enumerate(names) if n.endswith("FootMod")]
    if head_height is not None and feet and "Head" not in bones:
        bones.append("Head"); src_p.append(-2); src_q.append(0)
    parents = [-1] + [0] * (len(bones) - 1)
    header = _bvh_header(bones, parents, [(0.0, 0.0, 0.0)] * len(bones), channels6=True)
    offs = np.asarray(seq_offsets, dtype=np.int64)
    h = np.sqrt(0.5)
    rq_inv = np.array([h, -h, 0.0, 0.0])   # the loader turns the file's frame by [h, h, 0, 0] (Y-up -> Z-up)
    files = []
    for s in range(len(offs) - 1):
        a, b = int(offs[s]), int(offs[s + 1])
        T = b - a
        rows = np.zeros((T, len(bones), 6))
        for j in range(1, len(bones)):
            if src_p[j] == -2:   # the synthetic head: above the root, head_height over the lower foot
                p = pos[a:b, 0].astype(np.float64).copy()
                p[:, 2] = pos[a:b, feet, 2].astype(np.float64).min(axis=1) + head_height
            else:
                p = pos[a:b, src_p[j]].astype(np.float64)
            rows[:, j, 0], rows[:, j, 1], rows[:, j, 2] = 100.0 * p[:, 0], 100.0 * p[:, 2], -100.0 * p[:, 1]   # inverse of (x, -z, y) / 100
            q = qmul(np.broadcast_to(rq_inv, (T, 4)), quat[a:b, src_q[j]].astype(np.float64))
            rows[:, j, 3:6] = R.from_quat(q[:, [1, 2, 3, 0]]).as_euler("ZYX", degrees=True)   # q = qz (x) qy (x) qx: the channel order
        f = os.path.join(folder, f"{prefix}{s:03d}.bvh")
        _write_rows(f, header, rows.reshape(T, -1))
        files.append(f)
    return files


def smplx_joint_arrays_from_keypoints(pos, quat, names: List[str]):
    """Body-model-shaped arrays whose adapter output (30 fps, no resampling) is the given key-points: torch ``pos [N, B, 3]``,
    ``quat [N, B, 4]`` (wxyz) with SMPL-X joint names -> (global_orient [N, 3], full_pose [N, 165], joints [N, 55, 3]) in float64 on the
    same device.  Joints a config does not name keep the identity orientation and the origin; a joint's axis-angle is the rotation that
    takes its parent's global orientation to its own (what smpl.py:179-196 chains back together)."""
    import torch
    from .smplx_adapter import SMPLX_JOINT_NAMES, SMPLX_PARENTS
    N, J, dev = int(pos.shape[0]), len(SMPLX_PARENTS), pos.device
    G = torch.zeros((N, J, 4), dtype=torch.float64, device=dev)
    G[..., 0] = 1.0
    P = torch.zeros((N, J, 3), dtype=torch.float64, device=dev)
    for c, n in enumerate(names):
        if n in SMPLX_JOINT_NAMES:
            j = SMPLX_JOINT_NAMES.index(n)
            q = quat[:, c].to(torch.float64)
            G[:, j] = q / q.norm(dim=-1, keepdim=True)
            P[:, j] = pos[:, c].to(torch.float64)
    par = torch.as_tensor(SMPLX_PARENTS, device=dev)
    Gp = G[:, par.clamp_min(0)]
    Gp[:, 0] = torch.tensor([1.0, 0.0, 0.0, 0.0], dtype=torch.float64, device=dev)
    L = _t_qmul(Gp * torch.tensor([1.0, -1.0, -1.0, -1.0], dtype=torch.float64, device=dev), G)
    L = torch.where(L[..., :1] < 0, -L, L)
    v = L[..., 1:]
    s = v.norm(dim=-1, keepdim=True)
    ang = 2.0 * torch.atan2(s, L[..., :1])
    rv = torch.where(s > 1e-12, v / s.clamp_min(1e-300) * ang, 2.0 * v)
    return rv[:, 0].contiguous(), rv.reshape(N, 3 * J).contiguous(), P


def write_smplx_joint_files(folder: str, pos, quat, names: List[str], seq_offsets, fps: float = 30.0, heights: Optional[Sequence[float]] = None,
                            prefix: str = "clip", dtype=np.float32) -> List[str]:
    """``smplx_joint_arrays_from_keypoints`` per clip as joint-array files (``gmr_amd.smplx_adapter.save_joint_file``; float32 like a body
    model's output unless ``dtype`` says otherwise).  ``heights``: written as betas[0] = (h - 1.66) / 0.1, what the loader inverts."""
    import os
    from .smplx_adapter import save_joint_file
    go, fp, jt = (a.cpu().numpy() for a in smplx_joint_arrays_from_keypoints(pos, quat, names))
    files = []
    for k, (a, b) in enumerate(zip(seq_offsets[:-1], seq_offsets[1:])):
        betas = np.zeros(16)
        if heights is not None:
            betas[0] = (float(heights[k]) - 1.66) / 0.1
        f = os.path.join(folder, f"{prefix}_{k:05d}.npz")
        save_joint_file(f, jt[a:b].astype(dtype), go[a:b].astype(dtype), fp[a:b].astype(dtype), fps, betas)
        files.append(f)
    return files
