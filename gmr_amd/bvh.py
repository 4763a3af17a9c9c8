"""BVH input adapter on the fast path.

Mirror of ``load_lafan1_file`` (reference general_motion_retargeting/utils/lafan1.py:8-71) with the reference's
own file semantics (``read_bvh``, utils/lafan_vendor/extract.py:43-166): the Euler order taken from the first joint's
channels, the channel count from the last joint's, root translation from the first three motion columns, non-root local
positions = joint offsets.  Both text stages are native host code of the library: the HIERARCHY section goes through a
token grammar (``gmr_bvh_parse_header``, gmr_amd/csrc/bvh_text.h), the MOTION block -- the bulk of the file, and the
regex + float() loop that dominates loading in the reference -- through ``gmr_bvh_parse_motion`` (correctly rounded like
``float()``); Euler -> quaternion, the quaternion FK, the Y-up -> Z-up turn, cm -> m and the ``LeftFootMod`` /
``RightFootMod`` synthesis run in one HIP kernel (``gmr_bvh_fk``) and the result stays on the GPU as the
``[T, B, 3]`` / ``[T, B, 4]`` tensors ``retarget_batch`` consumes -- no per-frame dicts unless asked for.

Differences from the reference: quaternion signs are not made continuous in time (``remove_quat_discontinuities``
only flips signs; every consumer is sign-insensitive); 9-channel files are rejected.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Tuple

import numpy as np
import torch

from . import _native

_CHANNEL = {"Xrotation": 0, "Yrotation": 1, "Zrotation": 2}


class BvhAnim:
    def __init__(self, names, parents, offsets, order, positions, eulers_deg, frametime):
        self.bones: List[str] = names
        self.parents: np.ndarray = parents          # int32 [J]
        self.offsets: np.ndarray = offsets          # [J,3]
        self.order: Tuple[int, int, int] = order    # axis index of the three listed rotation channels
        self.pos: np.ndarray = positions            # local positions [T,J,3]
        self.eulers_deg: np.ndarray = eulers_deg    # [T,J,3]
        self.frametime = frametime


def _parse_motion(block: bytes, fnum: int, max_cols: int, filename: str) -> np.ndarray:
    """The first ``fnum`` non-empty lines of the motion block as a float64 ``[fnum, columns]`` array (native parser).
    ``max_cols`` bounds the row length (the hierarchy fixes it); longer rows are reported as malformed."""
    lib = _native.load()
    cap = fnum * max_cols + 1
    out = np.empty(cap, dtype=np.float64)
    n_lines, n_cols = C.c_int64(0), C.c_int64(0)
    n = lib.gmr_bvh_parse_motion(block, len(block), fnum, out.ctypes.data, cap, C.byref(n_lines), C.byref(n_cols))
    if n < 0:
        raise ValueError(f"{filename}: malformed motion block (bad number or ragged rows)")
    if n_lines.value < fnum:
        raise ValueError(f"{filename}: {n_lines.value} motion rows, header says {fnum}")
    return out[:n].reshape(fnum, n_cols.value)


def _parse_header(raw: bytes, filename: str):
    """HIERARCHY section + MOTION header through the library's tokenizer (``gmr_bvh_parse_header``, grammar in
    gmr_amd/csrc/bvh_text.h).  Returns (names, parents, offsets, channels per joint, euler order, n_frames, frame time, motion offset)."""
    lib = _native.load()
    vp = C.c_void_p
    max_j = 64
    while True:
        names_buf = C.create_string_buffer(64 * max_j)
        parents = np.empty(max_j, dtype=np.int32)
        offsets = np.empty((max_j, 3), dtype=np.float64)
        channels = np.empty(max_j, dtype=np.int32)
        order = np.empty(3, dtype=np.int32)
        fnum, ftime, moff = C.c_int64(0), C.c_double(0.0), C.c_size_t(0)
        n = lib.gmr_bvh_parse_header(raw, len(raw), max_j, names_buf, len(names_buf), parents.ctypes.data_as(vp), offsets.ctypes.data_as(vp),
                                     channels.ctypes.data_as(vp), order.ctypes.data_as(vp), C.byref(fnum), C.byref(ftime), C.byref(moff))
        if n == -2 and max_j < (1 << 16):
            max_j *= 4
            continue
        if n <= 0:
            raise ValueError(f"{filename}: not a BVH file this loader understands")
        names = names_buf.raw.split(b"\0")[:n]
        return ([x.decode("ascii") for x in names], parents[:n].copy(), offsets[:n].copy(), channels[:n].copy(),
                tuple(int(x) for x in order), int(fnum.value), float(ftime.value), int(moff.value))


def read_bvh(filename: str) -> BvhAnim:
    with open(filename, "rb") as f:
        raw = f.read()
    names, parents, offsets, chan, order, fnum, frametime, moff = _parse_header(raw, filename)
    motion_block = raw[moff:]
    channels = int(chan[-1])  # the reference shapes the motion rows by the LAST joint's channel count (extract.py:104-106)
    if (channels == 3 and (chan[0] not in (3, 6) or np.any(chan[1:] != 3))) or (channels == 6 and np.any(chan != 6)) \
            or (channels == 9 and (chan[0] != 3 or np.any(chan[1:] != 9))):
        raise NotImplementedError(f"{filename}: joints with mixed channel counts are not supported")
    J = len(names)
    data = _parse_motion(motion_block, fnum, 9 * J + 3, filename)  # the ctypes call releases the GIL: files parse in parallel threads
    offs = np.asarray(offsets, dtype=np.float64)
    positions = np.repeat(offs[None], fnum, axis=0)
    if channels == 3 and chan[0] == 3:  # a root without translation channels: the reference reads its first three columns as one anyway
        raise NotImplementedError(f"{filename}: a 3-channel root is not supported")
    if channels == 3:
        if data.shape[1] != 3 + 3 * J:
            raise ValueError(f"{filename}: expected {3 + 3 * J} columns, found {data.shape[1]}")
        positions[:, 0] = data[:, 0:3]
        rotations = data[:, 3:].reshape(fnum, J, 3)
    elif channels == 6:
        if data.shape[1] != 6 * J:
            raise ValueError(f"{filename}: expected {6 * J} columns, found {data.shape[1]}")
        blk = data.reshape(fnum, J, 6)
        positions = blk[:, :, 0:3].copy()
        rotations = blk[:, :, 3:6].copy()
    elif channels == 9:
        # extract.py:152-156: three root position values, then (position, rotation, scale) per non-root joint; a joint's local
        # position is its offset plus position * scale, the root keeps a zero rotation
        if len(names) < 2 or data.shape[1] != 3 + 9 * (J - 1):
            raise ValueError(f"{filename}: expected {3 + 9 * (J - 1)} columns, found {data.shape[1]}")
        positions[:, 0] = data[:, 0:3]
        blk = data[:, 3:].reshape(fnum, J - 1, 9)
        positions[:, 1:] += blk[:, :, 0:3] * blk[:, :, 6:9]
        rotations = np.zeros((fnum, J, 3))
        rotations[:, 1:] = blk[:, :, 3:6]
    else:
        raise NotImplementedError(f"{filename}: {channels}-channel joints are not supported")
    return BvhAnim(names, np.asarray(parents, dtype=np.int32), offs, order, positions, rotations, frametime)


class BvhClip:
    """Global joint poses of one BVH clip on the GPU, in the layout ``retarget_batch`` takes."""

    def __init__(self, pos: torch.Tensor, quat: torch.Tensor, names: List[str], human_height: float, frametime):
        self.pos, self.quat, self.body_names, self.human_height, self.frametime = pos, quat, names, human_height, frametime

    def __len__(self):
        return int(self.pos.shape[0])

    def frames(self) -> List[Dict[str, Tuple[np.ndarray, np.ndarray]]]:
        """The reference's return shape: one dict ``{bone: (position[3], quat_wxyz[4])}`` per frame."""
        p, q = self.pos.cpu().numpy(), self.quat.cpu().numpy()
        return [{n: (p[t, i], q[t, i]) for i, n in enumerate(self.body_names)} for t in range(p.shape[0])]


def _estimate_height(last: Dict[str, np.ndarray]) -> float:
    """lafan1.py:45-69 on the last frame's positions."""
    if not last:
        return 1.75
    if "Head" in last:
        feet = [last[k][2] for k in ("LeftFootMod", "RightFootMod", "LeftFoot", "RightFoot") if k in last]
        h = last["Head"][2] - (min(feet) if feet else min(v[2] for v in last.values()))
    else:
        z = [v[2] for v in last.values()]
        h = max(z) - min(z)
    if not np.isfinite(h) or h < 0.9 or h > 2.3:
        h = 1.75
    return float(h)


def load_lafan1_file(bvh_file: str, device: int = 0) -> BvhClip:
    """BVH file -> global poses (metres, Z-up, wxyz) on ``cuda:device`` + the reference's height estimate."""
    anim = read_bvh(bvh_file)
    lib = _native.load()
    dev = torch.device("cuda", device)
    T, J = anim.pos.shape[0], len(anim.bones)
    extra_names, extra_pos, extra_rot = [], [], []
    for side in ("Left", "Right"):
        if f"{side}Foot" in anim.bones and f"{side}Toe" in anim.bones:  # lafan1.py:36-39
            extra_names.append(f"{side}FootMod")
            extra_pos.append(anim.bones.index(f"{side}Foot"))
            extra_rot.append(anim.bones.index(f"{side}Toe"))
    E = len(extra_names)
    lp = torch.from_numpy(np.ascontiguousarray(anim.pos)).to(dev)
    er = torch.from_numpy(np.ascontiguousarray(np.radians(anim.eulers_deg))).to(dev)
    pos = torch.empty((T, J + E, 3), dtype=torch.float64, device=dev)
    quat = torch.empty((T, J + E, 4), dtype=torch.float64, device=dev)
    parents = np.ascontiguousarray(anim.parents, dtype=np.int32)
    order = np.asarray(anim.order, dtype=np.int32)
    ep, erot = np.asarray(extra_pos, dtype=np.int32), np.asarray(extra_rot, dtype=np.int32)
    vp = C.c_void_p
    rc = lib.gmr_bvh_fk(parents.ctypes.data_as(vp), J, order.ctypes.data_as(vp), ep.ctypes.data_as(vp) if E else None,
                        erot.ctypes.data_as(vp) if E else None, E, vp(lp.data_ptr()), vp(er.data_ptr()), T, 0.01,
                        vp(pos.data_ptr()), vp(quat.data_ptr()), vp(torch.cuda.current_stream(dev).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"gmr_bvh_fk failed with status {rc} ({J} joints, {E} extra)")
    names = list(anim.bones) + extra_names
    height = 1.75
    if T > 0:
        last = pos[-1].cpu().numpy()
        height = _estimate_height({n: last[i] for i, n in enumerate(names)})
    return BvhClip(pos, quat, names, height, anim.frametime)


class BvhBatch:
    """Several BVH clips on the GPU as one batch: ``pos [N, B, 3]``, ``quat [N, B, 4]`` (concatenated clips), ``seq_offsets``,
    one height estimate per clip -- the arguments ``retarget_batch(..., seq_offsets=..., human_heights=...)`` takes."""

    def __init__(self, pos, quat, names, seq_offsets, heights, frametimes, files):
        self.pos, self.quat, self.body_names = pos, quat, names
        self.seq_offsets, self.human_heights, self.frametimes, self.files = seq_offsets, heights, frametimes, files

    def __len__(self):
        return len(self.files)


def load_lafan1_files(bvh_files, device: int = 0, threads: int = 8) -> BvhBatch:
    """A folder's worth of BVH files -> one GPU batch (the file loop of scripts/bvh_to_robot_dataset.py:59-80, where every file
    is parsed with regexes and turned into per-frame dicts one after the other).  The text of the files is parsed on ``threads``
    host threads (both native parsers release the GIL), the clips are concatenated, and ONE ``gmr_bvh_fk`` launch does Euler ->
    quaternion, the skeleton FK, Y-up -> Z-up, cm -> m and the FootMod synthesis for all of them.  All files must share one
    skeleton (names, parents, Euler order), as a dataset does; the height estimate of every clip (lafan1.py:45-69, from its own
    last frame) comes back with the batch."""
    from concurrent.futures import ThreadPoolExecutor
    files = [str(f) for f in bvh_files]
    if not files:
        raise ValueError("no files")
    with ThreadPoolExecutor(max_workers=max(1, min(threads, len(files)))) as ex:
        anims = list(ex.map(read_bvh, files))
    a0 = anims[0]
    for f, a in zip(files, anims):
        if a.bones != a0.bones or not np.array_equal(a.parents, a0.parents) or a.order != a0.order:
            raise ValueError(f"{f}: skeleton differs from {files[0]} (one batch = one skeleton)")
    lib = _native.load()
    dev = torch.device("cuda", device)
    lens = np.array([a.pos.shape[0] for a in anims], dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    N, J = int(offs[-1]), len(a0.bones)
    extra_names, extra_pos, extra_rot = [], [], []
    for side in ("Left", "Right"):
        if f"{side}Foot" in a0.bones and f"{side}Toe" in a0.bones:  # lafan1.py:36-39
            extra_names.append(f"{side}FootMod")
            extra_pos.append(a0.bones.index(f"{side}Foot"))
            extra_rot.append(a0.bones.index(f"{side}Toe"))
    E = len(extra_names)
    lp_h = torch.empty((N, J, 3), dtype=torch.float64, pin_memory=True)
    er_h = torch.empty((N, J, 3), dtype=torch.float64, pin_memory=True)
    for a, o in zip(anims, offs[:-1]):
        n = a.pos.shape[0]
        lp_h[o:o + n] = torch.from_numpy(a.pos)
        np.radians(a.eulers_deg, out=er_h[o:o + n].numpy())
    lp, er = lp_h.to(dev, non_blocking=True), er_h.to(dev, non_blocking=True)
    pos = torch.empty((N, J + E, 3), dtype=torch.float64, device=dev)
    quat = torch.empty((N, J + E, 4), dtype=torch.float64, device=dev)
    names = list(a0.bones) + extra_names
    if N > 0:
        parents = np.ascontiguousarray(a0.parents, dtype=np.int32)
        order = np.asarray(a0.order, dtype=np.int32)
        ep, erot = np.asarray(extra_pos, dtype=np.int32), np.asarray(extra_rot, dtype=np.int32)
        vp = C.c_void_p
        rc = lib.gmr_bvh_fk(parents.ctypes.data_as(vp), J, order.ctypes.data_as(vp), ep.ctypes.data_as(vp) if E else None,
                            erot.ctypes.data_as(vp) if E else None, E, vp(lp.data_ptr()), vp(er.data_ptr()), N, 0.01,
                            vp(pos.data_ptr()), vp(quat.data_ptr()), vp(torch.cuda.current_stream(dev).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"gmr_bvh_fk failed with status {rc} ({J} joints, {E} extra)")
    heights = []
    if N > 0:
        last = pos[torch.from_numpy(offs[1:][lens > 0] - 1).to(dev)].cpu().numpy()  # every clip's last frame
        k = 0
        for n in lens:
            if n == 0:
                heights.append(1.75)
                continue
            heights.append(_estimate_height({nm: last[k, i] for i, nm in enumerate(names)}))
            k += 1
    else:
        heights = [1.75] * len(files)
    return BvhBatch(pos, quat, names, offs, heights, [a.frametime for a in anims], files)
