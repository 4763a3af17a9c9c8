"""BVH input adapter on the fast path.

Mirror of ``load_lafan1_file`` (reference general_motion_retargeting/utils/lafan1.py:8-71) with the reference's
own file semantics (``read_bvh``, utils/lafan_vendor/extract.py:43-166): the Euler order taken from the first joint's
channels, the channel count from the last joint's, root translation from the first three motion columns, non-root local
positions = joint offsets.  Both text stages are native host code of the library: the HIERARCHY section goes through a
token grammar (``gmr_bvh_parse_header``, gmr_amd/csrc/bvh_text.h), the MOTION block -- the bulk of the file, and the
regex + float() loop that dominates loading in the reference -- through ``gmr_bvh_parse_motion`` (correctly rounded like
``float()``); the parsed rows go to the GPU as they stand, and row slicing (3-, 6- and 9-channel layouts), degrees -> radians,
Euler -> quaternion, the quaternion FK, the Y-up -> Z-up turn, cm -> m and the ``LeftFootMod`` / ``RightFootMod`` synthesis run in
one HIP kernel (``gmr_bvh_fk_rows``); the result stays on the GPU as the ``[T, B, 3]`` / ``[T, B, 4]`` tensors ``retarget_batch``
consumes -- no per-frame dicts unless asked for, and with ``columns=`` only the bones an IK config reads.

Difference from the reference: quaternion signs are not made continuous in time (``remove_quat_discontinuities``
only flips signs; every consumer is sign-insensitive).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Tuple

import numpy as np
import torch

from . import _native

_CHANNEL = {"Xrotation": 0, "Yrotation": 1, "Zrotation": 2}


class BvhAnim:
    """One parsed BVH file: the skeleton of the HIERARCHY section and the MOTION rows as they stand in the file (degrees, file
    units).  ``pos`` / ``eulers_deg`` are the reference's ``Anim`` arrays (extract.py:140-166), sliced from the rows on demand;
    the GPU path never builds them -- ``gmr_bvh_fk_rows`` reads the rows directly."""

    def __init__(self, names, parents, offsets, order, rows, channels, frametime):
        self.bones: List[str] = names
        self.parents: np.ndarray = parents          # int32 [J]
        self.offsets: np.ndarray = offsets          # [J,3]
        self.order: Tuple[int, int, int] = order    # axis index of the three listed rotation channels
        self.rows: np.ndarray = rows                # [T, ncol] float64
        self.channels: int = channels               # 3, 6 or 9 (row layout, include/gmr_amd.h gmr_bvh_fk_rows)
        self.frametime = frametime

    def __len__(self):
        return int(self.rows.shape[0])

    @property
    def pos(self) -> np.ndarray:
        """local positions [T,J,3]"""
        T, J, data = self.rows.shape[0], len(self.bones), self.rows
        if self.channels == 6:
            return data.reshape(T, J, 6)[:, :, 0:3].copy()
        positions = np.repeat(np.asarray(self.offsets, dtype=np.float64)[None], T, axis=0)
        positions[:, 0] = data[:, 0:3]
        if self.channels == 9:
            # extract.py:152-156: three root position values, then (position, rotation, scale) per non-root joint; a joint's local
            # position is its offset plus position * scale, the root keeps a zero rotation
            blk = data[:, 3:].reshape(T, J - 1, 9)
            positions[:, 1:] += blk[:, :, 0:3] * blk[:, :, 6:9]
        return positions

    @property
    def eulers_deg(self) -> np.ndarray:
        """channel angles [T,J,3], degrees"""
        T, J, data = self.rows.shape[0], len(self.bones), self.rows
        if self.channels == 3:
            return data[:, 3:].reshape(T, J, 3)
        if self.channels == 6:
            return data.reshape(T, J, 6)[:, :, 3:6].copy()
        rotations = np.zeros((T, J, 3))
        rotations[:, 1:] = data[:, 3:].reshape(T, J - 1, 9)[:, :, 3:6]
        return rotations


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
    if channels == 3 and chan[0] == 3:  # a root without translation channels: the reference reads its first three columns as one anyway
        raise NotImplementedError(f"{filename}: a 3-channel root is not supported")
    if channels not in (3, 6, 9):
        raise NotImplementedError(f"{filename}: {channels}-channel joints are not supported")
    want = {3: 3 + 3 * J, 6: 6 * J, 9: 3 + 9 * (J - 1)}[channels]
    if (channels == 9 and J < 2) or data.shape[1] != want:
        raise ValueError(f"{filename}: expected {want} columns, found {data.shape[1]}")
    return BvhAnim(names, np.asarray(parents, dtype=np.int32), np.asarray(offsets, dtype=np.float64), order, data, channels, frametime)


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


def _foot_mods(bones: List[str]):
    """lafan1.py:36-39: LeftFootMod / RightFootMod = the foot's position with the toe's orientation."""
    names, pos_src, rot_src = [], [], []
    for side in ("Left", "Right"):
        if f"{side}Foot" in bones and f"{side}Toe" in bones:
            names.append(f"{side}FootMod")
            pos_src.append(bones.index(f"{side}Foot"))
            rot_src.append(bones.index(f"{side}Toe"))
    return names, pos_src, rot_src


def _device_fk(a0: BvhAnim, rows: torch.Tensor, dev: torch.device, columns=None):
    """``gmr_bvh_fk_rows`` on motion rows already on the device (all of one skeleton).  ``columns``: names to emit (joints or
    FootMod entries, in this order) or None for everything.  Returns (pos [N,B,3], quat [N,B,4], names)."""
    lib = _native.load()
    N, J = int(rows.shape[0]), len(a0.bones)
    extra_names, extra_pos, extra_rot = _foot_mods(a0.bones)
    E = len(extra_names)
    all_names = list(a0.bones) + extra_names
    if columns is None:
        names, cols, B = all_names, None, J + E
    else:
        names = [str(c) for c in columns]
        try:
            cols = np.asarray([all_names.index(c) for c in names], dtype=np.int32)
        except ValueError as ex:
            raise KeyError(f"{ex.args[0].split(' is not')[0]}: no such bone in the BVH skeleton") from None
        B = len(names)
    pos = torch.empty((N, B, 3), dtype=torch.float64, device=dev)
    quat = torch.empty((N, B, 4), dtype=torch.float64, device=dev)
    if N > 0:
        parents = np.ascontiguousarray(a0.parents, dtype=np.int32)
        order = np.asarray(a0.order, dtype=np.int32)
        ep, erot = np.asarray(extra_pos, dtype=np.int32), np.asarray(extra_rot, dtype=np.int32)
        offs_d = torch.from_numpy(np.ascontiguousarray(a0.offsets, dtype=np.float64)).to(dev)
        vp = C.c_void_p
        rc = lib.gmr_bvh_fk_rows(parents.ctypes.data_as(vp), J, order.ctypes.data_as(vp), ep.ctypes.data_as(vp) if E else None,
                                 erot.ctypes.data_as(vp) if E else None, E, int(a0.channels), vp(offs_d.data_ptr()), vp(rows.data_ptr()),
                                 int(rows.shape[1]), N, 0.01, cols.ctypes.data_as(vp) if cols is not None else None, B,
                                 vp(pos.data_ptr()), vp(quat.data_ptr()), vp(torch.cuda.current_stream(dev).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"gmr_bvh_fk_rows failed with status {rc} ({J} joints, {E} extra, {a0.channels}-channel rows)")
    return pos, quat, names


def _height_columns(a0: BvhAnim) -> List[str]:
    """The entries the height estimate of lafan1.py:45-69 looks at (all of them when there is no Head)."""
    extra_names, _, _ = _foot_mods(a0.bones)
    if "Head" not in a0.bones:
        return list(a0.bones) + extra_names
    return ["Head"] + [k for k in ("LeftFootMod", "RightFootMod", "LeftFoot", "RightFoot") if k in a0.bones or k in extra_names]


def _clip_heights(a0: BvhAnim, rows: torch.Tensor, offs: np.ndarray, dev: torch.device, full=None) -> List[float]:
    """lafan1.py:45-69 per clip, from each clip's own last frame (the few last rows go through the FK kernel once more when the
    batch was emitted with a column selection)."""
    lens = np.diff(offs)
    if int(lens.sum()) == 0:
        return [1.75] * len(lens)
    last_rows = torch.from_numpy(offs[1:][lens > 0] - 1).to(dev)
    if full is not None:
        pos, names = full
        last = pos[last_rows].cpu().numpy()
    else:
        names = _height_columns(a0)
        last, _, _ = _device_fk(a0, rows[last_rows].contiguous(), dev, names)
        last = last.cpu().numpy()
    heights, k = [], 0
    for n in lens:
        if n == 0:
            heights.append(1.75)
            continue
        heights.append(_estimate_height({nm: last[k, i] for i, nm in enumerate(names)}))
        k += 1
    return heights


def load_lafan1_file(bvh_file: str, device: int = 0, columns=None) -> BvhClip:
    """BVH file -> global poses (metres, Z-up, wxyz) on ``cuda:device`` + the reference's height estimate.
    ``columns``: emit only these entries (e.g. the bones an IK config consumes), in this order."""
    anim = read_bvh(bvh_file)
    dev = torch.device("cuda", device)
    rows = torch.from_numpy(np.ascontiguousarray(anim.rows)).to(dev)
    pos, quat, names = _device_fk(anim, rows, dev, columns)
    offs = np.array([0, len(anim)], dtype=np.int64)
    height = _clip_heights(anim, rows, offs, dev, full=(pos, names) if columns is None else None)[0]
    return BvhClip(pos, quat, names, height, anim.frametime)


class BvhBatch:
    """Several BVH clips on the GPU as one batch: ``pos [N, B, 3]``, ``quat [N, B, 4]`` (concatenated clips), ``seq_offsets``,
    one height estimate per clip -- the arguments ``retarget_batch(..., seq_offsets=..., human_heights=...)`` takes."""

    def __init__(self, pos, quat, names, seq_offsets, heights, frametimes, files):
        self.pos, self.quat, self.body_names = pos, quat, names
        self.seq_offsets, self.human_heights, self.frametimes, self.files = seq_offsets, heights, frametimes, files

    def __len__(self):
        return len(self.files)


def _check_one_skeleton(files, anims):
    a0 = anims[0]
    for f, a in zip(files, anims):
        if a.bones != a0.bones or not np.array_equal(a.parents, a0.parents) or a.order != a0.order or a.channels != a0.channels \
                or not np.array_equal(a.offsets, a0.offsets):
            raise ValueError(f"{f}: skeleton differs from {files[0]} (one batch = one skeleton)")


def load_lafan1_files(bvh_files, device: int = 0, threads: int = 8, columns=None) -> BvhBatch:
    """A folder's worth of BVH files -> one GPU batch (the file loop of scripts/bvh_to_robot_dataset.py:59-80, where every file
    is parsed with regexes and turned into per-frame dicts one after the other).  The text of the files is parsed on ``threads``
    host threads (both native parsers release the GIL) straight into one pinned row array, and ONE ``gmr_bvh_fk_rows`` launch does
    row slicing, degrees -> radians, Euler -> quaternion, the skeleton FK, Y-up -> Z-up, cm -> m and the FootMod synthesis for all
    of them.  All files must share one skeleton (names, parents, offsets, Euler order, channel layout), as a dataset does; the
    height estimate of every clip (lafan1.py:45-69, from its own last frame) comes back with the batch."""
    from concurrent.futures import ThreadPoolExecutor
    files = [str(f) for f in bvh_files]
    if not files:
        raise ValueError("no files")
    with ThreadPoolExecutor(max_workers=max(1, min(threads, len(files)))) as ex:
        anims = list(ex.map(read_bvh, files))
    a0 = anims[0]
    _check_one_skeleton(files, anims)
    dev = torch.device("cuda", device)
    lens = np.array([len(a) for a in anims], dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    N, ncol = int(offs[-1]), int(a0.rows.shape[1])
    rows_h = torch.empty((N, ncol), dtype=torch.float64, pin_memory=True)
    for a, o in zip(anims, offs[:-1]):
        rows_h[o:o + len(a)] = torch.from_numpy(a.rows)
    rows = rows_h.to(dev, non_blocking=True)
    pos, quat, names = _device_fk(a0, rows, dev, columns)
    heights = _clip_heights(a0, rows, offs, dev, full=(pos, names) if columns is None else None)
    return BvhBatch(pos, quat, names, offs, heights, [a.frametime for a in anims], files)
