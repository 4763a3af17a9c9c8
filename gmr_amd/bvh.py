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
import os
from typing import Dict, List, Optional, Tuple

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


def _buf(raw):
    """(pointer argument, length) of a bytes object or a uint8 numpy array."""
    if isinstance(raw, np.ndarray):
        return C.c_void_p(raw.ctypes.data), int(raw.nbytes)
    return raw, len(raw)


def _parse_motion(block, fnum: int, max_cols: int, filename: str) -> np.ndarray:
    """The first ``fnum`` non-empty lines of the motion block as a float64 ``[fnum, columns]`` array (native parser).
    ``max_cols`` bounds the row length (the hierarchy fixes it); longer rows are reported as malformed."""
    lib = _native.load()
    cap = fnum * max_cols + 1
    out = np.empty(cap, dtype=np.float64)
    n_lines, n_cols = C.c_int64(0), C.c_int64(0)
    ptr, nbytes = _buf(block)
    n = lib.gmr_bvh_parse_motion(ptr, nbytes, fnum, out.ctypes.data, cap, C.byref(n_lines), C.byref(n_cols))
    if n < 0:
        raise ValueError(f"{filename}: malformed motion block (bad number or ragged rows)")
    if n_lines.value < fnum:
        raise ValueError(f"{filename}: {n_lines.value} motion rows, header says {fnum}")
    return out[:n].reshape(fnum, n_cols.value)


def _parse_header(raw, filename: str):
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
        ptr, nbytes = _buf(raw)
        n = lib.gmr_bvh_parse_header(ptr, nbytes, max_j, names_buf, len(names_buf), parents.ctypes.data_as(vp), offsets.ctypes.data_as(vp),
                                     channels.ctypes.data_as(vp), order.ctypes.data_as(vp), C.byref(fnum), C.byref(ftime), C.byref(moff))
        if n == -2 and max_j < (1 << 16):
            max_j *= 4
            continue
        if n <= 0:
            raise ValueError(f"{filename}: not a BVH file this loader understands")
        names = names_buf.raw.split(b"\0")[:n]
        return ([x.decode("ascii") for x in names], parents[:n].copy(), offsets[:n].copy(), channels[:n].copy(),
                tuple(int(x) for x in order), int(fnum.value), float(ftime.value), int(moff.value))


def _layout(names, chan, filename: str):
    """Row layout of a parsed hierarchy: (channels, numbers per row); rejects what the loader cannot lay out."""
    channels = int(chan[-1])  # the reference shapes the motion rows by the LAST joint's channel count (extract.py:104-106)
    if (channels == 3 and (chan[0] not in (3, 6) or np.any(chan[1:] != 3))) or (channels == 6 and np.any(chan != 6)) \
            or (channels == 9 and (chan[0] != 3 or np.any(chan[1:] != 9))):
        raise NotImplementedError(f"{filename}: joints with mixed channel counts are not supported")
    J = len(names)
    if channels == 3 and chan[0] == 3:  # a root without translation channels: the reference reads its first three columns as one anyway
        raise NotImplementedError(f"{filename}: a 3-channel root is not supported")
    if channels not in (3, 6, 9):
        raise NotImplementedError(f"{filename}: {channels}-channel joints are not supported")
    if channels == 9 and J < 2:
        raise ValueError(f"{filename}: a 9-channel file needs a non-root joint")
    return channels, {3: 3 + 3 * J, 6: 6 * J, 9: 3 + 9 * (J - 1)}[channels]


def read_bvh(filename: str) -> BvhAnim:
    with open(filename, "rb") as f:
        raw = f.read()
    names, parents, offsets, chan, order, fnum, frametime, moff = _parse_header(raw, filename)
    channels, want = _layout(names, chan, filename)
    J = len(names)
    data = _parse_motion(raw[moff:], fnum, 9 * J + 3, filename)  # the ctypes call releases the GIL: files parse in parallel threads
    if data.shape[1] != want:
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
    one height estimate per clip -- the arguments ``retarget_batch(..., seq_offsets=..., human_heights=...)`` takes.  ``files`` are the
    clips' files in batch order; ``skipped`` lists (file, reason) of files left out (``skip_errors=True``)."""

    def __init__(self, pos, quat, names, seq_offsets, heights, frametimes, files, skipped=None):
        self.pos, self.quat, self.body_names = pos, quat, names
        self.seq_offsets, self.human_heights, self.frametimes, self.files = seq_offsets, heights, frametimes, files
        self.skipped = skipped or []

    def __len__(self):
        return len(self.files)


def _check_one_skeleton(files, anims):
    a0 = anims[0]
    for f, a in zip(files, anims):
        if a.bones != a0.bones or not np.array_equal(a.parents, a0.parents) or a.order != a0.order or a.channels != a0.channels \
                or not np.array_equal(a.offsets, a0.offsets):
            raise ValueError(f"{f}: skeleton differs from {files[0]} (one batch = one skeleton)")


class _FileText:
    """What the host keeps of a batch of files whose MOTION blocks are parsed on the device: the files' bytes in ONE page-locked
    array (read straight into it) and each file's parsed header."""

    def __init__(self, files, buf, starts, sizes, heads, total, skipped=None):
        self.files, self.buf, self.starts, self.sizes, self.heads, self.total = files, buf, starts, sizes, heads, total   # starts: per file
        self.skipped: List[Tuple[str, str]] = skipped or []   # (file, reason) of files left out (skip_errors)

    def anim0(self) -> BvhAnim:
        names, parents, offsets, chan, order, fnum, frametime, moff = self.heads[0]
        channels, ncol = _layout(names, chan, self.files[0])
        return BvhAnim(names, np.asarray(parents, dtype=np.int32), np.asarray(offsets, dtype=np.float64), order, np.zeros((0, ncol)), channels, frametime)


_PINNED_TEXT: Dict[int, torch.Tensor] = {}


def _pinned_bytes(n: int, slot: int) -> torch.Tensor:
    """A grow-only page-locked byte buffer per slot (two slots alternate when batches are read ahead): page-locking is what a
    fresh pinned allocation costs, so it is paid once per process, not once per batch."""
    t = _PINNED_TEXT.get(slot)
    if t is None or t.numel() < n:
        t = torch.empty(max(n, 1 << 20) * 5 // 4, dtype=torch.uint8, pin_memory=True)
        _PINNED_TEXT[slot] = t
    return t


def _read_files(files: List[str], threads: int, slot: int = 0, skip_errors: bool = False) -> _FileText:
    """Read the files into one pinned byte array (``readinto``: no intermediate bytes objects) and parse their HIERARCHY sections,
    on ``threads`` host threads (file reads and the native header parser release the GIL).  ``skip_errors``: a file that cannot be
    read or whose header / layout the loader does not understand is left out and reported in ``.skipped`` -- the per-file
    ``try / except: print; continue`` of scripts/bvh_to_robot_dataset.py:75-80 -- instead of failing the batch."""
    from concurrent.futures import ThreadPoolExecutor
    skipped = []
    if skip_errors:
        ok = []
        for f in files:
            try:
                os.path.getsize(f)
                ok.append(f)
            except OSError as ex:
                skipped.append((f, str(ex)))
        files = ok
    sizes = np.array([os.path.getsize(f) for f in files], dtype=np.int64)
    starts = np.concatenate([[0], np.cumsum((sizes + 63) // 64 * 64)]).astype(np.int64)  # every file on a 64-byte boundary
    buf = _pinned_bytes(int(starts[-1]) + 64, slot)
    host = buf.numpy()

    def one(k):
        a, n = int(starts[k]), int(sizes[k])
        view = host[a:a + n]
        with open(files[k], "rb", buffering=0) as f:
            got = 0
            while got < n:
                r = f.readinto(memoryview(view)[got:])
                if not r:
                    raise ValueError(f"{files[k]}: file shrank while it was read")
                got += r
        head = _parse_header(view, files[k])
        _layout(head[0], head[3], files[k])
        return head

    def guarded(k):
        try:
            return one(k)
        except (ValueError, NotImplementedError, OSError) as ex:
            if not skip_errors:
                raise
            return ex

    with ThreadPoolExecutor(max_workers=max(1, min(threads, max(1, len(files))))) as ex:
        heads = list(ex.map(guarded, range(len(files))))
    if skip_errors and any(isinstance(h, Exception) for h in heads):
        # the first readable file sets the batch's skeleton; files of another skeleton are left out like broken ones
        keep = [k for k, h in enumerate(heads) if not isinstance(h, Exception)]
        skipped += [(files[k], str(h)) for k, h in enumerate(heads) if isinstance(h, Exception)]
        return _FileText([files[k] for k in keep], buf, starts[:-1][keep], sizes[keep], [heads[k] for k in keep], int(starts[-1]), skipped)
    return _FileText(files, buf, starts[:-1], sizes, heads, int(starts[-1]), skipped)


def _rows_on_device(ft: _FileText, dev: torch.device, stats: Optional[dict] = None, skip_errors: bool = False):
    """The batch's MOTION blocks -> ``rows [N, ncol]`` float64 on the device (``gmr_bvh_parse_motion_device``): one H2D copy of the
    files as they are, three launches.  Tokens off the exact fast path are parsed by the host parser and patched in; a file whose
    structure the device rejects goes through the host parser whole (which raises what it always raised -- or, with ``skip_errors``,
    has the file left out and reported).  Returns (a0, rows, offs, files, frame times, skipped)."""
    lib = _native.load()
    skipped = list(ft.skipped)
    if not ft.files:
        raise ValueError("no readable BVH file in the batch: " + "; ".join(f"{f}: {r}" for f, r in skipped))
    a0 = ft.anim0()
    ncol = int(a0.rows.shape[1])
    keep = []
    for k, (f, h) in enumerate(zip(ft.files, ft.heads)):
        names, parents, offsets, chan, order, fnum, frametime, moff = h
        ch, nc = _layout(names, chan, f)
        if names != a0.bones or not np.array_equal(parents, a0.parents) or tuple(order) != tuple(a0.order) or ch != a0.channels \
                or not np.array_equal(np.asarray(offsets, dtype=np.float64), a0.offsets):
            if not skip_errors:
                raise ValueError(f"{f}: skeleton differs from {ft.files[0]} (one batch = one skeleton)")
            skipped.append((f, f"skeleton differs from {ft.files[0]}"))
        else:
            keep.append(k)
    if len(keep) != len(ft.files):
        ft = _FileText([ft.files[k] for k in keep], ft.buf, ft.starts[keep], ft.sizes[keep], [ft.heads[k] for k in keep], ft.total)
    files, heads = ft.files, ft.heads
    lens = np.array([h[5] for h in heads], dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    N, nf = int(offs[-1]), len(files)
    total = int(ft.total)
    text = torch.empty(max(total, 1), dtype=torch.uint8, device=dev)
    text[:total].copy_(ft.buf[:total], non_blocking=True)
    rows = torch.empty((N, ncol), dtype=torch.float64, device=dev)
    seg_b = (ft.starts + np.array([h[7] for h in heads], dtype=np.int64)).astype(np.int64)
    seg_e = (ft.starts + ft.sizes).astype(np.int64)
    status = np.zeros(nf, dtype=np.int32)
    ntok = np.zeros(nf, dtype=np.int64)
    max_slow = 1 << 16
    slow = np.zeros((max_slow, 3), dtype=np.int64)
    n_slow = C.c_int64(0)
    vp = C.c_void_p
    row_b = np.ascontiguousarray(offs[:-1])
    rc = lib.gmr_bvh_parse_motion_device(vp(text.data_ptr()), total, nf, seg_b.ctypes.data_as(vp), seg_e.ctypes.data_as(vp), lens.ctypes.data_as(vp), ncol,
                                         row_b.ctypes.data_as(vp), vp(rows.data_ptr()), status.ctypes.data_as(vp), ntok.ctypes.data_as(vp),
                                         slow.ctypes.data_as(vp), max_slow, C.byref(n_slow), vp(torch.cuda.current_stream(dev).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"gmr_bvh_parse_motion_device failed with status {rc}")
    host = ft.buf.numpy()
    redo = set(int(k) for k in np.nonzero(status)[0])
    ns = int(n_slow.value)
    if ns > max_slow:
        redo |= set(range(nf))
    elif ns > 0:
        # tokens off the fast path: the host parser's slow path (strtod) on each, patched into the rows
        idx, val = [], []
        out1 = np.empty(2, dtype=np.float64)
        nl, nc = C.c_int64(0), C.c_int64(0)
        for k, t, b in slow[:ns]:
            if int(k) in redo:
                continue
            e = int(b)
            end = int(seg_e[k])
            while e < end and host[e] not in (32, 9, 13, 10):
                e += 1
            got = lib.gmr_bvh_parse_motion(vp(host.ctypes.data + int(b)), e - int(b), 1, out1.ctypes.data, 1, C.byref(nl), C.byref(nc))
            if got != 1:
                redo.add(int(k))
                continue
            idx.append(int(offs[k]) * ncol + int(t))
            val.append(float(out1[0]))
        if idx:
            rows.view(-1)[torch.from_numpy(np.asarray(idx, dtype=np.int64)).to(dev)] = torch.from_numpy(np.asarray(val, dtype=np.float64)).to(dev)
    bad = {}
    for k in sorted(redo):  # the host parser decides (and words the error for a malformed file)
        a, n, moff = int(ft.starts[k]), int(ft.sizes[k]), int(heads[k][7])
        try:
            data = _parse_motion(host[a + moff:a + n], int(lens[k]), 9 * len(a0.bones) + 3, files[k])
            if data.shape[1] != ncol:
                raise ValueError(f"{files[k]}: expected {ncol} columns, found {data.shape[1]}")
        except ValueError as ex:
            if not skip_errors:
                raise
            bad[k] = str(ex)
            continue
        rows[int(offs[k]):int(offs[k + 1])] = torch.from_numpy(data).to(dev)
    if stats is not None:
        stats.update({"text_bytes": int(ft.sizes.sum()), "numbers": int(N * ncol), "slow_tokens": ns, "files_reparsed_on_host": len(redo)})
    frametimes = [h[6] for h in heads]
    if bad:  # leave the broken files' rows out (rare: one device copy of the good ones)
        good = [k for k in range(nf) if k not in bad]
        skipped += [(files[k], bad[k]) for k in sorted(bad)]
        rows = torch.cat([rows[int(offs[k]):int(offs[k + 1])] for k in good]) if good else rows[:0]
        offs = np.concatenate([[0], np.cumsum(lens[good])]).astype(np.int64)
        files, frametimes = [files[k] for k in good], [frametimes[k] for k in good]
    return a0, rows, offs, files, frametimes, skipped


def load_lafan1_files(bvh_files, device: int = 0, threads: int = 8, columns=None, parse: str = "device", stats: Optional[dict] = None,
                      skip_errors: bool = False, _slot: int = 0) -> BvhBatch:
    """A folder's worth of BVH files -> one GPU batch (the file loop of scripts/bvh_to_robot_dataset.py:59-80, where every file
    is parsed with regexes and turned into per-frame dicts one after the other).  All files must share one skeleton (names,
    parents, offsets, Euler order, channel layout), as a dataset does; the height estimate of every clip (lafan1.py:45-69, from its
    own last frame) comes back with the batch.

    ``parse="device"`` (default): the files are read into page-locked memory and copied to the GPU as they are; the MOTION blocks are
    parsed there (``gmr_bvh_parse_motion_device``: same numbers as the host parser, bit for bit; what is off its exact path is
    decided by the host parser) and ONE ``gmr_bvh_fk_rows`` launch does row slicing, degrees -> radians, Euler -> quaternion, the
    skeleton FK, Y-up -> Z-up, cm -> m and the FootMod synthesis for all clips -- the host only reads files and HIERARCHY sections.
    ``parse="host"``: the MOTION blocks are parsed by ``gmr_bvh_parse_motion`` on ``threads`` host threads (round 2's path).
    ``skip_errors`` (device path): a file that cannot be read or parsed, or whose skeleton differs from the first good file's, is left
    out and listed in ``batch.skipped`` -- the per-file ``try / except: print; continue`` of scripts/bvh_to_robot_dataset.py:75-80."""
    files = [str(f) for f in bvh_files]
    if not files:
        raise ValueError("no files")
    dev = torch.device("cuda", device)
    skipped = []
    if parse == "device":
        ft = _read_files(files, threads, _slot, skip_errors)
        a0, rows, offs, files, frametimes, skipped = _rows_on_device(ft, dev, stats, skip_errors)
    elif parse == "host":
        if skip_errors:
            raise ValueError("skip_errors needs parse='device'")
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=max(1, min(threads, len(files)))) as ex:
            anims = list(ex.map(read_bvh, files))
        a0 = anims[0]
        _check_one_skeleton(files, anims)
        lens = np.array([len(a) for a in anims], dtype=np.int64)
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        N, ncol = int(offs[-1]), int(a0.rows.shape[1])
        rows_h = torch.empty((N, ncol), dtype=torch.float64, pin_memory=True)
        for a, o in zip(anims, offs[:-1]):
            rows_h[o:o + len(a)] = torch.from_numpy(a.rows)
        rows = rows_h.to(dev, non_blocking=True)
        frametimes = [a.frametime for a in anims]
    else:
        raise ValueError("parse must be 'device' or 'host'")
    pos, quat, names = _device_fk(a0, rows, dev, columns)
    heights = _clip_heights(a0, rows, offs, dev, full=(pos, names) if columns is None else None)
    return BvhBatch(pos, quat, names, offs, heights, frametimes, files, skipped)


def iter_lafan1_batches(bvh_files, batch_files: int = 32, device: int = 0, threads: int = 8, columns=None, skip_errors: bool = False):
    """The folder in batches of ``batch_files`` files, read ahead: while the caller works on batch k (its ``retarget_batch`` call,
    writing the results), a background thread reads batch k + 1's files and parses their headers into the other pinned buffer.
    ``skip_errors``: broken files are left out of their batch and listed in ``batch.skipped``; a batch without a good file is skipped."""
    from concurrent.futures import ThreadPoolExecutor
    files = [str(f) for f in bvh_files]
    groups = [files[i:i + batch_files] for i in range(0, len(files), max(1, batch_files))]
    if not groups:
        return
    dev = torch.device("cuda", device)
    with ThreadPoolExecutor(max_workers=1) as bg:
        nxt = bg.submit(_read_files, groups[0], threads, 0, skip_errors)
        for g in range(len(groups)):
            ft = nxt.result()
            if g + 1 < len(groups):
                nxt = bg.submit(_read_files, groups[g + 1], threads, (g + 1) & 1, skip_errors)
            if skip_errors and not ft.files:
                yield BvhBatch(torch.empty((0, 0, 3), dtype=torch.float64, device=dev), torch.empty((0, 0, 4), dtype=torch.float64, device=dev), [],
                               np.zeros(1, dtype=np.int64), [], [], [], list(ft.skipped))
                continue
            a0, rows, offs, files_g, frametimes, skipped = _rows_on_device(ft, dev, None, skip_errors)
            try:
                pos, quat, names = _device_fk(a0, rows, dev, columns)
            except KeyError as ex:  # the skeleton lacks a bone the caller asked for (the reference fails at its first retarget(): KeyError)
                if not skip_errors:
                    raise
                yield BvhBatch(torch.empty((0, 0, 3), dtype=torch.float64, device=dev), torch.empty((0, 0, 4), dtype=torch.float64, device=dev), [],
                               np.zeros(1, dtype=np.int64), [], [], [], list(skipped) + [(f, str(ex.args[0])) for f in files_g])
                continue
            heights = _clip_heights(a0, rows, offs, dev, full=(pos, names) if columns is None else None)
            yield BvhBatch(pos, quat, names, offs, heights, frametimes, files_g, skipped)
