"""Dataset path: clips in -> the reference's motion dicts / pickles out, post-processing on the GPU.

Reproduces what ``process_file`` does after the retarget loop (reference
scripts/smplx_to_robot_dataset.py:93-146; the BVH variant scripts/bvh_to_robot_dataset.py:106-151 is the same
with both adjustments off):

* ``root_rot`` wxyz -> xyzw (:101-102), ``dof_pos = qpos[:, 7:]`` (:103)
* ``local_body_pos``: FK with zero root position and identity root rotation, float32 (:106-112)
* HEIGHT_ADJUST: FK with the real root, clip-global ``min z`` over all bodies, ``root_pos.z -= min`` (:118-126)
* ROOT_ORIGIN_OFFSET: subtract the first frame's root xy (:128-131)
* schema ``{fps, root_pos, root_rot, dof_pos, local_body_pos, link_body_list}`` (:134-141), read back by
  general_motion_retargeting/data_loader.py:4-18 and validated by scripts/smoke_test.py:19-72.

All clips of a batch go through two FK launches (``gmr_fk``, ``gmr_fk_min_height``); nothing is looped per clip
on the device.
"""
from __future__ import annotations

import os
import pickle
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .motion_retarget import GeneralMotionRetargeting


def motions_from_qpos(gmr: GeneralMotionRetargeting, qpos: torch.Tensor, seq_offsets: Sequence[int], fps,
                      height_adjust: bool = True, root_origin_offset: bool = True, ground_offset: float = 0.0) -> List[Dict]:
    """qpos ``[N, nq]`` float64 on the GPU (concatenated clips) -> one motion dict per clip.

    The arrays of the returned dicts are row slices of four batch-sized PAGE-LOCKED host arrays (no per-clip copy): any surviving
    motion dict keeps its whole batch pinned -- gigabytes for a dataset-sized batch.  Write the clips (``MotionWriter``) and drop
    them, or ``copy()`` the arrays of a clip that has to outlive its batch."""
    if gmr.model.planar_base:
        # the dataset scripts read a free-joint root out of qpos (root_pos = qpos[:3], root_rot = qpos[3:7],
        # scripts/smplx_to_robot_dataset.py:97-103); the reference has no such path for galaxea_r1pro either
        raise NotImplementedError("the dataset post-processing assumes a free-joint root; use retarget_batch for a planar-base robot")
    eng = gmr._engine
    offs = np.asarray(seq_offsets, dtype=np.int64)
    N = int(qpos.shape[0])
    if offs[0] != 0 or offs[-1] != N:
        raise ValueError("seq_offsets must span [0, N]")
    fps_list = list(fps) if isinstance(fps, (list, tuple, np.ndarray)) else [fps] * (len(offs) - 1)
    root_pos = qpos[:, 0:3].clone()
    root_rot = qpos[:, [4, 5, 6, 3]].contiguous()          # wxyz -> xyzw
    dof_pos = qpos[:, 7:].contiguous()
    dof32 = dof_pos.to(torch.float32)
    zeros = torch.zeros((N, 3), dtype=torch.float32, device=qpos.device)
    ident = torch.zeros((N, 4), dtype=torch.float32, device=qpos.device)
    ident[:, 3] = 1.0
    local_body_pos, _ = eng.fk(zeros, ident, dof32, want_rot=False)
    if height_adjust and N > 0:
        lowest = eng.fk_min_height(root_pos.to(torch.float32), root_rot.to(torch.float32), dof32, offs).to(torch.float64)
        lens = torch.from_numpy(np.diff(offs)).to(qpos.device)
        root_pos[:, 2] = root_pos[:, 2] - torch.repeat_interleave(lowest, lens) + ground_offset
    if root_origin_offset and N > 0:
        nonempty = np.diff(offs) > 0
        first = torch.zeros((len(offs) - 1, 2), dtype=torch.float64, device=qpos.device)
        first[torch.from_numpy(nonempty).to(qpos.device)] = root_pos[torch.from_numpy(offs[:-1][nonempty]).to(qpos.device), :2]
        lens = torch.from_numpy(np.diff(offs)).to(qpos.device)
        root_pos[:, :2] = root_pos[:, :2] - torch.repeat_interleave(first, lens, dim=0)
    # to the host through pinned arrays, all four copies in flight together: a fresh pageable array costs more in first-touch
    # page faults than the copy itself, and torch's host allocator hands the pinned blocks back to the next batch once these
    # arrays are dropped (pickled and released)
    host = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in (root_pos, root_rot, dof_pos, local_body_pos)]
    for h, t in zip(host, (root_pos, root_rot, dof_pos, local_body_pos)):
        h.copy_(t, non_blocking=True)
    torch.cuda.current_stream(qpos.device).synchronize()
    rp, rr, dp, lb = (h.numpy() for h in host)
    names = list(gmr.model.body_names)
    out = []
    for s in range(len(offs) - 1):
        a, b = int(offs[s]), int(offs[s + 1])
        # row slices of the batch arrays (C-contiguous views): no second host copy; pickling a slice stores only the slice
        out.append({"fps": fps_list[s], "root_pos": rp[a:b], "root_rot": rr[a:b], "dof_pos": dp[a:b],
                    "local_body_pos": lb[a:b], "link_body_list": names})
    return out


def retarget_clips(gmr: GeneralMotionRetargeting, pos, quat, body_names: Sequence[str], seq_offsets: Sequence[int], fps=30,
                   height_adjust: bool = True, root_origin_offset: bool = True, chunk: int = 0, burn_in: int = 0,
                   human_heights: Optional[Sequence[float]] = None, clip_start: str = "qpos0") -> List[Dict]:
    """The whole ``process_file`` compute path for a batch of clips: batched IK, FK, post-processing.  ``human_heights``:
    one ``actual_human_height`` per clip (the per-file ``GMR(..., actual_human_height=...)`` of
    scripts/smplx_to_robot_dataset.py:79-83).  ``clip_start``: ``retarget_batch``'s (``"root_target"`` is the opt-in departure
    from the reference that spares wound-up clips their slow start)."""
    tpos = torch.from_numpy(np.ascontiguousarray(pos)) if isinstance(pos, np.ndarray) else pos
    tquat = torch.from_numpy(np.ascontiguousarray(quat)) if isinstance(quat, np.ndarray) else quat
    qpos = gmr.retarget_batch(tpos.to(gmr.device), tquat.to(gmr.device), body_names, seq_offsets=seq_offsets, chunk=chunk, burn_in=burn_in,
                              human_heights=human_heights, clip_start=clip_start)  # (raises on non-finite qpos / a capped QP)
    return motions_from_qpos(gmr, qpos, seq_offsets, fps, height_adjust=height_adjust, root_origin_offset=root_origin_offset)


# ------------------------------------------------------------------ writing motion files (row f-3)
class _RawBytes:
    """The data of a C-contiguous array standing in for the ``bytes`` object numpy's reduce would copy it into."""
    __slots__ = ("view",)

    def __init__(self, arr: np.ndarray):
        self.view = memoryview(arr).cast("B")

    def __len__(self):
        return self.view.nbytes


class _MotionPickler(pickle._Pickler):
    """``pickle.dump(motion, f)`` byte for byte (protocol 4, the default the reference's ``pickle.dump`` uses,
    scripts/smplx_to_robot_dataset.py:143-146), without the two things that make it slow for motion dicts: ``ndarray.__reduce__``
    first copies every array into a ``bytes`` object (under the GIL), and only then is that copy written.  Here an array's reduce
    tuple is rebuilt around a view of its own memory and the payload goes from that memory straight into ``file.write`` (which
    releases the GIL), so a pool of threads writes clips in parallel.  ``fast_pickle_ok()`` checks the byte identity once per
    process against the stock pickler; where it does not hold the stock pickler is used."""

    def reducer_override(self, obj):
        if type(obj) is np.ndarray and obj.flags.c_contiguous and obj.dtype.kind in "fiub" and obj.nbytes >= 1 << 16:
            fn, args, state = np.empty((0,) * obj.ndim, dtype=obj.dtype).__reduce__()
            return fn, args, (state[0], obj.shape, obj.dtype, False, _RawBytes(obj))
        return NotImplemented

    def _save_raw(self, obj):  # pickle._Pickler.save_bytes for a payload that is a view, not a bytes object
        n = len(obj)
        if n > 0xFFFFFFFF:
            self._write_large_bytes(pickle.BINBYTES8 + n.to_bytes(8, "little"), obj.view)
        else:
            self._write_large_bytes(pickle.BINBYTES + n.to_bytes(4, "little"), obj.view)
        self.memoize(obj)

    dispatch = dict(pickle._Pickler.dispatch)
    dispatch[_RawBytes] = _save_raw


_FAST_PICKLE_OK: Optional[bool] = None


def fast_pickle_ok() -> bool:
    """True when ``_MotionPickler`` reproduces the stock ``pickle.dump`` byte for byte on a motion-shaped dict here (checked once:
    it rests on numpy's reduce format and the pickler's framing rules, both of which a new version could change)."""
    global _FAST_PICKLE_OK
    if _FAST_PICKLE_OK is None:
        import io
        rng = np.random.default_rng(0)
        probe = {"fps": 30, "root_pos": rng.random((4000, 3)), "root_rot": rng.random((4000, 4)), "dof_pos": rng.random((4000, 29)),
                 "local_body_pos": rng.random((4000, 38, 3)).astype(np.float32), "link_body_list": ["a", "b"], "small": rng.random((5, 3))}
        f = io.BytesIO()
        try:
            _MotionPickler(f, pickle.DEFAULT_PROTOCOL).dump(probe)
            _FAST_PICKLE_OK = pickle.DEFAULT_PROTOCOL >= 4 and f.getvalue() == pickle.dumps(probe)
        except Exception:
            _FAST_PICKLE_OK = False
    return _FAST_PICKLE_OK


def save_motion(path: str, motion: Dict, override: bool = False) -> bool:
    """Pickle one motion dict; like the scripts, skip files that already exist unless ``override`` (:219).  The file is
    what ``pickle.dump(motion, f)`` writes, byte for byte (see ``_MotionPickler``)."""
    if os.path.exists(path) and not override:
        return False
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    if str(path).endswith(".pt"):  # the torch twin of the schema (scripts/convert_motion_pkl_to_pt.py:40-48): arrays as tensors
        torch.save({k: torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v for k, v in motion.items()}, path)
        return True
    if fast_pickle_ok():
        _write_all(path, motion_stream(motion))
    else:
        with open(path, "wb") as f:
            pickle.dump(motion, f)
    return True


class _Pieces:
    """File object of ``_MotionPickler`` that keeps what it is given: small pieces as bytes, array payloads as views."""

    def __init__(self):
        self.pieces = []

    def write(self, b):
        self.pieces.append(b if isinstance(b, memoryview) and b.nbytes >= 1 << 16 else bytes(b))
        return len(b)


_STREAM_TEMPLATES: Dict = {}
_SMALL_MIN = 256  # arrays below this many bytes are part of a layout's key, like the scalars


class _Template:
    """The pickle of one motion layout with its numbers taken out: ``pieces`` are the stream's buffers in order -- bytes (opcodes,
    frame headers, strings, ...), or the KEY of a large array whose memory goes there as it is; ``patches`` say where, inside the
    bytes pieces, the raw data of the layout's small arrays lie (piece, offset, size, key).  Built from the first clip of a layout,
    compared once against the full pickler run of the second clip (``verified``), used from the third on."""
    __slots__ = ("pieces", "patches", "verified")

    def __init__(self, pieces, patches):
        self.pieces, self.patches, self.verified = pieces, patches, False

    def fill(self, motion: Dict) -> list:
        out = list(self.pieces)
        touched = {}
        for pi, off, n, k in self.patches:
            b = touched.get(pi)
            if b is None:
                b = touched[pi] = bytearray(out[pi])
                out[pi] = b
            b[off:off + n] = memoryview(motion[k]).cast("B")
        return [memoryview(motion[t]).cast("B") if isinstance(t, str) else t for t in out]


def _is_plain(v) -> bool:
    return type(v) is np.ndarray and v.flags.c_contiguous and v.dtype.kind in "fiub"


def _full_stream(motion: Dict):
    out = _Pieces()
    _MotionPickler(out, pickle.DEFAULT_PROTOCOL).dump(motion)
    return out.pieces


def _build_template(motion: Dict, big, small) -> Optional[_Template]:
    """Pickle a SHADOW of the clip -- same layout, the small arrays filled with random bytes -- and find those bytes in the stream: a
    clip's own numbers can be degenerate (a block of zeros also matches the zero bytes of the length field in front of it)."""
    rng = np.random.default_rng(0x5EED)
    shadow = dict(motion)
    marks = []
    for k, v in small:
        m = np.frombuffer(rng.bytes(v.nbytes), dtype=v.dtype).reshape(v.shape).copy()
        shadow[k] = m
        marks.append((k, m))
    pieces = _full_stream(shadow)
    small = marks
    order = {id(v): k for k, v in big}
    tp = []
    for p in pieces:
        if isinstance(p, memoryview) and id(p.obj) in order:
            tp.append(order[id(p.obj)])
        else:
            tp.append(bytes(p))
    if sum(isinstance(t, str) for t in tp) != len(big):
        return None
    patches, pi, pos = [], 0, 0
    for k, v in small:  # in dict order = stream order: search on from where the previous one ended
        raw = memoryview(v).cast("B").tobytes()
        while pi < len(tp):
            at = tp[pi].find(raw, pos) if isinstance(tp[pi], bytes) else -1
            if at >= 0:
                patches.append((pi, at, len(raw), k))
                pos = at + len(raw)
                break
            pi, pos = pi + 1, 0
        else:
            return None
    return _Template(tp, patches)


def motion_stream(motion: Dict) -> list:
    """The pickle of a motion dict as a list of buffers (headers as bytes, array payloads as views of the arrays' own memory).
    Clips of one batch differ in their numbers only, so the stream is kept per LAYOUT -- keys, array shapes and dtypes, and the pickle
    of everything that is not an array -- with the numbers taken out (``_Template``): a clip of a known layout costs a dict lookup and
    a copy of its small arrays (those below the pickler's 64 KB frame size, which travel inside frames), not a run of the pure-Python
    pickler (~0.5 ms under the GIL: every clip shorter than 2 731 frames has such arrays).  A layout's template is checked against the
    full pickler on the second clip that uses it before it is trusted."""
    big = [(k, v) for k, v in motion.items() if _is_plain(v) and v.nbytes >= 1 << 16]
    small = [(k, v) for k, v in motion.items() if _is_plain(v) and _SMALL_MIN <= v.nbytes < 1 << 16]
    if sum(v.nbytes for _, v in big) + sum(v.nbytes for _, v in small) < 1 << 15 or any(_is_plain(v) and v.nbytes < _SMALL_MIN for v in motion.values()):
        return [pickle.dumps(motion, pickle.DEFAULT_PROTOCOL)]  # a handful of frames: the stock pickler's copy costs nothing
    arrays = {k for k, _ in big} | {k for k, _ in small}
    try:
        rest = pickle.dumps([v for k, v in motion.items() if k not in arrays], pickle.DEFAULT_PROTOCOL)
    except Exception:
        return _full_stream(motion)
    key = (tuple(motion.keys()), tuple((k, v.shape, v.dtype.str) for k, v in big), tuple((k, v.shape, v.dtype.str) for k, v in small), rest)
    tmpl = _STREAM_TEMPLATES.get(key)
    if tmpl is False:  # a layout the template cannot express
        return _full_stream(motion)
    if tmpl is not None and tmpl.verified:
        return tmpl.fill(motion)
    pieces = _full_stream(motion)
    if tmpl is None:
        if len(_STREAM_TEMPLATES) >= 256:
            _STREAM_TEMPLATES.clear()
        _STREAM_TEMPLATES[key] = _build_template(motion, big, small) or False
    else:  # second clip of the layout: the template must reproduce the pickler's stream exactly
        a = b"".join(bytes(memoryview(x).cast("B")) for x in tmpl.fill(motion))
        b = b"".join(bytes(memoryview(x).cast("B")) for x in pieces)
        if a == b:
            tmpl.verified = True
        else:
            _STREAM_TEMPLATES[key] = False
    return pieces


def _write_all(path: str, pieces: list) -> None:
    """One gather-write of the whole file: a single system call, which is also the single stretch a writer thread spends without
    the GIL (threads that re-take the GIL between several writes of one file queue up behind each other's Python code)."""
    fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o666)
    try:
        views = [memoryview(p).cast("B") for p in pieces]
        while views:
            n = os.writev(fd, views[:1024])
            while n > 0 and views:  # (a short write: drop what went out, go on with the rest)
                if n >= views[0].nbytes:
                    n -= views[0].nbytes
                    views.pop(0)
                else:
                    views[0] = views[0][n:]
                    n = 0
            while views and views[0].nbytes == 0:
                views.pop(0)
    finally:
        os.close(fd)


class MotionWriter:
    """Writes motion files on a pool of threads while the caller goes on to the next batch (the reference writes each clip from
    the process that solved it, scripts/smplx_to_robot_dataset.py:134-146, 241-242: one pickle per clip, skipped when it exists).

        with MotionWriter(workers=8) as w:
            for batch in batches:                        # batch k + 1 is solved while batch k is being written
                motions = retarget_clips(gmr, ...)
                w.submit(motions, paths)
        w.written, w.skipped

    ``submit`` returns at once while at most ``max_pending`` earlier batches are unwritten, else it waits for the oldest of them:
    the motion dicts (row slices of the batch's pinned result arrays, ``motions_from_qpos``) stay alive until their files are closed,
    so a disk slower than the GPU must hold the producer back instead of piling up page-locked batches.  An error in a worker is
    raised by the next ``submit`` / ``close``."""

    def __init__(self, workers: int = 8, override: bool = False, max_pending: int = 3):
        from concurrent.futures import ThreadPoolExecutor
        self._pool = ThreadPoolExecutor(max_workers=max(1, int(workers)))
        self._override = override
        self._max_pending = max(1, int(max_pending))
        self._batches = []  # one list of futures per submitted batch, oldest first
        self._futures = []
        self.written = 0
        self.skipped = 0
        fast_pickle_ok()  # (decide once, before the threads start)

    def _reap(self, wait: bool):
        keep = []
        for f in self._futures:
            if wait or f.done():
                if f.result():
                    self.written += 1
                else:
                    self.skipped += 1
            else:
                keep.append(f)
        self._futures = keep

    def submit(self, motions: Sequence[Dict], paths: Sequence[str]) -> None:
        if len(motions) != len(paths):
            raise ValueError("one path per motion")
        self._reap(False)
        self._batches = [b for b in self._batches if not all(f.done() for f in b)]
        while len(self._batches) >= self._max_pending:   # back-pressure: wait for the oldest unwritten batch
            for f in self._batches.pop(0):
                f.exception()  # (waits; the error itself is raised by _reap below)
            self._reap(False)
        fs = [self._pool.submit(save_motion, p, m, self._override) for m, p in zip(motions, paths)]
        self._futures += fs
        self._batches.append(fs)

    def close(self) -> None:
        try:
            self._reap(True)
        finally:
            self._pool.shutdown(wait=True)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


def save_motions(motions: Sequence[Dict], paths: Sequence[str], workers: int = 8, override: bool = False) -> int:
    """Write a batch of motion files (``.pkl`` or ``.pt`` by extension) on ``workers`` threads; returns how many were written
    (existing files are skipped unless ``override``).  Files are byte-identical to those of a serial ``save_motion`` loop."""
    with MotionWriter(workers, override) as w:
        w.submit(motions, paths)
    return w.written


def load_robot_motion(motion_file: str):
    """Reader with the contract of general_motion_retargeting/data_loader.py:4-18 (root_rot returned as wxyz)."""
    if str(motion_file).endswith(".pt"):  # (:50-58 of the converter: tensors back to arrays; files this package wrote)
        d = {k: v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v for k, v in torch.load(motion_file, map_location="cpu", weights_only=True).items()}
    else:
        with open(motion_file, "rb") as f:
            d = pickle.load(f)
    root_rot = d["root_rot"][:, [3, 0, 1, 2]]
    return d, d["fps"], d["root_pos"], root_rot, d["dof_pos"], d["local_body_pos"], d["link_body_list"]


def validate_motion(motion: Dict, nq: Optional[int] = None) -> None:
    """The structural checks of scripts/smoke_test.py:19-72."""
    for k in ("fps", "root_pos", "root_rot", "dof_pos"):
        if k not in motion:
            raise KeyError(k)
    T = motion["root_pos"].shape[0]
    if motion["root_pos"].shape != (T, 3) or motion["root_rot"].shape != (T, 4) or motion["dof_pos"].shape[0] != T:
        raise ValueError("bad motion shapes")
    if nq is not None and motion["dof_pos"].shape[1] != nq - 7:
        raise ValueError("dof count does not match the model")
    n = np.linalg.norm(motion["root_rot"], axis=1)
    if T and (n.min() < 0.5 or n.max() > 1.5):
        raise ValueError("root_rot is not a quaternion track")
