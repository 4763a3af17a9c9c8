"""SMPL-X key-point adapter on the fast path.

``get_smplx_data_offline_fast`` of the reference (general_motion_retargeting/utils/smpl.py:109-198) minus the Python:
the SMPL-X body model itself (licensed assets, smpl.py:12-34) stays with the caller and hands over
``global_orient [T,3]``, ``full_pose [T,J*3]`` (axis-angle), ``joints [T,>=J,3]`` and ``parents [J]``; this module
aligns them to the target frame rate (slerp per joint, lerp per coordinate) and chains the orientations down the
kinematic tree on the GPU (``gmr_smplx_keypoints``), returning the ``[T', J, 3]`` / ``[T', J, 4]`` tensors
``retarget_batch`` consumes, with the SMPL-X joint names as columns.

The reference module cannot be imported here (it needs the ``smplx`` package), so this row is "parity unpinned";
tests pin it against a scipy restatement of the cited lines.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _native

# smplx.joint_names.JOINT_NAMES[:55] (body, jaw, eyes, hands) -- the names the smplx_to_*.json configs refer to
SMPLX_JOINT_NAMES: List[str] = [
    "pelvis", "left_hip", "right_hip", "spine1", "left_knee", "right_knee", "spine2", "left_ankle", "right_ankle", "spine3",
    "left_foot", "right_foot", "neck", "left_collar", "right_collar", "head", "left_shoulder", "right_shoulder", "left_elbow",
    "right_elbow", "left_wrist", "right_wrist", "jaw", "left_eye_smplhf", "right_eye_smplhf",
    "left_index1", "left_index2", "left_index3", "left_middle1", "left_middle2", "left_middle3", "left_pinky1", "left_pinky2",
    "left_pinky3", "left_ring1", "left_ring2", "left_ring3", "left_thumb1", "left_thumb2", "left_thumb3",
    "right_index1", "right_index2", "right_index3", "right_middle1", "right_middle2", "right_middle3", "right_pinky1", "right_pinky2",
    "right_pinky3", "right_ring1", "right_ring2", "right_ring3", "right_thumb1", "right_thumb2", "right_thumb3",
]
# kinematic tree of the SMPL-X model (body_model.parents)
SMPLX_PARENTS: List[int] = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 15, 15, 15,
                            20, 25, 26, 20, 28, 29, 20, 31, 32, 20, 34, 35, 20, 37, 38,
                            21, 40, 41, 21, 43, 44, 21, 46, 47, 21, 49, 50, 21, 52, 53]


def get_smplx_data_offline_fast(global_orient, full_pose, joints, parents: Sequence[int] = SMPLX_PARENTS, src_fps: float = 30.0,
                                tgt_fps: float = 30.0, joint_names: Sequence[str] = SMPLX_JOINT_NAMES,
                                device: int = 0, columns: Optional[Sequence[str]] = None, out=None) -> Tuple[torch.Tensor, torch.Tensor, List[str], float]:
    """-> (pos [T',J,3], quat [T',J,4] wxyz, joint names, aligned_fps); float64 CUDA tensors.
    ``columns``: emit only these joints, in this order (e.g. ``ik_columns(config)``: the 14 an smplx_to_*.json config reads) --
    their ancestors are chained inside the kernel, the rest of the 55 is neither read nor written.
    ``out``: (pos, quat) contiguous float64 tensors of exactly the result's shapes to write into (rows of a batch)."""
    lib = _native.load()
    dev = torch.device("cuda", device)
    # float32 arrays (a body model's output) go to the kernel as they are; anything else is made float64
    f32 = all((a.dtype == torch.float32) if isinstance(a, torch.Tensor) else (np.asarray(a).dtype == np.float32) for a in (global_orient, full_pose, joints))
    dt = torch.float32 if f32 else torch.float64
    as_t = lambda a: (a if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a))).detach().to(dev, dt)
    parents = np.ascontiguousarray(parents, dtype=np.int32)
    J = len(parents)
    go = as_t(global_orient).reshape(-1, 3).contiguous()
    T = int(go.shape[0])
    fp = as_t(full_pose).reshape(T, -1, 3)
    if fp.shape[1] < J:
        raise ValueError("full_pose has fewer joints than parents")
    fp = fp[:, :J].contiguous()
    jt = as_t(joints).reshape(T, -1, 3).contiguous()
    if jt.shape[1] < J or len(joint_names) < J:
        raise ValueError("joints / joint_names shorter than parents")
    frame_skip = int(src_fps / tgt_fps)  # smpl.py:119
    if tgt_fps < src_fps:
        T_out = T // frame_skip          # :127
        resample = 1
        aligned_fps = T_out / T * src_fps if T else tgt_fps  # :172
    else:
        T_out, resample, aligned_fps = T, 0, tgt_fps
    names = list(joint_names[:J])
    cols = None
    if columns is not None:
        sel = [str(c) for c in columns]
        missing = [c for c in sel if c not in names]
        if missing:
            raise KeyError(missing[0])
        cols = np.asarray([names.index(c) for c in sel], dtype=np.int32)
        names = sel
    B = len(names)
    if out is None:
        pos = torch.empty((T_out, B, 3), dtype=torch.float64, device=dev)
        quat = torch.empty((T_out, B, 4), dtype=torch.float64, device=dev)
    else:
        pos, quat = out
        for t, k in ((pos, 3), (quat, 4)):
            if tuple(t.shape) != (T_out, B, k) or t.dtype != torch.float64 or t.device != dev or not t.is_contiguous():
                raise ValueError(f"out tensors must be contiguous float64 [{T_out}, {B}, 3 / 4] on {dev}")
    vp = C.c_void_p
    if T_out > 0:
        rc = lib.gmr_smplx_keypoints_in(parents.ctypes.data_as(vp), J, int(jt.shape[1]), vp(go.data_ptr()), vp(fp.data_ptr()), vp(jt.data_ptr()),
                                        _native.GMR_DTYPE_F32 if f32 else _native.GMR_DTYPE_F64, T, T_out, resample,
                                        cols.ctypes.data_as(vp) if cols is not None else None, B,
                                        vp(pos.data_ptr()), vp(quat.data_ptr()), vp(torch.cuda.current_stream(dev).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"gmr_smplx_keypoints_in failed with status {rc}")
    return pos, quat, names, float(aligned_fps)


# ---------------------------------------------------------------------------------------------------------------- joint-array files
# The file side of this row.  scripts/smplx_to_robot_dataset.py:63-87 goes AMASS .npz -> smplx body model (licensed assets, stays with
# the caller) -> get_smplx_data_offline_fast -> retarget, per file.  A caller that owns the body model dumps its outputs ONCE
# (`save_joint_file`) and the whole rest of the script -- frame-rate alignment, orientation chaining, IK, FK,
# post-processing, pickles -- runs here over folders of such files, batched like the BVH folder path (gmr_amd.bvh.iter_lafan1_batches).
JOINT_FILE_KEYS = ("joints", "global_orient", "full_pose", "mocap_frame_rate", "betas")


def human_height_from_betas(betas) -> float:
    """load_smplx_file's height estimate (utils/smpl.py:37-40): 1.66 + 0.1 * betas[0] (betas [16] or [1, 16])."""
    b = np.asarray(betas, dtype=np.float64)
    return float(1.66 + 0.1 * (b[0] if b.ndim == 1 else b[0, 0]))


def save_joint_file(path, joints, global_orient, full_pose, mocap_frame_rate, betas, n_joints: int = len(SMPLX_PARENTS)) -> None:
    """One clip's body-model outputs as an uncompressed .npz: ``joints [T, >= n_joints, 3]`` (only the first ``n_joints`` are kept: the
    model emits 127, the adapter reads 55), ``global_orient [T, 3]``, ``full_pose [T, >= 3 n_joints]`` (axis-angle), the AMASS file's
    ``mocap_frame_rate`` and ``betas``.  Arrays keep their dtype (the model's float32 halves the file)."""
    j = np.asarray(joints)
    T = j.shape[0]
    fp = np.asarray(full_pose).reshape(T, -1)
    np.savez(path, joints=np.ascontiguousarray(j.reshape(T, -1, 3)[:, :n_joints]), global_orient=np.asarray(global_orient).reshape(T, 3),
             full_pose=np.ascontiguousarray(fp[:, :3 * n_joints]), mocap_frame_rate=np.asarray(mocap_frame_rate), betas=np.asarray(betas))


class SmplxBatch:
    """Several SMPL-X clips on the GPU as one batch: ``pos [N, B, 3]``, ``quat [N, B, 4]`` at the target frame rate (concatenated
    clips), ``seq_offsets``, ``human_heights`` (one per clip, from its betas) and ``fps`` (each clip's aligned frame rate, what the
    reference stores in the pickle) -- the arguments of ``retarget_batch`` / ``dataset.retarget_clips``.  ``skipped``: (file, reason)."""

    def __init__(self, pos, quat, names, seq_offsets, heights, fps, files, skipped=None):
        self.pos, self.quat, self.body_names = pos, quat, names
        self.seq_offsets, self.human_heights, self.fps, self.files = seq_offsets, heights, fps, files
        self.skipped = skipped or []

    def __len__(self):
        return len(self.files)


def _zip_directory(view: memoryview, path: str):
    """{name: (dtype, shape, byte offset of the data in `view`)} of an UNCOMPRESSED .npz (np.savez): the end-of-central-directory
    record, the central directory and the .npy headers are parsed in place; the arrays themselves are not touched.  (np.load walks every
    byte through zipfile's CRC loop under the GIL: 0.3 GB/s, slower with threads; a plain read of the same file runs at 5 GB/s.)"""
    import struct
    n = len(view)
    tail = bytes(view[max(0, n - 65557):])
    k = tail.rfind(b"PK\x05\x06")
    if k < 0:
        raise ValueError(f"{path}: not a zip archive")
    total, cd_size, cd_off = struct.unpack_from("<HII", tail, k + 10)
    if cd_off == 0xFFFFFFFF or total == 0xFFFF:
        raise ValueError(f"{path}: zip64 directory")
    out, p = {}, cd_off
    for _ in range(total):
        sig, method, csize, usize, nlen, xlen, clen, lho = struct.unpack_from("<4s6xH8xIIHHH8xI", view, p)
        if sig != b"PK\x01\x02":
            raise ValueError(f"{path}: bad central directory")
        name = bytes(view[p + 46:p + 46 + nlen]).decode()
        p += 46 + nlen + xlen + clen
        if method != 0:
            raise ValueError(f"{path}: member {name} is compressed (write joint files with np.savez / save_joint_file)")
        if usize == 0xFFFFFFFF or lho == 0xFFFFFFFF:
            raise ValueError(f"{path}: zip64 member")
        sig2, nlen2, xlen2 = struct.unpack_from("<4s22xHH", view, lho)
        if sig2 != b"PK\x03\x04":
            raise ValueError(f"{path}: bad local header")
        d0 = lho + 30 + nlen2 + xlen2
        head = bytes(view[d0:d0 + min(usize, 4096)])
        if head[:6] != b"\x93NUMPY":
            raise ValueError(f"{path}: member {name} is not a .npy array")
        major = head[6]
        hlen, hoff = (struct.unpack_from("<H", head, 8)[0], 10) if major == 1 else (struct.unpack_from("<I", head, 8)[0], 12)
        import ast
        d = ast.literal_eval(head[hoff:hoff + hlen].decode("latin1"))
        dt = np.dtype(d["descr"])
        if d["fortran_order"] or dt.hasobject:
            raise ValueError(f"{path}: member {name} is not a plain C-ordered array")
        out[name[:-4] if name.endswith(".npy") else name] = (dt, tuple(d["shape"]), d0 + hoff + hlen)
    return out


class _JointFiles:
    """A batch of joint files on the host: their bytes in one page-locked array and, per good file, where its arrays lie."""

    def __init__(self, files, buf, total, metas, skipped):
        self.files, self.buf, self.total, self.metas, self.skipped = files, buf, total, metas, skipped


def _read_joint_files(files, threads: int, n_joints: int, skip_errors: bool, slot: int = 0) -> _JointFiles:
    """Read the files into one pinned byte array (``readinto``, GIL released) and locate their arrays, on ``threads`` host threads."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    from .bvh import _pinned_bytes
    skipped, ok = [], []
    for f in files:
        try:
            ok.append((f, os.path.getsize(f)))
        except OSError as ex:
            if not skip_errors:
                raise
            skipped.append((f, str(ex)))
    sizes = np.array([n for _, n in ok], dtype=np.int64)
    starts = np.concatenate([[0], np.cumsum((sizes + 255) // 256 * 256)]).astype(np.int64)
    buf = _pinned_bytes(int(starts[-1]) + 256, 2 + slot)  # (slots 0 / 1 belong to the BVH reader)
    host = buf.numpy()

    def one(k):
        f, n = ok[k]
        a = int(starts[k])
        view = memoryview(host[a:a + n])
        with open(f, "rb", buffering=0) as fh:
            got = 0
            while got < n:
                r = fh.readinto(view[got:])
                if not r:
                    raise ValueError(f"{f}: file shrank while it was read")
                got += r
        mem = _zip_directory(view, f)
        missing = [key for key in JOINT_FILE_KEYS if key not in mem]
        if missing:
            raise ValueError(f"{f}: no '{missing[0]}' array (a joint file holds {', '.join(JOINT_FILE_KEYS)})")
        (jd, js, jo), (gd, gs, go_), (fd, fs, fo) = mem["joints"], mem["global_orient"], mem["full_pose"]
        T = js[0] if len(js) == 3 else -1
        if len(js) != 3 or js[1] < n_joints or js[2] != 3 or gs != (T, 3) or len(fs) != 2 or fs[0] != T or fs[1] < 3 * n_joints:
            raise ValueError(f"{f}: array shapes do not describe {n_joints} joints over {T} frames")
        for dt in (jd, gd, fd):
            if dt not in (np.dtype("<f4"), np.dtype("<f8")):
                raise ValueError(f"{f}: arrays must be float32 or float64")
        small = {}
        for key in ("mocap_frame_rate", "betas"):
            dt, shp, off = mem[key]
            cnt = int(np.prod(shp)) if shp else 1
            small[key] = np.frombuffer(view, dtype=dt, count=cnt, offset=off).reshape(shp).copy()
        fps = float(np.asarray(small["mocap_frame_rate"]).reshape(-1)[0])
        if not (fps > 0):
            raise ValueError(f"{f}: mocap_frame_rate must be positive")
        for (dt, shp, off) in (mem["joints"], mem["global_orient"], mem["full_pose"]):
            if off + int(np.prod(shp)) * dt.itemsize > n:
                raise ValueError(f"{f}: truncated")
        return {"T": T, "fps": fps, "height": human_height_from_betas(small["betas"]),
                "arrays": {key: (mem[key][0], mem[key][1], a + mem[key][2]) for key in ("joints", "global_orient", "full_pose")}}

    def guarded(k):
        try:
            return one(k)
        except Exception as ex:  # a broken file must not take the folder down (the reference prints and continues, :63-69)
            if not skip_errors:
                raise
            return ex
    with ThreadPoolExecutor(max_workers=max(1, min(threads, max(1, len(ok))))) as ex:
        metas = list(ex.map(guarded, range(len(ok))))
    good = [k for k, m in enumerate(metas) if not isinstance(m, Exception)]
    skipped += [(ok[k][0], repr(m)) for k, m in enumerate(metas) if isinstance(m, Exception)]
    return _JointFiles([ok[k][0] for k in good], buf, int(starts[-1]), [metas[k] for k in good], skipped)


def _joint_batch(jf: _JointFiles, dev, tgt_fps, columns, parents, joint_names) -> SmplxBatch:
    J = len(parents)
    names = list(joint_names[:J]) if columns is None else [str(c) for c in columns]
    t_out = [m["T"] // int(m["fps"] / tgt_fps) if tgt_fps < m["fps"] else m["T"] for m in jf.metas]  # smpl.py:119,127
    offs = np.concatenate([[0], np.cumsum(t_out)]).astype(np.int64)
    B = len(names)
    pos = torch.empty((int(offs[-1]), B, 3), dtype=torch.float64, device=dev)
    quat = torch.empty((int(offs[-1]), B, 4), dtype=torch.float64, device=dev)
    raw = jf.buf[: jf.total].to(dev, non_blocking=True)  # ONE copy of the files as they are; the kernel reads float32 or float64

    def arr(meta):
        dt, shp, off = meta
        nbytes = int(np.prod(shp)) * dt.itemsize
        t = raw[off:off + nbytes]
        if off % dt.itemsize:
            t = t.clone()  # (np.savez happens to align its members; a foreign writer may not)
        return t.view(torch.float32 if dt.itemsize == 4 else torch.float64).reshape(shp)
    out_fps = []
    for k, m in enumerate(jf.metas):
        a = m["arrays"]
        _, _, _, afps = get_smplx_data_offline_fast(arr(a["global_orient"]), arr(a["full_pose"]), arr(a["joints"]), parents, src_fps=m["fps"], tgt_fps=tgt_fps,
                                                    joint_names=joint_names, device=dev.index or 0, columns=columns, out=(pos[offs[k]:offs[k + 1]], quat[offs[k]:offs[k + 1]]))
        out_fps.append(afps)
    torch.cuda.current_stream(dev).synchronize()  # the pinned buffer is reused by the batch after next
    return SmplxBatch(pos, quat, names, offs, [m["height"] for m in jf.metas], out_fps, list(jf.files), list(jf.skipped))


def load_joint_files(files, device: int = 0, tgt_fps: float = 30.0, columns: Optional[Sequence[str]] = None, threads: int = 8, skip_errors: bool = False,
                     parents: Sequence[int] = SMPLX_PARENTS, joint_names: Sequence[str] = SMPLX_JOINT_NAMES) -> SmplxBatch:
    """A folder's worth of joint-array files (``save_joint_file``) -> one ``SmplxBatch`` on the GPU: files read on ``threads`` host
    threads, every clip aligned to ``tgt_fps`` and chained by the adapter kernel straight into its rows of the batch tensors."""
    files = [str(f) for f in files]
    dev = torch.device("cuda", device)
    return _joint_batch(_read_joint_files(files, threads, len(parents), skip_errors), dev, tgt_fps, columns, parents, joint_names)


def iter_joint_batches(files, batch_files: int = 256, device: int = 0, tgt_fps: float = 30.0, columns: Optional[Sequence[str]] = None, threads: int = 8,
                       skip_errors: bool = False, parents: Sequence[int] = SMPLX_PARENTS, joint_names: Sequence[str] = SMPLX_JOINT_NAMES):
    """The folder in batches of ``batch_files`` files, read ahead: while the caller solves and writes batch k, a background thread reads
    batch k + 1's files.  A batch without a good file is yielded empty (``len(batch) == 0``) with its ``skipped`` list."""
    from concurrent.futures import ThreadPoolExecutor
    files = [str(f) for f in files]
    groups = [files[i:i + batch_files] for i in range(0, len(files), max(1, batch_files))]
    if not groups:
        return
    dev = torch.device("cuda", device)
    with ThreadPoolExecutor(max_workers=1) as bg:
        nxt = bg.submit(_read_joint_files, groups[0], threads, len(parents), skip_errors, 0)
        for g in range(len(groups)):
            loaded = nxt.result()
            if g + 1 < len(groups):
                nxt = bg.submit(_read_joint_files, groups[g + 1], threads, len(parents), skip_errors, (g + 1) & 1)
            yield _joint_batch(loaded, dev, tgt_fps, columns, parents, joint_names)
