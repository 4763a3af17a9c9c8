"""One process per GPU over ``torch.distributed`` (backend "nccl" = RCCL on ROCm; "gloo" in CPU tests).

The retarget path shards by clip with no exchange on the data path (clips are independent:
a fresh solver state per file in the reference, scripts/smplx_to_robot_dataset.py:79,241-242).
Two collectives exist around it:

* ``broadcast_blob``  -- rank 0 compiles the model (MJCF + JSON -> blob, a few KB) and broadcasts it,
  so every rank runs the identical packed model.
* ``gather_rows``     -- all-gather of per-rank result rows (qpos: 288 B/frame) back into clip order,
  for callers that want the whole dataset on every rank (or on rank 0).
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist

from .schedule import partition_clips


def init_from_env(backend: Optional[str] = None) -> tuple:
    """Initialise the default process group from RANK/WORLD_SIZE/MASTER_* if WORLD_SIZE > 1. Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def _comm_device() -> torch.device:
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def broadcast_blob(blob: Optional[bytes], src: int = 0) -> bytes:
    """Broadcast the packed model from ``src``; other ranks may pass ``None``."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        assert blob is not None
        return blob
    dev = _comm_device()
    n = torch.tensor([len(blob) if dist.get_rank() == src else 0], dtype=torch.int64, device=dev)
    dist.broadcast(n, src)
    if dist.get_rank() == src:
        buf = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
    else:
        buf = torch.empty(int(n.item()), dtype=torch.uint8, device=dev)
    dist.broadcast(buf, src)
    return bytes(buf.cpu().numpy().tobytes())


def my_clips(lengths: Sequence[int], rank: Optional[int] = None, world: Optional[int] = None) -> List[int]:
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    return partition_clips(lengths, world)[rank]


def plan_gather(lengths: Sequence[int], world: int, itemrow_bytes: int, slab_bytes: int):
    """Host-side plan of ``gather_rows``: clip-level arrays only (nothing per row).

    Returns ``(parts, slabs)``: ``parts[r]`` = clip indices of rank r (``partition_clips``); every slab is a dict with, per rank, the
    local clip range ``clips[r] = (c0, c1)``, the local row range ``rows[r] = (a, b)`` and ``pad`` = the largest row count among the
    ranks (what each rank contributes to the collective, zero-padded).  A slab never receives more than ``slab_bytes`` per rank."""
    ln = np.asarray(lengths, dtype=np.int64).reshape(-1)
    parts = [np.asarray(p, dtype=np.int64) for p in partition_clips(ln, world)]
    cum = [np.concatenate([[0], np.cumsum(ln[p])]) for p in parts]      # local row offset of every local clip
    nloc = max(len(p) for p in parts) if parts else 0
    rows_cap = max(1, int(slab_bytes) // max(1, world * itemrow_bytes))  # rows one rank may contribute to one slab
    slabs, c0 = [], 0
    while c0 < nloc:
        # the largest c1 with max_r rows(c0:c1) <= rows_cap (at least one clip per slab)
        c1 = nloc
        for r in range(world):
            k0 = min(c0, len(parts[r]))
            k1 = int(np.searchsorted(cum[r], cum[r][k0] + rows_cap, side="right")) - 1
            if k1 < len(parts[r]):
                c1 = min(c1, max(k1, c0 + 1))
        clips = [(min(c0, len(parts[r])), min(c1, len(parts[r]))) for r in range(world)]
        rows = [(int(cum[r][a]), int(cum[r][b])) for r, (a, b) in enumerate(clips)]
        slabs.append({"clips": clips, "rows": rows, "pad": max(b - a for a, b in rows)})
        c0 = c1
    return parts, slabs


def gather_rows(local_rows: torch.Tensor, lengths: Sequence[int], out: Optional[torch.Tensor] = None, slab_bytes: int = 8 << 30) -> torch.Tensor:
    """All-gather per-rank rows (concatenated clips, in the order of ``my_clips``) into global clip order.

    ``local_rows`` is ``[sum(lengths[i] for i in my_clips), D]``; returns ``[sum(lengths), D]`` on every rank (``out`` if given).
    The host plans per clip (``plan_gather``); rows are placed by ONE ``index_copy_`` per slab on the communication device -- on the
    clip axis of a ``[clips, T * D]`` view when all clips have the same length T, else on rows through an index expanded on the
    device (``repeat_interleave``).  Slabs bound the receive buffer (``slab_bytes`` per rank per collective), the result is whole."""
    ln = np.asarray(lengths, dtype=np.int64).reshape(-1)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        if out is not None:
            out.copy_(local_rows)
            return out
        return local_rows
    rank, world = dist.get_rank(), dist.get_world_size()
    D = int(local_rows.shape[1])
    parts, slabs = plan_gather(ln, world, D * local_rows.element_size(), slab_bytes)
    if local_rows.shape[0] != int(ln[parts[rank]].sum()):
        raise ValueError("local_rows does not match this rank's clips")
    dev = _comm_device()
    total = int(ln.sum())
    offs = np.concatenate([[0], np.cumsum(ln)]).astype(np.int64)
    res = out if (out is not None and out.device == dev) else torch.empty((total, D), dtype=local_rows.dtype, device=dev)
    if res.shape != (total, D) or not res.is_contiguous():
        raise ValueError("out must be a contiguous [sum(lengths), D] tensor")
    equal = ln.size > 0 and bool(np.all(ln == ln[0])) and int(ln[0]) > 0
    T = int(ln[0]) if equal else 0
    src_all = local_rows if local_rows.device == dev else None
    for sl in slabs:
        pad = sl["pad"]
        if pad == 0:
            continue
        a, b = sl["rows"][rank]
        if src_all is not None and b - a == pad:
            send = src_all[a:b]                                   # contiguous slice of the caller's rows: no staging copy
        else:
            send = torch.zeros((pad, D), dtype=local_rows.dtype, device=dev)
            send[: b - a] = local_rows[a:b].to(dev)
        recv = torch.empty((world * pad, D), dtype=local_rows.dtype, device=dev)
        dist.all_gather_into_tensor(recv, send.contiguous())
        if equal:
            # clip axis: received clip (r, j) -> global clip parts[r][c0 + j]
            k = pad // T
            src_idx = np.concatenate([r * k + np.arange(c1 - c0) for r, (c0, c1) in enumerate(sl["clips"])])
            dst_idx = np.concatenate([parts[r][c0:c1] for r, (c0, c1) in enumerate(sl["clips"])])
            rv = recv.view(world * k, T * D)
            if len(src_idx) != world * k:
                rv = rv.index_select(0, torch.from_numpy(src_idx).to(dev))
            res.view(-1, T * D).index_copy_(0, torch.from_numpy(dst_idx).to(dev), rv)
        else:
            # rows: per received clip its length, first source row and first destination row; expanded on the device
            cl, s0, d0 = [], [], []
            for r, (c0, c1) in enumerate(sl["clips"]):
                L = ln[parts[r][c0:c1]]
                cl.append(L)
                s0.append(r * pad + np.cumsum(L) - L)
                d0.append(offs[parts[r][c0:c1]])
            cl, s0, d0 = (torch.from_numpy(np.concatenate(x)).to(dev) for x in (cl, s0, d0))
            nrow = int(cl.sum().item())
            if nrow == 0:
                continue
            first = torch.cumsum(cl, 0) - cl                       # position of every clip's first row in the expanded list
            within = torch.arange(nrow, dtype=torch.int64, device=dev) - torch.repeat_interleave(first, cl)
            src_rows = torch.repeat_interleave(s0, cl) + within
            dst_rows = torch.repeat_interleave(d0, cl) + within
            res.index_copy_(0, dst_rows, recv.index_select(0, src_rows))
        del recv
    if out is not None and res is not out:
        out.copy_(res)
        return out
    return res if out is not None else res.to(local_rows.device)


# ------------------------------------------------------------------ few, long clips on several GPUs (BASELINE config 3)
def _world():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)


def split_items(items: np.ndarray, world: int) -> List[tuple]:
    """Contiguous ranges [lo, hi) of a chunk-item list, one per rank, balanced by frames processed (burn-in included)."""
    cost = (items["n_burn"] + items["n_out"]).astype(np.int64)
    cum = np.concatenate([[0], np.cumsum(cost)])
    total = int(cum[-1])
    cuts = [int(np.searchsorted(cum, total * r / world, side="left")) for r in range(world)] + [len(items)]
    cuts = np.maximum.accumulate(np.minimum(cuts, len(items)))
    return [(int(cuts[r]), int(cuts[r + 1])) for r in range(world)]


def _allgather_ranges(t: torch.Tensor, ranges: Sequence[tuple]) -> None:
    """In place: rank r holds valid rows ranges[r] = [lo, hi) of the full-size ``t``; afterwards every rank holds all of them."""
    rank, world = _world()
    if world == 1:
        return
    dev = _comm_device()
    pad = max(hi - lo for lo, hi in ranges)
    if pad == 0:
        return
    lo, hi = ranges[rank]
    send = torch.zeros((pad,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
    send[: hi - lo] = t[lo:hi].to(dev)
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send)
    for r, (a, b) in enumerate(ranges):
        if r != rank and b > a:
            t[a:b] = recv[r][: b - a].to(t.device)


def solve_chunked_sharded(solve, n_frames: int, nq: int, seq_offsets, chunk: int, burn_in: int, device, height_scales=None):
    """Verified parallel-in-time solve of a set of long clips with the chunks of EVERY clip spread over all ranks
    (``Engine.ik_solve_chunked`` on one GPU; SURVEY 8(e) "contiguous time shards", BASELINE config 3).

    Every rank holds the whole input (392 B/frame: a LAFAN1-sized set is ~160 MB) and calls this with the same arguments.
    ``solve(items, qpos_init, qpos_final, out, iters, frames_done)`` runs work items on the rank's own device, writing into the
    full-size buffers it is handed (``Engine.ik_solve`` bound to the inputs; the CPU oracle in the gloo test).

    1. chunk items (``schedule.make_items(track=True)``) are split into contiguous ranges balanced by frames (``split_items``);
       each rank solves its range: speculative chunk starts make the chunks independent of each other, so there is no
       exchange inside this phase;
    2. all-gather of what the chunks produced: qpos rows (288 B/frame), solve counts, and the per-chunk B / F states
       (2 x 288 B per chunk) -- the one real exchange step of this path;
    3. the verification walk of a clip runs on the clip's owner (``schedule.partition_clips``); walks only re-solve chunks
       whose speculative start missed, and
    4. those chunks' rows travel in a second, small all-gather (owner -> everyone); the re-solved-chunk mask is an all-reduce.

    Returns (qpos [N, nq], iters [N], info) on every rank, equal to the sequential run to the walk's tolerance.
    """
    from . import _native
    from .schedule import make_items, plan_walks
    rank, world = _world()
    offs = np.asarray(seq_offsets, dtype=np.int64)
    items = make_items(offs, chunk=chunk, burn_in=burn_in, track=True, height_scales=height_scales)
    n = len(items)
    out = torch.full((n_frames, nq), float("nan"), dtype=torch.float64, device=device)
    iters = torch.zeros(n_frames, dtype=torch.int32, device=device)
    qf = torch.zeros((max(2 * n, 1), nq), dtype=torch.float64, device=device)
    info = {"chunks": n, "resolved_frames": 0, "resolved_chunks": 0, "ranks": world}
    if n == 0:
        return out, iters, info
    out_begin = (items["frame_begin"] + items["n_burn"]).astype(np.int64)
    out_end = out_begin + items["n_out"]
    ranges = split_items(items, world)
    lo, hi = ranges[rank]
    if hi > lo:
        solve(items[lo:hi], None, qf, out, iters, None)
    # -- exchange 1: chunk outputs and boundary states
    frame_ranges = [(int(out_begin[a]), int(out_end[b - 1])) if b > a else (0, 0) for a, b in ranges]
    _allgather_ranges(out, frame_ranges)
    iv = iters.view(-1, 1)
    _allgather_ranges(iv, frame_ranges)
    _allgather_ranges(qf, ranges)                                        # F rows: row i of chunk i
    _allgather_ranges(qf, [(n + a, n + b) for a, b in ranges])          # B rows: row n + i
    # -- walks on the clips' owners
    walks = plan_walks(items, offs, chunk)
    if len(walks) == 0:
        return out, iters, info
    first_chunk = walks["init_row"].astype(np.int64)                     # (plan_walks: init_row = the clip's first chunk)
    walk_len = walks["n_out"].astype(np.int64)
    owner = np.zeros(len(walks), dtype=np.int64)
    for r, part in enumerate(partition_clips(list(walk_len), world)):
        owner[part] = r
    mine = np.nonzero(owner == rank)[0]
    resolved = torch.zeros(n, dtype=torch.int32, device=device)
    if len(mine):
        b_before = qf[n:2 * n].clone()
        done = torch.zeros(len(mine), dtype=torch.int32, device=device)
        solve(walks[mine], qf, qf, out, iters, done)
        changed = (qf[n:2 * n] != b_before).any(dim=1)                    # a re-solved chunk got its B row rewritten (gmr_blob.h)
        resolved = changed.to(torch.int32)
        info["resolved_frames"] = int(done.sum().item())
    if world > 1:
        cdev = _comm_device()
        rs = resolved.to(cdev)
        dist.all_reduce(rs, op=dist.ReduceOp.MAX)
        res_all = rs.cpu().numpy().astype(bool)
        tot = torch.tensor([info["resolved_frames"]], dtype=torch.int64, device=cdev)
        dist.all_reduce(tot)
        info["resolved_frames"] = int(tot.item())
        # -- exchange 2: rows of the re-solved chunks, owner -> everyone (chunk -> clip -> owner is known to all ranks)
        chunk_owner = np.full(n, -1, dtype=np.int64)
        nchunks = (walk_len + chunk - 1) // chunk
        for k in range(len(walks)):
            chunk_owner[first_chunk[k] + 1: first_chunk[k] + 1 + nchunks[k]] = owner[k]
        idx = [np.nonzero(res_all & (chunk_owner == r))[0] for r in range(world)]
        frames = [np.concatenate([np.arange(out_begin[c], out_end[c]) for c in ix]) if len(ix) else np.zeros(0, np.int64) for ix in idx]
        pad_f, pad_c = max(len(f) for f in frames), max(len(ix) for ix in idx)
        if pad_c > 0:
            sendq = torch.zeros((pad_f, nq + 1), dtype=torch.float64, device=cdev)
            sends = torch.zeros((pad_c, 2 * nq), dtype=torch.float64, device=cdev)
            fi = torch.from_numpy(frames[rank]).to(device)
            ci = torch.from_numpy(idx[rank]).to(device)
            if len(frames[rank]):
                sendq[: len(fi), :nq] = out[fi].to(cdev)
                sendq[: len(fi), nq] = iters[fi].to(torch.float64).to(cdev)   # (solve counts < 2^31: exact in float64)
                sends[: len(ci), :nq] = qf[ci].to(cdev)
                sends[: len(ci), nq:] = qf[n + ci].to(cdev)
            rq = [torch.empty_like(sendq) for _ in range(world)]
            rsb = [torch.empty_like(sends) for _ in range(world)]
            dist.all_gather(rq, sendq)
            dist.all_gather(rsb, sends)
            for r in range(world):
                if r == rank or len(idx[r]) == 0:
                    continue
                fr = torch.from_numpy(frames[r]).to(device)
                cr = torch.from_numpy(idx[r]).to(device)
                blk = rq[r][: len(fr)].to(device)
                out[fr] = blk[:, :nq]
                iters[fr] = blk[:, nq].to(torch.int32)
                sb = rsb[r][: len(cr)].to(device)
                qf[cr] = sb[:, :nq]
                qf[n + cr] = sb[:, nq:]
        info["resolved_chunks"] = int(res_all.sum())
    else:
        info["resolved_chunks"] = int(resolved.sum().item())
    return out, iters, info
