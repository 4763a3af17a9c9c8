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


def gather_rows(local_rows: torch.Tensor, lengths: Sequence[int]) -> torch.Tensor:
    """All-gather per-rank rows (concatenated clips, in the order of ``my_clips``) into global clip order.

    ``local_rows`` is ``[sum(lengths[i] for i in my_clips), D]``; returns ``[sum(lengths), D]`` on every rank.
    """
    lengths = [int(x) for x in lengths]
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local_rows
    world = dist.get_world_size()
    parts = partition_clips(lengths, world)
    counts = [sum(lengths[i] for i in p) for p in parts]
    if local_rows.shape[0] != counts[dist.get_rank()]:
        raise ValueError("local_rows does not match this rank's clips")
    dev = _comm_device()
    D = local_rows.shape[1]
    pad = max(counts)
    send = torch.zeros((pad, D), dtype=local_rows.dtype, device=dev)
    send[: local_rows.shape[0]] = local_rows.to(dev)
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send)
    offs = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    out = torch.empty((int(offs[-1]), D), dtype=local_rows.dtype, device=dev)
    for r, p in enumerate(parts):
        cur = 0
        for i in p:
            out[offs[i]:offs[i + 1]] = recv[r][cur:cur + lengths[i]]
            cur += lengths[i]
    return out.to(local_rows.device)
