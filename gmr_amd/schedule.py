"""Work-item scheduling: clips -> runs of consecutive frames, one wavefront each.

The reference is sequential in time inside a clip (``self.configuration`` persists across
``retarget()`` calls, motion_retarget.py:75,150,185) and independent across clips (a fresh
``GeneralMotionRetargeting`` per file, scripts/smplx_to_robot_dataset.py:79).  So:

* ``chunk == 0`` -- one item per clip: exactly the reference's semantics.
* ``chunk > 0``  -- a clip is cut into runs of ``chunk`` output frames; every run except a clip's
  first starts ``burn_in`` frames early from ``qpos0`` and discards those frames.  This trades
  redundant work for parallelism when there are fewer clips than wavefronts (a single 3k-frame
  clip); it is an approximation whose residual against the sequential run is measured in tests
  and reported by bench.py, never assumed.

Multi-GPU: clips are independent, so ranks take disjoint sets of clips (longest-first greedy
bin packing by frame count); no data-path collective is needed.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np

from ._native import WORK_ITEM_DTYPE


def make_items(seq_offsets: Sequence[int], chunk: int = 0, burn_in: int = 0, track: bool = False) -> np.ndarray:
    offs = np.asarray(seq_offsets, dtype=np.int64)
    if offs.ndim != 1 or offs.size < 1 or np.any(np.diff(offs) < 0):
        raise ValueError("seq_offsets must be a non-decreasing 1-D array")
    rows = []
    for s in range(offs.size - 1):
        a, b = int(offs[s]), int(offs[s + 1])
        if b == a:
            continue
        if chunk <= 0:
            rows.append((a, 0, b - a, -1, -1, -1, 0))
            continue
        for start in range(a, b, chunk):
            n_out = min(chunk, b - start)
            burn = min(burn_in, start - a)
            rows.append((start - burn, burn, n_out, -1, -1, -1, 0))
    items = np.array(rows, dtype=WORK_ITEM_DTYPE) if rows else np.zeros(0, dtype=WORK_ITEM_DTYPE)
    if track:  # final state of item i -> row i, state after burn-in -> row n + i (see verified_chunked_solve)
        n = len(items)
        items["final_row"] = np.arange(n)
        items["burn_row"] = n + np.arange(n)
    return items


def partition_clips(lengths: Sequence[int], world_size: int) -> List[List[int]]:
    """Greedy longest-first assignment of clips to ranks; returns clip indices per rank (each sorted)."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    load = [0] * world_size
    parts: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        parts[r].append(i)
        load[r] += int(lengths[i])
    return [sorted(p) for p in parts]
