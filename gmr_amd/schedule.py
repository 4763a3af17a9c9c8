"""Work-item scheduling: clips -> runs of consecutive frames, one wavefront each.

The reference is sequential in time inside a clip (``self.configuration`` persists across
``retarget()`` calls, motion_retarget.py:75,150,185) and independent across clips (a fresh
``GeneralMotionRetargeting`` per file, scripts/smplx_to_robot_dataset.py:79).  So:

* ``chunk == 0`` -- one item per clip: exactly the reference's semantics.
* ``chunk > 0``  -- a clip is cut into runs of ``chunk`` output frames; every run except a clip's
  first starts ``burn_in`` frames early from a *speculative* state and discards those frames:
  ``qpos0`` with the floating base placed on the root task's target of its first frame
  (``INIT_ROOT_TARGET``, gmr_blob.h; measured on 9000-frame clips: 0-0.5 % of the chunks end their
  burn-in in another IK basin than the sequential run, against 1-26 % when started from ``qpos0`` at
  the world origin).  This trades redundant work for parallelism when there are fewer clips than
  wavefronts; alone it is an approximation -- ``plan_walks`` adds the verification pass that makes the
  result the sequential one (Engine.ik_solve_chunked).

Multi-GPU: clips are independent, so ranks take disjoint sets of clips (longest-first greedy
bin packing by frame count); no data-path collective is needed.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np

from ._native import INIT_QPOS0, INIT_ROOT_TARGET, WORK_ITEM_DTYPE


def make_items(seq_offsets: Sequence[int], chunk: int = 0, burn_in: int = 0, track: bool = False,
               height_scales: Optional[Sequence[float]] = None, chunk_init: int = INIT_ROOT_TARGET, clip_init: int = INIT_QPOS0) -> np.ndarray:
    """Work items of a batch of clips.  ``height_scales[s]`` is clip s's factor on the human scale table (its own
    ``actual_human_height`` over the height the model was compiled with, motion_retarget.py:36-43); ``chunk_init`` is the
    start state of every chunk but a clip's first.  ``clip_init`` is the start state of the clip itself: ``INIT_QPOS0`` is the
    reference (a fresh ``GeneralMotionRetargeting`` per file); ``INIT_ROOT_TARGET`` is an opt-in departure from it -- the robot is
    put where the human is before the first frame, which spares a clip that starts facing away from ``qpos0`` the reference's slow,
    sometimes never-ending, start-up."""
    offs = np.asarray(seq_offsets, dtype=np.int64)
    if offs.ndim != 1 or offs.size < 1 or np.any(np.diff(offs) < 0):
        raise ValueError("seq_offsets must be a non-decreasing 1-D array")
    hs = np.ones(offs.size - 1) if height_scales is None else np.asarray(height_scales, dtype=np.float64)
    if hs.shape != (offs.size - 1,) or not np.all(np.isfinite(hs)) or np.any(hs <= 0):
        raise ValueError("height_scales must hold one positive factor per clip")
    a, b = offs[:-1], offs[1:]
    keep = b > a  # empty clips have no item
    a, b, hk = a[keep], b[keep], hs[keep]
    if chunk <= 0:
        items = np.zeros(a.size, dtype=WORK_ITEM_DTYPE)
        items["frame_begin"], items["n_out"], items["init_row"] = a, b - a, clip_init
    else:
        nch = (b - a + chunk - 1) // chunk                       # chunks per clip
        clip = np.repeat(np.arange(a.size), nch)                 # clip of every chunk
        k = np.arange(int(nch.sum())) - np.repeat(np.cumsum(nch) - nch, nch)  # index of the chunk inside its clip
        start = a[clip] + k * chunk
        burn = np.minimum(burn_in, start - a[clip])
        items = np.zeros(start.size, dtype=WORK_ITEM_DTYPE)
        items["frame_begin"], items["n_burn"], items["n_out"] = start - burn, burn, np.minimum(chunk, b[clip] - start)
        items["init_row"] = np.where(k == 0, clip_init, chunk_init)
        hk = hk[clip]
    items["final_row"], items["burn_row"], items["check_stride"], items["height_scale"] = -1, -1, 0, hk
    if track:  # final state of item i -> row i, state after burn-in -> row n + i (see plan_walks)
        n = len(items)
        items["final_row"] = np.arange(n)
        items["burn_row"] = n + np.arange(n)
    return items


def auto_chunk(seq_offsets: Sequence[int], slots: int = 2048):
    """(chunk, burn_in) for ``make_items`` / ``Engine.ik_solve_chunked`` when the caller does not choose: measured on one MI355X
    (tools/experiments/chunk_sweep.py, ``profiles/r03_chunk_sweep.txt``).  Two regimes.  Few frames (fewer chunks than wavefront
    slots): the launch lasts as long as ONE chunk with its burn-in, the walk as long as the chunk boundaries of the longest clip
    take one after the other -- short chunks for short clips, ~sqrt(longest clip) / 2.  Many frames: chunks queue for slots, the launch
    lasts as long as the work, and every burn-in frame is redundant work -- long chunks, about two per slot.  A burn-in of 24 frames
    verified as often as 32 on every set and costs a quarter less.  (0, 0) = do not chunk: more than one clip per four slots."""
    offs = np.asarray(seq_offsets, dtype=np.int64)
    lens = np.diff(offs)
    if lens.size == 0 or int(lens.max(initial=0)) == 0:
        return 16, 24
    if 4 * lens.size > slots:
        # More than a clip per four wavefront slots: whole clips.  Cutting can gain at most slots / clips (the idle part of the chip) and
        # only if the chunks verify; clips that do not (wound-up starts: a few per cent of distinct clips, whatever the burn-in) are re-solved by their walk one after the
        # other, and then the burn-in frames and the second launch are pure loss -- measured on distinct 3000-frame clips, half of them
        # noisy / over-reaching: 256 / 1024 / 2048 / 4096 / 8192 clips whole 234 / 275 / 282 / 413 / 611 ms, in chunks 238-253 / 291-317 /
        # 375-397 / 536-568 / 725-880 ms (tools/experiments/auto_chunk_threshold.py, profiles/r03_unshaped_breakdown.md).  Below the
        # threshold the possible gain (4x and more: a LAFAN1-sized set of 77 long clips runs 16x faster in chunks) outweighs that loss.
        return 0, 0
    n, longest = int(lens.sum()), int(lens.max())
    c_latency = 8 * int(round(np.sqrt(longest) / 16.0))
    c_fill = 8 * int(round(n / (2.0 * slots) / 8.0))
    return int(min(128, max(16, c_latency, c_fill))), 24


def plan_walks(items: np.ndarray, seq_offsets: Sequence[int], chunk: int) -> np.ndarray:
    """Verification walks for the tracked chunk items of ``make_items(..., chunk, track=True)``: one item per clip that has
    more than one chunk, running from the clip's second chunk to its end with ``check_stride = chunk`` (gmr_blob.h): it starts
    from the first chunk's final state (exact: that chunk started from ``qpos0`` like the reference) and at every boundary
    either adopts the chunk solved speculatively or re-solves it from the true state."""
    offs = np.asarray(seq_offsets, dtype=np.int64)
    n = len(items)
    out_begin = items["frame_begin"] + items["n_burn"]
    first = np.nonzero(np.isin(out_begin, offs[:-1]) & (items["n_burn"] == 0))[0]  # first chunk of every (non-empty) clip
    last = np.append(first[1:], n)                                                   # one past its last chunk
    multi = last - first > 1
    walks = np.zeros(int(multi.sum()), dtype=WORK_ITEM_DTYPE)
    c0 = first[multi]
    clip_end = offs[np.searchsorted(offs, out_begin[c0], side="right")]
    walks["frame_begin"], walks["n_out"] = out_begin[c0 + 1], clip_end - out_begin[c0 + 1]
    walks["init_row"], walks["final_row"], walks["burn_row"] = c0, c0 + 1, n + c0 + 1
    walks["check_stride"], walks["height_scale"] = chunk, items["height_scale"][c0]
    return walks


def partition_clips(lengths: Sequence[int], world_size: int) -> List[List[int]]:
    """Greedy longest-first assignment of clips to ranks; returns clip indices per rank (each sorted).

    Ties go to the lowest clip index and the lowest rank, so equal-length clips are dealt round-robin (clip i -> rank
    i mod world) -- that case is answered without the greedy loop; the general one runs on a heap (O(n log world))."""
    import heapq
    ln = np.asarray(lengths, dtype=np.int64).reshape(-1)
    n = int(ln.size)
    if n == 0:
        return [[] for _ in range(world_size)]
    if world_size == 1:
        return [list(range(n))]
    if np.all(ln == ln[0]) and ln[0] > 0:
        return [list(range(r, n, world_size)) for r in range(world_size)]
    order = np.lexsort((np.arange(n), -ln))
    heap = [(0, r) for r in range(world_size)]
    owner = np.empty(n, dtype=np.int64)
    lo = ln.tolist()
    for i in order.tolist():
        load, r = heapq.heappop(heap)
        owner[i] = r
        heapq.heappush(heap, (load + lo[i], r))
    return [np.nonzero(owner == r)[0].tolist() for r in range(world_size)]
