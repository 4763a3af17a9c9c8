"""Torch-facing wrapper of one native model handle.

PyTorch is plumbing here: device buffers, the current HIP stream, and (elsewhere)
``torch.distributed``.  All arithmetic happens in libgmr_amd.so's kernels.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _native
from ._native import IKParams, IKStats, ModelInfo
from .model import CompiledModel

_ERR = {-1: "invalid argument", -2: "HIP runtime error", -3: "model not supported by the kernels", -4: "model has no IK config"}


class EngineError(RuntimeError):
    pass


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class Engine:
    def __init__(self, cm: CompiledModel, device: int = 0, _borrowed_handle=None):
        if not torch.cuda.is_available():
            raise EngineError("no HIP device visible to torch: the gmr_amd engine has no CPU path")
        self._lib = _native.load()
        self.cm = cm
        self.device_index = int(device)
        self.device = torch.device("cuda", self.device_index)
        self._owns = _borrowed_handle is None
        if _borrowed_handle is not None:  # a member of an EngineGroup: the group owns the handle
            self._h = _borrowed_handle
        else:
            err = C.create_string_buffer(512)
            self._h = self._lib.gmr_model_create(cm.blob, len(cm.blob), self.device_index, err, len(err))
            if not self._h:
                raise EngineError(f"gmr_model_create: {err.value.decode()}")
        info = ModelInfo()
        self._lib.gmr_model_info_get(self._h, C.byref(info))
        self.info = info
        self.nq, self.nv, self.nbody = info.nq, info.nv, info.nbody

    def close(self):
        if getattr(self, "_h", None):
            if self._owns:
                self._lib.gmr_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc != 0:
            msg = self._lib.gmr_last_error(self._h)
            raise EngineError(f"{what}: {_ERR.get(rc, rc)}: {msg.decode() if msg else ''}")

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def session(self, slot_col: np.ndarray, n_cols: int, params: Optional[IKParams] = None, dtype=np.float64) -> "Session":
        return Session(self, slot_col, n_cols, params, dtype)

    # ------------------------------------------------------------------
    def ik_solve(self, pos: torch.Tensor, quat: torch.Tensor, slot_col: np.ndarray, items: np.ndarray,
                 params: Optional[IKParams] = None, qpos_init: Optional[torch.Tensor] = None, n_final: int = 0,
                 want_iters: bool = True, out: Optional[torch.Tensor] = None, qpos_final: Optional[torch.Tensor] = None,
                 iters: Optional[torch.Tensor] = None, frames_done: Optional[torch.Tensor] = None, launch_order="auto",
                 _host_out: bool = False):
        """pos [N,B,3], quat [N,B,4] (float32 or float64 CUDA tensors) -> qpos [N,nq] float64.

        Frames not covered by any item's output range are left as NaN.  Returns (qpos, iters or None, qpos_final or None).
        ``frames_done`` (int32 [n_items] on the device) receives the output frames each item solved (repair runs, gmr_blob.h).
        ``launch_order``: an int32 device tensor from :meth:`plan_order`, ``None`` (array order, longer items first) or ``"auto"``
        -- plan an order when that pays: more plain items than wavefront slots, long enough for the probe to be a small fraction of
        the work (``PROBE_*`` below).  The order only moves work in time; results are identical.
        """
        if pos.device != self.device or quat.device != self.device:
            raise EngineError("inputs must live on the engine's device")
        if pos.dtype != quat.dtype or pos.dtype not in (torch.float32, torch.float64):
            raise EngineError("pos/quat must both be float32 or both float64")
        if pos.dim() != 3 or quat.dim() != 3 or pos.shape[2] != 3 or quat.shape[2] != 4 or pos.shape[:2] != quat.shape[:2]:
            raise EngineError(f"bad input shapes {tuple(pos.shape)} / {tuple(quat.shape)}")
        pos, quat = pos.contiguous(), quat.contiguous()
        N, B = int(pos.shape[0]), int(pos.shape[1])
        items = np.ascontiguousarray(items, dtype=_native.WORK_ITEM_DTYPE)
        slot_col = np.ascontiguousarray(slot_col, dtype=np.int32)
        if slot_col.shape != (self.info.nslot,):
            raise EngineError("slot_col has the wrong length")
        prm = params or IKParams()
        if out is None:
            out = torch.full((N, self.nq), float("nan"), dtype=torch.float64, device=self.device)
        elif out.shape != (N, self.nq) or out.dtype != torch.float64 or not out.is_contiguous() or \
                (out.device != self.device and not (_host_out and out.is_pinned())):
            raise EngineError("out must be a contiguous float64 [N, nq] tensor on the engine's device")
        if iters is None:
            iters = torch.zeros(N, dtype=torch.int32, device=self.device) if want_iters else None
        if qpos_final is not None:
            if qpos_final.dtype != torch.float64 or qpos_final.dim() != 2 or qpos_final.shape[1] != self.nq or not qpos_final.is_contiguous() \
                    or qpos_final.device != self.device:
                raise EngineError("qpos_final must be a contiguous float64 [R, nq] tensor on the engine's device")
            qfin, n_final = qpos_final, int(qpos_final.shape[0])
        else:
            qfin = torch.zeros((n_final, self.nq), dtype=torch.float64, device=self.device) if n_final > 0 else None
        if qpos_init is not None:
            if qpos_init.dtype != torch.float64 or qpos_init.dim() != 2 or qpos_init.shape[1] != self.nq or qpos_init.device != self.device:
                raise EngineError("qpos_init must be float64 [R, nq] on the engine's device")
            if not qpos_init.is_contiguous():
                qpos_init = qpos_init.contiguous()
            if len(items) and int(items["init_row"].max()) >= qpos_init.shape[0]:
                raise EngineError("init_row outside qpos_init")
        if len(items):
            reach = np.where(items["check_stride"] > 0, (np.maximum(items["n_out"], 1) - 1) // np.maximum(items["check_stride"], 1), 0)
            if int(max((items["final_row"] + reach).max(), (items["burn_row"] + reach).max())) >= n_final:
                raise EngineError("final_row outside qpos_final")
        if frames_done is not None and (frames_done.dtype != torch.int32 or frames_done.device != self.device or not frames_done.is_contiguous()
                                        or frames_done.numel() < len(items)):
            raise EngineError("frames_done must be a contiguous int32 [n_items] tensor on the engine's device")
        stats = IKStats()
        self.last_stats = stats
        if N == 0 or len(items) == 0:  # nothing to launch (empty tensors have no device pointer)
            return out, iters, qfin
        if isinstance(launch_order, str):
            if launch_order != "auto":
                raise EngineError("launch_order must be a tensor, None or 'auto'")
            pf = self._probe_frames(items)
            launch_order = self.plan_order(pos, quat, slot_col, items, prm, qpos_init, probe_frames=pf) if pf else None
        dt = _native.GMR_DTYPE_F64 if pos.dtype == torch.float64 else _native.GMR_DTYPE_F32
        if launch_order is None:
            rc = self._lib.gmr_ik_solve(
                self._h, _ptr(pos), _ptr(quat), dt, B, slot_col.ctypes.data_as(C.c_void_p), N, items.ctypes.data_as(C.c_void_p), len(items),
                C.byref(prm), _ptr(qpos_init), _ptr(qfin), _ptr(out), _ptr(iters), _ptr(frames_done), C.byref(stats), self._stream())
            self._check(rc, "gmr_ik_solve")
        else:
            if launch_order.dtype != torch.int32 or launch_order.device != self.device or not launch_order.is_contiguous() \
                    or launch_order.numel() != len(items):
                raise EngineError("launch_order must be a contiguous int32 [n_items] tensor on the engine's device")
            rc = self._lib.gmr_ik_solve_ordered(
                self._h, _ptr(pos), _ptr(quat), dt, B, slot_col.ctypes.data_as(C.c_void_p), N, items.ctypes.data_as(C.c_void_p), len(items),
                C.byref(prm), _ptr(qpos_init), _ptr(qfin), _ptr(out), _ptr(iters), _ptr(frames_done), C.byref(stats), _ptr(launch_order),
                self._stream())
            self._check(rc, "gmr_ik_solve_ordered")
        self.last_stats = stats
        return out, iters, qfin

    # Launch order by predicted cost (gmr_ik_plan_order: solves of an item's first frames x its length): when it is worth a probe, and of
    # how many frames.  Measured on 8192 distinct clips per launch, any heading, probe and device sort included
    # (tools/experiments/short_clip_probe.py, unshaped_probe_order.py; profiles/r03_unshaped_breakdown.md):
    #   equal lengths (nothing else tells the clips apart): 100 / 150 / 200 frames 32.4 -> 30.2, 47.9 -> 45.0, 65.3 -> 57.6 ms with a 4-frame probe;
    #     300 / 600 / 1000 / 3000 frames 94.8 -> 81.5, 183.7 -> 150.3, 306.5 -> 240.9, 608 -> 549 ms with 32 frames;
    #   lengths U(T/3, 5T/3) (the length order gmr_ik_solve applies by itself is most of the cost order): 300 / 600 frames: any probe loses
    #     (74.9 -> 78.6+, 147.3 -> 150.0+); 1000 / 1500 / 2000 / 3000 frames: 245.5 -> 243.0, 365.6 -> 351.1, 493.7 -> 470.8, 728.7 -> 691.8 ms with 32
    #     frames, shorter probes lose -- a few per cent of the clips need ~12 solves per frame against a median of 6.6 (twice the cost of their length,
    #     DESIGN 6) and have to start first, but it takes 32 frames to tell them from the start-up every clip goes through.
    PROBE_FRAMES = 32
    PROBE_FRAMES_SHORT = 4           # equal-length items of fewer than PROBE_SHORT_BELOW frames
    PROBE_SHORT_BELOW = 256
    PROBE_MIN_ITEMS_PER_SLOT = 1.0   # at most one item per wavefront slot: everything starts at once, order is irrelevant
    PROBE_MAX_LENGTH_SPREAD = 0.10   # coefficient of variation of the item lengths below which lengths carry no cost information
    PROBE_MIN_LENGTH_EQUAL = 64      # mean item length from which a probe pays: equal lengths ...
    PROBE_MIN_LENGTH = 1000          # ... and lengths that differ

    def _probe_frames(self, items: np.ndarray) -> int:
        """Frames of every item to probe before an ``launch_order="auto"`` launch; 0 = launch in length order without a probe."""
        if len(items) == 0 or np.any(items["check_stride"] != 0) or np.any(items["n_burn"] > 0):
            return 0  # walks cannot be probed; speculative chunks (burn-in) are short and alike within a clip: 1.84e7 -> 1.69e7 frames/s with a probe
        slots = 8 * torch.cuda.get_device_properties(self.device).multi_processor_count  # two wavefronts per SIMD
        if len(items) <= self.PROBE_MIN_ITEMS_PER_SLOT * slots:
            return 0
        ln = (items["n_burn"] + items["n_out"]).astype(np.float64)
        mean = float(ln.mean())
        if ln.std() <= self.PROBE_MAX_LENGTH_SPREAD * mean:
            if mean < self.PROBE_MIN_LENGTH_EQUAL:
                return 0
            return self.PROBE_FRAMES if mean >= self.PROBE_SHORT_BELOW else self.PROBE_FRAMES_SHORT
        return self.PROBE_FRAMES if mean >= self.PROBE_MIN_LENGTH else 0

    def _order_pays(self, items: np.ndarray) -> bool:
        return self._probe_frames(items) > 0

    def plan_order(self, pos: torch.Tensor, quat: torch.Tensor, slot_col: np.ndarray, items: np.ndarray, params: Optional[IKParams] = None,
                   qpos_init: Optional[torch.Tensor] = None, probe_frames: Optional[int] = None) -> torch.Tensor:
        """Items by predicted cost, most expensive first: int32 ``[n_items]`` on the device, for ``ik_solve(launch_order=...)``.
        Solves the first ``probe_frames`` frames of every item for their cost alone; asynchronous on the current stream."""
        items = np.ascontiguousarray(items, dtype=_native.WORK_ITEM_DTYPE)
        slot_col = np.ascontiguousarray(slot_col, dtype=np.int32)
        order = torch.empty(len(items), dtype=torch.int32, device=self.device)
        if len(items) == 0:
            return order
        prm = params or IKParams()
        rc = self._lib.gmr_ik_plan_order(
            self._h, _ptr(pos), _ptr(quat), _native.GMR_DTYPE_F64 if pos.dtype == torch.float64 else _native.GMR_DTYPE_F32, int(pos.shape[1]),
            slot_col.ctypes.data_as(C.c_void_p), int(pos.shape[0]), items.ctypes.data_as(C.c_void_p), len(items), C.byref(prm), _ptr(qpos_init),
            int(probe_frames or self.PROBE_FRAMES), _ptr(order), self._stream())
        self._check(rc, "gmr_ik_plan_order")
        return order

    def ik_solve_host(self, pos: np.ndarray, quat: np.ndarray, slot_col: np.ndarray, seq_offsets, params: Optional[IKParams] = None,
                      height_scales=None, first_batch_clips: Optional[int] = None, max_batch_frames: Optional[int] = None, want_iters: bool = True,
                      check: bool = True, out: Optional[np.ndarray] = None):
        """Whole clips from HOST arrays to a HOST result, pipelined: what the dataset scripts hand over
        (scripts/smplx_to_robot_dataset.py:84-89 builds host key-points per file) without a serial copy-in / solve / copy-out.

        * in: the caller's pageable arrays are read in place by the copy engine (all their columns; ``slot_col`` picks the ones the
          config consumes on the device), batch by batch, into two alternating device buffers on two HIP streams, so that batch
          k+1 crosses PCIe while batch k's kernel runs.  The first batch is small -- ``first_batch_clips``, by default one clip per
          wavefront slot -- because its copy is the only one nothing hides; the rest goes in batches of up to ``max_batch_frames``
          frames (default: what fits 16 GiB of staged key-points per buffer), big enough for the engine's cost-ordered launch.
        * out: there is no copy-out.  The kernel writes every frame's qpos straight into the pinned host result (288 B per frame
          of posted PCIe writes against the ~46 us a wavefront spends on a frame); a fresh pageable result array would cost more in
          page faults than the kernel takes, a device buffer + copy engine leaves the last batch's copy exposed.
        Results are bitwise those of ``ik_solve`` on resident tensors (same kernel, same per-clip work items).  With ``check``
        every batch's solve counts are inspected on the device: bit 31 (a non-finite qpos) -> FloatingPointError, bit 30 (a
        capped QP) -> RuntimeError.  Returns (qpos [N, nq] float64, iters [N] int32 or None) as numpy arrays backed by pinned
        memory.  ``out``: the qpos array of an earlier call of the same size, to be overwritten -- page-locking a fresh multi-GB
        result costs more than the solve (7 GB: 0.7 s), so loops over many batches should hand the previous result back (or
        simply drop it before the next call: the allocator then reuses its pinned block).  Measured rates: DESIGN.md.
        """
        from .schedule import make_items
        if pos.dtype != quat.dtype or pos.dtype not in (np.float32, np.float64):
            raise EngineError("pos/quat must both be float32 or both float64")
        if pos.ndim != 3 or quat.ndim != 3 or pos.shape[2] != 3 or quat.shape[2] != 4 or pos.shape[:2] != quat.shape[:2]:
            raise EngineError(f"bad input shapes {pos.shape} / {quat.shape}")
        offs = np.asarray(seq_offsets, dtype=np.int64)
        N, B = int(pos.shape[0]), int(pos.shape[1])
        if offs[0] != 0 or offs[-1] != N:
            raise EngineError("seq_offsets must span [0, N]")
        slot_col = np.ascontiguousarray(slot_col, dtype=np.int32)
        tdt = torch.float32 if pos.dtype == np.float32 else torch.float64
        hs = None if height_scales is None else np.asarray(height_scales, dtype=np.float64)
        if out is not None:
            if not isinstance(out, np.ndarray) or out.shape != (N, self.nq) or out.dtype != np.float64 or not out.flags.c_contiguous:
                raise EngineError("out must be a C-contiguous float64 [N, nq] array (the result of an earlier call)")
            out = torch.from_numpy(out)
            if not out.is_pinned():
                raise EngineError("out must be pinned host memory: pass the result of an earlier ik_solve_host call")
        else:
            out = torch.empty((N, self.nq), dtype=torch.float64, pin_memory=True)
        iters = torch.empty(N, dtype=torch.int32, pin_memory=True) if want_iters else None
        if N == 0:
            return out.numpy(), (iters.numpy() if want_iters else None)
        n_clips = len(offs) - 1
        # batch bounds (clip indices): a first batch of one clip per wavefront slot, then batches of up to max_batch_frames frames
        slots = 8 * torch.cuda.get_device_properties(self.device).multi_processor_count
        per = max(1, int(max_batch_frames)) if max_batch_frames else max(1, (16 << 30) // (B * 7 * pos.dtype.itemsize))
        bounds = [0, min(n_clips, max(1, int(first_batch_clips) if first_batch_clips else slots))]
        while bounds[-1] < n_clips:
            nxt = int(np.searchsorted(offs, offs[bounds[-1]] + per, side="right")) - 1
            bounds.append(min(n_clips, max(nxt, bounds[-1] + 1)))
        cap = max(int(offs[bounds[k + 1]] - offs[bounds[k]]) for k in range(len(bounds) - 1))
        tpos, tquat = torch.from_numpy(np.ascontiguousarray(pos)), torch.from_numpy(np.ascontiguousarray(quat))
        nbuf = min(2, len(bounds) - 1)
        st = [torch.cuda.Stream(self.device) for _ in range(nbuf)]
        dp = [torch.empty((cap, B, 3), dtype=tdt, device=self.device) for _ in range(nbuf)]
        dq = [torch.empty((cap, B, 4), dtype=tdt, device=self.device) for _ in range(nbuf)]
        want_i = want_iters or check
        di = [torch.empty(cap, dtype=torch.int32, device=self.device) for _ in range(nbuf)] if want_i else None
        flags = torch.zeros((nbuf, 2), dtype=torch.int32, device=self.device)
        cur = torch.cuda.current_stream(self.device)
        for s_ in st:
            s_.wait_stream(cur)
        for k in range(len(bounds) - 1):
            b = k % nbuf
            c0, c1 = bounds[k], bounds[k + 1]
            f0, f1 = int(offs[c0]), int(offs[c1])
            n = f1 - f0
            if n == 0:
                continue
            items = make_items(offs[c0:c1 + 1] - f0, height_scales=None if hs is None else hs[c0:c1])
            with torch.cuda.stream(st[b]):
                # pageable -> device: the runtime stages the copy and returns when the host data has been consumed; the other
                # stream's kernel keeps running meanwhile
                dp[b][:n].copy_(tpos[f0:f1], non_blocking=True)
                dq[b][:n].copy_(tquat[f0:f1], non_blocking=True)
                self.ik_solve(dp[b][:n], dq[b][:n], slot_col, items, params=params, out=out[f0:f1], iters=di[b][:n] if want_i else None,
                              want_iters=want_i, _host_out=True)
                if check:
                    flags[b, 0] |= (di[b][:n] >> 31).ne(0).any().to(torch.int32)
                    flags[b, 1] |= ((di[b][:n] >> 30) & 1).ne(0).any().to(torch.int32)
                if want_iters:
                    iters[f0:f1].copy_(di[b][:n], non_blocking=True)
        for s_ in st:
            s_.synchronize()
            cur.wait_stream(s_)
        if check:
            bad = flags.sum(0).cpu().numpy()
            if bad[0]:
                raise FloatingPointError("non-finite qpos")
            if bad[1]:
                raise RuntimeError("a box QP hit its iteration cap (the reference would assert on a failed QP)")
        return out.numpy(), (iters.numpy() if want_iters else None)

    def ik_solve_chunked(self, pos: torch.Tensor, quat: torch.Tensor, slot_col: np.ndarray, seq_offsets, chunk: int, burn_in: int,
                         params: Optional[IKParams] = None, eps: float = 1e-7, height_scales=None,
                         chunk_init: int = _native.INIT_ROOT_TARGET, clip_init: int = _native.INIT_QPOS0):
        """Parallel-in-time solve of long clips with *verified* chunk boundaries, in two launches.

        Launch 1 solves every chunk of ``chunk`` frames concurrently, each (but a clip's first) warmed up over
        ``burn_in`` earlier frames from a speculative state (``chunk_init``: qpos0 with the floating base on the root task's
        target, gmr_blob.h), and records per chunk the state B its first output frame started from and its final state F.
        Launch 2 is one *verification walk* per clip (``schedule.plan_walks``): starting from the exact first chunk it
        compares, boundary by boundary, the true state with the next chunk's B; if they agree to ``eps`` the chunk's stored
        frames are what the sequential run produces and the walk jumps to its F, otherwise it solves that chunk itself from
        the true state.  The result therefore follows the reference's sequential warm-start semantics to ``eps`` however
        good the speculative start was; its quality only decides how much of a clip the walk has to re-solve.
        Returns (qpos [N,nq], iters [N], info dict).
        """
        from .schedule import make_items, plan_walks
        offs = np.asarray(seq_offsets, dtype=np.int64)
        items = make_items(offs, chunk=chunk, burn_in=burn_in, track=True, height_scales=height_scales, chunk_init=chunk_init, clip_init=clip_init)
        n = len(items)
        prm = params or IKParams()
        prm = IKParams(prm.damping, prm.tol, prm.limit_gain, prm.lm_damping, prm.max_iter, prm.offset_to_ground, eps)
        out, iters, qf = self.ik_solve(pos, quat, slot_col, items, params=prm, n_final=2 * n)
        info = {"chunks": n, "passes": 0, "resolved_frames": 0}
        if n == 0:
            return out, iters, info
        walks = plan_walks(items, offs, chunk)
        if len(walks):
            done = torch.zeros(len(walks), dtype=torch.int32, device=self.device)
            self.ik_solve(pos, quat, slot_col, walks, params=prm, qpos_init=qf, qpos_final=qf, out=out, iters=iters, frames_done=done)
            info["resolved_frames"] = int(done.sum().item())
            info["passes"] = 1
        return out, iters, info

    def ik_solve_chunked_sharded(self, pos: torch.Tensor, quat: torch.Tensor, slot_col: np.ndarray, seq_offsets, chunk: int, burn_in: int,
                                 params: Optional[IKParams] = None, eps: float = 1e-7, height_scales=None):
        """``ik_solve_chunked`` with the chunks of every clip spread over all ranks of the default process group
        (``distributed.solve_chunked_sharded``: BASELINE config 3, few long clips on several GPUs).  Every rank passes the
        same full inputs and receives the full result."""
        from .distributed import solve_chunked_sharded
        prm = params or IKParams()
        prm = IKParams(prm.damping, prm.tol, prm.limit_gain, prm.lm_damping, prm.max_iter, prm.offset_to_ground, eps)

        def solve(items, qinit, qfinal, out, iters, done):
            self.ik_solve(pos, quat, slot_col, items, params=prm, qpos_init=qinit, qpos_final=qfinal, out=out, iters=iters, frames_done=done)

        return solve_chunked_sharded(solve, int(pos.shape[0]), self.nq, seq_offsets, chunk, burn_in, self.device, height_scales=height_scales)

    def evaluate(self, qpos: torch.Tensor, pos: Optional[torch.Tensor] = None, quat: Optional[torch.Tensor] = None,
                 slot_col: Optional[np.ndarray] = None, offset_to_ground: bool = False, want_errors: bool = True, want_poses: bool = False,
                 height_scale: Optional[torch.Tensor] = None, want_task_errors: bool = False):
        """Stage errors [N,2] and/or MuJoCo-convention body poses (xpos [N,nb,3], xquat [N,nb,4] wxyz) at ``qpos`` [N,nq].
        With ``want_task_errors`` a fourth value is returned: the per-task 6-vectors [N, ntask1+ntask2, 6]."""
        if qpos.device != self.device or qpos.dtype != torch.float64 or qpos.dim() != 2 or qpos.shape[1] != self.nq:
            raise EngineError("qpos must be float64 [N, nq] on the engine's device")
        qpos = qpos.contiguous()
        N = int(qpos.shape[0])
        err = xp = xq = None
        B, dt = 0, 0
        if want_errors:
            if pos is None or quat is None or slot_col is None:
                raise EngineError("errors need the human key-points and slot_col")
            if pos.device != self.device or quat.device != self.device or pos.dtype != quat.dtype or pos.dtype not in (torch.float32, torch.float64) \
                    or pos.shape[0] != N or pos.shape[:2] != quat.shape[:2] or pos.shape[2] != 3 or quat.shape[2] != 4:
                raise EngineError("bad key-point tensors")
            pos, quat = pos.contiguous(), quat.contiguous()
            slot_col = np.ascontiguousarray(slot_col, dtype=np.int32)
            if slot_col.shape != (self.info.nslot,):
                raise EngineError("slot_col has the wrong length")
            B, dt = int(pos.shape[1]), _native.GMR_DTYPE_F64 if pos.dtype == torch.float64 else _native.GMR_DTYPE_F32
            err = torch.empty((N, 2), dtype=torch.float64, device=self.device)
        terr = None
        if want_task_errors:
            if not want_errors:
                raise EngineError("task errors need want_errors")
            terr = torch.zeros((N, self.info.ntask[0] + self.info.ntask[1], 6), dtype=torch.float64, device=self.device)
        if want_poses:
            xp = torch.empty((N, self.nbody, 3), dtype=torch.float64, device=self.device)
            xq = torch.empty((N, self.nbody, 4), dtype=torch.float64, device=self.device)
        if N == 0:
            return (err, xp, xq, terr) if want_task_errors else (err, xp, xq)
        if height_scale is not None and (height_scale.dtype != torch.float64 or height_scale.device != self.device or height_scale.shape != (N,)
                                         or not height_scale.is_contiguous()):
            raise EngineError("height_scale must be a contiguous float64 [N] tensor on the engine's device")
        rc = self._lib.gmr_evaluate(self._h, _ptr(qpos), N, _ptr(pos) if want_errors else None, _ptr(quat) if want_errors else None, dt, B,
                                    slot_col.ctypes.data_as(C.c_void_p) if want_errors else None, int(bool(offset_to_ground)), _ptr(height_scale),
                                    _ptr(err), _ptr(terr), _ptr(xp), _ptr(xq), self._stream())
        self._check(rc, "gmr_evaluate")
        return (err, xp, xq, terr) if want_task_errors else (err, xp, xq)

    def fk(self, root_pos: torch.Tensor, root_rot_xyzw: torch.Tensor, dof: torch.Tensor, want_rot: bool = True,
           out_pos: Optional[torch.Tensor] = None, out_rot: Optional[torch.Tensor] = None, fitted_shape: Optional[torch.Tensor] = None):
        """``KinematicsModel.forward_kinematics``: body_pos [T,nbody,3] (and body_rot [T,nbody,4] xyzw), float32.  ``out_pos`` /
        ``out_rot``: caller-owned result tensors (a fresh multi-GB allocation costs more than the kernel).  ``fitted_shape``: float32
        [nbody] or [nbody, 3] on the device, the per-body scale of the local translations (kinematics_model.py:225)."""
        for t in (root_pos, root_rot_xyzw, dof):
            if t.device != self.device or t.dtype != torch.float32:
                raise EngineError("fk inputs must be float32 tensors on the engine's device")
        T = int(root_pos.shape[0])
        if root_pos.shape != (T, 3) or root_rot_xyzw.shape != (T, 4) or dof.shape != (T, self.nq - 7):
            raise EngineError("bad fk input shapes")
        root_pos, root_rot_xyzw, dof = root_pos.contiguous(), root_rot_xyzw.contiguous(), dof.contiguous()
        def _out(t, k):
            if t is None:
                return torch.empty((T, self.nbody, k), dtype=torch.float32, device=self.device)
            if t.shape != (T, self.nbody, k) or t.dtype != torch.float32 or t.device != self.device or not t.is_contiguous():
                raise EngineError("fk output tensors must be contiguous float32 [T, nbody, 3 / 4] on the engine's device")
            return t
        bp = _out(out_pos, 3)
        br = _out(out_rot, 4) if want_rot else None
        if fitted_shape is None:
            rc = self._lib.gmr_fk(self._h, _ptr(root_pos), _ptr(root_rot_xyzw), _ptr(dof), T, _ptr(bp), _ptr(br), self._stream())
        else:
            sh = fitted_shape
            if sh.device != self.device or sh.dtype != torch.float32 or tuple(sh.shape) not in ((self.nbody,), (self.nbody, 1), (self.nbody, 3)):
                raise EngineError("fitted_shape must be a float32 [nbody] or [nbody, 3] tensor on the engine's device")
            sh = sh.contiguous()
            rc = self._lib.gmr_fk_shape(self._h, _ptr(root_pos), _ptr(root_rot_xyzw), _ptr(dof), _ptr(sh), 3 if sh.dim() == 2 and sh.shape[1] == 3 else 1,
                                        T, _ptr(bp), _ptr(br), self._stream())
        self._check(rc, "gmr_fk")
        return bp, br

    def _kin_op(self, name: str, x: torch.Tensor, in_shape, out_shape, out: Optional[torch.Tensor]):
        if x.device != self.device or x.dtype != torch.float32 or tuple(x.shape[1:]) != tuple(in_shape):
            raise EngineError(f"{name}: input must be a float32 [T, {', '.join(map(str, in_shape))}] tensor on the engine's device")
        T = int(x.shape[0])
        x = x.contiguous()
        if out is None:
            out = torch.empty((T,) + tuple(out_shape), dtype=torch.float32, device=self.device)
        elif tuple(out.shape) != (T,) + tuple(out_shape) or out.dtype != torch.float32 or out.device != self.device or not out.is_contiguous():
            raise EngineError(f"{name}: the output tensor must be contiguous float32 [T, {', '.join(map(str, out_shape))}] on the engine's device")
        rc = getattr(self._lib, name)(self._h, _ptr(x), T, _ptr(out), self._stream())
        self._check(rc, name)
        return out

    def dof_to_rot(self, dof: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``KinematicsModel.dof_to_rot`` (kinematics_model.py:172-182): [T, ndof] -> [T, nbody-1, 4] xyzw."""
        return self._kin_op("gmr_dof_to_rot", dof, (self.nq - 7,), (self.nbody - 1, 4), out)

    def rot_to_dof(self, joint_rot: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``KinematicsModel.rot_to_dof`` (kinematics_model.py:184-197): [T, nbody-1, 4] -> [T, ndof], clamped to the joint limits."""
        return self._kin_op("gmr_rot_to_dof", joint_rot, (self.nbody - 1, 4), (self.nq - 7,), out)

    def local_rot_to_global(self, local_rot: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``KinematicsModel.convert_local_rot_to_global`` (kinematics_model.py:199-211): [T, nbody, 4] -> [T, nbody, 4]."""
        return self._kin_op("gmr_local_rot_to_global", local_rot, (self.nbody, 4), (self.nbody, 4), out)

    def fk_min_height(self, root_pos: torch.Tensor, root_rot_xyzw: torch.Tensor, dof: torch.Tensor, seq_offsets) -> torch.Tensor:
        for t in (root_pos, root_rot_xyzw, dof):
            if t.device != self.device or t.dtype != torch.float32:
                raise EngineError("fk inputs must be float32 tensors on the engine's device")
        offs = np.ascontiguousarray(seq_offsets, dtype=np.int64)
        T = int(root_pos.shape[0])
        if offs[-1] != T or root_pos.shape != (T, 3) or root_rot_xyzw.shape != (T, 4) or dof.shape != (T, self.nq - 7):
            raise EngineError("bad fk_min_height shapes")
        out = torch.empty(len(offs) - 1, dtype=torch.float32, device=self.device)
        rc = self._lib.gmr_fk_min_height(self._h, _ptr(root_pos.contiguous()), _ptr(root_rot_xyzw.contiguous()), _ptr(dof.contiguous()),
                                         offs.ctypes.data_as(C.c_void_p), len(offs) - 1, _ptr(out), self._stream())
        self._check(rc, "gmr_fk_min_height")
        return out


class EngineGroup:
    """Several robots' batches in ONE launch (``gmr_group_*``; BASELINE config 4, "heterogeneous trees in one launch").

    The members are built for one common kernel variant; ``ik_solve`` takes one batch per member and runs all their work items
    in a single grid.  ``engines[i]`` is member i as an ordinary ``Engine`` (FK, evaluate, sessions, its own launches).
    """

    def __init__(self, cms, device: int = 0):
        if not torch.cuda.is_available():
            raise EngineError("no HIP device visible to torch: the gmr_amd engine has no CPU path")
        self._lib = _native.load()
        self.device_index = int(device)
        self.device = torch.device("cuda", self.device_index)
        n = len(cms)
        blobs = (C.c_char_p * n)(*[cm.blob for cm in cms])
        sizes = (C.c_size_t * n)(*[len(cm.blob) for cm in cms])
        err = C.create_string_buffer(512)
        self._g = self._lib.gmr_group_create(blobs, sizes, n, self.device_index, err, len(err))
        if not self._g:
            raise EngineError(f"gmr_group_create: {err.value.decode()}")
        self.engines = [Engine(cm, device, _borrowed_handle=self._lib.gmr_group_model(self._g, i)) for i, cm in enumerate(cms)]

    def close(self):
        if getattr(self, "_g", None):
            for e in self.engines:
                e.close()
            self._lib.gmr_group_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def ik_solve(self, batches, params: Optional[IKParams] = None):
        """``batches[i] = (pos, quat, slot_col, items)`` for member i (or ``None``: no work) -> list of (qpos, iters) per member."""
        if len(batches) != len(self.engines):
            raise EngineError("one batch (or None) per group member")
        prm = params or IKParams()
        inputs = (_native.GroupInput * len(batches))()
        keep, outs = [], []
        for i, (eng, b) in enumerate(zip(self.engines, batches)):
            if b is None:
                outs.append((None, None))
                continue
            pos, quat, slot_col, items = b
            if pos.device != self.device or quat.device != self.device or pos.dtype != quat.dtype or pos.dtype not in (torch.float32, torch.float64) \
                    or pos.dim() != 3 or quat.dim() != 3 or pos.shape[2] != 3 or quat.shape[2] != 4 or pos.shape[:2] != quat.shape[:2]:
                raise EngineError(f"member {i}: bad key-point tensors")
            pos, quat = pos.contiguous(), quat.contiguous()
            N, B = int(pos.shape[0]), int(pos.shape[1])
            items = np.ascontiguousarray(items, dtype=_native.WORK_ITEM_DTYPE)
            slot_col = np.ascontiguousarray(slot_col, dtype=np.int32)
            if slot_col.shape != (eng.info.nslot,):
                raise EngineError(f"member {i}: slot_col has the wrong length")
            if len(items) and (int(items["init_row"].max()) >= 0 or int(items["final_row"].max()) >= 0 or int(items["burn_row"].max()) >= 0):
                raise EngineError("group launches take plain per-clip items (no state rows)")
            out = torch.full((N, eng.nq), float("nan"), dtype=torch.float64, device=self.device)
            iters = torch.zeros(N, dtype=torch.int32, device=self.device)
            outs.append((out, iters))
            keep += [pos, quat, items, slot_col]
            if N == 0 or len(items) == 0:
                continue
            inputs[i].human_pos, inputs[i].human_quat = pos.data_ptr(), quat.data_ptr()
            inputs[i].in_dtype = _native.GMR_DTYPE_F64 if pos.dtype == torch.float64 else _native.GMR_DTYPE_F32
            inputs[i].n_cols, inputs[i].slot_col, inputs[i].n_frames = B, slot_col.ctypes.data, N
            inputs[i].items, inputs[i].n_items = items.ctypes.data, len(items)
            inputs[i].qpos_out, inputs[i].iters_out = out.data_ptr(), iters.data_ptr()
        rc = self._lib.gmr_group_ik_solve(self._g, inputs, C.byref(prm), C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        if rc != 0:
            msg = self._lib.gmr_group_last_error(self._g)
            raise EngineError(f"gmr_group_ik_solve: {_ERR.get(rc, rc)}: {msg.decode() if msg else ''}")
        return outs


class Session:
    """One live sequence (``gmr_session_*``): a frame in, a qpos out, warm start kept on the device.

    The latency path behind ``GeneralMotionRetargeting.retarget`` -- what scripts/optitrack_to_robot.py:37-46 and the
    interactive viewers drive once per captured frame.  Host numpy in, host numpy out; no torch tensors on the path.
    """

    def __init__(self, engine: Engine, slot_col: np.ndarray, n_cols: int, params: Optional[IKParams] = None, dtype=np.float64):
        self._e = engine  # keeps the model alive for as long as the session
        self._lib = engine._lib
        self.dtype = np.dtype(dtype)
        if self.dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
            raise TypeError("session inputs must be float32 or float64")
        self.n_cols = int(n_cols)
        sc = np.ascontiguousarray(slot_col, dtype=np.int32)
        prm = params or IKParams()
        self._h = self._lib.gmr_session_create(engine._h, int(self.dtype == np.float64), self.n_cols, sc.ctypes.data, C.byref(prm))
        if not self._h:
            msg = self._lib.gmr_last_error(engine._h)
            raise EngineError(f"gmr_session_create: {msg.decode() if msg else ''}")
        self._q = np.empty(engine.nq, dtype=np.float64)
        self._solves = C.c_int32(0)

    def step(self, pos: np.ndarray, quat: np.ndarray, offset_to_ground: bool = False):
        """pos ``[n_cols, 3]``, quat ``[n_cols, 4]`` wxyz -> (qpos ``[nq]`` float64 (fresh copy), solves spent)."""
        p = np.ascontiguousarray(pos, dtype=self.dtype)
        q = np.ascontiguousarray(quat, dtype=self.dtype)
        if p.shape != (self.n_cols, 3) or q.shape != (self.n_cols, 4):
            raise ValueError(f"expected pos [{self.n_cols},3] and quat [{self.n_cols},4], got {p.shape} and {q.shape}")
        rc = self._lib.gmr_session_step(self._h, p.ctypes.data, q.ctypes.data, int(bool(offset_to_ground)), self._q.ctypes.data,
                                        C.addressof(self._solves))
        self._e._check(rc, "gmr_session_step")
        return self._q.copy(), int(self._solves.value)

    def set_persistent(self, idle_ms: int = 200):
        """Serve the following steps from one resident wavefront fed through a pinned mailbox (``gmr_session_set_persistent``):
        lower latency per frame, identical results.  ``idle_ms = 0`` switches back to one launch per frame."""
        self._e._check(self._lib.gmr_session_set_persistent(self._h, int(idle_ms)), "gmr_session_set_persistent")

    def reset(self, qpos: Optional[np.ndarray] = None):
        q = None if qpos is None else np.ascontiguousarray(qpos, dtype=np.float64).reshape(self._e.nq)
        self._e._check(self._lib.gmr_session_reset(self._h, None if q is None else q.ctypes.data), "gmr_session_reset")

    def state(self) -> np.ndarray:
        out = np.empty(self._e.nq, dtype=np.float64)
        self._e._check(self._lib.gmr_session_state(self._h, out.ctypes.data), "gmr_session_state")
        return out

    def close(self):
        if getattr(self, "_h", None) and getattr(self._e, "_h", None):
            self._lib.gmr_session_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
