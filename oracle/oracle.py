"""ctypes front end of the CPU oracle (``oracle/gmr_oracle.c``).

TEST INFRASTRUCTURE ONLY -- imported by ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg; never by the ``gmr_amd`` product package.
Parity status: FK (KinematicsModel convention) pinned by reference-generated golden
vectors; IK side "parity unpinned" (mink/mujoco/daqp absent) -- see the C header.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "libgmr_oracle.so")


class IKParams(C.Structure):
    """Mirror of ``gmr_ik_params`` (include/gmr_blob.h)."""

    _fields_ = [
        ("damping", C.c_double), ("tol", C.c_double), ("limit_gain", C.c_double), ("lm_damping", C.c_double),
        ("max_iter", C.c_int32), ("offset_to_ground", C.c_int32), ("check_tol", C.c_double),
    ]

    def __init__(self, damping=0.5, tol=1e-3, limit_gain=0.95, lm_damping=1.0, max_iter=10, offset_to_ground=0, check_tol=1e-7):
        super().__init__(damping, tol, limit_gain, lm_damping, max_iter, int(offset_to_ground), check_tol)


WORK_ITEM_DTYPE = np.dtype(
    [("frame_begin", "<i8"), ("n_burn", "<i4"), ("n_out", "<i4"), ("init_row", "<i4"), ("final_row", "<i4"),
     ("burn_row", "<i4"), ("check_stride", "<i4"), ("height_scale", "<f8")], align=True
)


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "gmr_oracle.c")
    hdr = os.path.join(os.path.dirname(HERE), "include", "gmr_blob.h")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["make", "-C", HERE, "-s"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        vp, dp, ip = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32)
        L.oracle_model_create.restype = vp
        L.oracle_model_create.argtypes = [C.c_char_p, C.c_size_t]
        L.oracle_model_destroy.argtypes = [vp]
        L.oracle_fk_mj.argtypes = [vp, dp, dp, dp]
        L.oracle_task_error_and_jacobian.argtypes = [vp, dp, C.c_int, dp, dp, dp, dp]
        L.oracle_integrate.argtypes = [vp, dp, dp]
        L.oracle_box_qp.restype = C.c_int
        L.oracle_box_qp.argtypes = [C.c_int, dp, dp, dp, dp, dp]
        L.oracle_prepare_targets.argtypes = [vp, dp, dp, C.c_int, dp, dp]
        L.oracle_retarget_frame.restype = C.c_int
        L.oracle_retarget_frame.argtypes = [vp, C.POINTER(IKParams), dp, dp, dp, dp]
        L.oracle_ik_solve.restype = C.c_int
        L.oracle_ik_solve.argtypes = [vp, C.POINTER(IKParams), vp, vp, C.c_int, C.c_int, ip, vp, C.c_int, dp, dp, dp, ip, ip, C.c_int]
        L.oracle_fk_kin.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int64,
                                    C.POINTER(C.c_float), C.POINTER(C.c_float)]
        fp = C.POINTER(C.c_float)
        L.oracle_fk_kin_shape.argtypes = [vp, fp, fp, fp, fp, C.c_int64, fp, fp]
        L.oracle_dof_to_rot.argtypes = [vp, fp, C.c_int64, fp]
        L.oracle_rot_to_dof.argtypes = [vp, fp, C.c_int64, fp]
        L.oracle_local_rot_to_global.argtypes = [vp, fp, C.c_int64, fp]
        L.oracle_stage_error.restype = C.c_double
        L.oracle_stage_error.argtypes = [vp, C.c_int, dp, dp, dp, dp]
        L.oracle_build_qp_at.argtypes = [vp, C.c_int, C.POINTER(IKParams), dp, dp, dp, dp, dp, dp, dp]
        L.oracle_set_mixed_assembly.argtypes = [C.c_int]
        for f in ("oracle_nq", "oracle_nv", "oracle_nbody"):
            getattr(L, f).restype = C.c_int
            getattr(L, f).argtypes = [vp]
        _lib = L
    return _lib


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _c64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Oracle:
    """One compiled model (blob from ``gmr_amd.model.compile_model``) on the CPU oracle."""

    def __init__(self, blob: bytes):
        self._h = lib().oracle_model_create(blob, len(blob))
        if not self._h:
            raise ValueError("oracle rejected the model blob")
        self.nq = lib().oracle_nq(self._h)
        self.nv = lib().oracle_nv(self._h)
        self.nbody = lib().oracle_nbody(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().oracle_model_destroy(self._h)
            self._h = None

    # --- kinematics ---------------------------------------------------
    def fk_mj(self, qpos):
        q = _c64(qpos)
        xpos = np.empty((self.nbody, 3))
        xquat = np.empty((self.nbody, 4))
        lib().oracle_fk_mj(self._h, _d(q), _d(xpos), _d(xquat))
        return xpos, xquat

    def fk_kin(self, root_pos, root_rot_xyzw, dof, want_rot=True, fitted_shape=None):
        rp = np.ascontiguousarray(root_pos, dtype=np.float32)
        rr = np.ascontiguousarray(root_rot_xyzw, dtype=np.float32)
        d = np.ascontiguousarray(dof, dtype=np.float32)
        T = rp.shape[0]
        bp = np.empty((T, self.nbody, 3), dtype=np.float32)
        br = np.empty((T, self.nbody, 4), dtype=np.float32) if want_rot else None
        sh = None
        if fitted_shape is not None:  # [nb] or [nb, 3] (kinematics_model.py:225)
            sh = np.ascontiguousarray(np.broadcast_to(np.asarray(fitted_shape, np.float32).reshape(self.nbody, -1), (self.nbody, 3)))
        lib().oracle_fk_kin_shape(self._h, _f(rp), _f(rr), _f(d), _f(sh) if sh is not None else None, T, _f(bp), _f(br) if want_rot else None)
        return bp, br

    def dof_to_rot(self, dof):
        d = np.ascontiguousarray(dof, dtype=np.float32)
        out = np.empty((d.shape[0], self.nbody - 1, 4), dtype=np.float32)
        lib().oracle_dof_to_rot(self._h, _f(d), d.shape[0], _f(out))
        return out

    def rot_to_dof(self, joint_rot):
        r = np.ascontiguousarray(joint_rot, dtype=np.float32)
        out = np.zeros((r.shape[0], self.nq - 7), dtype=np.float32)
        lib().oracle_rot_to_dof(self._h, _f(r), r.shape[0], _f(out))
        return out

    def local_rot_to_global(self, local_rot):
        r = np.ascontiguousarray(local_rot, dtype=np.float32)
        out = np.empty_like(r)
        lib().oracle_local_rot_to_global(self._h, _f(r), r.shape[0], _f(out))
        return out

    def task_error_and_jacobian(self, qpos, body, tpos, tquat):
        q, tp, tq = _c64(qpos), _c64(tpos), _c64(tquat)
        e = np.empty(6)
        J = np.empty((6, self.nv))
        lib().oracle_task_error_and_jacobian(self._h, _d(q), int(body), _d(tp), _d(tq), _d(e), _d(J))
        return e, J

    def integrate(self, qpos, dq):
        q = _c64(qpos).copy()
        v = _c64(dq)
        lib().oracle_integrate(self._h, _d(q), _d(v))
        return q

    # --- IK pieces ------------------------------------------------------
    def prepare_targets(self, hp, hq, offset_to_ground=False):
        hp, hq = _c64(hp), _c64(hq)
        tp = np.empty_like(hp)
        tq = np.empty_like(hq)
        lib().oracle_prepare_targets(self._h, _d(hp), _d(hq), int(offset_to_ground), _d(tp), _d(tq))
        return tp, tq

    def stage_error(self, tab, qpos, tp, tq, ntask):
        q, tp, tq = _c64(qpos), _c64(tp), _c64(tq)
        e = np.empty(6 * ntask)
        n = lib().oracle_stage_error(self._h, tab, _d(q), _d(tp), _d(tq), _d(e))
        return n, e.reshape(ntask, 6)

    def build_qp(self, tab, qpos, tp, tq, params=None):
        prm = params or IKParams()
        q, tp, tq = _c64(qpos), _c64(tp), _c64(tq)
        H = np.empty((self.nv, self.nv))
        c, lo, hi = np.empty(self.nv), np.empty(self.nv), np.empty(self.nv)
        lib().oracle_build_qp_at(self._h, tab, C.byref(prm), _d(q), _d(tp), _d(tq), _d(H), _d(c), _d(lo), _d(hi))
        return H, c, lo, hi

    def retarget_frame(self, qpos, hp, hq, params=None):
        """One ``retarget()`` call; returns (new qpos, solves, [err1, err2])."""
        prm = params or IKParams()
        q = _c64(qpos).copy()
        hp, hq = _c64(hp), _c64(hq)
        errs = np.zeros(2)
        s = lib().oracle_retarget_frame(self._h, C.byref(prm), _d(q), _d(hp), _d(hq), _d(errs))
        if s < 0:
            raise RuntimeError("oracle QP failed")
        return q, s, errs

    def ik_solve(self, pos, quat, slot_col, items, qpos_init=None, params=None, n_threads=1, want_final=False, qpos_final=None,
                 want_done=False, out=None, iters=None):
        """Batch solve with the work-item semantics of ``gmr_ik_solve``.

        pos ``[N, n_cols, 3]``, quat ``[N, n_cols, 4]`` (float32 or float64, same dtype),
        items: structured array of ``WORK_ITEM_DTYPE``.  Returns (qpos_out [N,nq], iters [N], qpos_final or None).
        ``out`` / ``iters``: caller-owned result arrays (a verification walk writes into the rows its chunks produced).
        """
        prm = params or IKParams()
        assert pos.dtype == quat.dtype and pos.dtype in (np.float32, np.float64)
        pos = np.ascontiguousarray(pos)
        quat = np.ascontiguousarray(quat)
        N, n_cols = pos.shape[0], pos.shape[1]
        slot_col = np.ascontiguousarray(slot_col, dtype=np.int32)
        items = np.ascontiguousarray(items, dtype=WORK_ITEM_DTYPE)
        qout = np.full((N, self.nq), np.nan) if out is None else out
        iters = np.zeros(N, dtype=np.int32) if iters is None else iters
        assert qout.dtype == np.float64 and qout.flags.c_contiguous and qout.shape == (N, self.nq) and iters.dtype == np.int32 and iters.shape == (N,)
        qi = _c64(qpos_init) if qpos_init is not None else None
        nfin = int(max(items["final_row"].max(), items["burn_row"].max())) + 1 if len(items) else 0
        qf = np.zeros((max(nfin, 1), self.nq)) if want_final else None
        if qpos_final is not None:  # caller-owned rows (repair runs read the stored chunk states from them)
            qf = qpos_final
        done = np.zeros(len(items), dtype=np.int32)
        rc = lib().oracle_ik_solve(
            self._h, C.byref(prm), pos.ctypes.data, quat.ctypes.data, int(pos.dtype == np.float64), n_cols, _i(slot_col),
            items.ctypes.data, len(items), _d(qi) if qi is not None else None, _d(qf) if qf is not None else None,
            _d(qout), _i(iters), _i(done), int(n_threads),
        )
        if rc != 0:
            raise RuntimeError("oracle QP failed")
        return (qout, iters, qf, done) if want_done else (qout, iters, qf)


def set_mixed_assembly(on: bool) -> None:
    """Experiment switch of the oracle (tools/experiments/mixed_precision_emulation.py): float32 H / c assembly."""
    lib().oracle_set_mixed_assembly(int(bool(on)))


def box_qp(H, c, lo, hi):
    H, c, lo, hi = _c64(H), _c64(c), _c64(lo), _c64(hi)
    n = c.shape[0]
    x = np.empty(n)
    it = lib().oracle_box_qp(n, _d(H), _d(c), _d(lo), _d(hi), _d(x))
    if it < 0:
        raise RuntimeError("oracle QP failed")
    return x, it
