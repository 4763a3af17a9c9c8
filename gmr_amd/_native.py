"""ctypes binding of libgmr_amd.so (the C ABI in include/gmr_amd.h).

Fails loudly: if the shared library is missing or does not export the expected ABI,
importing anything that needs it raises -- there is no Python/CPU fallback path.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .build import LIB_PATH as _DEFAULT_LIB_PATH

LIB_PATH = os.environ.get("GMR_AMD_LIB") or _DEFAULT_LIB_PATH  # override only for A/B diagnostics of variant builds

ABI_VERSION = 5
GMR_DTYPE_F32, GMR_DTYPE_F64 = 0, 1

WORK_ITEM_DTYPE = np.dtype(
    [("frame_begin", "<i8"), ("n_burn", "<i4"), ("n_out", "<i4"), ("init_row", "<i4"), ("final_row", "<i4"),
     ("burn_row", "<i4"), ("check_stride", "<i4"), ("height_scale", "<f8")], align=True
)
assert WORK_ITEM_DTYPE.itemsize == 40
INIT_QPOS0, INIT_ROOT_TARGET = -1, -2

EXPORTS = ["gmr_abi_version", "gmr_model_create", "gmr_model_destroy", "gmr_last_error", "gmr_model_info_get",
           "gmr_ik_solve", "gmr_fk", "gmr_fk_shape", "gmr_dof_to_rot", "gmr_rot_to_dof", "gmr_local_rot_to_global", "gmr_fk_min_height", "gmr_bvh_fk", "gmr_bvh_parse_header", "gmr_bvh_parse_motion", "gmr_evaluate", "gmr_smplx_keypoints", "gmr_smplx_keypoints_cols", "gmr_smplx_keypoints_in", "gmr_bvh_fk_rows", "gmr_bvh_parse_motion_device",
           "gmr_session_create", "gmr_session_destroy", "gmr_session_reset", "gmr_session_step", "gmr_session_state", "gmr_session_set_persistent", "gmr_ik_plan_order", "gmr_ik_solve_ordered",
           "gmr_group_create", "gmr_group_destroy", "gmr_group_size", "gmr_group_model", "gmr_group_last_error", "gmr_group_ik_solve"]


class IKParams(C.Structure):
    """``gmr_ik_params`` (include/gmr_blob.h); defaults = the reference's constants."""

    _fields_ = [
        ("damping", C.c_double), ("tol", C.c_double), ("limit_gain", C.c_double), ("lm_damping", C.c_double),
        ("max_iter", C.c_int32), ("offset_to_ground", C.c_int32), ("check_tol", C.c_double),
    ]

    def __init__(self, damping=0.5, tol=1e-3, limit_gain=0.95, lm_damping=1.0, max_iter=10, offset_to_ground=0, check_tol=1e-7):
        super().__init__(float(damping), float(tol), float(limit_gain), float(lm_damping), int(max_iter), int(offset_to_ground),
                         float(check_tol))


class ModelInfo(C.Structure):
    _fields_ = [
        ("nbody", C.c_int32), ("nq", C.c_int32), ("nv", C.c_int32), ("nslot", C.c_int32), ("ntask", C.c_int32 * 2),
        ("n_active_dof", C.c_int32), ("nv_padded", C.c_int32), ("lds_bytes", C.c_int32), ("device", C.c_int32),
        ("reserved", C.c_int32 * 6),
    ]


class GroupInput(C.Structure):
    """``gmr_group_input`` (include/gmr_amd.h): one member's arguments of a group launch."""

    _fields_ = [
        ("human_pos", C.c_void_p), ("human_quat", C.c_void_p), ("in_dtype", C.c_int32), ("n_cols", C.c_int32),
        ("slot_col", C.c_void_p), ("n_frames", C.c_int64), ("items", C.c_void_p), ("n_items", C.c_int32), ("reserved", C.c_int32),
        ("qpos_init", C.c_void_p), ("qpos_final", C.c_void_p), ("qpos_out", C.c_void_p), ("iters_out", C.c_void_p), ("frames_done", C.c_void_p),
    ]


class IKStats(C.Structure):
    _fields_ = [("n_items", C.c_int64), ("n_frames_total", C.c_int64), ("n_frames_out", C.c_int64), ("reserved", C.c_int32 * 4)]


class NativeLibraryError(RuntimeError):
    pass


_lib = None


def load():
    """Return the loaded library, raising ``NativeLibraryError`` if it is unusable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} is missing: build it with `python -m gmr_amd.build` (hipcc, gfx950). "
            "gmr_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    missing = [s for s in EXPORTS if not hasattr(L, s)]
    if missing:
        raise NativeLibraryError(f"{LIB_PATH} does not export {missing}")
    vp = C.c_void_p
    L.gmr_abi_version.restype = C.c_int
    if L.gmr_abi_version() != ABI_VERSION:
        raise NativeLibraryError(f"ABI version mismatch: library {L.gmr_abi_version()}, binding {ABI_VERSION}")
    L.gmr_model_create.restype = vp
    L.gmr_model_create.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_char_p, C.c_size_t]
    L.gmr_model_destroy.argtypes = [vp]
    L.gmr_last_error.restype = C.c_char_p
    L.gmr_last_error.argtypes = [vp]
    L.gmr_model_info_get.argtypes = [vp, C.POINTER(ModelInfo)]
    L.gmr_ik_solve.restype = C.c_int
    L.gmr_ik_solve.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, C.c_int64, vp, C.c_int, C.POINTER(IKParams), vp, vp, vp, vp, vp,
                               C.POINTER(IKStats), vp]
    L.gmr_fk.restype = C.c_int
    L.gmr_fk.argtypes = [vp, vp, vp, vp, C.c_int64, vp, vp, vp]
    L.gmr_fk_shape.restype = C.c_int
    L.gmr_fk_shape.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int64, vp, vp, vp]
    for f in ("gmr_dof_to_rot", "gmr_rot_to_dof", "gmr_local_rot_to_global"):
        getattr(L, f).restype = C.c_int
        getattr(L, f).argtypes = [vp, vp, C.c_int64, vp, vp]
    L.gmr_fk_min_height.restype = C.c_int
    L.gmr_fk_min_height.argtypes = [vp, vp, vp, vp, vp, C.c_int, vp, vp]
    L.gmr_group_create.restype = vp
    L.gmr_group_create.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int, C.c_int, C.c_char_p, C.c_size_t]
    L.gmr_group_destroy.argtypes = [vp]
    L.gmr_group_size.restype = C.c_int
    L.gmr_group_size.argtypes = [vp]
    L.gmr_group_model.restype = vp
    L.gmr_group_model.argtypes = [vp, C.c_int]
    L.gmr_group_last_error.restype = C.c_char_p
    L.gmr_group_last_error.argtypes = [vp]
    L.gmr_group_ik_solve.restype = C.c_int
    L.gmr_ik_plan_order.restype = C.c_int
    L.gmr_ik_plan_order.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, C.c_int64, vp, C.c_int, C.POINTER(IKParams), vp, C.c_int, vp, vp]
    L.gmr_ik_solve_ordered.restype = C.c_int
    L.gmr_ik_solve_ordered.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, C.c_int64, vp, C.c_int, C.POINTER(IKParams), vp, vp, vp, vp, vp,
                                       C.POINTER(IKStats), vp, vp]
    L.gmr_group_ik_solve.argtypes = [vp, C.POINTER(GroupInput), C.POINTER(IKParams), vp]
    L.gmr_bvh_parse_header.restype = C.c_int
    L.gmr_bvh_parse_header.argtypes = [vp, C.c_size_t, C.c_int, vp, C.c_size_t, vp, vp, vp, vp, vp, vp, vp]
    L.gmr_evaluate.restype = C.c_int
    L.gmr_evaluate.argtypes = [vp, vp, C.c_int64, vp, vp, C.c_int, C.c_int, vp, C.c_int, vp, vp, vp, vp, vp, vp]
    L.gmr_smplx_keypoints.restype = C.c_int
    L.gmr_smplx_keypoints.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, C.c_int64, C.c_int64, C.c_int, vp, vp, vp]
    L.gmr_smplx_keypoints_cols.restype = C.c_int
    L.gmr_smplx_keypoints_cols.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, C.c_int64, C.c_int64, C.c_int, vp, C.c_int, vp, vp, vp]
    L.gmr_smplx_keypoints_in.restype = C.c_int
    L.gmr_smplx_keypoints_in.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, C.c_int, C.c_int64, C.c_int64, C.c_int, vp, C.c_int, vp, vp, vp]
    L.gmr_bvh_fk_rows.restype = C.c_int
    L.gmr_bvh_fk_rows.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int, C.c_int, vp, vp, C.c_int64, C.c_int64, C.c_double, vp, C.c_int, vp, vp, vp]
    L.gmr_bvh_parse_motion_device.restype = C.c_int
    L.gmr_bvh_parse_motion_device.argtypes = [vp, C.c_int64, C.c_int, vp, vp, vp, C.c_int64, vp, vp, vp, vp, vp, C.c_int64, vp, vp]
    L.gmr_bvh_fk.restype = C.c_int
    L.gmr_bvh_fk.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, C.c_int64, C.c_double, vp, vp, vp]
    L.gmr_bvh_parse_motion.restype = C.c_int64
    L.gmr_bvh_parse_motion.argtypes = [vp, C.c_size_t, C.c_int64, vp, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.gmr_session_create.restype = vp
    L.gmr_session_create.argtypes = [vp, C.c_int, C.c_int, vp, C.POINTER(IKParams)]
    L.gmr_session_destroy.argtypes = [vp]
    L.gmr_session_reset.restype = C.c_int
    L.gmr_session_reset.argtypes = [vp, vp]
    L.gmr_session_step.restype = C.c_int
    L.gmr_session_step.argtypes = [vp, vp, vp, C.c_int, vp, vp]
    L.gmr_session_state.restype = C.c_int
    L.gmr_session_state.argtypes = [vp, vp]
    L.gmr_session_set_persistent.restype = C.c_int
    L.gmr_session_set_persistent.argtypes = [vp, C.c_int]
    _lib = L
    return L
