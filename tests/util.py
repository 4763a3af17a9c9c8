"""Shared helpers for the test-suite (model construction, work items)."""
import functools

import numpy as np

from gmr_amd import params
from gmr_amd.ik_config import load_ik_config
from gmr_amd.mjcf import load_robot
from gmr_amd.model import compile_model

CONFIG_ROBOTS = ["unitree_g1", "unitree_g1_with_hands", "booster_t1", "stanford_toddy", "fourier_n1", "engineai_pm01"]


@functools.lru_cache(maxsize=None)
def compiled(src="smplx", robot="unitree_g1", height=None):
    rob = load_robot(params.ROBOT_XML_DICT[robot], name=robot)
    cfg = load_ik_config(params.IK_CONFIG_DICT[src][robot])
    return compile_model(rob, cfg, height)


def make_items(seq_offsets, dtype):
    """One work item per clip, no burn-in (exact reference semantics)."""
    n = len(seq_offsets) - 1
    items = np.zeros(n, dtype=dtype)
    items["frame_begin"] = seq_offsets[:-1]
    items["n_burn"] = 0
    items["n_out"] = np.diff(seq_offsets)
    items["init_row"] = -1
    items["final_row"] = -1
    items["burn_row"] = -1
    return items


def quat_angle(a, b):
    """Geodesic angle between wxyz unit quaternions (vectorised)."""
    d = np.abs(np.sum(a * b, axis=-1)).clip(0, 1)
    return 2 * np.arccos(d)
