"""Compile (robot tree, IK config, human height) into the packed model blob.

Mirrors what ``GeneralMotionRetargeting.__init__`` / ``setup_retarget_configuration``
derive once per clip (reference motion_retarget.py:13-114): the height ratio scales
every ``human_scale_table`` entry (:36-43), tasks with both weights zero are dropped
(:82,100), ``pos_offset - ground_height*z`` and ``rot_offset`` are stored per *human
body* for table 1 (:91-94), and targets for both tables are later built with the
table-1 offsets only (:121).  The result is one contiguous buffer in the layout of
``include/gmr_blob.h``.
"""
from __future__ import annotations

import dataclasses
import struct
from typing import Dict, List, Optional, Sequence

import numpy as np

from .ik_config import IKConfig, IKTask, rot_offset_unit
from .mjcf import RobotModel

BLOB_MAGIC = 0x42524D47
BLOB_VERSION = 1
_HEADER_FMT = "<4I" + "12i" + "23I" + "3I"  # see gmr_blob_header
HEADER_BYTES = struct.calcsize(_HEADER_FMT)

MAX_BODIES = 64
MAX_TASKS = 32
MAX_SLOTS = 32


@dataclasses.dataclass
class CompiledModel:
    robot: RobotModel
    config: Optional[IKConfig]
    ratio: float
    slot_names: List[str]            # human bodies kept (scale-table keys that have table-1 offsets)
    unoffset_scale_keys: List[str]   # scale-table keys with NO table-1 offsets (KeyError if present in data)
    root_slot: int
    tasks: List[List[IKTask]]        # per table, non-zero-weight tasks in table order
    task_body: List[np.ndarray]
    task_slot: List[np.ndarray]
    slot_scale: np.ndarray           # [nslot] human_scale_table * ratio
    slot_pos_off: np.ndarray         # [nslot,3] table-1 pos_offset - ground
    slot_rot_off: np.ndarray         # [nslot,4] table-1 rot_offset, unit wxyz
    blob: bytes

    @property
    def nslot(self) -> int:
        return len(self.slot_names)

    def slot_columns(self, body_names: Sequence[str]) -> np.ndarray:
        """Column of every slot in a ``[T, len(body_names), .]`` input array.

        Raises ``KeyError`` exactly where the reference would: missing human root
        (motion_retarget.py:212), a task whose human body is absent (:129,135), or a
        scale-table body present in the data without table-1 offsets (:241).
        """
        col = {n: i for i, n in enumerate(body_names)}
        cfg = self.config
        if cfg is None:
            raise RuntimeError("model was compiled without an IK config")
        if cfg.human_root_name not in col:
            raise KeyError(cfg.human_root_name)
        for k in self.unoffset_scale_keys:
            if k in col:
                raise KeyError(k)
        needed = set()
        for tab, use in zip(self.tasks, (cfg.use_ik_match_table1, cfg.use_ik_match_table2)):
            if use:
                needed.update(t.human for t in tab)
        out = np.empty(self.nslot, dtype=np.int32)
        for s, n in enumerate(self.slot_names):
            if n in col:
                out[s] = col[n]
            elif n in needed:
                raise KeyError(n)
            else:
                out[s] = col[cfg.human_root_name]  # dropped by scale_human_data; never read by a task
        return out


def _align8(n: int) -> int:
    return (n + 7) & ~7


def compile_model(robot: RobotModel, config: Optional[IKConfig], actual_human_height: Optional[float] = None) -> CompiledModel:
    if robot.nbody > MAX_BODIES:
        raise NotImplementedError(f"{robot.name}: {robot.nbody} bodies > {MAX_BODIES} supported by the IK kernel")
    slot_names: List[str] = []
    unoffset: List[str] = []
    tasks: List[List[IKTask]] = [[], []]
    task_body = [np.zeros(0, np.int32), np.zeros(0, np.int32)]
    task_slot = [np.zeros(0, np.int32), np.zeros(0, np.int32)]
    ratio = 1.0
    slot_scale = np.zeros(0)
    slot_pos_off = np.zeros((0, 3))
    slot_rot_off = np.zeros((0, 4))
    slot_is_foot = np.zeros(0, np.int32)
    root_slot = 0
    use = [0, 0]
    if config is not None:
        if actual_human_height is not None:
            ratio = actual_human_height / config.human_height_assumption  # :36-37
        scale = {k: v * ratio for k, v in config.human_scale_table.items()}  # :42-43
        ground = config.ground_height * np.array([0.0, 0.0, 1.0])  # :54
        tabs = [
            [t for t in config.table1 if t.pos_weight != 0 or t.rot_weight != 0],  # :82
            [t for t in config.table2 if t.pos_weight != 0 or t.rot_weight != 0],  # :100
        ]
        for k, tab in enumerate(tabs):
            humans = [t.human for t in tab]
            if len(set(humans)) != len(humans):
                raise ValueError(f"ik_match_table{k+1}: two robot frames share one human body; the reference would leave a task without target")
            if len(tab) > MAX_TASKS:
                raise NotImplementedError("too many tasks")
        off1: Dict[str, IKTask] = {t.human: t for t in tabs[0]}  # pos_offsets1 / rot_offsets1, :90-94
        # a table that is switched off is never given targets nor solved (:132,163): its frames need not even exist in the robot
        # (smplx_to_r1pro.json's table 2 names G1 links), so it contributes no tasks -- only table 1's offsets above
        tabs = [tab if on else [] for tab, on in zip(tabs, (config.use_ik_match_table1, config.use_ik_match_table2))]
        for name in scale.keys():
            (slot_names if name in off1 else unoffset).append(name)
        if config.human_root_name not in scale:
            raise KeyError(config.human_root_name)  # human_scale_table[human_root_name], :215
        if config.human_root_name not in off1:
            raise KeyError(config.human_root_name)  # rot_offsets[root], :241 (root always survives scaling)
        if len(slot_names) > MAX_SLOTS:
            raise NotImplementedError("too many human bodies")
        root_slot = slot_names.index(config.human_root_name)
        slot_scale = np.array([scale[n] for n in slot_names], dtype=np.float64)
        slot_pos_off = np.array([np.asarray(off1[n].pos_offset, dtype=np.float64) - ground for n in slot_names]).reshape(-1, 3)
        slot_rot_off = np.array([rot_offset_unit(off1[n].rot_offset) for n in slot_names]).reshape(-1, 4)
        slot_is_foot = np.array([int("Foot" in n or "foot" in n) for n in slot_names], dtype=np.int32)  # :260
        sidx = {n: i for i, n in enumerate(slot_names)}
        for k, tab in enumerate(tabs):
            for t in tab:
                if t.human not in sidx:
                    # human_data[body_name] KeyError in update_targets (:129,135): body dropped by scaling
                    raise KeyError(t.human)
            tasks[k] = tab
            task_body[k] = np.array([robot.body_index(t.frame) for t in tab], dtype=np.int32)
            task_slot[k] = np.array([sidx[t.human] for t in tab], dtype=np.int32)
        use = [int(config.use_ik_match_table1), int(config.use_ik_match_table2)]

    nb = robot.nbody
    arrays = [
        ("parent", robot.parent.astype("<i4")),
        ("jnt_type", robot.jnt_type.astype("<i4")),
        ("qpos_adr", robot.qpos_adr.astype("<i4")),
        ("dof_adr", robot.dof_adr.astype("<i4")),
        ("jnt_limited", robot.jnt_limited.astype("<i4")),
        ("body_pos", robot.body_pos.astype("<f8")),
        ("body_quat", robot.body_quat.astype("<f8")),
        ("body_quat_raw", robot.body_quat_raw.astype("<f8")),
        ("jnt_axis", robot.jnt_axis.astype("<f8")),
        ("jnt_range", robot.jnt_range.astype("<f8")),
        ("qpos0", robot.qpos0.astype("<f8")),
        ("slot_scale", slot_scale.astype("<f8")),
        ("slot_pos_off", slot_pos_off.astype("<f8")),
        ("slot_rot_off", slot_rot_off.astype("<f8")),
        ("slot_is_foot", slot_is_foot.astype("<i4")),
        ("task_body0", task_body[0].astype("<i4")),
        ("task_body1", task_body[1].astype("<i4")),
        ("task_slot0", task_slot[0].astype("<i4")),
        ("task_slot1", task_slot[1].astype("<i4")),
        ("task_wp0", np.array([t.pos_weight for t in tasks[0]], dtype="<f8")),
        ("task_wp1", np.array([t.pos_weight for t in tasks[1]], dtype="<f8")),
        ("task_wr0", np.array([t.rot_weight for t in tasks[0]], dtype="<f8")),
        ("task_wr1", np.array([t.rot_weight for t in tasks[1]], dtype="<f8")),
    ]
    offs = {}
    cur = _align8(HEADER_BYTES)
    body = bytearray()
    for name, a in arrays:
        offs[name] = cur
        raw = np.ascontiguousarray(a).tobytes()
        pad = _align8(len(raw)) - len(raw)
        body += raw + b"\0" * pad
        cur += len(raw) + pad
    total = cur
    header = struct.pack(
        _HEADER_FMT,
        BLOB_MAGIC, BLOB_VERSION, total, 0,
        nb, robot.nq, robot.nv, len(slot_names), len(tasks[0]), len(tasks[1]), use[0], use[1], root_slot,
        0 if robot.root_dofs == 0x3F else int(robot.root_dofs), 0, 0,
        offs["parent"], offs["jnt_type"], offs["qpos_adr"], offs["dof_adr"], offs["jnt_limited"],
        offs["body_pos"], offs["body_quat"], offs["body_quat_raw"], offs["jnt_axis"], offs["jnt_range"], offs["qpos0"],
        offs["slot_scale"], offs["slot_pos_off"], offs["slot_rot_off"], offs["slot_is_foot"],
        offs["task_body0"], offs["task_body1"], offs["task_slot0"], offs["task_slot1"],
        offs["task_wp0"], offs["task_wp1"], offs["task_wr0"], offs["task_wr1"],
        0, 0, 0,
    )
    blob = header + b"\0" * (_align8(HEADER_BYTES) - HEADER_BYTES) + bytes(body)
    assert len(blob) == total
    return CompiledModel(
        robot=robot, config=config, ratio=ratio, slot_names=slot_names, unoffset_scale_keys=unoffset,
        root_slot=root_slot, tasks=tasks, task_body=task_body, task_slot=task_slot,
        slot_scale=slot_scale, slot_pos_off=slot_pos_off, slot_rot_off=slot_rot_off, blob=blob,
    )
