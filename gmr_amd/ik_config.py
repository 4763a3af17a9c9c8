"""IK-config plugin surface: ``ik_configs/<src>_to_<robot>.json``.

Accepts the reference's JSON schema as is (reference
general_motion_retargeting/ik_configs/*.json, consumed at motion_retarget.py:30-54,
80-114) and the normalised ``gmr_amd.ikconfig.v1`` form this repo ships under
``gmr_amd/packs/`` (same information, produced by ``tools/compile_packs.py``).
"""
from __future__ import annotations

import dataclasses
import json
import os
from typing import Dict, List

import numpy as np


@dataclasses.dataclass
class IKTask:
    frame: str          # robot body the FrameTask tracks
    human: str          # human body supplying the target
    pos_weight: float
    rot_weight: float
    pos_offset: List[float]
    rot_offset: List[float]  # wxyz

    def as_entry(self) -> list:
        """The reference's table entry ``[human, wp, wr, pos_off, rot_off]``."""
        return [self.human, self.pos_weight, self.rot_weight, list(self.pos_offset), list(self.rot_offset)]


@dataclasses.dataclass
class IKConfig:
    robot_root_name: str
    human_root_name: str
    ground_height: float
    human_height_assumption: float
    use_ik_match_table1: bool
    use_ik_match_table2: bool
    human_scale_table: Dict[str, float]
    table1: List[IKTask]
    table2: List[IKTask]
    source: str = ""

    @property
    def ik_match_table1(self) -> Dict[str, list]:
        return {t.frame: t.as_entry() for t in self.table1}

    @property
    def ik_match_table2(self) -> Dict[str, list]:
        return {t.frame: t.as_entry() for t in self.table2}

    def to_dict(self) -> dict:
        def tab(ts):
            return [
                {"frame": t.frame, "human": t.human, "wp": t.pos_weight, "wr": t.rot_weight,
                 "pos_off": list(t.pos_offset), "rot_off": list(t.rot_offset)}
                for t in ts
            ]
        return {
            "format": "gmr_amd.ikconfig.v1",
            "source": self.source,
            "robot_root": self.robot_root_name,
            "human_root": self.human_root_name,
            "ground_height": self.ground_height,
            "human_height_assumption": self.human_height_assumption,
            "use_table": [bool(self.use_ik_match_table1), bool(self.use_ik_match_table2)],
            "scale": dict(self.human_scale_table),
            "tables": [tab(self.table1), tab(self.table2)],
        }


def _table_from_reference(tab: dict) -> List[IKTask]:
    out = []
    for frame, entry in tab.items():
        human, wp, wr, poff, roff = entry
        out.append(IKTask(frame, human, float(wp), float(wr), [float(x) for x in poff], [float(x) for x in roff]))
    return out


def ik_config_from_dict(d: dict, source: str = "") -> IKConfig:
    if d.get("format") == "gmr_amd.ikconfig.v1":
        tabs = [
            [IKTask(t["frame"], t["human"], float(t["wp"]), float(t["wr"]), list(t["pos_off"]), list(t["rot_off"])) for t in tab]
            for tab in d["tables"]
        ]
        return IKConfig(
            robot_root_name=d["robot_root"], human_root_name=d["human_root"], ground_height=float(d["ground_height"]),
            human_height_assumption=float(d["human_height_assumption"]),
            use_ik_match_table1=bool(d["use_table"][0]), use_ik_match_table2=bool(d["use_table"][1]),
            human_scale_table={k: float(v) for k, v in d["scale"].items()}, table1=tabs[0], table2=tabs[1],
            source=d.get("source", source),
        )
    # reference schema: KeyError on a missing key, like json.load + dict access there
    return IKConfig(
        robot_root_name=d["robot_root_name"], human_root_name=d["human_root_name"],
        ground_height=float(d["ground_height"]), human_height_assumption=float(d["human_height_assumption"]),
        use_ik_match_table1=bool(d["use_ik_match_table1"]), use_ik_match_table2=bool(d["use_ik_match_table2"]),
        human_scale_table={k: float(v) for k, v in d["human_scale_table"].items()},
        table1=_table_from_reference(d["ik_match_table1"]), table2=_table_from_reference(d["ik_match_table2"]),
        source=source,
    )


def load_ik_config(path) -> IKConfig:
    path = os.fspath(path)
    with open(path) as f:
        d = json.load(f)
    return ik_config_from_dict(d, source=os.path.basename(path))


def rot_offset_unit(q) -> np.ndarray:
    """scipy ``Rotation.from_quat`` normalises (motion_retarget.py:92-94)."""
    q = np.asarray(q, dtype=np.float64)
    return q / np.linalg.norm(q)
