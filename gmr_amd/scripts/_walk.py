"""The folder walk both dataset scripts share: which files to convert and where each result goes."""
from __future__ import annotations

import os
import re
from typing import Callable, List, Sequence, Tuple


def natural_key(name: str):
    """natsort's default order for plain file names (scripts/smplx_to_robot_dataset.py:207 sorts with natsorted): digit runs compare as numbers."""
    return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", name)]


def plan_files(src_folder: str, tgt_folder: str, want: Callable[[str], bool], src_ext: str, override: bool, natural: bool = False) -> Tuple[List[str], List[str], int]:
    """(sources, targets, skipped): every file below ``src_folder`` that ``want`` accepts, folder by folder in ``os.walk`` order with the
    names sorted (bvh_to_robot_dataset.py:59-60: ``sorted``; smplx_to_robot_dataset.py:206-207: ``natsorted``), its target
    ``path.replace(src_folder, tgt_folder).replace(src_ext, ".pkl")`` (:67, :212), left out when the target exists unless ``override`` (:69-71, :213)."""
    srcs, tgts, skipped = [], [], 0
    for dirpath, _, filenames in os.walk(src_folder):
        for filename in sorted(filenames, key=natural_key if natural else None):
            if not want(filename):
                continue
            s = os.path.join(dirpath, filename)
            t = s.replace(src_folder, tgt_folder).replace(src_ext, ".pkl")
            if os.path.exists(t) and not override:
                skipped += 1
                continue
            srcs.append(s)
            tgts.append(t)
    return srcs, tgts, skipped


def hard_motion_names(paths: Sequence[str]) -> List[str]:
    """Motion names listed in the reference's ``assets/hard_motions/*.txt`` (``Motion: <path>, ...`` lines; smplx_to_robot_dataset.py:193-203)."""
    out = []
    for p in paths:
        if not os.path.exists(p):
            continue
        with open(p, "r") as f:
            for line in f:
                if "Motion:" not in line:
                    continue
                out.append(line.split(":")[1].strip().split(",")[0].strip().split(".")[0])
    return out
