#!/usr/bin/env python3
"""Golden vectors for the BVH adapter, from the reference's own loader (build container only).

    python tests/golden/make_bvh_golden.py

Inputs written next to this script (data, not code):
* ``bvh_canonical_40f.bvh``  -- the first 40 motion rows of the reference's out/test_canonical.bvh (101-joint
  Character-Creator skeleton; no LAFAN foot names, so no FootMod entries and the 1.75 m height fallback)
* ``bvh_lafan_like.bvh``     -- a synthetic 22-joint skeleton with LAFAN1 bone names (exercises LeftFootMod /
  RightFootMod and the Head-minus-foot height estimate)
* ``bvh_nine_channel.bvh``   -- the same skeleton in the legacy 9-channel layout (positions-only root; position, rotation
  and scale per joint), the third row shape the reference's reader accepts
Outputs: ``bvh_*.npz`` with ``names``, ``pos [T,B,3]``, ``quat [T,B,4]`` (wxyz), ``human_height`` as returned by
``general_motion_retargeting.utils.lafan1.load_lafan1_file``.
"""
import os
import sys
import types

import numpy as np

REF = os.environ.get("GMR_ROOT", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
pkg = types.ModuleType("general_motion_retargeting")
pkg.__path__ = [os.path.join(REF, "general_motion_retargeting")]
sys.modules["general_motion_retargeting"] = pkg
from general_motion_retargeting.utils.lafan1 import load_lafan1_file  # noqa: E402

LAFAN = [("Hips", -1), ("LeftUpLeg", 0), ("LeftLeg", 1), ("LeftFoot", 2), ("LeftToe", 3), ("RightUpLeg", 0), ("RightLeg", 5),
         ("RightFoot", 6), ("RightToe", 7), ("Spine", 0), ("Spine1", 9), ("Spine2", 10), ("Neck", 11), ("Head", 12),
         ("LeftShoulder", 11), ("LeftArm", 14), ("LeftForeArm", 15), ("LeftHand", 16), ("RightShoulder", 11), ("RightArm", 18),
         ("RightForeArm", 19), ("RightHand", 20)]
OFFS = {"Hips": (0, 0, 0), "LeftUpLeg": (10, -5, 0), "LeftLeg": (0, -42, 0), "LeftFoot": (0, -40, 0), "LeftToe": (0, -6, 14),
        "RightUpLeg": (-10, -5, 0), "RightLeg": (0, -42, 0), "RightFoot": (0, -40, 0), "RightToe": (0, -6, 14), "Spine": (0, 8, 0),
        "Spine1": (0, 12, 0), "Spine2": (0, 12, 0), "Neck": (0, 22, 0), "Head": (0, 10, 0), "LeftShoulder": (4, 18, 0),
        "LeftArm": (14, 0, 0), "LeftForeArm": (28, 0, 0), "LeftHand": (25, 0, 0), "RightShoulder": (-4, 18, 0), "RightArm": (-14, 0, 0),
        "RightForeArm": (-28, 0, 0), "RightHand": (-25, 0, 0)}


def write_lafan_like(path, T=30, seed=0):
    rng = np.random.default_rng(seed)
    children = {i: [j for j, (_, p) in enumerate(LAFAN) if p == i] for i in range(len(LAFAN))}
    out = ["HIERARCHY"]

    def emit(i, depth):
        name, parent = LAFAN[i]
        ind = "\t" * depth
        out.append(f"{ind}{'ROOT' if parent < 0 else 'JOINT'} {name}")
        out.append(ind + "{")
        o = OFFS[name]
        out.append(f"{ind}\tOFFSET {o[0]:.6f} {o[1]:.6f} {o[2]:.6f}")
        if parent < 0:
            out.append(f"{ind}\tCHANNELS 6 Xposition Yposition Zposition Zrotation Yrotation Xrotation")
        else:
            out.append(f"{ind}\tCHANNELS 3 Zrotation Yrotation Xrotation")
        if not children[i]:
            out.append(f"{ind}\tEnd Site")
            out.append(ind + "\t{")
            out.append(f"{ind}\t\tOFFSET 0.000000 5.000000 0.000000")
            out.append(ind + "\t}")
        for c in children[i]:
            emit(c, depth + 1)
        out.append(ind + "}")

    emit(0, 0)
    out += ["MOTION", f"Frames: {T}", "Frame Time: 0.0333333"]
    for t in range(T):
        root = [3.0 * t, 92.0 + np.sin(t / 5.0), 1.5 * t]
        eul = rng.normal(0, 8.0, (len(LAFAN), 3))
        out.append(" ".join(f"{v:.6f}" for v in root + eul.reshape(-1).tolist()))
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")


def write_nine_channel(path, T=12, seed=3):
    """The legacy layout of extract.py:152-156: a positions-only root, (position, rotation, scale) per other joint."""
    rng = np.random.default_rng(seed)
    children = {i: [j for j, (_, p) in enumerate(LAFAN) if p == i] for i in range(len(LAFAN))}
    out = ["HIERARCHY"]

    def emit(i, depth):
        name, parent = LAFAN[i]
        ind = "  " * depth
        out.append(f"{ind}{'ROOT' if parent < 0 else 'JOINT'} {name}")
        out.append(ind + "{")
        o = OFFS[name]
        out.append(f"{ind}  OFFSET {o[0]:.6f} {o[1]:.6f} {o[2]:.6f}")
        if parent < 0:
            out.append(f"{ind}  CHANNELS 3 Xposition Yposition Zposition")
        else:
            out.append(f"{ind}  CHANNELS 9 Xposition Yposition Zposition Zrotation Yrotation Xrotation Xscale Yscale Zscale")
        if not children[i]:
            out.extend([f"{ind}  End Site", ind + "  {", f"{ind}    OFFSET 0.000000 5.000000 0.000000", ind + "  }"])
        for c in children[i]:
            emit(c, depth + 1)
        out.append(ind + "}")

    emit(0, 0)
    out += ["MOTION", f"Frames: {T}", "Frame Time: 0.0333333"]
    for t in range(T):
        row = [2.0 * t, 92.0 + np.cos(t / 4.0), -1.0 * t]
        for _ in range(len(LAFAN) - 1):
            row += rng.normal(0, 0.5, 3).tolist() + rng.normal(0, 8.0, 3).tolist() + rng.uniform(0.9, 1.1, 3).tolist()
        out.append(" ".join(f"{v:.6f}" for v in row))
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")


def dump(bvh, npz):
    frames, h = load_lafan1_file(bvh)
    names = list(frames[0].keys())
    pos = np.array([[f[n][0] for n in names] for f in frames])
    quat = np.array([[f[n][1] for n in names] for f in frames])
    np.savez_compressed(npz, names=np.array(names), pos=pos, quat=quat, human_height=h)
    print(os.path.basename(bvh), pos.shape, h)


def cut(src_name, dst_stem, first, count):
    """Motion rows [first, first + count) of one of the reference's clips as a small fixture + the reference loader's output."""
    lines = open(os.path.join(REF, "out", src_name)).read().split("\n")
    k = next(i for i, ln in enumerate(lines) if ln.startswith("Frames:"))
    head, rows = lines[:k], [ln for ln in lines[k + 2:] if ln.strip()][first:first + count]
    small = os.path.join(HERE, dst_stem + ".bvh")
    with open(small, "w") as f:
        f.write("\n".join(head + [f"Frames: {len(rows)}", lines[k + 1]] + rows) + "\n")
    dump(small, os.path.join(HERE, dst_stem + ".npz"))


def main():
    cut("test_canonical.bvh", "bvh_canonical_40f", 0, 40)
    # the 87-joint pruned clip (another hierarchy: fewer finger / twist bones), frames from the middle of the motion
    cut("test_canonical_pruned.bvh", "bvh_pruned_mid_24f", 120, 24)
    syn = os.path.join(HERE, "bvh_lafan_like.bvh")
    write_lafan_like(syn)
    dump(syn, os.path.join(HERE, "bvh_lafan_like.npz"))
    nine = os.path.join(HERE, "bvh_nine_channel.bvh")
    write_nine_channel(nine)
    dump(nine, os.path.join(HERE, "bvh_nine_channel.npz"))


if __name__ == "__main__":
    main()
