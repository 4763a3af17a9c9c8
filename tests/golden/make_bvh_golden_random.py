#!/usr/bin/env python3
"""More golden vectors for the BVH adapter from the reference's own loader (build container only): random skeletons in the Euler orders
and layouts the four hand-made goldens do not cover.

    python tests/golden/make_bvh_golden_random.py

Files (data): ``bvh_random_<k>.bvh`` + ``.npz`` (names, pos, quat wxyz, human_height as general_motion_retargeting.utils.lafan1.load_lafan1_file
returns them): k = 0..5 -- Euler orders XYZ, YZX, ZXY, XZY, YXZ, ZYX x layouts 3 / 6 / 9 / 3 / 6 / 9, 9-17 joints in random trees with End Sites,
some bones carrying the LAFAN1 names the loader treats specially (LeftFoot / LeftToe / RightFoot / RightToe -> FootMod entries, Head -> the
height estimate), root heights that put the estimate inside and outside its 0.9-2.3 m window.
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

SPECIAL = [["LeftFoot", "LeftToe", "RightFoot", "RightToe", "Head"], ["LeftFoot", "LeftToe", "Head"], ["Head"], [], ["RightFoot", "RightToe"], ["LeftFoot", "LeftToe", "RightFoot", "RightToe", "Head"]]


def write(path, k):
    rng = np.random.default_rng(100 + k)
    order = ["XYZ", "YZX", "ZXY", "XZY", "YXZ", "ZYX"][k]
    layout = [3, 6, 9, 3, 6, 9][k]
    J = int(rng.integers(9, 18))
    parents = [-1] + [int(rng.integers(0, j)) for j in range(1, J)]
    names = [f"Bone{j}" for j in range(J)]
    names[0] = "Hips"
    for n, j in zip(SPECIAL[k], rng.permutation(np.arange(1, J))[: len(SPECIAL[k])]):
        names[int(j)] = n
    children = {j: [c for c in range(J) if parents[c] == j] for j in range(J)}
    out = ["HIERARCHY"]
    rot = " ".join(a + "rotation" for a in order)

    def emit(j, depth):
        ind = "\t" * depth
        out.append(f"{ind}{'ROOT' if parents[j] < 0 else 'JOINT'} {names[j]}")
        out.append(ind + "{")
        o = rng.normal(0, 18, 3) if j else np.zeros(3)
        out.append(f"{ind}\tOFFSET {o[0]:.6f} {o[1]:.6f} {o[2]:.6f}")
        if layout == 9:
            out.append(f"{ind}\tCHANNELS 3 Xposition Yposition Zposition" if parents[j] < 0 else f"{ind}\tCHANNELS 9 Xposition Yposition Zposition {rot} Xscale Yscale Zscale")
        elif layout == 6 or parents[j] < 0:
            out.append(f"{ind}\tCHANNELS 6 Xposition Yposition Zposition {rot}")
        else:
            out.append(f"{ind}\tCHANNELS 3 {rot}")
        for c in children[j]:
            emit(c, depth + 1)
        if not children[j]:
            out.extend([f"{ind}\tEnd Site", ind + "\t{", f"{ind}\t\tOFFSET 0.000000 4.000000 0.000000", ind + "\t}"])
        out.append(ind + "}")
    emit(0, 0)
    T = 8
    out += ["MOTION", f"Frames: {T}", "Frame Time: 0.0333333"]
    for t in range(T):
        root = [2.0 * t, [95.0, 40.0, 160.0][k % 3] + np.sin(t / 3.0), -1.5 * t]
        if layout == 3:
            row = root + rng.normal(0, 35.0, 3 * J).tolist()
        elif layout == 6:
            row = []
            for j in range(J):
                row += (root if j == 0 else rng.normal(0, 12.0, 3).tolist()) + rng.normal(0, 35.0, 3).tolist()
        else:
            row = list(root)
            for _ in range(J - 1):
                row += rng.normal(0, 2.0, 3).tolist() + rng.normal(0, 35.0, 3).tolist() + rng.uniform(0.8, 1.2, 3).tolist()
        out.append(" ".join(f"{v:.6f}" for v in row))
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")


def main():
    for k in range(6):
        bvh = os.path.join(HERE, f"bvh_random_{k}.bvh")
        write(bvh, k)
        frames, h = load_lafan1_file(bvh)
        names = list(frames[0].keys())
        pos = np.array([[f[n][0] for n in names] for f in frames])
        quat = np.array([[f[n][1] for n in names] for f in frames])
        np.savez_compressed(bvh[:-4] + ".npz", names=np.array(names), pos=pos, quat=quat, human_height=h)
        print(os.path.basename(bvh), pos.shape, h, [n for n in names if not n.startswith("Bone")])


if __name__ == "__main__":
    main()
