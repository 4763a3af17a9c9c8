#!/usr/bin/env python3
"""Differential run of gmr_amd.bvh.read_bvh (native tokenizer + number parser) against the REFERENCE's own read_bvh on random files.

Run in the build container only (needs /root/reference; never on the GPU box):

    python tests/golden/fuzz_bvh_vs_reference.py [seconds] [seed]

Same import recipe as make_bvh_golden.py (stub parent package; utils.lafan_vendor.extract / utils need only numpy).  Random skeletons
(1-40 joints, random trees, End Sites anywhere), the three row layouts the reference reads (3-channel joints behind a 6-channel root,
6 channels everywhere, the 9-channel layout with a positions-only root), all six Euler orders, numbers in the spellings the
reference's regexes accept (OFFSET takes [-0-9.e] only, MOTION rows are split at single blanks), 1-12 frames.  Compared, exactly:
bone names, parents, offsets, local positions, and the quaternions the reference derives (its own euler_to_quat +
remove_quat_discontinuities applied to OUR channel angles must give ITS rotations bit for bit: same angles, same order).
"""
import os
import sys
import tempfile
import time
import types

import numpy as np

REF = os.environ.get("GMR_ROOT", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
for name, sub in (("general_motion_retargeting", ""), ("general_motion_retargeting.utils", "utils"), ("general_motion_retargeting.utils.lafan_vendor", "utils/lafan_vendor")):
    m = types.ModuleType(name)
    m.__path__ = [os.path.join(REF, "general_motion_retargeting", sub)]
    sys.modules[name] = m
from general_motion_retargeting.utils.lafan_vendor import extract as ref_extract, utils as ref_utils  # noqa: E402

from gmr_amd.bvh import read_bvh  # noqa: E402  (host code only: no GPU involved)

AX = "XYZ"


def fmt(rng, v):
    k = int(rng.integers(0, 5))
    return ["%.6f" % v, "%.4f" % v, "%d" % round(v), "%.10g" % v, "%.3e" % v][k].replace("e+", "e")   # (no '+', no 'E': the reference's OFFSET regex)


def make_file(rng, path, special=()):
    """``special``: bone names to plant on random non-root joints (the names load_lafan1_file treats specially)."""
    J = int(rng.integers(1, 41))
    parents = [-1] + [int(rng.integers(0, j)) for j in range(1, J)]
    layout = int(rng.choice([3, 6, 9])) if J > 1 else int(rng.choice([3, 6]))
    order = [AX[i] for i in rng.permutation(3)]
    names = ["j%d_%s" % (j, "".join(rng.choice(list("abcXYZ_09"), size=int(rng.integers(0, 5))))) for j in range(J)]
    for n, j in zip(special, rng.permutation(np.arange(1, J))[: len(special)] if J > 1 else []):
        names[int(j)] = n
    children = {j: [c for c in range(J) if parents[c] == j] for j in range(J)}
    # the reference numbers joints in file order = depth-first order of the text: emit recursively and record that order
    file_order, lines = [], ["HIERARCHY"]

    def emit(j, depth):
        ind = "\t" * depth if rng.random() < 0.5 else "  " * depth
        file_order.append(j)
        lines.append(f"{ind}{'ROOT' if parents[j] < 0 else 'JOINT'} {names[j]}")
        lines.append(ind + "{")
        off = rng.normal(0, 10, 3) * (rng.random(3) < 0.9)
        lines.append(f"{ind}\tOFFSET {fmt(rng, off[0])} {fmt(rng, off[1])} {fmt(rng, off[2])}")
        rot = " ".join(a + "rotation" for a in order)
        if layout == 9:
            if parents[j] < 0:
                lines.append(f"{ind}\tCHANNELS 3 Xposition Yposition Zposition")
            else:
                lines.append(f"{ind}\tCHANNELS 9 Xposition Yposition Zposition {rot} Xscale Yscale Zscale")
        elif layout == 6 or parents[j] < 0:
            lines.append(f"{ind}\tCHANNELS 6 Xposition Yposition Zposition {rot}")
        else:
            lines.append(f"{ind}\tCHANNELS 3 {rot}")
        for c in children[j]:
            emit(c, depth + 1)
        if not children[j] or rng.random() < 0.2:
            lines.append(f"{ind}\tEnd Site")
            lines.append(ind + "\t{")
            lines.append(f"{ind}\t\tOFFSET {fmt(rng, rng.normal())} 0 {fmt(rng, rng.normal())}")
            lines.append(ind + "\t}")
        lines.append(ind + "}")
    emit(0, 0)
    T = int(rng.integers(1, 13))
    ncol = {3: 3 + 3 * J, 6: 6 * J, 9: 3 + 9 * (J - 1)}[layout]
    lines += ["MOTION", f"Frames: {T}", "Frame Time: %s" % ["0.033333", "0.008333", "0.0166667"][int(rng.integers(0, 3))]]
    for _ in range(T):
        vals = rng.normal(0, 60, ncol)
        lines.append(" ".join(fmt(rng, v) for v in vals))
    eol = "\n" if rng.random() < 0.8 else "\r\n"
    with open(path, "w", newline="") as f:
        f.write(eol.join(lines) + eol)
    return J, layout


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    d = tempfile.mkdtemp()
    path = os.path.join(d, "f.bvh")
    t0, runs, frames, bad, by_layout = time.time(), 0, 0, 0, {3: 0, 6: 0, 9: 0}
    while time.time() - t0 < seconds:
        J, layout = make_file(rng, path)
        ref = ref_extract.read_bvh(path)
        mine = read_bvh(path)
        ok = list(ref.bones) == list(mine.bones) and np.array_equal(np.asarray(ref.parents), mine.parents) and np.array_equal(ref.offsets, mine.offsets) \
            and np.array_equal(ref.pos, mine.pos)
        order = "".join("xyz"[i] for i in mine.order)
        q = ref_utils.remove_quat_discontinuities(ref_utils.euler_to_quat(np.radians(mine.eulers_deg), order=order))
        ok = ok and np.array_equal(q, ref.quats)
        runs += 1
        frames += len(mine)
        by_layout[layout] += 1
        if not ok:
            bad += 1
            print(f"MISMATCH J={J} layout={layout}; file kept as {path}.{bad}", flush=True)
            os.replace(path, f"{path}.{bad}")
    print(f"bvh parser vs the reference's read_bvh: {runs} random files ({by_layout[3]} / {by_layout[6]} / {by_layout[9]} in the 3- / 6- / 9-channel layouts), {frames} frames, "
          f"{bad} mismatches (names, parents, offsets, positions and the reference's quaternions of our angles compared exactly), {time.time() - t0:.0f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
