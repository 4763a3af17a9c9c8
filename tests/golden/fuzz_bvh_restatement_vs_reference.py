#!/usr/bin/env python3
"""The numpy restatement the GPU tests and tools/fuzz_adapters.py check `bvh_fk_kernel` with (tests/test_gpu_adapters.py::_bvh_restatement)
against the REFERENCE's own load_lafan1_file, on random files -- the link between "kernel == restatement on random skeletons" (GPU box) and
"== the reference" (only possible here).

Run in the build container only (needs /root/reference):

    python tests/golden/fuzz_bvh_restatement_vs_reference.py [seconds] [seed]

Random skeletons / layouts / Euler orders as in fuzz_bvh_vs_reference.py, some bones carrying the names the loader treats specially.  Compared per
file: every frame's positions (1e-12 relative) and quaternions up to sign (1e-12) for every bone and FootMod entry, and the height estimate.
"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_bvh_vs_reference as gen  # noqa: E402  (also registers the stub parent packages of the reference)
from general_motion_retargeting.utils.lafan1 import load_lafan1_file as ref_load  # noqa: E402

from gmr_amd.bvh import _estimate_height, _foot_mods, read_bvh  # noqa: E402
from tests.test_gpu_adapters import _bvh_restatement  # noqa: E402

SPECIAL = ["LeftFoot", "LeftToe", "RightFoot", "RightToe", "Head"]


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    path = os.path.join(tempfile.mkdtemp(), "f.bvh")
    t0, runs, frames, bad, worst_p, worst_q, with_mods = time.time(), 0, 0, 0, 0.0, 0.0, 0
    while time.time() - t0 < seconds:
        sp = [n for n in SPECIAL if rng.random() < 0.6]
        gen.make_file(rng, path, special=sp)
        ref_frames, ref_h = ref_load(path)
        a = read_bvh(path)
        extra_names, extra_pos, extra_rot = _foot_mods(a.bones)
        gp, gq = _bvh_restatement(a.parents, a.order, a.pos, np.radians(a.eulers_deg), extra_pos, extra_rot, 0.01)
        names = list(a.bones) + list(extra_names)
        ok = list(ref_frames[0].keys()) == names
        if ok:
            rp = np.array([[f[n][0] for n in names] for f in ref_frames])
            rq = np.array([[f[n][1] for n in names] for f in ref_frames])
            dp = float(np.abs(gp - rp).max() / max(1.0, np.abs(rp).max()))
            dq = float(np.minimum(np.abs(gq - rq).max(-1), np.abs(gq + rq).max(-1)).max())
            h = _estimate_height({n: gp[-1, i] for i, n in enumerate(names)})
            worst_p, worst_q = max(worst_p, dp), max(worst_q, dq)
            ok = dp < 1e-12 and dq < 1e-12 and abs(h - ref_h) < 1e-9
        runs += 1
        frames += len(ref_frames)
        with_mods += bool(extra_names)
        if not ok:
            bad += 1
            print(f"MISMATCH: file kept as {path}.{bad}", flush=True)
            os.replace(path, f"{path}.{bad}")
    print(f"numpy restatement of the BVH adapter vs the reference's load_lafan1_file: {runs} random files ({with_mods} with FootMod entries), {frames} frames, {bad} mismatches; "
          f"worst relative position difference {worst_p:.2e}, worst quaternion difference (up to sign) {worst_q:.2e}, heights equal; {time.time() - t0:.0f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
