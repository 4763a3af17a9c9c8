"""The file checks of scripts/smoke_test.py (:19-72, :105-120) without the viewer: every .pkl of a folder must load and hold a motion the
robot's model can play -- keys, shapes, hinge count, frame count, quaternion norms.

    python -m gmr_amd.scripts.smoke_test [--folder out] [--robot unitree_g1]
"""
from __future__ import annotations

import argparse
import os


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    ap.add_argument("--folder", default="out", help="folder of motion .pkl files (the reference looks into <repo>/out)")
    ap.add_argument("--robot", default=None, help="robot whose hinge count the motions must match (default: the registry's first, like the reference)")
    args = ap.parse_args(argv)
    from .. import dataset, params
    from ..mjcf import load_robot
    robot = args.robot or next(iter(params.ROBOT_XML_DICT.keys()))
    nq = load_robot(str(params.ROBOT_XML_DICT[robot])).nq
    if not os.path.isdir(args.folder):
        print(f"no folder {args.folder}: nothing to validate")
        return 0
    ok = bad = 0
    for name in sorted(os.listdir(args.folder)):
        if not name.endswith(".pkl"):
            continue
        path = os.path.join(args.folder, name)
        try:
            motion = dataset.load_robot_motion(path)[0]
            if motion["dof_pos"].shape[1] != nq - 7:   # (:44-47: a structural pass, playback skipped)
                print(f"WARN {name}: dof mismatch motion({motion['dof_pos'].shape[1]}) model({nq - 7})")
                dataset.validate_motion(motion)
            else:
                dataset.validate_motion(motion, nq=nq)
            if motion["dof_pos"].shape[0] == 0:
                raise ValueError("zero frames")
            print(f"OK {name}: frames={motion['dof_pos'].shape[0]} ndof={motion['dof_pos'].shape[1]}")
            ok += 1
        except ValueError as ex:
            if "quaternion" in str(ex):   # (:55-58: suspect norms are a warning there)
                print(f"WARN {name}: {ex}")
                ok += 1
            else:
                print(f"FAIL {name}: {ex!r}")
                bad += 1
        except Exception as ex:
            print(f"FAIL {name}: {ex!r}")
            bad += 1
    print(f"{ok} ok, {bad} failed")
    return 1 if bad else 0


if __name__ == "__main__":
    raise SystemExit(main())
