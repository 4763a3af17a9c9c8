#!/usr/bin/env python3
"""Compile a GMR checkout's robot models and IK configs into gmr_amd packs.

    python tools/compile_packs.py [--gmr-root /root/reference]

Reads ``<root>/assets/<robot>/*.xml`` (kinematic subset only; meshes are never
opened) and ``<root>/general_motion_retargeting/ik_configs/*.json`` and writes
``gmr_amd/packs/robots/<robot>.json`` (``gmr_amd.robot.v1``) and
``gmr_amd/packs/ik_configs/<name>.json`` (``gmr_amd.ikconfig.v1``).  The packs are
what the engine loads when no GMR checkout is reachable through ``GMR_ROOT``.
Robots the kinematic compiler does not support (slide joints, fixed base) are
reported and skipped.
"""
import argparse
import json
import os
import pathlib
import sys

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))

from gmr_amd import params  # noqa: E402
from gmr_amd.ik_config import load_ik_config  # noqa: E402
from gmr_amd.mjcf import MjcfError, load_mjcf  # noqa: E402


def _dump(obj, path):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump(obj, f, separators=(",", ":"))
        f.write("\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gmr-root", default=os.environ.get("GMR_ROOT", "/root/reference"))
    args = ap.parse_args()
    root = pathlib.Path(args.gmr_root)
    out = params.PACK_ROOT
    for robot, rel in params._ROBOT_XML_REL.items():
        src = root / "assets" / rel
        try:
            model = load_mjcf(src, name=robot)
        except (MjcfError, FileNotFoundError) as e:
            print(f"[skip] {robot}: {e}")
            continue
        _dump(model.to_dict(), str(out / "robots" / f"{robot}.json"))
        print(f"[ok]   {robot}: {model.nbody} bodies, nq {model.nq}, nv {model.nv}")
    seen = set()
    for src_h, tab in params._IK_CONFIG_REL.items():
        for robot, fn in tab.items():
            if fn in seen:
                continue
            seen.add(fn)
            p = root / "general_motion_retargeting" / "ik_configs" / fn
            if not p.exists():
                print(f"[skip] {fn}: not found")
                continue
            cfg = load_ik_config(p)
            _dump(cfg.to_dict(), str(out / "ik_configs" / params.pack_name(fn)))
            print(f"[ok]   {fn}: {len(cfg.table1)}+{len(cfg.table2)} table entries")


if __name__ == "__main__":
    main()
