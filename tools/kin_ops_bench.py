#!/usr/bin/env python3
"""bench.kin_ops_leg on its own: dof_to_rot / rot_to_dof / local_rot_to_global kernels against the HBM roofline (G1, random in-limit angles).

    python tools/kin_ops_bench.py [frames]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from gmr_amd import params
    from gmr_amd.engine import Engine
    from gmr_amd.mjcf import load_robot
    from gmr_amd.model import compile_model
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
    robot = sys.argv[2] if len(sys.argv) > 2 else "unitree_g1"
    dev = torch.device("cuda", 0)
    rob = load_robot(str(params.ROBOT_XML_DICT[robot]))
    eng = Engine(compile_model(rob, None), 0)
    lo, hi = (torch.tensor(x, dtype=torch.float32, device=dev) for x in rob.dof_limits())
    dof = lo + torch.rand((T, eng.nq - 7), device=dev) * (hi - lo)
    print(json.dumps(bench.kin_ops_leg(eng, dof)))


if __name__ == "__main__":
    main()
