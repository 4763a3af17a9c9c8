#!/usr/bin/env python3
"""The adapters leg of bench.py on its own (rows f-1, f-2): gmr::bvh_fk_kernel / gmr::smplx_keypoints_kernel against the HBM roofline.

    python tools/adapter_bench.py [--steps K] [--bvh-frames N] [--smplx-frames N]

Prints one JSON object (bench.adapters_leg's record).  Profiled by tools/gpu_adapters.sh (kernel-trace stats + FETCH_SIZE /
WRITE_SIZE passes, each in its own run)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--bvh-frames", type=int, default=4_000_000)
    ap.add_argument("--smplx-frames", type=int, default=1_000_000)
    a = ap.parse_args()
    import torch
    import bench
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    print(json.dumps(bench.adapters_leg(dev, a.bvh_frames, a.smplx_frames, steps=a.steps)))


if __name__ == "__main__":
    main()
