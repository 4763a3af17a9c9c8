"""Verified-chunked solve: chunk / burn-in sweep on the two long-clip sets of the bench (LAFAN1-sized: 77 clips of 2000-9000 frames;
a small folder: 24 clips of 4000 frames), to place `chunk="auto"`."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from gmr_amd import synth
dev = torch.device("cuda", 0)
sets = {"lafan_sized_77": bench.long_clip_set(None, synth, dev, yaw0=1.0)}
lc = sets["lafan_sized_77"]
from gmr_amd import params
from gmr_amd.ik_config import load_ik_config
from gmr_amd.mjcf import load_robot
from gmr_amd.model import compile_model
cmb = compile_model(load_robot(params.ROBOT_XML_DICT["unitree_g1"], name="unitree_g1"), load_ik_config(params.IK_CONFIG_DICT["bvh"]["unitree_g1"]))
p, q, names, offs = synth.synth_clips_torch(cmb, np.full(24, 4000), seed=33, device=dev, hard=np.arange(24) % 2 == 1, yaw0=1.0)
sets["folder_24x4000"] = {"eng": lc["eng"], "pos": p, "quat": q, "sc": cmb.slot_columns(names), "offs": offs}
p, q, names, offs = synth.synth_clips_torch(cmb, np.full(4, 9000), seed=34, device=dev, hard=np.arange(4) % 2 == 1, yaw0=1.0)
sets["four_x_9000"] = {"eng": lc["eng"], "pos": p, "quat": q, "sc": cmb.slot_columns(names), "offs": offs}
out = {}
for name, s in sets.items():
    N = int(s["offs"][-1])
    for chunk in (16, 24, 32, 48, 64, 96, 128):
        for burn in (24, 32):
            ts = []
            for _ in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                _, _, info = s["eng"].ik_solve_chunked(s["pos"], s["quat"], s["sc"], s["offs"], chunk=chunk, burn_in=burn)
                torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            out[f"{name} chunk {chunk} burn {burn}"] = (round(N / np.median(ts)), info["resolved_frames"], info["chunks"])
            print(f"{name:16s} N {N:7d} chunk {chunk:3d} burn {burn}: {N / np.median(ts):.3e} f/s  resolved {info['resolved_frames']:5d}  chunks {info['chunks']}", flush=True)
