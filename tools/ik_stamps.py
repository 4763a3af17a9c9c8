#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the IK kernel (separate -DGMR_IK_STAMPS build, never the shipped library).
Read the SHARES, not the run time: stamps serialise phases the real kernel overlaps."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gmr_amd import build, _native
lib_path = os.path.join(ROOT, "gpurun_out", "libgmr_amd_stamps.so")
os.makedirs(os.path.dirname(lib_path), exist_ok=True)
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DGMR_IK_VARIANTS", "-DGMR_IK_STAMPS", f"-I{build.INCLUDE}",
                       "-Wno-unused-value", "-o", lib_path, os.path.join(build.CSRC, "api.hip")])
build.LIB_PATH = _native.LIB_PATH = lib_path
from gmr_amd import synth
from gmr_amd.engine import Engine
from gmr_amd.schedule import make_items
from tests.util import compiled
cm = compiled("smplx", "unitree_g1")
eng = Engine(cm, 0)
S, T = int(os.environ.get("STAMP_CLIPS", 2048)), 300
pe, qe, names, _, _ = synth.synth_clips(cm, 32, T, seed=1000, hard=False)
ph, qh, _, _, _ = synth.synth_clips(cm, 32, T, seed=2000, hard=True)
pos = torch.from_numpy(np.concatenate([pe, ph])).cuda().repeat(S // 64, 1, 1)
quat = torch.from_numpy(np.concatenate([qe, qh])).cuda().repeat(S // 64, 1, 1)
offs = np.arange(S + 1, dtype=np.int64) * T
if os.environ.get("STAMP_UNSHAPED"):  # the un-shaped mix of bench.py: distinct clips, any heading, lengths U(T/3, 5T/3), half of them hard
    T = 900
    lens = np.random.default_rng(7).integers(T // 3, 5 * T // 3 + 1, size=S)
    pos, quat, names, offs = synth.synth_clips_torch(cm, lens, seed=4242, device=torch.device("cuda", 0), hard=(np.arange(S) % 2 == 1), yaw0=np.pi)
eng.ik_solve(pos, quat, cm.slot_columns(names), make_items(offs))
torch.cuda.synchronize()
out = (C.c_ulonglong * 16)()
eng._lib.gmr_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p]
eng._lib.gmr_debug_read_stamps(eng._h, out)
_, it, _ = eng.ik_solve(pos, quat, cm.slot_columns(names), make_items(offs))
torch.cuda.synchronize()
eng._lib.gmr_debug_read_stamps(eng._h, out)
v = np.array(list(out), dtype=np.float64)
names_ = ["prep", "fk", "residual", "task_block", "screws", "composites", "F_c_limits", "H_assemble", "box_qp", "integrate", "output"]
solves = float((it & 0x3FFFFFFF).sum().item())
qp_iters = v[15]; v[15] = 0
with_ws = v[14]; v[14] = 0
print(f"total solves {solves:.0f}; QP iterations per solve {qp_iters / solves:.3f}; solves ending with active bounds {with_ws / solves:.3f}; cycles per solve per wave: {v.sum() / solves:.0f}")
for n, x in zip(names_, v):
    print(f"  {n:12s} {100 * x / v.sum():6.2f} %   {x / solves:9.0f} cyc/solve")
