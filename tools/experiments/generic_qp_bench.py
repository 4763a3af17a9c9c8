"""Generic-QP kernel variants (ik_kernel<NVP, false>, NVP 36-64): launch time on synthetic robots the structured back end cannot
take, for an A/B of two builds of the library (GMR_AMD_LIB=...).

    python tools/experiments/generic_qp_bench.py            # prints one JSON line
"""
import json
import os
import pathlib
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from gmr_amd import synth
from gmr_amd.engine import Engine
from gmr_amd.schedule import make_items
from tests.test_gpu_parity import _synthetic_robot
from tests.util import compiled

dev = torch.device("cuda", 0)
out = {"lib": os.environ.get("GMR_AMD_LIB", "default")}
tmp = pathlib.Path(tempfile.mkdtemp())
cases = [("48-dof six limbs (NVP 48)", lambda: _synthetic_robot(tmp, [7, 7, 7, 7, 7, 7], 2)),
         ("62-dof four limbs (NVP 64)", lambda: _synthetic_robot(tmp, [14, 14, 14, 14], 4)),
         ("38-dof four limbs (NVP 40, structured)", lambda: _synthetic_robot(tmp, [9, 9, 8, 8], 3))]
if os.environ.get("GMR_AMD_GENERIC_QP") == "1":
    cases.append(("unitree_g1 forced generic (NVP 36)", lambda: compiled("smplx", "unitree_g1")))
for label, make in cases:
    cm = make()
    eng = Engine(cm, 0)
    pos, quat, names, offs, _ = synth.synth_clips(cm, 16, 100, seed=4, hard=True, dtype=np.float32, amp=0.2)
    S = 2048
    tp = torch.from_numpy(pos).to(dev).repeat(S // 16, 1, 1)
    tq = torch.from_numpy(quat).to(dev).repeat(S // 16, 1, 1)
    items = make_items(np.arange(S + 1, dtype=np.int64) * 100)
    sc = cm.slot_columns(names)
    q, it, _ = eng.ik_solve(tp, tq, sc, items)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); eng.ik_solve(tp, tq, sc, items, out=q); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    out[label] = {"nv_padded": int(eng.info.nv_padded), "structured": bool(eng.info.reserved[0]), "ms": float(np.median(ts)), "frames_per_s": S * 100 / (np.median(ts) * 1e-3),
                  "solves_per_frame": float((it & 0x3FFFFFFF).double().mean()), "checksum": float(q.abs().sum())}
print(json.dumps(out))
