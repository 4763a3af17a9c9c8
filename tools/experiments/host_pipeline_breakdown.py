"""Where Engine.ik_solve_host spends its time (8192 clips x 3000 frames from pageable arrays)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gmr_amd import synth
from gmr_amd.engine import Engine
from gmr_amd.schedule import make_items
from tests.util import compiled
cm = compiled("smplx", "unitree_g1"); eng = Engine(cm, 0)
S, T, D = int(os.environ.get("S", 8192)), 3000, 32
pe, qe, names, _, _ = synth.synth_clips(cm, D, T, seed=1000, hard=False, dtype=np.float32)
pos = np.tile(pe, (S // D, 1, 1)); quat = np.tile(qe, (S // D, 1, 1))
offs = np.arange(S + 1, dtype=np.int64) * T
sc = cm.slot_columns(names)
N = S * T
res = {}
def t(fn):
    torch.cuda.synchronize(); a = time.perf_counter(); r = fn(); torch.cuda.synchronize(); return time.perf_counter() - a, r
for name, kw in (("default", {}), ("2 batches", {"max_batches": 2}), ("8 batches", {"max_batches": 8, "min_batch_clips": 1024}), ("1 batch", {"max_batches": 1})):
    ts = []
    for rep in range(4):
        dt, r = t(lambda: eng.ik_solve_host(pos, quat, sc, offs, want_iters=False, **kw))
        ts.append(dt); del r
    res[name] = {"s": ts, "frames_per_s_best": N / min(ts)}
# pieces
dt, out = t(lambda: torch.empty((N, 36), dtype=torch.float64, pin_memory=True)); res["pinned_alloc_first_s"] = dt; del out
dt, out = t(lambda: torch.empty((N, 36), dtype=torch.float64, pin_memory=True)); res["pinned_alloc_cached_s"] = dt
dp = torch.empty((N, 14, 3), dtype=torch.float32, device="cuda"); dq = torch.empty((N, 14, 4), dtype=torch.float32, device="cuda")
dt, _ = t(lambda: (dp.copy_(torch.from_numpy(pos)), dq.copy_(torch.from_numpy(quat)))); res["h2d_pageable_s"] = dt
do = torch.empty((N, 36), dtype=torch.float64, device="cuda")
dt, _ = t(lambda: eng.ik_solve(dp, dq, sc, make_items(offs), out=do, want_iters=False)); res["kernel_s"] = dt
dt, _ = t(lambda: out.copy_(do, non_blocking=True)); res["d2h_pinned_s"] = dt
print(json.dumps(res, indent=1))
