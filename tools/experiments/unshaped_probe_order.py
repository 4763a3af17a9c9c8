"""Un-shaped workload (8192 distinct clips, lengths U(1000, 5000), any heading): the launch in length order (what gmr_ik_solve applies by itself) against
the probe's predicted-cost order (gmr_ik_plan_order: solves of the first 32 frames x length), probe time included."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gmr_amd import synth
from gmr_amd.engine import Engine
from gmr_amd.schedule import make_items
from tests.util import compiled
cm = compiled("smplx", "unitree_g1"); eng = Engine(cm, 0); dev = eng.device
S, T = 8192, 3000
rng = np.random.default_rng(7)
lens = rng.integers(T // 3, 5 * T // 3 + 1, size=S)
pos, quat, names, offs = synth.synth_clips_torch(cm, lens, seed=4242, device=dev, hard=np.arange(S) % 2 == 1, yaw0=np.pi)
items, sc = make_items(offs), cm.slot_columns(names)
out = torch.empty((int(offs[-1]), eng.nq), dtype=torch.float64, device=dev)
def timed(fn):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))
res = {"length_order_ms": timed(lambda: eng.ik_solve(pos, quat, sc, items, out=out, launch_order=None))}
for pf in (16, 32, 64):
    res[f"probe{pf}_order_ms_incl_probe"] = timed(lambda: eng.ik_solve(pos, quat, sc, items, out=out, launch_order=eng.plan_order(pos, quat, sc, items, probe_frames=pf)))
print(json.dumps(res))
