"""Many SHORT clips (AMASS-like: a few hundred frames): is a cost probe worth its share of the work?  8192 distinct clips of T frames, any heading;
length order (none for equal lengths: array order) vs the order of a p-frame probe, probe included."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gmr_amd import synth
from gmr_amd.engine import Engine
from gmr_amd.schedule import make_items
from tests.util import compiled
cm = compiled("smplx", "unitree_g1"); eng = Engine(cm, 0); dev = eng.device
S = 8192
def timed(fn):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))
for T, var in [(int(a.split(':')[0]), a.endswith(':v')) for a in (sys.argv[1:] or ['150:e', '300:e', '300:v', '600:e', '600:v'])]:
    rng = np.random.default_rng(3)
    lens = rng.integers(T // 3, 5 * T // 3 + 1, size=S) if var else np.full(S, T)
    pos, quat, names, offs = synth.synth_clips_torch(cm, lens, seed=99, device=dev, hard=np.arange(S) % 2 == 1, yaw0=np.pi)
    items, sc = make_items(offs), cm.slot_columns(names)
    out = torch.empty((int(offs[-1]), eng.nq), dtype=torch.float64, device=dev)
    row = {"T": T, "variable_lengths": var, "no_probe_ms": round(timed(lambda: eng.ik_solve(pos, quat, sc, items, out=out, launch_order=None)), 2)}
    for pf in (4, 8, 16, 32):
        row[f"probe{pf}_ms"] = round(timed(lambda: eng.ik_solve(pos, quat, sc, items, out=out, launch_order=eng.plan_order(pos, quat, sc, items, probe_frames=pf))), 2)
    print(json.dumps(row), flush=True)
    del pos, quat, out
