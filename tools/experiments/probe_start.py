"""Probe variants for the launch order: start state (qpos0 like the clip itself, or the root-target start that skips the start-up
transient) x probe length, judged by the time of the ordered main launch (8192 x 3000 bench workload)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gmr_amd import synth
from gmr_amd.engine import Engine
from gmr_amd.schedule import make_items
from gmr_amd._native import INIT_ROOT_TARGET
from tests.util import compiled
cm = compiled('smplx', 'unitree_g1'); eng = Engine(cm); dev = torch.device('cuda', 0)
S, T, D = 8192, 3000, 64
pe, qe, names, _, _ = synth.synth_clips(cm, D // 2, T, seed=1000, hard=False, dtype=np.float32)
ph, qh, _, _, _ = synth.synth_clips(cm, D - D // 2, T, seed=2000, hard=True, dtype=np.float32)
pos = torch.from_numpy(np.concatenate([pe, ph])).to(dev).repeat(S // D, 1, 1).contiguous()
quat = torch.from_numpy(np.concatenate([qe, qh])).to(dev).repeat(S // D, 1, 1).contiguous()
items = make_items(np.arange(S + 1, dtype=np.int64) * T)
sc = cm.slot_columns(names)
out = torch.empty((S * T, eng.nq), dtype=torch.float64, device=dev)
def ev(): return torch.cuda.Event(enable_timing=True)
for start in ("qpos0", "root_target"):
    pit = items.copy()
    if start == "root_target":
        pit["init_row"] = INIT_ROOT_TARGET
    for P in (8, 16, 32):
        best = None
        for _ in range(2):
            a, b, c = ev(), ev(), ev()
            a.record(); order = eng.plan_order(pos, quat, sc, pit, probe_frames=P); b.record()
            eng.ik_solve(pos, quat, sc, items, out=out, want_iters=False, launch_order=order); c.record(); torch.cuda.synchronize()
            t = (a.elapsed_time(b), b.elapsed_time(c))
            best = t if best is None or sum(t) < sum(best) else best
        print(f"probe from {start:11s} {P:2d} frames: probe {best[0]:5.2f} ms + main {best[1]:6.1f} ms = {sum(best):6.1f} ms")
