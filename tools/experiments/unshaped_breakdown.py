"""What the un-shaped workload's extra time per solve is made of: clip lengths (equal / U(T/3, 5T/3)) x initial heading (within 1 rad /
anywhere) x distinct clips, each as one ik_solve launch (launch_order="auto"), reported as ns per solve."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gmr_amd import synth
from gmr_amd.engine import Engine
from gmr_amd.schedule import make_items
from tests.util import compiled

S, T = 8192, 3000
cm = compiled("smplx", "unitree_g1")
eng = Engine(cm, 0)
dev = eng.device
res = {}
for name, var_len, yaw in (("equal_len_heading1", False, 1.0), ("equal_len_any_heading", False, np.pi), ("var_len_heading1", True, 1.0), ("var_len_any_heading", True, np.pi)):
    rng = np.random.default_rng(7)
    lens = rng.integers(T // 3, 5 * T // 3 + 1, size=S) if var_len else np.full(S, T)
    pos, quat, names, offs = synth.synth_clips_torch(cm, lens, seed=4242, device=dev, hard=np.arange(S) % 2 == 1, yaw0=yaw)
    items, sc = make_items(offs), cm.slot_columns(names)
    out = torch.empty((int(offs[-1]), eng.nq), dtype=torch.float64, device=dev)
    for order in ("auto", None):
        eng.ik_solve(pos, quat, sc, items, out=out, launch_order=order); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); _, iters, _ = eng.ik_solve(pos, quat, sc, items, out=out, launch_order=order); b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b)
        solves = int((iters & 0x3FFFFFFF).sum().item())
        per_clip = torch.zeros(S, device=dev, dtype=torch.float64).index_add_(0, torch.repeat_interleave(torch.arange(S, device=dev), torch.as_tensor(np.diff(offs), device=dev)), (iters & 0x3FFFFFFF).double())
        res[f"{name}/{order}"] = {"ms": ms, "frames": int(offs[-1]), "solves_per_frame": solves / int(offs[-1]), "ns_per_solve": ms * 1e6 / solves,
                                  "clip_cost_max_over_mean": float(per_clip.max() / per_clip.mean()), "ideal_ms_at_shaped_rate": None}
    it = (iters & 0x3FFFFFFF).cpu().numpy()
    first = np.array([it[a:min(a + 32, b)].sum() for a, b in zip(offs[:-1], offs[1:])])
    np.savez_compressed(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", f"clipcost_{name}.npz"), lens=np.diff(offs), total=per_clip.cpu().numpy(), first32=first)
    del pos, quat, out
print(json.dumps(res, indent=1))
