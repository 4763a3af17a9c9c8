"""Does cutting the clips of a many-clip workload into verified chunks pay as LOAD BALANCING?  (8192 distinct clips: the per-clip cost
spread leaves a tail that 4 clips per wavefront slot cannot level.)  Whole clips vs ik_solve_chunked at several chunk sizes."""
import json, os, sys
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
for name, var_len, yaw in (("equal_len_heading1", False, 1.0), ("var_len_any_heading", True, np.pi)):
    rng = np.random.default_rng(7)
    lens = rng.integers(T // 3, 5 * T // 3 + 1, size=S) if var_len else np.full(S, T)
    pos, quat, names, offs = synth.synth_clips_torch(cm, lens, seed=4242, device=dev, hard=np.arange(S) % 2 == 1, yaw0=yaw)
    items, sc = make_items(offs), cm.slot_columns(names)
    out = torch.empty((int(offs[-1]), eng.nq), dtype=torch.float64, device=dev)

    def timed(fn):
        fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); r = fn(); b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b), r
    ms, _ = timed(lambda: eng.ik_solve(pos, quat, sc, items, out=out, launch_order="auto"))
    ref = out.clone()
    res[f"{name}/whole_clips"] = {"ms": ms}
    for chunk in (1500, 1000, 750, 500):
        ms, (q, it, info) = timed(lambda: eng.ik_solve_chunked(pos, quat, sc, offs, chunk, 24))
        res[f"{name}/chunk{chunk}"] = {"ms": ms, "resolved_frames": info["resolved_frames"], "chunks": info["chunks"], "max_abs_diff": float((q - ref).abs().max().item())}
    del pos, quat, out, ref
print(json.dumps(res, indent=1))
