"""From how many clips on is cutting them into verified chunks a loss?  (schedule.auto_chunk returns (0, 0) from one clip per wavefront slot
on.)  S distinct 3000-frame clips, whole vs chunked (128 / 24, what auto_chunk picks when chunks queue), one MI355X."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gmr_amd import synth
from gmr_amd.engine import Engine
from gmr_amd.schedule import make_items
from tests.util import compiled
cm = compiled("smplx", "unitree_g1"); eng = Engine(cm, 0); dev = eng.device
T = 3000
for S in (256, 512, 1024, 1536, 2048, 3072, 4096):
    pos, quat, names, offs = synth.synth_clips_torch(cm, np.full(S, T), seed=77, device=dev, hard=np.arange(S) % 2 == 1, yaw0=1.0)
    items, sc = make_items(offs), cm.slot_columns(names)

    def timed(fn):
        fn(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(2):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); r = fn(); b.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b))
        return best, r
    ms_w, _ = timed(lambda: eng.ik_solve(pos, quat, sc, items, launch_order="auto"))
    row = {"clips": S, "whole_ms": round(ms_w, 1)}
    for chunk in (128, 256, 512):
        ms_c, (q, it, info) = timed(lambda: eng.ik_solve_chunked(pos, quat, sc, offs, chunk, 24))
        row[f"chunk{chunk}_ms"] = round(ms_c, 1); row[f"chunk{chunk}_resolved"] = info["resolved_frames"]
    print(json.dumps(row), flush=True)
    del pos, quat
