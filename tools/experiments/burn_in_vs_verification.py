"""Do the clips that fail verification (it turns out: the wound-up ones, ~12 solves on every frame, easy and hard clips alike) verify with a longer burn-in?
256 distinct 3000-frame clips, half of them hard; chunks of 512 and 256 with burn-in 24 ... 384; per-kind re-solved frames."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gmr_amd import synth
from gmr_amd.engine import Engine
from gmr_amd.schedule import make_items, plan_walks
from tests.util import compiled
cm = compiled("smplx", "unitree_g1"); eng = Engine(cm, 0); dev = eng.device
S, T = 256, 3000
hard = np.arange(S) % 2 == 1
pos, quat, names, offs = synth.synth_clips_torch(cm, np.full(S, T), seed=77, device=dev, hard=hard, yaw0=1.0)
sc = cm.slot_columns(names)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
eng.ik_solve(pos, quat, sc, make_items(offs)); torch.cuda.synchronize()
a.record(); q_ref, it_ref, _ = eng.ik_solve(pos, quat, sc, make_items(offs)); b.record(); torch.cuda.synchronize()
print(json.dumps({"whole_ms": round(a.elapsed_time(b), 1), "solves_per_frame_easy": float((it_ref & 0x3fffffff).view(S, T)[~hard].float().mean()), "solves_per_frame_hard": float((it_ref & 0x3fffffff).view(S, T)[hard].float().mean())}), flush=True)
for chunk in (512, 256):
    for burn in (24, 48, 96, 192, 384):
        items = make_items(offs, chunk=chunk, burn_in=burn, track=True)
        walks = plan_walks(items, offs, chunk)
        from gmr_amd.engine import IKParams
        prm = IKParams(check_tol=1e-7)
        a.record()
        out, iters, qf = eng.ik_solve(pos, quat, sc, items, params=prm, n_final=2 * len(items))
        done = torch.zeros(len(walks), dtype=torch.int32, device=dev)
        eng.ik_solve(pos, quat, sc, walks, params=prm, qpos_init=qf, qpos_final=qf, out=out, iters=iters, frames_done=done)
        b.record(); torch.cuda.synchronize()
        d = done.cpu().numpy()
        print(json.dumps({"chunk": chunk, "burn_in": burn, "ms": round(a.elapsed_time(b), 1), "resolved_easy": int(d[~hard].sum()), "resolved_hard": int(d[hard].sum()),
                          "worst_clip_resolved": int(d.max()), "max_abs_diff": float((out - q_ref).abs().max())}), flush=True)
