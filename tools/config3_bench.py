"""BASELINE config 3 shape on one GPU: a LAFAN1-sized set (77 clips of 2000..9000 frames, bvh_to_g1.json).

Few, long clips leave most wavefront slots empty when every clip is one work item; verified parallel-in-time chunking
(Engine.ik_solve_chunked) fills the chip.  Prints frames/s for the sequential schedule and a sweep of (chunk, burn_in)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gmr_amd import params, synth
from gmr_amd.mjcf import load_robot
from gmr_amd.ik_config import load_ik_config
from gmr_amd.model import compile_model
from gmr_amd.engine import Engine
from gmr_amd.schedule import make_items

n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 77
cm = compile_model(load_robot(params.ROBOT_XML_DICT["unitree_g1"], name="unitree_g1"), load_ik_config(params.IK_CONFIG_DICT["bvh"]["unitree_g1"]))
eng = Engine(cm, 0)
rng = np.random.default_rng(3)
base_T, n_base = 9000, 8
pos, quat, names, _, _ = synth.synth_clips(cm, n_base, base_T, seed=33, hard=False, dtype=np.float32)
hpos, hquat, _, _, _ = synth.synth_clips(cm, n_base, base_T, seed=34, hard=True, dtype=np.float32)
pos, quat = np.concatenate([pos, hpos]), np.concatenate([quat, hquat])
lengths = rng.integers(2000, 9001, size=n_clips)
idx = np.concatenate([np.arange(b * base_T, b * base_T + L) for b, L in zip(rng.integers(2 * n_base, size=n_clips), lengths)])
offs = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
dev = eng.device
tp, tq = torch.from_numpy(pos).to(dev)[torch.from_numpy(idx).to(dev)], torch.from_numpy(quat).to(dev)[torch.from_numpy(idx).to(dev)]
sc = cm.slot_columns(names)
N = int(offs[-1])

def timed(fn, reps=2):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t)
    return best, r

t_seq, (q_seq, _, _) = timed(lambda: eng.ik_solve(tp, tq, sc, make_items(offs)))
res = {"clips": n_clips, "frames": N, "sequential": {"s": t_seq, "frames_per_s": N / t_seq}, "chunked": []}
for chunk, burn in ((16, 24), (32, 24), (32, 48), (64, 32), (64, 64), (128, 64), (256, 64)):
    t, (q, _, info) = timed(lambda: eng.ik_solve_chunked(tp, tq, sc, offs, chunk=chunk, burn_in=burn))
    res["chunked"].append({"chunk": chunk, "burn_in": burn, "s": t, "frames_per_s": N / t, "passes": info["passes"], "resolved_frames": info["resolved_frames"],
                           "max_abs_diff_vs_sequential": float((q - q_seq).abs().max().item())})
print(json.dumps(res))
