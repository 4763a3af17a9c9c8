"""GPU check of a GMR_IK_MIXED variant build (GMR_AMD_LIB=.../libmixed36.so): difference to the float64 oracle on the bench's
clips, solve-count differences, and the launch time next to the shipped float64 build's."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gmr_amd import synth
from gmr_amd.engine import Engine
from gmr_amd.schedule import make_items
from oracle.oracle import Oracle
from tests.util import compiled

cm = compiled("smplx", "unitree_g1")
eng, orc = Engine(cm, 0), Oracle(cm.blob)
T, D = 3000, 32
pe, qe, names, _, _ = synth.synth_clips(cm, D // 2, T, seed=1000, hard=False, dtype=np.float32)
ph, qh, _, _, _ = synth.synth_clips(cm, D // 2, T, seed=2000, hard=True, dtype=np.float32)
pos, quat = np.concatenate([pe, ph]), np.concatenate([qe, qh])
sc = cm.slot_columns(names)
items = make_items(np.arange(D + 1) * T)
q_ref, it_ref, _ = orc.ik_solve(pos, quat, sc, items, n_threads=16)
tp, tq = torch.from_numpy(pos).cuda(), torch.from_numpy(quat).cuda()
q, it, _ = eng.ik_solve(tp, tq, sc, items)
q, it = q.cpu().numpy(), it.cpu().numpy() & 0x3FFFFFFF
d = np.abs(q - q_ref)
d[:, 3:7] = np.minimum(d[:, 3:7], np.abs(q[:, 3:7] + q_ref[:, 3:7]))
per = d.max(axis=1)
S = 8192
big_p, big_q = tp.repeat(S // D, 1, 1), tq.repeat(S // D, 1, 1)
big_items = make_items(np.arange(S + 1) * T)
out = torch.empty((S * T, eng.nq), dtype=torch.float64, device="cuda")
eng.ik_solve(big_p, big_q, sc, big_items, out=out)
torch.cuda.synchronize()
ts = []
for _ in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); eng.ik_solve(big_p, big_q, sc, big_items, out=out); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
print(json.dumps({"lib": os.environ.get("GMR_AMD_LIB", "default"), "frames_checked": int(len(per)), "max_abs_dq_vs_f64_oracle": float(per.max()),
                  "p999": float(np.quantile(per, 0.999)), "frames_with_different_solve_count": int((it != it_ref).sum()),
                  "nan": bool(np.isnan(q).any()), "kernel_ms_8192x3000": float(np.median(ts)), "frames_per_s": S * T / np.median(ts) * 1e3}))
