import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from gmr_amd import synth
from gmr_amd.engine import Engine
from gmr_amd.schedule import make_items
from oracle.oracle import Oracle
from tests.util import compiled, quat_angle
dev = torch.device("cuda", 0)
cm = compiled("smplx", "unitree_g1")
eng = Engine(cm, 0)
print("info", eng.info.n_active_dof, eng.info.nv_padded, eng.info.lds_bytes, flush=True)
orc = Oracle(cm.blob)
for hard in (False, True):
    pos, quat, names, offs, qtrue = synth.synth_clips(cm, 2, 30, seed=21, hard=hard, dtype=np.float32)
    sc = cm.slot_columns(names); items = make_items(offs)
    q_ref, it_ref, _ = orc.ik_solve(pos, quat, sc, items)
    q, it, _ = eng.ik_solve(torch.from_numpy(pos).to(dev), torch.from_numpy(quat).to(dev), sc, items)
    torch.cuda.synchronize()
    q = q.cpu().numpy(); it = it.cpu().numpy()
    d = np.abs(q - q_ref)
    print("hard", hard, "nan", np.isnan(q).sum(), "max diff", np.nanmax(d), "iters gpu", it[:12], "ref", it_ref[:12], flush=True)
    print(" per-frame maxdiff", np.nanmax(d, axis=1)[:12])
