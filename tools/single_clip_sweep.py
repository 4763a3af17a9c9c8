import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from gmr_amd import synth
from gmr_amd.engine import Engine
from gmr_amd.schedule import make_items
from tests.util import compiled
cm = compiled("smplx", "unitree_g1"); eng = Engine(cm, 0)
for hard in (False, True):
    pos, quat, names, offs, _ = synth.synth_clips(cm, 1, 3000, seed=11, hard=hard, dtype=np.float32)
    tp, tq = torch.from_numpy(pos).cuda(), torch.from_numpy(quat).cuda()
    sc = cm.slot_columns(names)
    qs, _, _ = eng.ik_solve(tp, tq, sc, make_items(offs)); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): eng.ik_solve(tp, tq, sc, make_items(offs)); torch.cuda.synchronize()
    tseq = (time.perf_counter() - t0) / 3
    print(f"hard={hard} sequential {3000 / tseq:.3e} f/s")
    for chunk, burn in ((8, 24), (4, 24), (8, 16), (16, 24), (8, 32), (6, 18), (12, 24), (4, 16)):
        q, it, info = eng.ik_solve_chunked(tp, tq, sc, offs, chunk, burn); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): q, it, info = eng.ik_solve_chunked(tp, tq, sc, offs, chunk, burn); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print(f"  chunk {chunk:2d} burn {burn:2d}: {3000 / dt:.3e} f/s  resolved {info['resolved_frames']:4d}  maxdiff {float((q - qs).abs().max()):.1e}")
