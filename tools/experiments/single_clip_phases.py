"""Where one 3000-frame clip's verified-chunked solve spends its time: host scheduling, launch 1 (all chunks), launch 2 (the walk).

Measured (round 3, one MI355X, chunk 16 / burn-in 24, 188 chunks): total 1.68 ms (easy clip) / 2.15 ms (hard) = host plan 0.07 + launch 1
1.32 / 1.80 + walk 0.26 (1.4 us per boundary when every chunk verifies; a re-solved chunk adds its 16 frames, ~0.45 ms).  Launch 1 lasts as long
as its SLOWEST chunk -- 40 frames at one wavefront's latency, more where a stretch needs twice the usual solves (bench.py's clips: 3.2 ms in all) --
so what is left is per-solve latency, not the walk: checking all boundaries in parallel would save at most the 0.26 ms."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gmr_amd import synth, _native
from gmr_amd.engine import Engine, IKParams
from gmr_amd.schedule import make_items, plan_walks
from tests.util import compiled
cm = compiled("smplx", "unitree_g1"); eng = Engine(cm, 0); dev = eng.device
for hard in (False, True):
    pos, quat, names, offs, _ = synth.synth_clips(cm, 1, 3000, seed=5, hard=hard, dtype=np.float32)
    tp, tq = torch.from_numpy(pos).to(dev), torch.from_numpy(quat).to(dev)
    sc = cm.slot_columns(names)
    for chunk, burn in ((16, 24), (24, 24), (32, 24)):
        prm = IKParams(check_tol=1e-7)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        best = None
        for rep in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            items = make_items(offs, chunk=chunk, burn_in=burn, track=True)
            walks = plan_walks(items, offs, chunk)
            t1 = time.perf_counter()
            n = len(items)
            ev[0].record()
            out, iters, qf = eng.ik_solve(tp, tq, sc, items, params=prm, n_final=2 * n)
            ev[1].record()
            done = torch.zeros(len(walks), dtype=torch.int32, device=dev)
            eng.ik_solve(tp, tq, sc, walks, params=prm, qpos_init=qf, qpos_final=qf, out=out, iters=iters, frames_done=done)
            ev[2].record()
            torch.cuda.synchronize(); t2 = time.perf_counter()
            rec = (1e3 * (t2 - t0), 1e3 * (t1 - t0), ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]), int(done.sum().item()))
            best = rec if best is None or rec[0] < best[0] else best
        print(f"hard={hard} chunk {chunk}: total {best[0]:.2f} ms = host plan {best[1]:.2f} + launch 1 {best[2]:.2f} (GPU timeline incl. launch gaps) + launch 2 (walk) {best[3]:.2f}; re-solved {best[4]} frames; {len(items)} chunks")
    # the packaged call (Engine.ik_solve_chunked), as bench.py's single_clip leg times it
    ts = []
    for rep in range(7):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        q, it, info = eng.ik_solve_chunked(tp, tq, sc, offs, chunk=16, burn_in=24)
        torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    print(f"hard={hard} Engine.ik_solve_chunked(16, 24): calls {', '.join(f'{t:.2f}' for t in ts)} ms")
