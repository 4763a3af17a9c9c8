"""Where the time of a verified chunked solve goes on the LAFAN1-sized set (bench.py long_clips, heading within 1 rad): host
planning, the chunk launch, the walk launch, the final read-back -- and what the chunk launch would take in cost order."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from gmr_amd import synth
from gmr_amd.engine import IKParams
from gmr_amd.schedule import make_items, plan_walks
dev = torch.device("cuda", 0)
lc = bench.long_clip_set(None, synth, dev, yaw0=1.0)
eng, pos, quat, sc, offs = lc["eng"], lc["pos"], lc["quat"], lc["sc"], lc["offs"]
chunk, burn = 64, 32
def ev(): return torch.cuda.Event(enable_timing=True)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    items = make_items(offs, chunk=chunk, burn_in=burn, track=True)
    walks = plan_walks(items, offs, chunk)
    t1 = time.perf_counter()
    n = len(items)
    prm = IKParams(check_tol=1e-7)
    e = [ev() for _ in range(3)]
    e[0].record()
    out, iters, qf = eng.ik_solve(pos, quat, sc, items, params=prm, n_final=2 * n)
    e[1].record()
    done = torch.zeros(len(walks), dtype=torch.int32, device=dev)
    eng.ik_solve(pos, quat, sc, walks, params=prm, qpos_init=qf, qpos_final=qf, out=out, iters=iters, frames_done=done)
    e[2].record()
    r = int(done.sum().item())
    t2 = time.perf_counter()
print(f"{len(items)} chunk items, {len(walks)} walks, {int(offs[-1])} frames: host plan {1e3 * (t1 - t0):.2f} ms, chunk launch {e[0].elapsed_time(e[1]):.2f} ms, "
      f"walk launch {e[1].elapsed_time(e[2]):.2f} ms, total wall {1e3 * (t2 - t0):.2f} ms, re-solved {r} frames")
# the chunk launch in cost order (true per-item cost from the run above)
it = (iters.to(torch.int64) & 0x3FFFFFFF)
cs = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(it, 0)])
ob = torch.from_numpy((items["frame_begin"] + items["n_burn"]).astype(np.int64)).to(dev)
cost = (cs[ob + torch.from_numpy(items["n_out"].astype(np.int64)).to(dev)] - cs[ob]).cpu().numpy()
order = torch.from_numpy(np.argsort(-cost, kind="stable").astype(np.int32)).to(dev)
for lbl, o in (("array order", None), ("true-cost order", order)):
    ts = []
    for _ in range(3):
        a, b = ev(), ev()
        a.record(); eng.ik_solve(pos, quat, sc, items, params=prm, n_final=2 * n, launch_order=o); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    print(f"chunk launch, {lbl}: {min(ts):.2f} ms")
