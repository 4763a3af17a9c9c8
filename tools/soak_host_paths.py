"""Soak of the host-facing paths: Engine.ik_solve_host (kernel writing the pinned result, two streams) against the resident solve,
and a persistent session against the batched solve of the same frames, repeated for the given number of seconds."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gmr_amd import synth
from gmr_amd.engine import Engine, IKParams
from gmr_amd.schedule import make_items
from tests.util import compiled
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
cm = compiled("smplx", "unitree_g1"); eng = Engine(cm); dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
S = 4608
lens = rng.integers(200, 700, size=S)
pos, quat, names, offs = synth.synth_clips_torch(cm, lens, seed=9, device=dev, hard=(np.arange(S) % 2 == 1), yaw0=1.0)
sc = cm.slot_columns(names)
ref, it_ref, _ = eng.ik_solve(pos, quat, sc, make_items(offs))
ref, it_ref = ref.cpu().numpy(), it_ref.cpu().numpy()
hp, hq = pos.cpu().numpy(), quat.cpu().numpy()
T = 4000
p1, q1, _, o1 = synth.synth_clips_torch(cm, [T], seed=10, device=dev, hard=True, yaw0=1.0)
qb, itb, _ = eng.ik_solve(p1, q1, sc, make_items(o1))
qb, itb, p1h, q1h = qb.cpu().numpy(), itb.cpu().numpy(), p1.cpu().numpy(), q1.cpu().numpy()
t0, n_host, n_live, out, last = time.time(), 0, 0, None, time.time()
while time.time() - t0 < seconds:
    first = int(rng.choice([256, 1024, 2048, 3000]))
    out, it = eng.ik_solve_host(hp, hq, sc, offs, first_batch_clips=first, max_batch_frames=int(rng.choice([200000, 600000, 1 << 25])), out=out)
    assert np.array_equal(out, ref) and np.array_equal(it, it_ref), f"host pipeline run {n_host} differs"
    n_host += 1
    s = eng.session(sc, p1h.shape[1], IKParams(), dtype=np.float32)
    s.set_persistent(int(rng.choice([5, 50, 200])))
    for f in range(T):
        q, k = s.step(p1h[f], q1h[f])
        if not (np.abs(q - qb[f]).max() < 1e-9 and k == itb[f]):
            raise SystemExit(f"persistent session frame {f} of run {n_live} differs")
        if f % 997 == 0:
            time.sleep(0.02)  # lets a short idle time-out expire: the next frame relaunches
    s.close(); n_live += 1
    if time.time() - last > 30:
        print(f"{n_host} host-pipeline runs and {n_live} x {T} live frames identical, {time.time() - t0:.0f} s", flush=True); last = time.time()
print(f"soak ok: {n_host} host-pipeline runs of {int(offs[-1])} frames bitwise equal to the resident solve, {n_live} persistent sessions of {T} frames equal to the batched solve, {time.time() - t0:.0f} s")
