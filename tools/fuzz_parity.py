"""Randomised differential run: the HIP path against the oracle over random robots, seeds, amplitudes, input dtypes, per-clip
heights, offset_to_ground and solver constants, for the given number of seconds.  Reports the worst difference and any frame whose
solve count differs."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gmr_amd import synth
from gmr_amd.engine import Engine, IKParams
from gmr_amd.schedule import make_items
from oracle.oracle import Oracle, IKParams as OParams
from tests.util import CONFIG_ROBOTS, compiled, quat_angle
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
robots = list(CONFIG_ROBOTS) + ["kuavo_s45", "hightorque_hi", "booster_k1", "galaxea_r1pro"]
cache = {}
t0, runs, frames, worst, mism, last = time.time(), 0, 0, 0.0, 0, time.time()
while time.time() - t0 < seconds:
    robot = robots[int(rng.integers(len(robots)))]
    src = "bvh" if robot in ("unitree_g1", "booster_t1", "fourier_n1") and rng.random() < 0.3 else "smplx"
    if (src, robot) not in cache:
        cm = compiled(src, robot)
        cache[(src, robot)] = (cm, Engine(cm), Oracle(cm.blob))
    cm, eng, orc = cache[(src, robot)]
    n, T = int(rng.integers(1, 12)), int(rng.integers(5, 120))
    dt = np.float32 if rng.random() < 0.6 else np.float64
    pos, quat, names, offs, _ = synth.synth_clips(cm, n, T, seed=int(rng.integers(1 << 30)), hard=bool(rng.random() < 0.6), dtype=dt,
                                                  amp=float(rng.uniform(0.1, 0.5)))
    if rng.random() < 0.3:  # face anywhere: the slow start-ups and wound-up clips
        ang = rng.uniform(-np.pi, np.pi, n).repeat(T)
        c, s = np.cos(ang)[:, None], np.sin(ang)[:, None]
        x, y = pos[:, :, 0].copy(), pos[:, :, 1].copy()
        pos[:, :, 0], pos[:, :, 1] = c * x - s * y, s * x + c * y
        spin = np.stack([np.cos(ang / 2), 0 * ang, 0 * ang, np.sin(ang / 2)], -1)[:, None].astype(quat.dtype)
        quat = synth.qmul(np.broadcast_to(spin, quat.shape), quat).astype(dt)
    prm = dict(offset_to_ground=int(rng.random() < 0.2))
    if rng.random() < 0.3:
        prm.update(damping=float(rng.choice([0.05, 0.5, 2.0])), max_iter=int(rng.choice([0, 3, 10, 15])))
    hs = rng.uniform(0.85, 1.2, n) if rng.random() < 0.3 else None
    items = make_items(offs, height_scales=hs, clip_init=-2 if rng.random() < 0.2 else -1)
    sc = cm.slot_columns(names)
    q, it, _ = eng.ik_solve(torch.from_numpy(pos).cuda(), torch.from_numpy(quat).cuda(), sc, items, params=IKParams(**prm))
    q_ref, it_ref, _ = orc.ik_solve(pos, quat, sc, items, params=OParams(**prm), n_threads=16)
    q, it = q.cpu().numpy(), it.cpu().numpy()
    d = max(np.abs(q[:, :3] - q_ref[:, :3]).max(), quat_angle(q[:, 3:7], q_ref[:, 3:7]).max(), np.abs(q[:, 7:] - q_ref[:, 7:]).max())
    bad = int(((it & 0x3FFFFFFF) != it_ref).sum())
    if d > 1e-6 or bad or (it >> 30).any():
        print(f"MISMATCH robot {robot} src {src} n {n} T {T} dtype {dt.__name__} prm {prm}: max diff {d:.3e}, {bad} frames with another solve count, flags {int((it >> 30).any())}", flush=True)
        mism += 1
    worst = max(worst, d); runs += 1; frames += n * T
    if time.time() - last > 30:
        print(f"{runs} runs, {frames} frames, worst {worst:.2e}, mismatching runs {mism}, {time.time() - t0:.0f} s", flush=True); last = time.time()
print(f"fuzz done: {runs} runs, {frames} frames, {len(cache)} (source, robot) pairs, worst difference {worst:.3e}, runs with a mismatch: {mism}")
