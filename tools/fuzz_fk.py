"""Randomised differential run of the KinematicsModel FK kernels (fk_pos_kernel, fk_kernel<0>, the min-height reduction) against the
oracle's float32 restatement (itself pinned by reference-generated goldens): random robots, frame counts (tile edges), angle ranges
up to +-8 rad, root positions up to +-50 m, non-unit root quaternions; and of the class's other operators (dof_to_rot, rot_to_dof,
local_rot_to_global -- the chain must be bit-exact --, fitted_shape)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gmr_amd.engine import Engine
from oracle.oracle import Oracle
from tests.util import compiled
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
robots = ["unitree_g1", "unitree_g1_with_hands", "booster_t1", "stanford_toddy", "fourier_n1", "engineai_pm01", "kuavo_s45", "hightorque_hi", "booster_k1"]
cache = {}
t0, runs, frames, worst_p, worst_r, worst_k, last = time.time(), 0, 0, 0.0, 0.0, 0.0, time.time()
while time.time() - t0 < seconds:
    robot = robots[int(rng.integers(len(robots)))]
    if robot not in cache:
        cm = compiled("smplx", robot)
        cache[robot] = (cm, Engine(cm), Oracle(cm.blob))
    cm, eng, orc = cache[robot]
    n = int(rng.choice([1, 63, 64, 65, 127, 128, 129, 1000, int(rng.integers(1, 5000))]))
    nd = cm.robot.nq - 7
    amp = float(rng.choice([0.3, 1.5, 8.0]))
    rp = (rng.normal(size=(n, 3)) * float(rng.choice([1.0, 50.0]))).astype(np.float32)
    rr = rng.normal(size=(n, 4)).astype(np.float32)
    rr /= np.linalg.norm(rr, axis=1, keepdims=True)
    if rng.random() < 0.3:
        rr *= rng.uniform(0.5, 2.0, (n, 1)).astype(np.float32)   # the reference does not normalise the root rotation
    dof = rng.uniform(-amp, amp, (n, nd)).astype(np.float32)
    bp_ref, br_ref = orc.fk_kin(rp, rr, dof)
    t = lambda a: torch.from_numpy(a).cuda()
    bp, br = eng.fk(t(rp), t(rr), t(dof))
    bp2, _ = eng.fk(t(rp), t(rr), t(dof), want_rot=False)
    sp, sr = max(1.0, np.abs(bp_ref).max()), max(1.0, np.abs(br_ref).max())
    dp = max(np.abs(bp.cpu().numpy() - bp_ref).max(), np.abs(bp2.cpu().numpy() - bp_ref).max()) / sp
    dr = np.abs(br.cpu().numpy() - br_ref).max() / sr
    offs = np.unique(np.concatenate([[0, n], rng.integers(0, n + 1, size=int(rng.integers(0, 4)))])).astype(np.int64)
    mz = eng.fk_min_height(t(rp), t(rr), t(dof), offs).cpu().numpy()
    mz_ref = np.array([bp_ref[a:b, :, 2].min() for a, b in zip(offs[:-1], offs[1:])])
    dz = np.abs(mz - mz_ref).max() / sp
    # the other KinematicsModel operators on the same draw
    nb = cm.robot.nbody
    jr = eng.dof_to_rot(t(dof))
    dk = float(np.abs(jr.cpu().numpy() - orc.dof_to_rot(dof)).max())
    lr = rng.normal(size=(n, nb, 4)).astype(np.float32)
    lr /= np.linalg.norm(lr, axis=-1, keepdims=True)
    if rng.random() < 0.3:
        lr[..., :3] *= np.float32(10.0 ** rng.uniform(-7, 0))   # small rotations, down through the 1e-5 axis threshold
        lr[..., 3] = np.sqrt(np.maximum(0.0, 1.0 - (lr[..., :3].astype(np.float64) ** 2).sum(-1))).astype(np.float32) * np.where(rng.random((n, nb)) < 0.5, -1, 1)
    dk = max(dk, float(np.abs(eng.rot_to_dof(t(np.ascontiguousarray(lr[:, 1:]))).cpu().numpy() - orc.rot_to_dof(lr[:, 1:])).max()) / 3.2)
    chain_equal = np.array_equal(eng.local_rot_to_global(t(lr)).cpu().numpy(), orc.local_rot_to_global(lr))
    shp = rng.uniform(0.5, 2.0, (nb, int(rng.choice([1, 3])))).astype(np.float32)
    bps, brs = eng.fk(t(rp), t(rr), t(dof), fitted_shape=t(shp[:, 0] if shp.shape[1] == 1 else shp))
    bps_ref, brs_ref = orc.fk_kin(rp, rr, dof, fitted_shape=shp)
    dp = max(dp, np.abs(bps.cpu().numpy() - bps_ref).max() / max(1.0, np.abs(bps_ref).max()))
    dr = max(dr, np.abs(brs.cpu().numpy() - brs_ref).max() / sr)
    worst_k = max(worst_k, dk)
    if dk > 2e-6 or not chain_equal:
        print(f"MISMATCH {robot} n {n}: kin ops {dk:.2e} chain bit-exact {chain_equal}", flush=True)
    if dp > 2e-6 or dr > 2e-6 or dz > 2e-6:
        print(f"MISMATCH {robot} n {n} amp {amp}: pos {dp:.2e} rot {dr:.2e} min-z {dz:.2e}", flush=True)
    worst_p, worst_r = max(worst_p, dp, dz), max(worst_r, dr)
    runs += 1; frames += n
    if time.time() - last > 30:
        print(f"{runs} runs, {frames} frames, worst pos {worst_p:.2e} rot {worst_r:.2e} (relative to the largest magnitude), {time.time() - t0:.0f} s", flush=True); last = time.time()
print(f"fk fuzz done: {runs} runs, {frames} frames, {len(cache)} robots, worst relative difference: positions / min-z {worst_p:.2e}, rotations {worst_r:.2e}, dof_to_rot / rot_to_dof {worst_k:.2e} (bound 2e-6); local_rot_to_global bit-exact in every run")
