"""BASELINE.json configs 3-5 at their full sizes, checked through size-independent properties.

The oracle needs seconds per few thousand frames, so at these sizes the GPU result is pinned by
  * a bounded oracle sample (first frames of a few clips): exact parity as in test_gpu_parity;
  * determinism (two launches are bitwise equal);
  * clip independence (a clip solved alone == its rows in the batch, bitwise: no state leaks between wavefronts);
  * the box constraint as an invariant of the path: every hinge stays inside its range (mink ConfigurationLimit, gain 0.95,
    can approach a bound but never cross it);
  * solve counts inside [2, 22] (two stages of 1 + at most 10 solves) and no QP iteration cap hit;
  * exactly reachable inputs are tracked: the stage errors at the solved configuration stay small.
Inputs are synthetic in the shape the configs name (no AMASS / LAFAN1 data exists offline).
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from gmr_amd import synth  # noqa: E402
from gmr_amd.schedule import make_items  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402
from tests.util import compiled  # noqa: E402


def _engine(cm):
    from gmr_amd.engine import Engine
    return Engine(cm, 0)


def _hinge_limits(cm):
    r = cm.robot
    hb = sorted(r.hinge_bodies(), key=lambda b: r.qpos_adr[b])  # hinge order of qpos[7:]
    lim = np.array(r.jnt_range, dtype=np.float64)[hb]
    return lim[:, 0], lim[:, 1]


def _check_invariants(cm, q, iters, n_frames):
    it = (iters & 0x3FFFFFFF)
    assert int((iters >> 30).sum().item()) == 0, "a QP hit its iteration cap"
    assert int(it.min().item()) >= 2 and int(it.max().item()) <= 22
    assert q.shape == (n_frames, cm.robot.nq) and bool(torch.isfinite(q).all().item())
    lo, hi = _hinge_limits(cm)
    lo_t, hi_t = torch.from_numpy(lo).to(q.device), torch.from_numpy(hi).to(q.device)
    hinges = q[:, 7:]
    assert float((lo_t - hinges).max().item()) <= 1e-9 and float((hinges - hi_t).max().item()) <= 1e-9
    assert float((q[:, 3:7].norm(dim=1) - 1.0).abs().max().item()) < 1e-12


def _tile_variable(pos, quat, base_T, lengths, rng):
    """Clips of the given lengths cut from randomly chosen base clips (each a prefix, so frame 0 is a valid start)."""
    nb = pos.shape[0] // base_T
    idx = []
    for L in lengths:
        b = int(rng.integers(nb))
        idx.append(np.arange(b * base_T, b * base_T + L))
    idx = np.concatenate(idx)
    offs = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    return idx, offs


def test_config3_lafan_sized_bvh_set():
    """LAFAN1-shaped: 77 clips of 2000..9000 frames through bvh_to_g1.json (scripts/bvh_to_robot_dataset.py path)."""
    cm = compiled("bvh", "unitree_g1")
    eng = _engine(cm)
    rng = np.random.default_rng(3)
    base_T, n_base = 9000, 8
    pos, quat, names, _, _ = synth.synth_clips(cm, n_base, base_T, seed=33, hard=False, dtype=np.float32)
    hpos, hquat, _, _, _ = synth.synth_clips(cm, n_base, base_T, seed=34, hard=True, dtype=np.float32)
    pos, quat = np.concatenate([pos, hpos]), np.concatenate([quat, hquat])  # bases 0..7 reachable, 8..15 noisy / over-reach
    lengths = rng.integers(2000, 9001, size=77)
    idx, offs = _tile_variable(pos, quat, base_T, lengths, rng)
    N = int(offs[-1])
    assert 3.0e5 < N < 6.0e5
    dev = eng.device
    tp, tq = torch.from_numpy(pos).to(dev)[torch.from_numpy(idx).to(dev)], torch.from_numpy(quat).to(dev)[torch.from_numpy(idx).to(dev)]
    sc = cm.slot_columns(names)
    q, it, _ = eng.ik_solve(tp, tq, sc, make_items(offs))
    _check_invariants(cm, q, it, N)
    q2, it2, _ = eng.ik_solve(tp, tq, sc, make_items(offs))
    assert torch.equal(q, q2) and torch.equal(it, it2)  # deterministic
    for c in (0, 38, 76):  # clip independence, bitwise
        a, b = int(offs[c]), int(offs[c + 1])
        qc, _, _ = eng.ik_solve(tp[a:b].contiguous(), tq[a:b].contiguous(), sc, make_items([0, b - a]))
        assert torch.equal(qc, q[a:b])
    orc = Oracle(cm.blob)
    for c in (1, 40):  # bounded oracle sample
        a = int(offs[c])
        q_ref, it_ref, _ = orc.ik_solve(tp[a:a + 150].cpu().numpy(), tq[a:a + 150].cpu().numpy(), sc, make_items([0, 150]))
        assert np.abs(q[a:a + 150].cpu().numpy() - q_ref).max() < 1e-6
        assert np.array_equal((it[a:a + 150] & 0x3FFFFFFF).cpu().numpy(), it_ref)


def test_config4_five_robots_concurrently_full_size():
    """5 robots x 64 clips x 1000 frames, one model handle and one HIP stream per robot, all in flight together."""
    robots = ["unitree_g1", "booster_t1", "stanford_toddy", "fourier_n1", "engineai_pm01"]
    jobs = []
    for r in robots:
        cm = compiled("smplx", r)
        eng = _engine(cm)
        pos, quat, names, offs8, _ = synth.synth_clips(cm, 8, 1000, seed=41, hard=False, dtype=np.float32)
        rep = lambda a: torch.from_numpy(a).to(eng.device).repeat(8, 1, 1)  # 64 clips from 8 distinct
        offs = np.arange(65, dtype=np.int64) * 1000
        jobs.append((cm, eng, rep(pos), rep(quat), names, offs, torch.cuda.Stream(eng.device)))
    torch.cuda.synchronize()
    outs = []
    for cm, eng, tp, tq, names, offs, st in jobs:
        with torch.cuda.stream(st):
            outs.append(eng.ik_solve(tp, tq, cm.slot_columns(names), make_items(offs)))
    torch.cuda.synchronize()
    for (cm, eng, tp, tq, names, offs, _), (q, it, _) in zip(jobs, outs):
        _check_invariants(cm, q, it, 64000)
        assert torch.equal(q[:8000], q[8000:16000]) and torch.equal(q[:8000], q[56000:])  # identical clips, identical rows
        # exactly reachable inputs are tracked: stage errors at the solved configuration (settled part of each clip)
        e, _, _ = eng.evaluate(q[:8000], tp[:8000], tq[:8000], cm.slot_columns(names))
        settled = torch.cat([e[c * 1000 + 30:(c + 1) * 1000] for c in range(8)])
        assert float(settled.max().item()) < 5e-2, float(settled.max().item())
        q_ref, it_ref, _ = Oracle(cm.blob).ik_solve(tp[:120].cpu().numpy(), tq[:120].cpu().numpy(), cm.slot_columns(names), make_items([0, 120]))
        assert np.abs(q[:120].cpu().numpy() - q_ref).max() < 1e-6


def test_config4_five_robots_in_one_launch():
    """BASELINE config 4 as worded: heterogeneous trees in ONE launch.  EngineGroup builds the five robots for a common kernel variant
    (NVP 36, structured QP) and gmr_group_ik_solve runs 5 x 64 clips x 1000 frames as one grid of 320 wavefronts, each looking up
    its own model, LDS layout and arrays.  Every member's result must be bitwise what its own single-model launch gives (members
    rebuilt for a larger NVP included: padding rows do not change the arithmetic), and the oracle's to 1e-6."""
    from gmr_amd.engine import EngineGroup
    robots = ["unitree_g1", "booster_t1", "stanford_toddy", "fourier_n1", "engineai_pm01"]
    cms = [compiled("smplx", r) for r in robots]
    grp = EngineGroup(cms, 0)
    assert len({e.info.nv_padded for e in grp.engines}) == 1 and grp.engines[0].info.nv_padded == 36
    dev = grp.device
    batches, offs = [], np.arange(65, dtype=np.int64) * 1000
    for cm in cms:
        pos, quat, names, _, _ = synth.synth_clips(cm, 8, 1000, seed=41, hard=True, dtype=np.float32)
        rep = lambda a: torch.from_numpy(a).to(dev).repeat(8, 1, 1)  # noqa: E731
        batches.append((rep(pos), rep(quat), cm.slot_columns(names), make_items(offs)))
    outs = grp.ik_solve(batches)
    torch.cuda.synchronize()
    for cm, eng, b, (q, it) in zip(cms, grp.engines, batches, outs):
        _check_invariants(cm, q, it, 64000)
        q1, it1, _ = _engine(cm).ik_solve(b[0], b[1], b[2], b[3])            # the member alone, built for its own variant
        assert torch.equal(q, q1) and torch.equal(it, it1)
        q2, it2, _ = eng.ik_solve(b[0], b[1], b[2], b[3])                     # the group's handle, single-model launch
        assert torch.equal(q, q2) and torch.equal(it, it2)
        q_ref, it_ref, _ = Oracle(cm.blob).ik_solve(b[0][:150].cpu().numpy(), b[1][:150].cpu().numpy(), b[2], make_items([0, 150]))
        assert np.abs(q[:150].cpu().numpy() - q_ref).max() < 1e-6 and np.array_equal(it[:150].cpu().numpy() & 0x3FFFFFFF, it_ref)
    # a member without work, and unequal batch sizes
    small = [None, (batches[1][0][:3000], batches[1][1][:3000], batches[1][2], make_items(offs[:4])), None, None,
             (batches[4][0][:1000], batches[4][1][:1000], batches[4][2], make_items(offs[:2]))]
    o2 = grp.ik_solve(small)
    assert o2[0] == (None, None) and torch.equal(o2[1][0], outs[1][0][:3000]) and torch.equal(o2[4][0], outs[4][0][:1000])
    grp.close()


def test_config5_hands_model_full_size():
    """unitree_g1_with_hands (43 hinges, deepest tree) over 2048 clips x 3000 frames = 6.1 M frames in one launch."""
    cm = compiled("smplx", "unitree_g1_with_hands")
    eng = _engine(cm)
    S, T, D = 2048, 3000, 16
    pos, quat, names, _, _ = synth.synth_clips(cm, D // 2, T, seed=51, hard=False, dtype=np.float32)
    hpos, hquat, _, _, _ = synth.synth_clips(cm, D // 2, T, seed=52, hard=True, dtype=np.float32)
    dev = eng.device
    tp = torch.from_numpy(np.concatenate([pos, hpos])).to(dev).repeat(S // D, 1, 1)
    tq = torch.from_numpy(np.concatenate([quat, hquat])).to(dev).repeat(S // D, 1, 1)
    offs = np.arange(S + 1, dtype=np.int64) * T
    sc = cm.slot_columns(names)
    q, it, _ = eng.ik_solve(tp, tq, sc, make_items(offs))
    torch.cuda.synchronize()
    _check_invariants(cm, q, it, S * T)
    # the 14 hand hinges have no task below them: exactly zero, always (reference quirk 8)
    g1 = compiled("smplx", "unitree_g1")  # only for its body names (the two MJCFs differ in link geometry, so motions do too)
    hb = sorted(cm.robot.hinge_bodies(), key=lambda b: cm.robot.qpos_adr[b])
    hand = [i for i, b in enumerate(hb) if cm.robot.body_names[b] not in set(g1.robot.body_names)]
    assert len(hand) == 14 and float(q[:, 7:][:, hand].abs().max().item()) == 0.0
    # tiling: every repetition of the 16 distinct clips gives the same rows
    block = D * T
    assert torch.equal(q[:block], q[block:2 * block]) and torch.equal(q[:block], q[(S // D - 1) * block:])
    q_ref, it_ref, _ = Oracle(cm.blob).ik_solve(tp[8 * T:8 * T + 150].cpu().numpy(), tq[8 * T:8 * T + 150].cpu().numpy(), sc, make_items([0, 150]))
    assert np.abs(q[8 * T:8 * T + 150].cpu().numpy() - q_ref).max() < 1e-6


def test_long_clips_any_heading_chunked_equals_sequential():
    """Clips that start at ANY heading: some leave the sequential run in a wound-up IK basin for their whole length, so
    the speculative chunks of those clips verify nowhere and the walk re-solves them -- the chunked result must still be the
    sequential one, values and solve counts, and clips in the natural basin must verify (few re-solved frames overall is NOT
    expected here).  Inputs from the GPU generator (synth_clips_torch), bvh_to_g1 config, per-clip heights on top."""
    cm = compiled("bvh", "unitree_g1")
    eng = _engine(cm)
    lens = [700, 450, 900, 600, 820, 510]
    pos, quat, names, offs = synth.synth_clips_torch(cm, lens, seed=33, device=eng.device, hard=[False, True] * 3, yaw0=np.pi)
    sc = cm.slot_columns(names)
    hs = [1.0, 0.95, 1.05, 1.0, 0.9, 1.1]
    q_seq, it_seq, _ = eng.ik_solve(pos, quat, sc, make_items(offs, height_scales=hs))
    q_chk, it_chk, info = eng.ik_solve_chunked(pos, quat, sc, offs, chunk=32, burn_in=24, height_scales=hs)
    assert float((q_chk - q_seq).abs().max().item()) < 1e-6 and torch.equal(it_chk & 0x3FFFFFFF, it_seq & 0x3FFFFFFF)
    assert 0 <= info["resolved_frames"] <= sum(lens)
    q_ref, it_ref, _ = Oracle(cm.blob).ik_solve(pos[: lens[0]].cpu().numpy(), quat[: lens[0]].cpu().numpy(), sc, make_items([0, lens[0]], height_scales=hs[:1]))
    assert np.abs(q_chk[: lens[0]].cpu().numpy() - q_ref).max() < 1e-6 and np.array_equal((it_chk[: lens[0]] & 0x3FFFFFFF).cpu().numpy(), it_ref)


def test_clip_start_on_root_target_is_opt_in():
    """retarget_batch(clip_start="root_target"): clips start with the base on their first root target (NOT the reference's qpos0
    start; opt-in).  Same as the oracle's INIT_ROOT_TARGET items; on clips that start facing away from qpos0 it spares the slow
    start-up (fewer solves), and every speculative chunk verifies."""
    from gmr_amd import GeneralMotionRetargeting as GMR
    from gmr_amd._native import INIT_ROOT_TARGET
    g = GMR("bvh", "unitree_g1")
    cm = g._cm
    lens = [400, 300, 500]
    pos, quat, names, offs = synth.synth_clips_torch(cm, lens, seed=33, device=g.device, hard=False, yaw0=np.pi)
    q_ref_start, it_ref_start = g.retarget_batch(pos, quat, names, seq_offsets=offs, return_iters=True)
    q_rt, it_rt = g.retarget_batch(pos, quat, names, seq_offsets=offs, return_iters=True, clip_start="root_target")
    sc = cm.slot_columns(names)
    items = make_items(offs, clip_init=INIT_ROOT_TARGET)
    q_o, it_o, _ = Oracle(cm.blob).ik_solve(pos.cpu().numpy(), quat.cpu().numpy(), sc, items)
    assert np.abs(q_rt.cpu().numpy() - q_o).max() < 1e-6 and np.array_equal((it_rt & 0x3FFFFFFF).cpu().numpy(), it_o)
    assert float((it_rt & 0x3FFFFFFF).float().mean()) <= float((it_ref_start & 0x3FFFFFFF).float().mean())
    q_c = g.retarget_batch(pos, quat, names, seq_offsets=offs, chunk=32, burn_in=24, clip_start="root_target")
    assert float((q_c - q_rt).abs().max().item()) < 1e-6 and g.last_chunk_info["resolved_frames"] <= 64
    with pytest.raises(ValueError):
        g.retarget_batch(pos, quat, names, clip_start="nowhere")


def test_chunked_solve_is_blind_to_target_quaternion_signs():
    """Key-points from files carry quaternions of either sign: the verified-chunked solve must neither re-solve the chunks whose
    start took the other sign (the wall it hit in round 3: half of a BVH folder re-solved) nor hand back base quaternions of the
    chunk's sign -- the result is the sequential run, bit for bit where chunks were adopted with the same sign, negated rows turned."""
    cm = compiled("bvh", "unitree_g1")
    eng = _engine(cm)
    lengths = np.array([900, 1500, 700])
    pos, quat, names, offs = synth.synth_clips_torch(cm, lengths, seed=9, device=eng.device, hard=np.array([False, True, False]), yaw0=1.0, dtype=torch.float64)
    sc = cm.slot_columns(names)
    flip = torch.from_numpy(np.random.default_rng(1).choice([-1.0, 1.0], size=tuple(quat.shape[:2]))).to(eng.device)
    quat_s = quat * flip[..., None]
    q_seq, it_seq, _ = eng.ik_solve(pos, quat_s, sc, make_items(offs))
    q_ref, _, _ = eng.ik_solve(pos, quat, sc, make_items(offs))
    assert torch.equal(q_seq, q_ref)                                       # the sequential solve does not see the signs
    q_c, it_c, info_c = eng.ik_solve_chunked(pos, quat, sc, offs, chunk=64, burn_in=32)
    q_s, it_s, info_s = eng.ik_solve_chunked(pos, quat_s, sc, offs, chunk=64, burn_in=32)
    assert info_s["resolved_frames"] <= info_c["resolved_frames"] + 64 and info_s["resolved_frames"] < 0.1 * int(offs[-1])
    assert float((q_s - q_seq).abs().max().item()) < 1e-6 and torch.equal(it_s & 0x3FFFFFFF, it_seq & 0x3FFFFFFF)   # signs of qpos[3:7] included
    # a bounded oracle sample of the scrambled set
    e = int(offs[1])
    q_o, it_o, _ = Oracle(cm.blob).ik_solve(pos[:e].cpu().numpy(), quat_s[:e].cpu().numpy(), sc, make_items(offs[:2]))
    assert np.abs(q_s[:e].cpu().numpy() - q_o).max() < 1e-6
