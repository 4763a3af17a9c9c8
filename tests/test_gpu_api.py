"""GPU tests of the drop-in API (GeneralMotionRetargeting / KinematicsModel) and the scheduling options."""
import json
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from gmr_amd import synth  # noqa: E402
from gmr_amd.schedule import make_items  # noqa: E402
from oracle.oracle import IKParams as OParams, Oracle  # noqa: E402
from tests.util import compiled, quat_angle  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _frames(pos, quat, names, f):
    return {n: (pos[f, i].astype(np.float64), quat[f, i].astype(np.float64)) for i, n in enumerate(names)}


def test_retarget_per_frame_is_stateful_and_matches_batch_and_oracle():
    from gmr_amd import GeneralMotionRetargeting as GMR
    g = GMR(src_human="smplx", tgt_robot="unitree_g1", actual_human_height=1.7)
    cm = g._cm
    assert abs(cm.ratio - 1.7 / 1.8) < 1e-15 and g.xml_file.endswith((".xml", ".json"))
    pos, quat, names, offs, _ = synth.synth_clips(cm, 1, 12, seed=5, hard=True, dtype=np.float32, pad_to=55)
    assert pos.shape[1] == 55
    q_ref, _, _ = Oracle(cm.blob).ik_solve(pos, quat, cm.slot_columns(names), make_items(offs))
    qs = []
    for f in range(12):
        d = _frames(pos, quat, names, f)
        d[names[0]] = (list(d[names[0]][0]), list(d[names[0]][1]))  # lists are converted in place like the reference
        q = g.retarget(d)
        assert isinstance(d[names[0]][0], np.ndarray) and q.shape == (36,) and q.dtype == np.float64
        q[:] = 0  # caller owns the returned copy
        qs.append(g.configuration.data.qpos.copy())
    qs = np.array(qs)
    assert np.abs(qs - q_ref).max() < 1e-6
    qb = g.retarget_batch(pos, quat, names)
    assert np.abs(qb - q_ref).max() < 1e-6
    g.setup_retarget_configuration()  # fresh state -> first frame reproduces
    assert np.abs(g.retarget(_frames(pos, quat, names, 0)) - q_ref[0]).max() < 1e-6
    sd = g.scaled_human_data
    tp, tq = Oracle(cm.blob).prepare_targets(pos[0][cm.slot_columns(names)].astype(np.float64), quat[0][cm.slot_columns(names)].astype(np.float64))
    for s, n in enumerate(cm.slot_names):
        assert np.abs(sd[n][0] - tp[s]).max() < 1e-12 and min(np.abs(sd[n][1] - tq[s]).max(), np.abs(sd[n][1] + tq[s]).max()) < 1e-12
    assert set(sd.keys()) == set(cm.slot_names)
    with pytest.raises(KeyError):
        bad = _frames(pos, quat, names, 0)
        del bad["pelvis"]
        g.retarget(bad)
    with pytest.raises(KeyError):
        GMR("smplx", "no_such_robot")


def test_offset_to_ground_f64_inputs_and_bvh_config():
    from gmr_amd import GeneralMotionRetargeting as GMR
    g = GMR("bvh", "unitree_g1")
    cm = g._cm
    pos, quat, names, offs, _ = synth.synth_clips(cm, 2, 15, seed=8, hard=True, dtype=np.float64)
    prm = OParams(offset_to_ground=1)
    q_ref, it_ref, _ = Oracle(cm.blob).ik_solve(pos, quat, cm.slot_columns(names), make_items(offs), params=prm)
    q, it = g.retarget_batch(pos, quat, names, seq_offsets=offs, offset_to_ground=True, return_iters=True)
    assert np.abs(q - q_ref).max() < 1e-6 and np.array_equal(it & 0x3FFFFFFF, it_ref)
    q0 = g.retarget_batch(pos, quat, names, seq_offsets=offs)
    assert np.abs(q0 - q_ref).max() > 1e-3  # the option does something


def test_full_length_clip_matches_oracle_and_chunking_residual():
    """BASELINE config 2 size: one 3000-frame clip.  Sequential GPU == sequential oracle; chunked GPU == chunked oracle;
    the chunked-vs-sequential residual (an approximation, see schedule.py) is measured and recorded, not assumed."""
    from gmr_amd.engine import Engine
    cm = compiled("smplx", "unitree_g1")
    eng, orc = Engine(cm, 0), Oracle(cm.blob)
    dev = eng.device
    rec = {}
    for hard in (False, True):
        pos, quat, names, offs, _ = synth.synth_clips(cm, 1, 3000, seed=31, hard=hard, dtype=np.float32)
        sc = cm.slot_columns(names)
        tp, tq = torch.from_numpy(pos).to(dev), torch.from_numpy(quat).to(dev)
        q_seq_ref, it_ref, _ = orc.ik_solve(pos, quat, sc, make_items(offs))
        q_seq, it, _ = eng.ik_solve(tp, tq, sc, make_items(offs))
        q_seq = q_seq.cpu().numpy()
        assert np.abs(q_seq - q_seq_ref).max() < 1e-6
        assert (it.cpu().numpy() != it_ref).sum() == 0
        for chunk, burn in ((16, 16), (16, 32), (16, 64), (8, 128)):
            items = make_items(offs, chunk=chunk, burn_in=burn)
            q_c, _, _ = eng.ik_solve(tp, tq, sc, items)
            q_c = q_c.cpu().numpy()
            assert not np.isnan(q_c).any()
            if (chunk, burn) == (16, 32):
                q_c_ref, _, _ = orc.ik_solve(pos, quat, sc, items, n_threads=8)
                assert np.abs(q_c - q_c_ref).max() < 1e-6
            d = np.abs(q_c - q_seq)
            rec[f"{'hard' if hard else 'easy'}_chunk{chunk}_burn{burn}"] = {
                "max_abs": float(d.max()), "p999": float(np.quantile(d.max(axis=1), 0.999)), "frames_over_1e-3": int((d.max(axis=1) > 1e-3).sum())}
        # verified parallel-in-time: same chunks, boundaries checked against the predecessor and repaired
        import time
        for chunk, burn in ((16, 32), (8, 24), (32, 32)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            q_v, it_v, info = eng.ik_solve_chunked(tp, tq, sc, offs, chunk=chunk, burn_in=burn)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            dv = np.abs(q_v.cpu().numpy() - q_seq)
            assert dv.max() < 1e-6, (chunk, burn, dv.max(), info)
            rec[f"{'hard' if hard else 'easy'}_verified_chunk{chunk}_burn{burn}"] = dict(info, max_abs=float(dv.max()), seconds=dt,
                                                                                     frames_per_s=3000 / dt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.ik_solve(tp, tq, sc, make_items(offs))
        torch.cuda.synchronize()
        rec[f"{'hard' if hard else 'easy'}_sequential_seconds"] = time.perf_counter() - t0
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "chunk_residual.json"), "w") as f:
        json.dump(rec, f, indent=1)


def test_kinematics_model_mirror(golden_dir):
    from gmr_amd import KinematicsModel
    from gmr_amd import params
    km = KinematicsModel(str(params.ROBOT_XML_DICT["unitree_g1"]), device="cuda:0")
    with open(os.path.join(golden_dir, "tree_unitree_g1.json")) as f:
        ref = json.load(f)
    assert km.body_names == ref["body_names"] and km.joint_dof_idx == ref["joint_dof_idx"] and km.num_dof == 29 and km.num_joint == 38
    assert km.parent_indices.tolist() == ref["parent_indices"]
    lo, hi = km.get_dof_limits()
    assert lo.dtype == torch.float32 and np.allclose(lo.cpu().numpy(), np.array(ref["lower"], np.float32))
    g = np.load(os.path.join(golden_dir, "fk_unitree_g1.npz"))
    bp, br = km.forward_kinematics(torch.from_numpy(g["root_pos"]), torch.from_numpy(g["root_rot"]), torch.from_numpy(g["dof_pos"]))
    assert bp.shape == (64, 38, 3) and br.shape == (64, 38, 4) and bp.is_cuda
    assert np.abs(bp.cpu().numpy() - g["body_pos"]).max() < 1e-5 and np.abs(br.cpu().numpy() - g["body_rot"]).max() < 2e-6
    bp2, _ = km.forward_kinematics(torch.from_numpy(g["root_pos"]).reshape(8, 8, 3), torch.from_numpy(g["root_rot"]).reshape(8, 8, 4),
                                   torch.from_numpy(g["dof_pos"]).reshape(8, 8, 29))
    assert bp2.shape == (8, 8, 38, 3) and torch.equal(bp2.reshape(64, 38, 3), bp)
    with pytest.raises(RuntimeError):
        KinematicsModel(str(params.ROBOT_XML_DICT["unitree_g1"]), device="cpu")


def test_fk_batch_on_the_retargeter_matches_kinematics_model(golden_dir):
    """GeneralMotionRetargeting.fk_batch (batched surface) == the KinematicsModel mirror == the reference golden."""
    from gmr_amd import GeneralMotionRetargeting as GMR
    g = GMR("smplx", "unitree_g1")
    gold = np.load(os.path.join(golden_dir, "fk_unitree_g1.npz"))
    bp = g.fk_batch(gold["root_pos"], gold["root_rot"], gold["dof_pos"])
    assert isinstance(bp, np.ndarray) and np.abs(bp - gold["body_pos"]).max() < 2e-6 * max(1.0, np.abs(gold["body_pos"]).max())
    bp_t, br_t = g.fk_batch(torch.from_numpy(gold["root_pos"]).cuda(), torch.from_numpy(gold["root_rot"]).cuda(),
                            torch.from_numpy(gold["dof_pos"]).cuda(), want_rot=True)
    assert bp_t.is_cuda and np.abs(br_t.cpu().numpy() - gold["body_rot"]).max() < 2e-6


def test_fk_large_and_min_height():
    from gmr_amd.engine import Engine
    cm = compiled("smplx", "unitree_g1_with_hands")
    eng, orc = Engine(cm, 0), Oracle(cm.blob)
    rng = np.random.default_rng(0)
    T = 100_003  # not a multiple of the block size
    lo, hi = cm.robot.dof_limits()
    dof = (lo + rng.uniform(0, 1, (T, lo.size)) * (hi - lo)).astype(np.float32)
    rp = rng.normal(0, 1, (T, 3)).astype(np.float32)
    rq = rng.normal(size=(T, 4))
    rq = (rq / np.linalg.norm(rq, axis=1, keepdims=True)).astype(np.float32)
    d = eng.device
    bp, br = eng.fk(torch.from_numpy(rp).to(d), torch.from_numpy(rq).to(d), torch.from_numpy(dof).to(d))
    idx = np.r_[0:500, T - 500:T]
    bp_ref, br_ref = orc.fk_kin(rp[idx], rq[idx], dof[idx])
    assert np.abs(bp.cpu().numpy()[idx] - bp_ref).max() < 5e-6 and np.abs(br.cpu().numpy()[idx] - br_ref).max() < 2e-6
    offs = np.array([0, 1, 1000, 1000, 70_000, T], dtype=np.int64)
    mz = eng.fk_min_height(torch.from_numpy(rp).to(d), torch.from_numpy(rq).to(d), torch.from_numpy(dof).to(d), offs).cpu().numpy()
    z = bp.cpu().numpy()[:, :, 2]
    for s in range(len(offs) - 1):
        if offs[s + 1] > offs[s]:
            assert mz[s] == z[offs[s]:offs[s + 1]].min()
        else:
            assert np.isinf(mz[s])


def test_bad_arguments_are_rejected():
    from gmr_amd.engine import Engine, EngineError
    cm = compiled("smplx", "unitree_g1")
    eng = Engine(cm, 0)
    pos = torch.zeros((4, 14, 3), device=eng.device)
    quat = torch.zeros((4, 14, 4), device=eng.device)
    quat[..., 0] = 1
    sc = np.arange(14, dtype=np.int32)
    with pytest.raises(EngineError):
        eng.ik_solve(pos, quat, sc, make_items([0, 9]))          # item beyond the data
    bad = sc.copy()
    bad[3] = 14
    with pytest.raises(EngineError):
        eng.ik_solve(pos, quat, bad, make_items([0, 4]))         # column out of range
    with pytest.raises(EngineError):
        eng.ik_solve(pos.cpu(), quat.cpu(), sc, make_items([0, 4]))
    fk_only = Engine(__import__("gmr_amd.model", fromlist=["compile_model"]).compile_model(cm.robot, None), 0)
    with pytest.raises(EngineError):
        fk_only.ik_solve(pos, quat, np.zeros(0, np.int32), make_items([0, 4]))
    out, it, _ = eng.ik_solve(pos, quat, sc, make_items([0, 0]))  # empty work: nothing written
    assert torch.isnan(out).all()
    import ctypes
    prm = __import__("gmr_amd._native", fromlist=["IKParams"]).IKParams()
    rc = eng._lib.gmr_ik_solve(eng._h, None, None, 0, 14, sc.ctypes.data_as(ctypes.c_void_p), 4, None, 0, ctypes.byref(prm), None, None, None, None, None, None, None)
    assert rc == -1 and b"null" in eng._lib.gmr_last_error(eng._h)  # the C ABI itself rejects null buffers


def test_dataset_path_matches_reference_postprocessing(tmp_path):
    """retarget_clips == (oracle IK -> reference-convention FK -> the arithmetic of smplx_to_robot_dataset.py:97-141)."""
    from gmr_amd import GeneralMotionRetargeting as GMR
    from gmr_amd import dataset
    g = GMR("smplx", "unitree_g1")
    cm = g._cm
    pos, quat, names, offs, _ = synth.synth_clips(cm, 3, 50, seed=3, hard=True, dtype=np.float32)
    offs = np.array([0, 50, 50, 110, 150])  # ragged, with one empty clip
    motions = dataset.retarget_clips(g, pos, quat, names, offs, fps=[30, 30, 60, 30])
    orc = Oracle(cm.blob)
    q_ref, _, _ = orc.ik_solve(pos, quat, cm.slot_columns(names), make_items(offs))
    assert len(motions) == 4 and motions[1]["root_pos"].shape == (0, 3)
    for s, mo in enumerate(motions):
        a, b = offs[s], offs[s + 1]
        q = q_ref[a:b]
        dataset.validate_motion(mo, nq=36)
        assert mo["fps"] == [30, 30, 60, 30][s] and mo["link_body_list"] == cm.robot.body_names
        if b == a:
            continue
        root_rot = q[:, [4, 5, 6, 3]]
        ident = np.tile(np.array([[0, 0, 0, 1]], np.float32), (b - a, 1))
        local, _ = orc.fk_kin(np.zeros((b - a, 3), np.float32), ident, q[:, 7:].astype(np.float32), want_rot=False)
        body, _ = orc.fk_kin(q[:, :3].astype(np.float32), root_rot.astype(np.float32), q[:, 7:].astype(np.float32), want_rot=False)
        root_pos = q[:, :3].copy()
        root_pos[:, 2] -= float(body[..., 2].min())
        root_pos[:, :2] -= root_pos[0, :2]
        assert mo["root_pos"].dtype == np.float64 and mo["local_body_pos"].dtype == np.float32
        assert np.abs(mo["root_pos"] - root_pos).max() < 5e-6   # float32 FK feeds the height
        assert np.abs(mo["root_rot"] - root_rot).max() < 1e-6 and np.abs(mo["dof_pos"] - q[:, 7:]).max() < 1e-6
        assert np.abs(mo["local_body_pos"] - local).max() < 5e-6
        assert abs(mo["root_pos"][0, 0]) < 1e-12 and abs(mo["root_pos"][0, 1]) < 1e-12
    p = tmp_path / "clip.pkl"
    assert dataset.save_motion(str(p), motions[0]) and not dataset.save_motion(str(p), motions[0])
    d, fps, rp, rr_wxyz, dp, lb, names_out = dataset.load_robot_motion(str(p))
    assert fps == 30 and np.array_equal(rr_wxyz[:, [1, 2, 3, 0]], motions[0]["root_rot"]) and names_out == cm.robot.body_names
    bvh = dataset.retarget_clips(g, pos[:50], quat[:50], names, [0, 50], height_adjust=False, root_origin_offset=False)
    assert np.abs(bvh[0]["root_pos"] - q_ref[:50, :3]).max() < 1e-6


@pytest.mark.parametrize("name", ["bvh_canonical_40f", "bvh_lafan_like", "bvh_pruned_mid_24f", "bvh_nine_channel"] + [f"bvh_random_{k}" for k in range(6)])
def test_bvh_adapter_matches_reference_loader(name, golden_dir):
    """gmr_amd.bvh.load_lafan1_file (host parse + gmr_bvh_fk) vs the reference's load_lafan1_file output (golden).  bvh_random_<k>
    (tests/golden/make_bvh_golden_random.py): random trees in the Euler orders XYZ, YZX, ZXY, XZY, YXZ, ZYX and all three row layouts, with
    and without the bone names the loader treats specially."""
    from gmr_amd.bvh import load_lafan1_file
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    clip = load_lafan1_file(os.path.join(golden_dir, name + ".bvh"))
    assert clip.body_names == [str(n) for n in g["names"]]
    assert abs(clip.human_height - float(g["human_height"])) < 1e-9
    pos, quat = clip.pos.cpu().numpy(), clip.quat.cpu().numpy()
    assert np.abs(pos - g["pos"]).max() < 1e-9
    # the reference removes sign flips along time on the local quaternions; rotations are identical up to sign
    d = np.minimum(np.abs(quat - g["quat"]).max(axis=-1), np.abs(quat + g["quat"]).max(axis=-1))
    assert d.max() < 1e-9
    fr = clip.frames()
    assert len(fr) == len(clip) and set(fr[0].keys()) == set(clip.body_names) and fr[3]["Hips" if "Hips" in fr[0] else clip.body_names[0]][0].shape == (3,)


def test_bvh_to_robot_end_to_end(golden_dir):
    """bvh_to_robot_dataset.py path: BVH file -> GPU adapter -> batched IK (bvh_to_g1 config) == oracle on the golden poses."""
    from gmr_amd import GeneralMotionRetargeting as GMR
    from gmr_amd.bvh import load_lafan1_file
    clip = load_lafan1_file(os.path.join(golden_dir, "bvh_lafan_like.bvh"))
    g = GMR(src_human="bvh", tgt_robot="unitree_g1", actual_human_height=clip.human_height)
    q = g.retarget_batch(clip.pos, clip.quat, clip.body_names)
    gold = np.load(os.path.join(golden_dir, "bvh_lafan_like.npz"))
    cm = g._cm
    q_ref, _, _ = Oracle(cm.blob).ik_solve(gold["pos"], gold["quat"], cm.slot_columns([str(n) for n in gold["names"]]), make_items([0, len(clip)]))
    assert np.abs(q.cpu().numpy() - q_ref).max() < 1e-6


def test_error_accessors_and_xpos_match_oracle():
    """error1()/error2() (motion_retarget.py:188-200) and configuration.data.xpos after a retarget() call."""
    from gmr_amd import GeneralMotionRetargeting as GMR
    g = GMR("smplx", "engineai_pm01")
    cm = g._cm
    orc = Oracle(cm.blob)
    pos, quat, names, offs, _ = synth.synth_clips(cm, 1, 4, seed=13, hard=True, dtype=np.float64)
    with pytest.raises(RuntimeError):
        g.error1()
    for f in range(4):
        q = g.retarget(_frames(pos, quat, names, f))
    sc = cm.slot_columns(names)
    tp, tq = orc.prepare_targets(pos[3][sc], quat[3][sc])
    e1, _ = orc.stage_error(0, q, tp, tq, len(cm.tasks[0]))
    e2, _ = orc.stage_error(1, q, tp, tq, len(cm.tasks[1]))
    assert abs(g.error1() - e1) < 1e-9 and abs(g.error2() - e2) < 1e-9
    xp, xq = orc.fk_mj(q)
    # MuJoCo layout: row 0 is the world body, row model.body(name).id the named body
    assert g.configuration.data.xpos.shape == (cm.robot.nbody + 1, 3) and not g.configuration.data.xpos[0].any()
    assert np.array_equal(g.configuration.data.xquat[0], [1, 0, 0, 0])
    assert np.abs(g.configuration.data.xpos[1:] - xp).max() < 1e-12
    assert np.minimum(np.abs(g.configuration.data.xquat[1:] - xq), np.abs(g.configuration.data.xquat[1:] + xq)).max() < 1e-12
    # batched evaluation through the engine
    qb = g.retarget_batch(pos, quat, names)
    err, _, _ = g._engine.evaluate(torch.from_numpy(qb).to(g.device) if isinstance(qb, np.ndarray) else qb,
                                   torch.from_numpy(pos).to(g.device), torch.from_numpy(quat).to(g.device), sc)
    assert err.shape == (4, 2) and abs(float(err[3, 1]) - e2) < 1e-9


def test_five_robots_concurrently_on_streams():
    """BASELINE config 4: heterogeneous trees side by side -- one model handle and one HIP stream per robot."""
    from gmr_amd.engine import Engine
    robots = ["unitree_g1", "booster_t1", "stanford_toddy", "fourier_n1", "engineai_pm01"]
    jobs = []
    for r in robots:
        cm = compiled("smplx", r)
        pos, quat, names, offs, _ = synth.synth_clips(cm, 4, 25, seed=17, hard=True, dtype=np.float32)
        eng = Engine(cm, 0)
        jobs.append((cm, eng, torch.from_numpy(pos).to(eng.device), torch.from_numpy(quat).to(eng.device), pos, quat, names, offs,
                     torch.cuda.Stream(eng.device)))
    torch.cuda.synchronize()
    outs = []
    for cm, eng, tp, tq, _, _, names, offs, st in jobs:
        with torch.cuda.stream(st):
            outs.append(eng.ik_solve(tp, tq, cm.slot_columns(names), make_items(offs))[0])
    torch.cuda.synchronize()
    nvps = set()
    for (cm, eng, _, _, pos, quat, names, offs, _), q in zip(jobs, outs):
        q_ref, _, _ = Oracle(cm.blob).ik_solve(pos, quat, cm.slot_columns(names), make_items(offs))
        assert np.abs(q.cpu().numpy() - q_ref).max() < 1e-6
        nvps.add(eng.info.nv_padded)
    assert nvps == {32, 36}


def test_degenerate_batches():
    from gmr_amd import GeneralMotionRetargeting as GMR
    g = GMR("smplx", "unitree_g1")
    cm = g._cm
    pos, quat, names, offs, _ = synth.synth_clips(cm, 1, 3, seed=1, dtype=np.float32)
    empty = g.retarget_batch(pos[:0], quat[:0], names, seq_offsets=[0, 0])
    assert empty.shape == (0, 36)
    ones = g.retarget_batch(pos, quat, names, seq_offsets=[0, 1, 2, 3])  # three 1-frame clips: each starts from qpos0
    first = g.retarget_batch(pos[:1], quat[:1], names)
    assert np.abs(ones[0] - first[0]).max() < 1e-12
    g2 = GMR("smplx", "unitree_g1")
    assert np.abs(g2.retarget(_frames(pos, quat, names, 1)) - ones[1]).max() < 1e-9


def _smplx_restatement(global_orient, full_pose, joints, parents, src_fps, tgt_fps):
    """scipy restatement of reference utils/smpl.py:75-198 (the module itself needs the absent smplx package)."""
    from scipy.interpolate import interp1d
    from scipy.spatial.transform import Rotation as R

    def slerp(r1, r2, t):
        q1, q2 = r1.as_quat(), r2.as_quat()
        q1, q2 = q1 / np.linalg.norm(q1), q2 / np.linalg.norm(q2)
        dot = np.sum(q1 * q2)
        if dot < 0.0:
            q2, dot = -q2, -dot
        if dot > 0.9995:
            return R.from_quat(q1 + t * (q2 - q1))
        th0 = np.arccos(dot)
        th = th0 * t
        return R.from_quat((np.cos(th) - dot * np.sin(th) / np.sin(th0)) * q1 + np.sin(th) / np.sin(th0) * q2)

    T, J = full_pose.shape[0], len(parents)
    if tgt_fps < src_fps:
        n = T // int(src_fps / tgt_fps)
        tt = np.linspace(0, T - 1, n)
        go, fp = [], np.zeros((n, J, 3))
        for k, t in enumerate(tt):
            i1 = int(np.floor(t)); i2 = min(i1 + 1, T - 1); a = t - i1
            go.append(slerp(R.from_rotvec(global_orient[i1]), R.from_rotvec(global_orient[i2]), a).as_rotvec())
            for j in range(J):
                fp[k, j] = slerp(R.from_rotvec(full_pose[i1, j]), R.from_rotvec(full_pose[i2, j]), a).as_rotvec()
        go = np.stack(go)
        jt = np.stack([interp1d(np.arange(T), joints[:, i, c], kind="linear")(tt) for i in range(joints.shape[1]) for c in range(3)], axis=1).reshape(n, -1, 3)
        fps = n / T * src_fps
    else:
        go, fp, jt, fps = global_orient, full_pose, joints, tgt_fps
    pos, quat = np.zeros((len(go), J, 3)), np.zeros((len(go), J, 4))
    for k in range(len(go)):
        rots = []
        for i in range(J):
            rots.append(R.from_rotvec(go[k]) if i == 0 else rots[parents[i]] * R.from_rotvec(fp[k, i]))
            pos[k, i], quat[k, i] = jt[k, i], rots[i].as_quat(scalar_first=True)
    return pos, quat, fps


@pytest.mark.parametrize("src_fps", [120.0, 30.0])
def test_smplx_keypoint_adapter_matches_scipy_restatement(src_fps):
    """parity unpinned (smplx absent): GPU adapter vs a scipy restatement of smpl.py:75-198; then straight into the IK."""
    from gmr_amd import GeneralMotionRetargeting as GMR
    from gmr_amd.smplx_adapter import SMPLX_JOINT_NAMES, SMPLX_PARENTS, get_smplx_data_offline_fast
    rng = np.random.default_rng(0)
    T = 50
    base = rng.normal(0, 0.4, (1, 55, 3))
    full_pose = base + np.cumsum(rng.normal(0, 0.03, (T, 55, 3)), axis=0)
    full_pose[:, 7] = 0.0  # an exactly-zero rotation exercises the small-angle branch
    global_orient = full_pose[:, 0].copy()
    joints = np.cumsum(rng.normal(0, 0.01, (T, 127, 3)), axis=0) + rng.normal(0, 0.5, (1, 127, 3))
    pos, quat, names, fps = get_smplx_data_offline_fast(global_orient, full_pose.reshape(T, -1), joints, SMPLX_PARENTS, src_fps=src_fps)
    p_ref, q_ref, fps_ref = _smplx_restatement(global_orient, full_pose, joints, SMPLX_PARENTS, src_fps, 30.0)
    assert names == SMPLX_JOINT_NAMES and len(names) == 55 and abs(fps - fps_ref) < 1e-12
    assert pos.shape == p_ref.shape and np.abs(pos.cpu().numpy() - p_ref).max() < 1e-12
    d = np.minimum(np.abs(quat.cpu().numpy() - q_ref).max(-1), np.abs(quat.cpu().numpy() + q_ref).max(-1))
    assert d.max() < 1e-9
    g = GMR("smplx", "unitree_g1")
    q = g.retarget_batch(pos, quat, names)  # 55 columns, the config picks its 14
    assert q.shape == (pos.shape[0], 36) and bool(torch.isfinite(q).all())


def test_session_single_sequence_mode_matches_oracle():
    """gmr_session_*: the live per-frame path (scripts/optitrack_to_robot.py:37-46): state carried in the session, f32 and
    f64 inputs, reset / state accessors, argument checks, a layout switch mid-stream keeps the warm start."""
    import time
    from gmr_amd.engine import Engine, EngineError
    from gmr_amd._native import IKParams
    cm = compiled("smplx", "unitree_g1")
    eng = Engine(cm)
    T = 40
    pos, quat, names, offs, _ = synth.synth_clips(cm, 1, T, seed=11, hard=True, dtype=np.float32)
    sc = cm.slot_columns(names)
    orc = Oracle(cm.blob)
    q_ref, it_ref, _ = orc.ik_solve(pos, quat, sc, make_items(offs))
    s32 = eng.session(sc, pos.shape[1], IKParams(), dtype=np.float32)
    lat = []
    for f in range(T):
        t0 = time.perf_counter()
        q, n = s32.step(pos[f], quat[f])
        lat.append(time.perf_counter() - t0)
        assert np.abs(q - q_ref[f]).max() < 1e-6 and (n & 0x3FFFFFFF) == it_ref[f]
    print(f"session step latency: median {1e6 * np.median(lat):.0f} us, max {1e6 * np.max(lat):.0f} us")
    assert np.array_equal(s32.state(), q)
    # reset to qpos0 reproduces frame 0; reset to a given configuration continues from it
    s32.reset()
    assert np.abs(s32.step(pos[0], quat[0])[0] - q_ref[0]).max() < 1e-6
    s32.reset(q_ref[9])
    assert np.abs(s32.step(pos[10], quat[10])[0] - q_ref[10]).max() < 1e-6
    # float64 inputs, offset_to_ground per step
    s64 = eng.session(sc, pos.shape[1], IKParams(), dtype=np.float64)
    p64, q64 = pos.astype(np.float64), quat.astype(np.float64)
    qg_ref, _, _ = orc.ik_solve(p64[:3], q64[:3], sc, make_items([0, 3]), params=OParams(offset_to_ground=1))
    for f in range(3):
        assert np.abs(s64.step(p64[f], q64[f], offset_to_ground=True)[0] - qg_ref[f]).max() < 1e-6
    with pytest.raises(ValueError):
        s64.step(p64[0][:-1], q64[0])
    with pytest.raises(EngineError):
        eng.session(np.full_like(sc, 99), pos.shape[1])
    with pytest.raises(EngineError):
        eng.session(sc, pos.shape[1], IKParams(damping=0.0))
    s32.close(); s64.close()
    # the class API rides on sessions: switching the dict layout mid-stream keeps the warm start
    from gmr_amd import GeneralMotionRetargeting as GMR
    g = GMR("smplx", "unitree_g1", actual_human_height=1.8)
    sub = [n for n in names]
    for f in range(4):
        g.retarget(_frames(pos, quat, names, f))
    rev = list(reversed(sub))
    d = {n: (pos[4][names.index(n)], quat[4][names.index(n)]) for n in rev}
    assert np.abs(g.retarget(d) - q_ref[4]).max() < 1e-6 and len(g._sessions) == 2


def test_verification_walk_matches_oracle():
    """check_stride work items through the C ABI: same speculative chunks, same walk, same adopted / re-solved chunks
    (frames_done), same stored states and the sequential result at the end -- GPU against the oracle's restatement."""
    from gmr_amd.engine import Engine
    from gmr_amd._native import IKParams, WORK_ITEM_DTYPE as W
    cm = compiled("smplx", "unitree_g1")
    eng, orc = Engine(cm), Oracle(cm.blob)
    pos, quat, names, offs, _ = synth.synth_clips(cm, 3, 120, seed=13, hard=True, dtype=np.float32)
    sc = cm.slot_columns(names)
    chunk, burn = 8, 8
    items = make_items(offs, chunk=chunk, burn_in=burn, track=True)
    n = len(items)
    from gmr_amd.schedule import plan_walks
    walks = plan_walks(items, offs, chunk)
    assert len(walks) == 3
    # oracle
    qf_o = np.zeros((2 * n, orc.nq))
    q_o, it_o, _ = orc.ik_solve(pos, quat, sc, items, qpos_final=qf_o)
    q_w, it_w, _, done_o = orc.ik_solve(pos, quat, sc, walks, qpos_init=qf_o.copy(), qpos_final=qf_o, want_done=True)
    m = ~np.isnan(q_w[:, 0])
    q_o[m], it_o[m] = q_w[m], it_w[m]
    # GPU
    tp, tq = torch.from_numpy(pos).to(eng.device), torch.from_numpy(quat).to(eng.device)
    q_g, it_g, qf_g = eng.ik_solve(tp, tq, sc, items, n_final=2 * n)
    done_g = torch.zeros(3, dtype=torch.int32, device=eng.device)
    eng.ik_solve(tp, tq, sc, walks, qpos_init=qf_g, qpos_final=qf_g, out=q_g, iters=it_g, frames_done=done_g)
    assert np.array_equal(done_g.cpu().numpy(), done_o) and 0 < done_o.sum() < 3 * 112
    assert np.abs(q_g.cpu().numpy() - q_o).max() < 1e-6 and np.array_equal((it_g & 0x3FFFFFFF).cpu().numpy(), it_o)
    assert np.abs(qf_g.cpu().numpy() - qf_o).max() < 1e-6
    q_seq, it_seq, _ = orc.ik_solve(pos, quat, sc, make_items(offs))
    assert np.abs(q_g.cpu().numpy() - q_seq).max() < 1e-6 and np.array_equal((it_g & 0x3FFFFFFF).cpu().numpy(), it_seq)
    # the packaged form
    q_c, it_c, info = eng.ik_solve_chunked(tp, tq, sc, offs, chunk=chunk, burn_in=burn)
    assert torch.equal(q_c, q_g) and info["resolved_frames"] == int(done_o.sum())
    with pytest.raises(Exception):
        bad = walks.copy(); bad["n_burn"] = 1
        eng.ik_solve(tp, tq, sc, bad, qpos_init=qf_g, qpos_final=qf_g)


def test_error1_equals_error2_for_same_map_tables():
    """fbx_to_g1.json maps the same frames in both tables: error1() == error2() after every retarget, the identity the reference's
    own error logs show on all 2 031 rows (tests/golden/ref_fixtures)."""
    from gmr_amd import GeneralMotionRetargeting as GMR
    g = GMR("fbx", "unitree_g1")
    pos, quat, names, offs, _ = synth.synth_clips(g._cm, 1, 5, seed=2, hard=True, dtype=np.float64)
    for f in range(5):
        g.retarget(_frames(pos, quat, names, f))
        assert g.error1() == g.error2() and g.error1() > 0


def test_bench_two_rank_path_rehearsal():
    """bench.py's N > 1 path, started the way a driver without a launcher would (`python bench.py --gpus 2`: the script spawns its
    own ranks): rank discovery, model broadcast, barrier + max-over-ranks timing, the timed exchange steps (all-gather of qpos,
    strong-scaling split, long clips with chunks sharded over the ranks), rank-0 JSON -- two ranks sharing this box's one GPU
    with gloo collectives, i.e. everything the 8-GPU run does except RCCL itself."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(GMR_BENCH_BACKEND="gloo", GMR_BENCH_SHARE_GPU="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--clips", "64", "--frames", "60"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["frames_per_step"] == 2 * 64 * 60
    assert d["value"] > 0 and abs(d["value"] - d["config"]["frames_per_step"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert d["value_unshaped"] > 0 and d["roofline"]["bound"] == "fp64-vector" and d["hbm"]["non_binding"]
    assert abs(d["n1_equivalent_value"] * 2 - d["value"]) < 1e-6 * d["value"]
    # the CPU path "in the same run" and rank 0's parity are on the N > 1 line too
    assert d["roofline"]["frac"] > 0 and d["roofline"]["kernel_ms"] > 0
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1
    assert d["parity"]["max_abs_qpos_err_vs_cpu"] < 1e-6 and d["parity"]["frames_with_different_solve_count"] == 0
    assert d["parity"]["unshaped"]["max_abs_qpos_err_vs_cpu"] < 1e-6
    c = d["collectives"]
    assert c["rccl_ranks"] == 2 and c["allgather_complete"] and c["allgather_qpos_ms"] > 0 and c["allgather_bytes_per_rank"] == 64 * 60 * 36 * 8
    assert c["allgather_rows_total"] == 2 * 64 * 60  # the whole output, not a sample
    assert d["strong"]["clips_total"] == 64 and d["strong"]["value"] > 0
    lc = d["long_clips_sharded"]
    assert lc["ranks"] == 2 and lc["clips"] == 77 and lc["frames_per_s"] > 0 and lc["resolved_frames"] < 0.05 * lc["frames"]  # (initial headings within 1 rad: speculative starts hit the right basin)


def test_c_abi_bvh_file_from_plain_c(golden_dir, tmp_path):
    """examples/c_abi_bvh_file.c: one BVH file to qpos in C99 through the round-3 entry points -- gmr_bvh_parse_header on the host, the
    MOTION block parsed on the device, gmr_bvh_fk_rows with the config's 14 columns, gmr_ik_solve -- against the Python loader
    (reference-golden-pinned) and the oracle; a file with one 25-digit number exercises the off-path patch."""
    import shutil
    import subprocess
    from gmr_amd import GeneralMotionRetargeting as GMR, _native
    from gmr_amd.bvh import load_lafan1_file
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc on this box")
    _native.load()
    libdir = os.path.dirname(_native.LIB_PATH)
    exe = tmp_path / "c_abi_bvh_file"
    subprocess.check_call([gcc, "-std=c99", "-D__HIP_PLATFORM_AMD__", f"-I{root}/include", "-I/opt/rocm/include",
                           os.path.join(root, "examples", "c_abi_bvh_file.c"), f"-L{libdir}", "-lgmr_amd", "-L/opt/rocm/lib", "-lamdhip64",
                           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
    g = GMR(src_human="bvh", tgt_robot="unitree_g1")
    cm = g._cm
    src = tmp_path / "clip.bvh"
    txt = open(os.path.join(golden_dir, "bvh_lafan_like.bvh"), "rb").read().split(b"\n")
    row = txt[-2].split()
    row[10] = row[10] + b"000000000000000000"   # the same value with 25 digits: off the device's exact path
    txt[-2] = b" ".join(row)
    src.write_bytes(b"\n".join(txt))
    d = tmp_path / "io"
    d.mkdir()
    (d / "model.blob").write_bytes(cm.blob)
    (d / "cols.txt").write_text(f"{cm.robot.nq} {len(g.ik_columns)}\n" + "\n".join(g.ik_columns) + "\n")
    out = subprocess.run([str(exe), str(src), str(d)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("ok:") and "1 off the exact path" in out.stdout, out.stderr[-2000:] + out.stdout[-500:]
    clip = load_lafan1_file(str(src), columns=g.ik_columns)
    T, B = len(clip), len(g.ik_columns)
    kp = np.fromfile(d / "keypoints.f64", dtype="<f8").reshape(T, B, 7)
    assert np.array_equal(kp[..., :3], clip.pos.cpu().numpy()) and np.array_equal(kp[..., 3:], clip.quat.cpu().numpy())
    q = np.fromfile(d / "qpos.f64", dtype="<f8").reshape(T, cm.robot.nq)
    it = np.fromfile(d / "iters.i32", dtype="<i4")
    q_ref, it_ref, _ = Oracle(cm.blob).ik_solve(kp[..., :3].copy(), kp[..., 3:].copy(), np.arange(B, dtype=np.int32), make_items([0, T]))
    assert np.abs(q - q_ref).max() < 1e-6 and np.array_equal(it & 0x3FFFFFFF, it_ref)


def test_caller_access_pattern_of_fbx_to_robot():
    """The attribute accesses of scripts/fbx_to_robot.py on a retargeter: ``model.body(name).id`` into
    ``configuration.data.xpos`` (:1040-1041, 1083-1084, 1157-1158), ``tasks1/2`` as objects with ``frame_name`` and
    ``compute_error(configuration)`` (:1129-1132), ``human_body_to_task1`` (:1012), against the oracle."""
    from gmr_amd import GeneralMotionRetargeting as GMR
    g = GMR("smplx", "unitree_g1")
    cm, orc = g._cm, Oracle(g._cm.blob)
    pos, quat, names, offs, _ = synth.synth_clips(cm, 1, 3, seed=5, hard=True, dtype=np.float64)
    for f in range(3):
        q = g.retarget(_frames(pos, quat, names, f))
    sc = cm.slot_columns(names)
    tp, tq = orc.prepare_targets(pos[2][sc], quat[2][sc])
    xp, _ = orc.fk_mj(q)
    # (1) body ids index xpos the MuJoCo way; the root body's row is qpos[:3]
    bid = g.model.body(g.robot_root_name).id
    assert bid == 1 and g.model.name2id("no_such_body") == -1 and g.model.body(0).name == "world"
    assert np.abs(g.configuration.data.xpos[bid] - q[:3]).max() < 1e-12
    for name in ("left_wrist_yaw_link", "right_wrist_yaw_link", "pelvis"):
        assert np.abs(g.configuration.data.xpos[g.model.body(name).id] - xp[cm.robot.body_index(name)]).max() < 1e-12
    with pytest.raises(KeyError):
        g.model.body("no_such_body")
    # (2) the per-task error breakdown of :1129-1132
    assert hasattr(g, "human_body_to_task1") and g.human_body_to_task1["pelvis"].frame_name == "pelvis"
    for tab, tasks in enumerate((g.tasks1, g.tasks2)):
        _, e_ref = orc.stage_error(tab, q, tp, tq, len(cm.tasks[tab]))
        errs = {t.frame_name: float(np.linalg.norm(t.compute_error(g.configuration))) for t in tasks}
        assert list(errs) == [t.frame for t in cm.tasks[tab]]
        for i, t in enumerate(tasks):
            assert np.abs(t.compute_error(g.configuration) - e_ref[i]).max() < 1e-9
            assert (t.position_cost, t.orientation_cost) == (cm.tasks[tab][i].pos_weight, cm.tasks[tab][i].rot_weight)
        assert abs(np.sqrt(sum(v * v for v in errs.values())) - (g.error1(), g.error2())[tab]) < 1e-9


def test_per_clip_human_heights():
    """clips of different subjects in one batch.  Clip s with ``human_heights[s]`` must equal a retargeter built with
    ``actual_human_height=human_heights[s]`` (one per file in scripts/smplx_to_robot_dataset.py:79-83), and both the oracle."""
    from gmr_amd import GeneralMotionRetargeting as GMR
    heights = [1.55, 1.8, 1.93]
    g = GMR("smplx", "unitree_g1", actual_human_height=1.7)  # the batch object's own height must not matter
    cm = g._cm
    pos, quat, names, offs, _ = synth.synth_clips(cm, 3, 20, seed=8, hard=True, dtype=np.float32)
    q, it = g.retarget_batch(pos, quat, names, seq_offsets=offs, human_heights=heights, return_iters=True)
    qc = g.retarget_batch(pos, quat, names, seq_offsets=offs, human_heights=heights, chunk=8, burn_in=8)
    assert np.abs(q - qc).max() < 1e-6
    for s, h in enumerate(heights):
        a, b = int(offs[s]), int(offs[s + 1])
        gs = GMR("smplx", "unitree_g1", actual_human_height=h)
        q_one = gs.retarget_batch(pos[a:b], quat[a:b], names)
        assert np.abs(q[a:b] - q_one).max() < 1e-9, (s, np.abs(q[a:b] - q_one).max())
        q_ref, it_ref, _ = Oracle(gs._cm.blob).ik_solve(pos[a:b], quat[a:b], gs._cm.slot_columns(names), make_items([0, b - a]))
        assert np.abs(q[a:b] - q_ref).max() < 1e-6 and np.array_equal(it[a:b] & 0x3FFFFFFF, it_ref)
    assert np.abs(q[:20] - q[20:40]).max() > 1e-3  # (the heights do change the result)
    with pytest.raises(ValueError):
        g.retarget_batch(pos, quat, names, seq_offsets=offs, human_heights=[1.7])


def test_one_handle_on_two_streams():
    """ADVICE r1 / VERDICT: launches of one handle on two streams must not share scheduling data (it is stream-ordered now)."""
    from gmr_amd.engine import Engine
    cm = compiled("smplx", "unitree_g1")
    eng, orc = Engine(cm, 0), Oracle(cm.blob)
    dev = eng.device
    sets = []
    for seed, (S, T) in ((41, (96, 60)), (42, (7, 300))):
        pos, quat, names, offs, _ = synth.synth_clips(cm, S, T, seed=seed, hard=True, dtype=np.float32)
        sets.append((torch.from_numpy(pos).to(dev), torch.from_numpy(quat).to(dev), cm.slot_columns(names), make_items(offs), pos, quat))
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    outs = [None, None]
    torch.cuda.synchronize()
    for rep in range(3):
        for k in (0, 1):
            with torch.cuda.stream(streams[k]):
                outs[k] = eng.ik_solve(sets[k][0], sets[k][1], sets[k][2], sets[k][3])
    torch.cuda.synchronize()
    for k in (0, 1):
        q_ref, it_ref, _ = orc.ik_solve(sets[k][4], sets[k][5], sets[k][2], sets[k][3], n_threads=8)
        assert np.abs(outs[k][0].cpu().numpy() - q_ref).max() < 1e-6 and np.array_equal(outs[k][1].cpu().numpy(), it_ref)


def test_host_pipeline_bitwise_equals_resident_path():
    """Engine.ik_solve_host (host arrays in, batches alternating between two streams, qpos written by the kernel straight into the
    pinned host result) against ik_solve on
    resident tensors: same kernel, same per-clip work items -> bitwise equal, whatever the batch split; with the 55-column SMPL-X
    layout (slot_col picks 14 columns on the device) and per-clip heights.  retarget_batch routes big numpy batches through it."""
    from gmr_amd import GeneralMotionRetargeting as GMR
    from gmr_amd.engine import Engine
    cm = compiled("smplx", "unitree_g1")
    eng = Engine(cm, 0)
    lens = [40, 7, 63, 21, 35, 12, 50, 9, 28]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    pos, quat, names, _, _ = synth.synth_clips(cm, 1, int(offs[-1]), seed=17, hard=True, dtype=np.float32, pad_to=55)
    perm = np.random.default_rng(0).permutation(55)                        # the used columns anywhere among the 55
    pos, quat, names = np.ascontiguousarray(pos[:, perm]), np.ascontiguousarray(quat[:, perm]), [names[i] for i in perm]
    sc = cm.slot_columns(names)
    hs = np.linspace(0.9, 1.1, len(lens))
    items = make_items(offs, height_scales=hs)
    q_res, it_res, _ = eng.ik_solve(torch.from_numpy(pos).cuda(), torch.from_numpy(quat).cuda(), sc, items)
    for first, per in ((2, 60), (1, 1), (3, 10 ** 6), (None, None)):        # 2 + 4 batches, one clip each, 3 + the rest, one batch
        q, it = eng.ik_solve_host(pos, quat, sc, offs, height_scales=hs, first_batch_clips=first, max_batch_frames=per)
        assert np.array_equal(q, q_res.cpu().numpy()) and np.array_equal(it, it_res.cpu().numpy())
    q[:] = 0.0
    q_again, _ = eng.ik_solve_host(pos, quat, sc, offs, height_scales=hs, out=q)   # the previous (pinned) result handed back
    assert q_again.ctypes.data == q.ctypes.data and np.array_equal(q, q_res.cpu().numpy())
    with pytest.raises(Exception):
        eng.ik_solve_host(pos, quat, sc, offs, out=np.zeros_like(q))               # pageable memory is refused
    g = GMR("smplx", "unitree_g1")
    g.HOST_PIPELINE_MIN_FRAMES = 1
    q2 = g.retarget_batch(pos, quat, names, seq_offsets=offs, human_heights=hs * cm.config.human_height_assumption)
    assert isinstance(q2, np.ndarray) and np.array_equal(q2, q_res.cpu().numpy())
    bad = pos.copy()
    bad[5, sc[3]] = np.nan
    with pytest.raises(FloatingPointError):
        eng.ik_solve_host(bad, quat, sc, offs)


def test_c_abi_from_plain_c(tmp_path):
    """examples/c_abi_retarget.c: the caller loop of scripts/smplx_to_robot_dataset.py:84-112 written against include/gmr_amd.h in
    C99 -- model from a blob file, hipMalloc'ed buffers, gmr_ik_solve + gmr_fk, no Python or torch in the process -- against the
    oracle on the same files."""
    import shutil
    import subprocess
    from gmr_amd import _native
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc on this box")
    _native.load()
    libdir = os.path.dirname(_native.LIB_PATH)
    exe = tmp_path / "c_abi_retarget"
    subprocess.check_call([gcc, "-std=c99", "-D__HIP_PLATFORM_AMD__", f"-I{root}/include", "-I/opt/rocm/include",
                           os.path.join(root, "examples", "c_abi_retarget.c"), f"-L{libdir}", "-lgmr_amd", "-L/opt/rocm/lib", "-lamdhip64",
                           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
    cm = compiled("smplx", "unitree_g1")
    lens = [50, 33, 81]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    pos, quat, names, _, _ = synth.synth_clips(cm, 1, int(offs[-1]), seed=23, hard=True, dtype=np.float32, pad_to=20)
    sc = cm.slot_columns(names)
    d = tmp_path / "io"
    d.mkdir()
    (d / "model.blob").write_bytes(cm.blob)
    (d / "meta.txt").write_text(f"{offs[-1]} {pos.shape[1]} {cm.nslot} {cm.robot.nq} {cm.robot.nbody} {len(lens)}\n")
    sc.astype("<i4").tofile(d / "slot_col.i32")
    offs.astype("<i8").tofile(d / "seq_offsets.i64")
    pos.astype("<f4").tofile(d / "pos.f32")
    quat.astype("<f4").tofile(d / "quat.f32")
    out = subprocess.run([str(exe), str(d)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("ok:"), out.stderr[-2000:] + out.stdout[-500:]
    q = np.fromfile(d / "qpos.f64", dtype="<f8").reshape(-1, cm.robot.nq)
    it = np.fromfile(d / "iters.i32", dtype="<i4")
    bp = np.fromfile(d / "body_pos.f32", dtype="<f4").reshape(-1, cm.robot.nbody, 3)
    orc = Oracle(cm.blob)
    q_ref, it_ref, _ = orc.ik_solve(pos, quat, sc, make_items(offs))
    assert np.abs(q - q_ref).max() < 1e-6 and np.array_equal(it & 0x3FFFFFFF, it_ref)
    N = q.shape[0]
    ident = np.zeros((N, 4), np.float32)
    ident[:, 3] = 1.0
    bp_ref, _ = orc.fk_kin(np.zeros((N, 3), np.float32), ident, q[:, 7:].astype(np.float32), want_rot=False)
    assert np.abs(bp - bp_ref).max() < 2e-6


def test_bvh_folder_to_robot_batch(golden_dir, tmp_path):
    """scripts/bvh_to_robot_dataset.py's file loop as one batch: load_lafan1_files (threaded native parse, one gmr_bvh_fk launch)
    -> retarget_batch with every clip's own height -> the same qpos as loading and retargeting the files one by one; poses equal
    the reference loader's goldens."""
    import shutil
    from gmr_amd import GeneralMotionRetargeting as GMR
    from gmr_amd.bvh import load_lafan1_file, load_lafan1_files
    src = os.path.join(golden_dir, "bvh_lafan_like.bvh")
    text = open(src).read().split("\n")
    k = next(i for i, ln in enumerate(text) if ln.startswith("Frame Time"))
    rows = [ln for ln in text[k + 1:] if ln.strip()]
    files = []
    for j, n in enumerate((len(rows), 11, 23)):                       # the golden clip and two shorter cuts of it
        p = tmp_path / f"clip{j}.bvh"
        p.write_text("\n".join(text[:k - 1] + [f"Frames: {n}", text[k]] + rows[:n]) + "\n")
        files.append(str(p))
    batch = load_lafan1_files(files, threads=3)
    gold = np.load(os.path.join(golden_dir, "bvh_lafan_like.npz"))
    assert batch.body_names == [str(n) for n in gold["names"]] and list(batch.seq_offsets) == [0, len(rows), len(rows) + 11, len(rows) + 34]
    assert np.abs(batch.pos[: len(rows)].cpu().numpy() - gold["pos"]).max() < 1e-9
    g = GMR(src_human="bvh", tgt_robot="unitree_g1")
    q = g.retarget_batch(batch.pos, batch.quat, batch.body_names, seq_offsets=batch.seq_offsets, human_heights=batch.human_heights)
    for j, f in enumerate(files):
        clip = load_lafan1_file(f)
        assert abs(clip.human_height - batch.human_heights[j]) < 1e-12
        q1 = GMR(src_human="bvh", tgt_robot="unitree_g1", actual_human_height=clip.human_height).retarget_batch(clip.pos, clip.quat, clip.body_names)
        a, b = int(batch.seq_offsets[j]), int(batch.seq_offsets[j + 1])
        assert float((q[a:b] - q1).abs().max().item()) < 1e-9
    with pytest.raises(ValueError):
        load_lafan1_files([files[0], os.path.join(golden_dir, "bvh_canonical_40f.bvh")])   # another skeleton


def test_planar_base_robot_through_the_class():
    """GMR("smplx", "galaxea_r1pro") -- reachable in the reference through scripts/smplx_to_robot.py:29 -- returns MuJoCo's own
    qpos layout [x, y, yaw, 24 hinges]: per-frame calls, the batched call and the oracle agree, and the heading accumulates past
    +-pi like a hinge coordinate instead of wrapping like a quaternion."""
    from gmr_amd import GeneralMotionRetargeting as GMR
    g = GMR("smplx", "galaxea_r1pro")
    cm = g._cm
    assert g.model.mj_nq == 27 and g.configuration.data.qpos.shape == (27,) and len(g.tasks1) == 10 and len(g.tasks2) == 0
    pos, quat, names, offs, q_true = synth.synth_clips(cm, 2, 90, seed=12, hard=False, dtype=np.float64)
    # spin the second clip's targets about z so that the base turns through more than a full revolution
    T = 90
    ang = np.linspace(0.0, 2.6 * np.pi, T)
    c, s_ = np.cos(ang), np.sin(ang)
    p2 = pos[T:].copy()
    pos[T:, :, 0], pos[T:, :, 1] = c[:, None] * p2[:, :, 0] - s_[:, None] * p2[:, :, 1], s_[:, None] * p2[:, :, 0] + c[:, None] * p2[:, :, 1]
    spin = np.stack([np.cos(ang / 2), 0 * ang, 0 * ang, np.sin(ang / 2)], -1)
    quat[T:] = synth.qmul(np.broadcast_to(spin[:, None], quat[T:].shape), quat[T:])
    qb = g.retarget_batch(pos, quat, names, seq_offsets=offs)
    assert qb.shape == (2 * T, 27)
    q_ref, _, _ = Oracle(cm.blob).ik_solve(pos, quat, cm.slot_columns(names), make_items(offs))
    mj_ref = cm.robot.to_mj_qpos(q_ref)
    assert np.abs(qb[:, :2] - mj_ref[:, :2]).max() < 1e-6 and np.abs(qb[:, 3:] - mj_ref[:, 3:]).max() < 1e-6
    assert np.abs(np.angle(np.exp(1j * (qb[:, 2] - mj_ref[:, 2])))).max() < 1e-6          # same heading ...
    assert np.abs(np.diff(qb[T:, 2])).max() < 1.0 and qb[-1, 2] - qb[T, 2] > 2 * np.pi       # ... accumulated, not wrapped
    assert np.abs(np.diff(qb[:T, 2])).max() < 1.0
    for f in range(T, 2 * T):  # the stateful per-frame API walks the same path
        q = g.retarget({n: (pos[f, i], quat[f, i]) for i, n in enumerate(names)})
        assert q.shape == (27,) and np.abs(q - qb[f]).max() < 1e-6, f
    assert np.abs(g.configuration.data.qpos - qb[-1]).max() < 1e-6 and g.configuration.data.xpos.shape == (26, 3)
    from gmr_amd import dataset
    with pytest.raises(NotImplementedError):
        dataset.retarget_clips(g, pos, quat, names, offs)


def test_persistent_session_is_the_batch_solver_fed_frame_by_frame():
    """gmr_session_set_persistent: frames through one resident wavefront and a pinned mailbox (SURVEY f-4).  The wavefront keeps its
    state in LDS between frames exactly like a multi-frame work item, so the result equals the batched solve of the same frames
    bit for bit (a launch per frame re-reads its state and differs in the last bit); it leaves when idle and the next frame brings
    a new one; reset / state / offset_to_ground / mode switches / destroy work while it is resident."""
    import time
    from gmr_amd.engine import Engine, IKParams
    cm = compiled("smplx", "unitree_g1")
    eng = Engine(cm)
    T = 80
    pos, quat, names, offs, _ = synth.synth_clips(cm, 1, T, seed=23, hard=True, dtype=np.float32)
    sc = cm.slot_columns(names)
    dev = torch.device("cuda", 0)
    q_batch, it_batch, _ = eng.ik_solve(torch.from_numpy(pos).to(dev), torch.from_numpy(quat).to(dev), sc, make_items(offs))
    q_batch, it_batch = q_batch.cpu().numpy(), it_batch.cpu().numpy()
    b = eng.session(sc, pos.shape[1], IKParams(), dtype=np.float32)
    b.set_persistent(200)
    lat = []
    for f in range(T):
        t0 = time.perf_counter()
        qb, nb = b.step(pos[f], quat[f])
        lat.append(time.perf_counter() - t0)
        # bit for bit while one wavefront stays resident; a relaunch re-reads the state like a launch per frame does (last bit)
        assert (np.array_equal(qb, q_batch[f]) if f <= 20 else np.abs(qb - q_batch[f]).max() < 1e-9) and nb == it_batch[f], f
        if f == 20:  # the resident wavefront idles out (idle 20 ms) and the next frame brings a new one, continuing from its state
            b.set_persistent(20)
            time.sleep(0.08)
        if f == 30:
            assert np.array_equal(b.state(), qb)
        if f == 45:
            b.set_persistent(0)    # back to one launch per frame (last-bit differences from here on) ...
        if f == 50:
            b.set_persistent(200)  # ... and resident again
            break
    for f in range(51, T):
        qb, nb = b.step(pos[f], quat[f])
        assert np.abs(qb - q_batch[f]).max() < 1e-9 and nb == it_batch[f], f
    print(f"persistent session step latency: median {1e6 * np.median(lat):.0f} us")
    # offset_to_ground rides on the posted word; a launch-per-frame session is the comparison
    a = eng.session(sc, pos.shape[1], IKParams(), dtype=np.float32)
    a.reset(); b.reset()
    for f in range(12):
        qa, na = a.step(pos[f], quat[f], offset_to_ground=f % 3 == 1)
        qb, nb = b.step(pos[f], quat[f], offset_to_ground=f % 3 == 1)
        assert np.abs(qa - qb).max() < 1e-9 and na == nb, f
    a.close()
    b.close()  # destroyed while the wavefront is resident
    torch.cuda.synchronize()
    # the class rides on it when asked to
    from gmr_amd import GeneralMotionRetargeting as GMR
    g0, g1 = GMR("smplx", "unitree_g1"), GMR("smplx", "unitree_g1", persistent_session_ms=100)
    for f in range(10):
        fr = {n: (pos[f, i].astype(np.float64), quat[f, i].astype(np.float64)) for i, n in enumerate(names)}
        assert np.abs(g0.retarget(dict(fr)) - g1.retarget(dict(fr))).max() < 1e-9
    g1.setup_retarget_configuration()  # closes the sessions (parks the wavefront)


def test_launch_order_by_probe_moves_work_in_time_only():
    """gmr_ik_plan_order / gmr_ik_solve_ordered: the probe's order is a permutation that ranks the expensive clips first, the ordered
    launch gives bit for bit what the plain launch gives (qpos, solve counts, frames_done), and `launch_order="auto"` plans only
    when there are more items than wavefront slots, long enough for the probe to be a small fraction of the work."""
    from gmr_amd.engine import Engine, EngineError
    cm = compiled("smplx", "unitree_g1")
    eng = Engine(cm)
    dev = torch.device("cuda", 0)
    T, D = 96, 16
    pe, qe, names, _, _ = synth.synth_clips(cm, D // 2, T, seed=31, hard=False, dtype=np.float32)
    ph, qh, _, _, _ = synth.synth_clips(cm, D // 2, T, seed=32, hard=True, dtype=np.float32)
    S = 2304  # > 8 x 256 wavefront slots
    pos = torch.from_numpy(np.concatenate([pe, ph])).to(dev).repeat(S // D, 1, 1).contiguous()
    quat = torch.from_numpy(np.concatenate([qe, qh])).to(dev).repeat(S // D, 1, 1).contiguous()
    offs = np.arange(S + 1, dtype=np.int64) * T
    items = make_items(offs)
    sc = cm.slot_columns(names)
    fd0 = torch.zeros(S, dtype=torch.int32, device=dev)
    q0, it0, _ = eng.ik_solve(pos, quat, sc, items, launch_order=None, frames_done=fd0)
    order = eng.plan_order(pos, quat, sc, items, probe_frames=24)
    o = order.cpu().numpy()
    assert sorted(o.tolist()) == list(range(S))
    work = (it0.reshape(S, T).to(torch.int64) & 0x3FFFFFFF).sum(1).cpu().numpy()
    first, last = work[o[: S // 4]].mean(), work[o[-S // 4:]].mean()
    assert first > 1.08 * last, (first, last)          # the expensive quarter leads
    fd1 = torch.zeros(S, dtype=torch.int32, device=dev)
    q1, it1, _ = eng.ik_solve(pos, quat, sc, items, launch_order=order, frames_done=fd1)
    assert torch.equal(q0, q1) and torch.equal(it0, it1) and torch.equal(fd0, fd1) and int(fd1.min()) == T
    # "auto" (Engine._probe_frames): more items than slots; equal lengths -> a 4-frame probe from 64 frames on, 32 frames from 256 on;
    # different lengths -> 32 frames from a mean of 1000 on, no probe below (the length order is better there)
    eq = lambda n, T_: make_items(np.arange(n + 1, dtype=np.int64) * T_)  # noqa: E731
    assert eng._probe_frames(items) == 4 and eng._probe_frames(eq(S, 600)) == 32 and eng._probe_frames(eq(S, 255)) == 4 and eng._probe_frames(eq(S, 48)) == 0
    assert eng._probe_frames(make_items(offs[:1025])) == 0 and not eng._order_pays(eq(2048, 3000))                    # no more items than slots
    var = lambda a, b: make_items(np.cumsum(np.r_[0, np.tile([a, b], S // 2)]))  # noqa: E731
    assert eng._probe_frames(var(600, 900)) == 0 and eng._probe_frames(var(800, 1600)) == 32
    assert eng._probe_frames(make_items(np.arange(65, dtype=np.int64) * 6400, chunk=104, burn_in=24)) == 0           # speculative chunks are not probed
    q2, it2, _ = eng.ik_solve(pos, quat, sc, items)  # launch_order="auto": the 96-frame items are probed (4 frames) -- same numbers
    assert torch.equal(q0, q2) and torch.equal(it0, it2)
    # items of different lengths (the un-shaped workload's case), forced through the probe: still the plain launch's numbers
    lens = np.tile([64, 96, 80, 144], S // 4)
    voffs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    vitems = make_items(voffs)
    vpos, vquat = pos[: voffs[-1]].contiguous(), quat[: voffs[-1]].contiguous()
    qa, ia, _ = eng.ik_solve(vpos, vquat, sc, vitems, launch_order=None)
    eng.PROBE_MIN_LENGTH = 64
    try:
        assert eng._probe_frames(vitems) == 32
        qb, ib, _ = eng.ik_solve(vpos, vquat, sc, vitems)
        assert torch.equal(qa, qb) and torch.equal(ia, ib)
    finally:
        del eng.PROBE_MIN_LENGTH
    with pytest.raises(EngineError):
        eng.ik_solve(pos, quat, sc, items, launch_order=order[:-1].contiguous())
    walk = items.copy(); walk["check_stride"][0] = 4
    with pytest.raises(EngineError):
        eng.plan_order(pos, quat, sc, walk)
    # small and degenerate inputs: one item, three items of different lengths (shorter than the probe), identical clips (one bucket)
    assert eng.plan_order(pos, quat, sc, items[:1]).cpu().tolist() == [0]
    short = make_items(np.array([0, 5, 25, 40], dtype=np.int64))
    assert sorted(eng.plan_order(pos, quat, sc, short).cpu().tolist()) == [0, 1, 2]
    same = make_items(np.arange(5, dtype=np.int64) * (D * T))  # four items over the same tiled data -> equal cost
    assert sorted(eng.plan_order(pos, quat, sc, same, probe_frames=8).cpu().tolist()) == [0, 1, 2, 3]
    q3, _, _ = eng.ik_solve(pos, quat, sc, short, launch_order=eng.plan_order(pos, quat, sc, short))
    q4, _, _ = eng.ik_solve(pos, quat, sc, short, launch_order=None)
    assert torch.equal(q3[:40], q4[:40])


def test_handles_on_concurrent_host_threads():
    """include/gmr_amd.h: one handle per host thread at a time, different handles are independent.  Four host threads, each with its own
    handle and stream, solve, run FK and a kin_op at the same time, twenty rounds each; every result equals the serial one
    (the library's shared state is the per-device scratch pool behind a mutex and nothing else)."""
    import threading
    from gmr_amd.engine import Engine
    cm = compiled("smplx", "unitree_g1")
    dev = torch.device("cuda", 0)
    n_thr, rounds = 4, 20
    sets = []
    for k in range(n_thr):
        pos, quat, names, offs, _ = synth.synth_clips(cm, 6 + k, 40 + 8 * k, seed=300 + k, hard=bool(k % 2), dtype=np.float32)
        sets.append((torch.from_numpy(pos).to(dev), torch.from_numpy(quat).to(dev), cm.slot_columns(names), make_items(offs)))
    ref_eng = Engine(cm)
    refs = []
    for p, q, sc, items in sets:
        out, it, _ = ref_eng.ik_solve(p, q, sc, items)
        d32 = out[:, 7:].float().contiguous()
        bp, br = ref_eng.fk(out[:, :3].float().contiguous(), out[:, [4, 5, 6, 3]].float().contiguous(), d32)
        refs.append((out.clone(), it.clone(), bp.clone(), br.clone(), ref_eng.dof_to_rot(d32).clone()))
    torch.cuda.synchronize()
    errors = []

    def work(k):
        try:
            eng = Engine(cm)
            st = torch.cuda.Stream(dev)
            p, q, sc, items = sets[k]
            with torch.cuda.stream(st):
                for _ in range(rounds):
                    out, it, _ = eng.ik_solve(p, q, sc, items)
                    d32 = out[:, 7:].float().contiguous()
                    bp, br = eng.fk(out[:, :3].float().contiguous(), out[:, [4, 5, 6, 3]].float().contiguous(), d32)
                    jr = eng.dof_to_rot(d32)
                    st.synchronize()
                    r = refs[k]
                    if not (torch.equal(out, r[0]) and torch.equal(it, r[1]) and torch.equal(bp, r[2]) and torch.equal(br, r[3]) and torch.equal(jr, r[4])):
                        errors.append(f"thread {k}: results differ from the serial run")
                        return
            eng.close()
        except Exception as ex:  # noqa: BLE001
            errors.append(f"thread {k}: {ex!r}")
    threads = [threading.Thread(target=work, args=(k,)) for k in range(n_thr)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not errors and not any(t.is_alive() for t in threads), errors
