"""CPU-only checks that pin the oracle (oracle/gmr_oracle.c).

FK (KinematicsModel convention) is pinned by golden vectors generated from the
reference's own kinematics_model.py (tests/golden/make_golden.py).  The IK side has no
runnable reference (mink/mujoco/daqp absent -> "parity unpinned"); it is pinned here by
mathematical invariants: finite-difference Jacobians, KKT + an independent bounded
least-squares solve, reachable-target recovery, limit behaviour, error1 == error2.
"""
import json
import os

import numpy as np
import pytest
from scipy.optimize import lsq_linear

from gmr_amd import synth
from oracle.oracle import Oracle, WORK_ITEM_DTYPE, box_qp
from tests.util import CONFIG_ROBOTS, compiled, make_items, quat_angle

# every registry robot the reference's KinematicsModel can parse (engineai_pm01 needs <include>, galaxea_r1pro and
# berkeley_humanoid_lite fail in its _parse_xml: no golden possible)
GOLDEN_ROBOTS = ["unitree_g1", "unitree_g1_with_hands", "booster_t1", "stanford_toddy", "fourier_n1", "kuavo_s45", "hightorque_hi", "booster_k1"]


@pytest.mark.parametrize("robot", GOLDEN_ROBOTS)
def test_tree_matches_reference(robot, golden_dir):
    cm = compiled("smplx", robot)
    with open(os.path.join(golden_dir, f"tree_{robot}.json")) as f:
        ref = json.load(f)
    rob = cm.robot
    assert rob.body_names == ref["body_names"]
    assert rob.parent.tolist() == ref["parent_indices"]
    dof_idx = [(-1 if rob.qpos_adr[b] < 7 else int(rob.qpos_adr[b] - 7)) for b in range(rob.nbody)]
    assert dof_idx == ref["joint_dof_idx"]
    lo, hi = rob.dof_limits()
    np.testing.assert_allclose(lo.astype(np.float32), np.array(ref["lower"], np.float32), rtol=0, atol=0)
    np.testing.assert_allclose(hi.astype(np.float32), np.array(ref["upper"], np.float32), rtol=0, atol=0)
    assert rob.nq - 7 == ref["num_dof"]


@pytest.mark.parametrize("robot", GOLDEN_ROBOTS)
def test_fk_kin_matches_reference_golden(robot, golden_dir):
    """oracle_fk_kin vs reference KinematicsModel.forward_kinematics; float32, tol 2e-6 (abs, metres / quat units)."""
    cm = compiled("smplx", robot)
    g = np.load(os.path.join(golden_dir, f"fk_{robot}.npz"))
    orc = Oracle(cm.blob)
    bp, br = orc.fk_kin(g["root_pos"], g["root_rot"], g["dof_pos"])
    assert np.abs(bp - g["body_pos"]).max() < 2e-6 * max(1.0, np.abs(g["body_pos"]).max())
    assert np.abs(br - g["body_rot"]).max() < 2e-6
    T = g["dof_pos"].shape[0]
    ident = np.tile(np.array([[0, 0, 0, 1]], np.float32), (T, 1))
    bp0, _ = orc.fk_kin(np.zeros((T, 3), np.float32), ident, g["dof_pos"])
    assert np.abs(bp0 - g["local_body_pos"]).max() < 2e-6


@pytest.mark.parametrize("robot", ["unitree_g1", "unitree_g1_with_hands", "booster_t1"])
def test_fk_kin_matches_reference_golden_wide_inputs(robot, golden_dir):
    """The same against the second reference-generated set (tests/golden/make_golden_wide.py): angles of +-7 rad, exact 0 / +-pi /
    2 pi, root positions of +-50 m, root quaternions that are not unit (the reference multiplies them as they come).  Tolerance
    2e-6 relative to the largest magnitude in play (positions up to 55 m, quaternion entries up to 4)."""
    cm = compiled("smplx", robot)
    g = np.load(os.path.join(golden_dir, f"fk_{robot}_wide.npz"))
    bp, br = Oracle(cm.blob).fk_kin(g["root_pos"], g["root_rot"], g["dof_pos"])
    assert np.abs(bp - g["body_pos"]).max() < 2e-6 * max(1.0, np.abs(g["body_pos"]).max())
    assert np.abs(br - g["body_rot"]).max() < 2e-6 * max(1.0, np.abs(g["body_rot"]).max())


@pytest.mark.parametrize("robot", GOLDEN_ROBOTS)
def test_kin_ops_match_reference_golden(robot, golden_dir):
    """dof_to_rot / rot_to_dof / convert_local_rot_to_global / forward_kinematics(fitted_shape=) of the reference's KinematicsModel
    (kinematics_model.py:172-246; tests/golden/make_golden_kin_ops.py) against the oracle's float32 restatements.  rot_to_dof's inputs
    include w < 0, rotations below the 1e-5 axis threshold and rotations next to pi; the result is clamped to the joint limits."""
    cm = compiled("smplx", robot)
    g = np.load(os.path.join(golden_dir, f"kin_ops_{robot}.npz"))
    orc = Oracle(cm.blob)
    assert np.abs(orc.dof_to_rot(g["dof_pos"]) - g["joint_rot"]).max() < 2e-7
    back = orc.rot_to_dof(g["rot_in"])
    assert back.shape == g["dof_back"].shape
    assert np.abs(back - g["dof_back"]).max() < 2e-6
    lo, hi = cm.robot.dof_limits()
    assert (back >= lo.astype(np.float32)).all() and (back <= hi.astype(np.float32)).all()
    assert np.abs(orc.local_rot_to_global(g["local_rot"]) - g["global_rot"]).max() < 2e-6
    for key in ("shape1", "shape3"):
        bp, br = orc.fk_kin(g["root_pos"], g["root_rot"], g["dof_pos"], fitted_shape=g[key])
        assert np.abs(bp - g[f"body_pos_{key}"]).max() < 2e-6 * max(1.0, np.abs(g[f"body_pos_{key}"]).max())
        assert np.abs(br - g[f"body_rot_{key}"]).max() < 2e-6


def test_quat_mul_convention(golden_dir):
    """wxyz Hamilton product agrees with reference rot_utils.quat_mul_np (golden)."""
    g = np.load(os.path.join(golden_dir, "quat_mul_wxyz.npz"))
    np.testing.assert_allclose(synth.qmul(g["a"], g["b"]), g["ab"], atol=1e-14)


@pytest.mark.parametrize("robot", CONFIG_ROBOTS)
def test_fk_mj_vs_fk_kin_conventions_agree(robot):
    """MuJoCo-convention f64 FK == KinematicsModel-convention f32 FK up to f32 rounding and
    the un-normalised XML quaternions the latter keeps (<= 5e-5 m on these models)."""
    cm = compiled("smplx", robot)
    orc = Oracle(cm.blob)
    rng = np.random.default_rng(3)
    q = synth.synth_robot_trajectory(cm.robot, 8, rng)
    for f in range(8):
        xp, xq = orc.fk_mj(q[f])
        xp2, xq2 = synth.fk_numpy(cm.robot, q[f:f + 1])
        assert np.abs(xp - xp2[0]).max() < 1e-12
        root_xyzw = q[f, [4, 5, 6, 3]]
        bp, _ = orc.fk_kin(q[f:f + 1, :3], root_xyzw[None], q[f:f + 1, 7:])
        assert np.abs(bp[0] - xp).max() < 5e-5


@pytest.mark.parametrize("robot", ["unitree_g1", "engineai_pm01", "unitree_g1_with_hands"])
def test_task_jacobian_finite_difference(robot):
    """J = d e / d dq: central differences of e(q (+) h e_k) vs the analytic J, all task bodies."""
    cm = compiled("smplx", robot)
    orc = Oracle(cm.blob)
    rng = np.random.default_rng(5)
    q = synth.synth_robot_trajectory(cm.robot, 4, rng)[3]
    h = 1e-6
    worst = 0.0
    for body in cm.task_body[0]:
        tp = rng.normal(0, 0.5, 3) + q[:3]
        tq = rng.normal(size=4)
        tq /= np.linalg.norm(tq)
        e0, J = orc.task_error_and_jacobian(q, body, tp, tq)
        Jfd = np.zeros_like(J)
        for k in range(orc.nv):
            dq = np.zeros(orc.nv)
            dq[k] = h
            ep, _ = orc.task_error_and_jacobian(orc.integrate(q, dq), body, tp, tq)
            em, _ = orc.task_error_and_jacobian(orc.integrate(q, -dq), body, tp, tq)
            Jfd[:, k] = (ep - em) / (2 * h)
        worst = max(worst, np.abs(J - Jfd).max())
    assert worst < 2e-6, worst


def _kkt(H, c, lo, hi, x):
    g = H @ x + c
    r = 0.0
    for i in range(len(x)):
        at_lo, at_hi = abs(x[i] - lo[i]) < 1e-12, abs(x[i] - hi[i]) < 1e-12
        if at_lo and not at_hi:
            r = max(r, max(0.0, -g[i]))
        elif at_hi and not at_lo:
            r = max(r, max(0.0, g[i]))
        elif not at_lo and not at_hi:
            r = max(r, abs(g[i]))
    return r


def test_box_qp_random_vs_bvls():
    rng = np.random.default_rng(7)
    for trial in range(40):
        n = int(rng.integers(3, 40))
        A = rng.normal(size=(n + 5, n))
        H = A.T @ A + 0.5 * np.eye(n)
        c = rng.normal(size=n) * 10
        lo = -np.abs(rng.normal(size=n)) * 0.2
        hi = np.abs(rng.normal(size=n)) * 0.2
        if trial % 5 == 0:  # some boxes that exclude 0
            lo[0], hi[0] = 0.05, 0.3
        x, _ = box_qp(H, c, lo, hi)
        assert np.all(x >= lo - 1e-12) and np.all(x <= hi + 1e-12)
        assert _kkt(H, c, lo, hi, x) < 1e-8 * (1 + np.abs(c).max())
        L = np.linalg.cholesky(H)
        ref = lsq_linear(L.T, -np.linalg.solve(L, c), bounds=(lo, hi), method="bvls", tol=1e-14).x
        np.testing.assert_allclose(x, ref, atol=1e-8)


@pytest.mark.parametrize("robot", CONFIG_ROBOTS)
def test_reachable_targets_recovered(robot):
    """Targets generated by FK of an in-limit trajectory are tracked: final error small,
    qpos recovered within 1e-3 rad after the first frames, error1 == error2 at the truth."""
    cm = compiled("smplx", robot)
    orc = Oracle(cm.blob)
    pos, quat, names, offs, qtrue = synth.synth_clips(cm, 1, 60, seed=11, dtype=np.float64, amp=0.2)
    sc = cm.slot_columns(names)
    q, iters, _ = orc.ik_solve(pos, quat, sc, make_items(offs, WORK_ITEM_DTYPE))
    assert iters.min() >= 2 and iters.max() <= 22
    active = np.zeros(cm.robot.nq, bool)  # coordinates some task constrains
    for b in cm.task_body[0]:
        while b >= 0:
            a = cm.robot.qpos_adr[b]
            if a >= 7:
                active[a] = True
            b = cm.robot.parent[b]
    d = q[20:] - qtrue[20:]
    assert np.abs(d[:, :3]).max() < 1e-3
    assert quat_angle(q[20:, 3:7], qtrue[20:, 3:7]).max() < 1e-3
    assert np.abs(d[:, active]).max() < 1e-3, np.abs(d[:, active]).max()
    tp, tq = orc.prepare_targets(pos[30][sc], quat[30][sc])
    e1, _ = orc.stage_error(0, qtrue[30], tp, tq, len(cm.tasks[0]))
    e2, _ = orc.stage_error(1, qtrue[30], tp, tq, len(cm.tasks[1]))
    assert e1 < 1e-9 and e2 < 1e-9


def test_hand_dofs_stay_zero():
    """unitree_g1_with_hands reuses the 29-DoF config: untasked hand hinges get zero gradient."""
    cm = compiled("smplx", "unitree_g1_with_hands")
    cm29 = compiled("smplx", "unitree_g1")
    orc = Oracle(cm.blob)
    pos, quat, names, offs, _ = synth.synth_clips(cm29, 1, 12, seed=2, hard=True, dtype=np.float64)
    q, _, _ = orc.ik_solve(pos, quat, cm.slot_columns(names), make_items(offs, WORK_ITEM_DTYPE))
    tasked = set()
    for b in cm.task_body[0]:
        while b >= 0:
            tasked.add(int(b))
            b = cm.robot.parent[b]
    hand = [cm.robot.qpos_adr[b] for b in cm.robot.hinge_bodies() if int(b) not in tasked]
    assert len(hand) == 14
    assert np.all(q[:, hand] == 0.0)


def test_limits_respected_and_gain_rule():
    """Hard targets push joints to their range: qpos never leaves [lo, hi] and every step obeys dq <= 0.95 (hi - q)."""
    cm = compiled("smplx", "unitree_g1")
    orc = Oracle(cm.blob)
    pos, quat, names, offs, _ = synth.synth_clips(cm, 1, 40, seed=4, hard=True, dtype=np.float64)
    sc = cm.slot_columns(names)
    q, _, _ = orc.ik_solve(pos, quat, sc, make_items(offs, WORK_ITEM_DTYPE))
    lo, hi = cm.robot.dof_limits()
    assert np.all(q[:, 7:] >= lo - 1e-12) and np.all(q[:, 7:] <= hi + 1e-12)
    tp, tq = orc.prepare_targets(pos[5][sc], quat[5][sc])
    H, c, blo, bhi = orc.build_qp(0, q[4], tp, tq)
    np.testing.assert_allclose(bhi[6:], 0.95 * (hi - q[4, 7:]), atol=1e-15)
    np.testing.assert_allclose(blo[6:], -0.95 * (q[4, 7:] - lo), atol=1e-15)
    x, _ = box_qp(H, c, blo, bhi)
    assert _kkt(H, c, blo, bhi, x) < 1e-8 * (1 + np.abs(c).max())
    assert np.allclose(H, H.T) and np.linalg.eigvalsh(H).min() >= 0.5 - 1e-9


def test_work_item_semantics_chunk_equals_sequential_prefix():
    """A work item with burn-in reproduces the sequential run when it starts at frame 0; init_row/final_row chain state."""
    cm = compiled("smplx", "unitree_g1")
    orc = Oracle(cm.blob)
    pos, quat, names, offs, _ = synth.synth_clips(cm, 1, 30, seed=9, dtype=np.float64)
    sc = cm.slot_columns(names)
    q_seq, _, _ = orc.ik_solve(pos, quat, sc, make_items(offs, WORK_ITEM_DTYPE))
    it = np.zeros(1, WORK_ITEM_DTYPE)
    it["frame_begin"], it["n_burn"], it["n_out"], it["init_row"], it["final_row"], it["burn_row"] = 0, 10, 20, -1, 0, 1
    q_b, _, qf = orc.ik_solve(pos, quat, sc, it, want_final=True)
    assert np.all(np.isnan(q_b[:10])) and np.array_equal(q_b[10:], q_seq[10:])
    np.testing.assert_array_equal(qf[0], q_seq[-1])
    np.testing.assert_array_equal(qf[1], q_seq[9])  # burn_row: the state the first output frame starts from
    it2 = np.zeros(1, WORK_ITEM_DTYPE)
    it2["frame_begin"], it2["n_burn"], it2["n_out"], it2["init_row"], it2["final_row"], it2["burn_row"] = 15, 0, 15, 0, -1, -1
    q_c, _, _ = orc.ik_solve(pos, quat, sc, it2, qpos_init=q_seq[14:15])
    np.testing.assert_array_equal(q_c[15:], q_seq[15:])


def test_verification_walk_equals_sequential():
    """check_stride items (gmr_blob.h): chunks solved from a speculative start, then one walk per clip (schedule.plan_walks) that
    adopts the consistent chunks and re-solves the others -- the result is the sequential run, whatever the speculation did.
    Started from qpos0 at the world origin (INIT_QPOS0, a deliberately poor start with this short burn-in) some chunks are wrong
    and get re-solved; started with the base on the root task's target (INIT_ROOT_TARGET, the default) fewer are."""
    from gmr_amd._native import INIT_QPOS0, INIT_ROOT_TARGET
    from gmr_amd.schedule import make_items as sched_items, plan_walks
    cm = compiled("smplx", "unitree_g1")
    orc = Oracle(cm.blob)
    pos, quat, names, offs, _ = synth.synth_clips(cm, 2, 90, seed=13, hard=True, dtype=np.float64)
    sc = cm.slot_columns(names)
    q_seq, it_seq, _ = orc.ik_solve(pos, quat, sc, make_items(offs, WORK_ITEM_DTYPE))
    chunk, burn = 8, 8
    resolved = {}
    for init in (INIT_QPOS0, INIT_ROOT_TARGET):
        items = np.ascontiguousarray(sched_items(offs, chunk=chunk, burn_in=burn, track=True, chunk_init=init), dtype=WORK_ITEM_DTYPE)
        assert set(items["init_row"]) == {INIT_QPOS0, init} and np.all(items["init_row"][items["n_burn"] == 0] == INIT_QPOS0)
        n = len(items)
        qf = np.zeros((2 * n, orc.nq))
        q0, it0, _ = orc.ik_solve(pos, quat, sc, items, qpos_final=qf)
        if init == INIT_QPOS0:
            assert np.abs(q0 - q_seq).max() > 1e-3  # the speculative chunks are wrong somewhere
        else:
            assert np.abs(q0 - q_seq).max() < 1e-3  # every chunk already sits in the sequential run's basin
        walks = np.ascontiguousarray(plan_walks(items, offs, chunk), dtype=WORK_ITEM_DTYPE)
        assert len(walks) == 2 and np.all(walks["check_stride"] == chunk)
        q1, it1, _, done = orc.ik_solve(pos, quat, sc, walks, qpos_init=qf.copy(), qpos_final=qf, want_done=True)
        solved = ~np.isnan(q1[:, 0])
        q0[solved], it0[solved] = q1[solved], it1[solved]
        assert np.abs(q0 - q_seq).max() < 1e-6 and np.array_equal(it0, it_seq)
        assert int(solved.sum()) == int(done.sum()) < 2 * 90 - 2 * chunk
        resolved[init] = int(done.sum())
    assert resolved[INIT_QPOS0] > 0 and resolved[INIT_ROOT_TARGET] <= resolved[INIT_QPOS0]


def test_verification_walk_is_blind_to_the_sign_of_target_quaternions():
    """Key-points that come from files carry quaternions of either sign (q and -q are one rotation).  A chunk started on its root
    target takes that sign, the sequential run keeps its own: the walk compares up to the sign and turns what it adopts, so the
    result is still the sequential run -- values, the base quaternion's sign included -- and no more is re-solved than with
    consistently signed inputs (gmr_blob.h)."""
    from gmr_amd.schedule import make_items as sched_items, plan_walks
    cm = compiled("bvh", "unitree_g1")
    orc = Oracle(cm.blob)
    pos, quat, names, offs, _ = synth.synth_clips(cm, 2, 120, seed=21, hard=False, dtype=np.float64)
    sc = cm.slot_columns(names)
    chunk, burn = 8, 12
    done_by = {}
    for label in ("consistent", "scrambled"):
        qq = quat.copy()
        if label == "scrambled":
            qq *= np.random.default_rng(0).choice([-1.0, 1.0], size=quat.shape[:2])[..., None]
        q_seq, it_seq, _ = orc.ik_solve(pos, qq, sc, make_items(offs, WORK_ITEM_DTYPE))
        items = np.ascontiguousarray(sched_items(offs, chunk=chunk, burn_in=burn, track=True), dtype=WORK_ITEM_DTYPE)
        n = len(items)
        qf = np.zeros((2 * n, orc.nq))
        q0, it0, _ = orc.ik_solve(pos, qq, sc, items, qpos_final=qf)
        walks = np.ascontiguousarray(plan_walks(items, offs, chunk), dtype=WORK_ITEM_DTYPE)
        _, _, _, done = orc.ik_solve(pos, qq, sc, walks, qpos_init=qf.copy(), qpos_final=qf, want_done=True, out=q0, iters=it0)
        assert np.abs(q0 - q_seq).max() < 1e-6 and np.array_equal(it0, it_seq), label   # (the sign of qpos[3:7] too)
        done_by[label] = int(done.sum())
    q_a, _, _ = orc.ik_solve(pos, quat, sc, make_items(offs, WORK_ITEM_DTYPE))
    assert np.abs(q_a - q_seq).max() < 1e-9   # the sequential run itself does not see the signs
    assert done_by["scrambled"] <= done_by["consistent"] + 2 * chunk


def test_root_target_init_and_height_scale_items():
    """The two per-item knobs of gmr_work_item (gmr_blob.h).  INIT_ROOT_TARGET: the first frame starts from qpos0 with the base on
    the prepared root-task target.  height_scale: the item is solved with the scale table of a model compiled for that height."""
    from gmr_amd._native import INIT_ROOT_TARGET
    cm = compiled("smplx", "unitree_g1")
    orc = Oracle(cm.blob)
    pos, quat, names, offs, _ = synth.synth_clips(cm, 1, 12, seed=3, hard=True, dtype=np.float64)
    sc = cm.slot_columns(names)
    it = make_items([0, 12], WORK_ITEM_DTYPE)
    it["frame_begin"], it["n_out"], it["init_row"] = 4, 8, INIT_ROOT_TARGET
    q_a, _, _ = orc.ik_solve(pos, quat, sc, it)
    tp, tq = orc.prepare_targets(pos[4][sc], quat[4][sc])
    rt = [t for t in cm.tasks[0] if cm.robot.body_index(t.frame) == 0][0]
    init = np.array(cm.robot.qpos0)
    init[:3], init[3:7] = tp[cm.slot_names.index(rt.human)], tq[cm.slot_names.index(rt.human)]
    it["init_row"] = 0
    q_b, _, _ = orc.ik_solve(pos, quat, sc, it, qpos_init=init[None])
    np.testing.assert_array_equal(q_a[4:], q_b[4:])
    # height_scale 1.6 / 1.8 on the default model == a model compiled with actual_human_height = 1.6
    cm16 = compiled("smplx", "unitree_g1", 1.6)
    it = make_items([0, 12], WORK_ITEM_DTYPE)
    it["height_scale"] = 1.6 / cm.config.human_height_assumption
    q_s, n_s, _ = orc.ik_solve(pos, quat, sc, it)
    q_m, n_m, _ = Oracle(cm16.blob).ik_solve(pos, quat, sc, make_items([0, 12], WORK_ITEM_DTYPE))
    assert np.abs(q_s - q_m).max() < 1e-12 and np.array_equal(n_s, n_m)
    q_1, _, _ = orc.ik_solve(pos, quat, sc, make_items([0, 12], WORK_ITEM_DTYPE))
    assert np.abs(q_s - q_1).max() > 1e-3


def test_offset_to_ground_and_height_ratio():
    cm = compiled("bvh", "unitree_g1", 1.6)
    assert abs(cm.ratio - 1.6 / cm.config.human_height_assumption) < 1e-15
    orc = Oracle(cm.blob)
    rng = np.random.default_rng(0)
    hp = rng.normal(size=(cm.nslot, 3))
    hq = rng.normal(size=(cm.nslot, 4))
    tp, tq = orc.prepare_targets(hp, hq, offset_to_ground=True)
    feet = [i for i, n in enumerate(cm.slot_names) if "Foot" in n or "foot" in n]
    assert len(feet) == 2 and abs(tp[feet, 2].min() - 0.1) < 1e-12
    tp0, _ = orc.prepare_targets(hp, hq)
    rs = cm.root_slot
    assert np.allclose(tp0[rs] - synth.qrot(tq[rs], cm.slot_pos_off[rs]), cm.slot_scale[rs] * hp[rs])


def test_reference_error_logs_error1_equals_error2(golden_dir):
    """The only IK-side numbers the reference ships: per-frame error logs of scripts/fbx_to_robot.py:983-1060 (errors.csv,
    test_errors.csv, copied as data to tests/golden/ref_fixtures).  fbx_to_g1.json maps the same frames in both tables, so
    error1 == error2 on every logged row; the restatement must reproduce that identity (SURVEY 8(c), weak pin (i))."""
    import csv
    rows = 0
    for name in ("errors.csv", "test_errors.csv"):
        with open(os.path.join(golden_dir, "ref_fixtures", name)) as f:
            for r in csv.DictReader(f):
                assert r["error1"] == r["error2"]
                rows += 1
    assert rows == 2031
    cm = compiled("fbx", "unitree_g1")
    assert [t.frame for t in cm.tasks[0]] == [t.frame for t in cm.tasks[1]]
    orc = Oracle(cm.blob)
    pos, quat, names, offs, _ = synth.synth_clips(cm, 1, 6, seed=2, hard=True, dtype=np.float64)
    sc = cm.slot_columns(names)
    q = np.array(cm.robot.qpos0, dtype=np.float64)
    for f in range(6):
        q, _, errs = orc.retarget_frame(q, pos[f][sc], quat[f][sc])
        tp, tq = orc.prepare_targets(pos[f][sc], quat[f][sc])
        e1, _ = orc.stage_error(0, q, tp, tq, len(cm.tasks[0]))
        e2, _ = orc.stage_error(1, q, tp, tq, len(cm.tasks[1]))
        assert e1 == e2 and e1 > 0


def _dumped_frame(golden_dir, cm):
    """Frame 0 of one of the fork author's fbx_to_robot.py runs (first_frame_debug.json, written by scripts/fbx_to_robot.py:
    779-788; copied as data): 101 CC_Base joints + the 14 synonym-filled IK names."""
    import json
    with open(os.path.join(golden_dir, "ref_fixtures", "first_frame_debug.json")) as f:
        d = json.load(f)
    hp = np.array([d[s]["pos"] for s in cm.slot_names], dtype=np.float64)
    hq = np.array([d[s]["quat_wxyz"] for s in cm.slot_names], dtype=np.float64)
    return hp, hq


def test_reference_error_logs_plateau(golden_dir):
    """Weak consistency with the reference's error logs -- NOT a pin (tests/golden/make_ik_pin.py documents why no logged row
    can be reproduced: default solver OSQP, source clip absent, free numeric flags).  On the one input frame the reference
    holds, held still, the restated loop settles on an error1 inside the band the 2 031 logged rows span, with the same
    signature: error1 == error2, the first frame spends the whole 22-solve budget, later frames 2-3 solves."""
    import csv
    logged = []
    for name in ("errors.csv", "test_errors.csv"):
        with open(os.path.join(golden_dir, "ref_fixtures", name)) as f:
            logged += [float(r["error1"]) for r in csv.DictReader(f)]
    cm = compiled("fbx", "unitree_g1", 1.75)  # both loaders fall back to 1.75 m on a CC_Base skeleton (lafan1.py:66-69)
    orc = Oracle(cm.blob)
    hp, hq = _dumped_frame(golden_dir, cm)
    tp, tq = orc.prepare_targets(hp, hq)
    q = np.array(cm.robot.qpos0, dtype=np.float64)
    e1s, solves = [], []
    for _ in range(40):
        q, s, _ = orc.retarget_frame(q, hp, hq)
        e1, _ = orc.stage_error(0, q, tp, tq, len(cm.tasks[0]))
        e2, _ = orc.stage_error(1, q, tp, tq, len(cm.tasks[1]))
        assert e1 == e2
        e1s.append(e1)
        solves.append(s)
    assert solves[0] == 22 and max(solves[10:]) <= 3
    assert min(logged) < e1s[-1] < max(logged), (min(logged), e1s[-1], max(logged))
    assert abs(e1s[-1] - e1s[-2]) < 1e-3
    with open(os.path.join(golden_dir, "ik_pin_attempt.json")) as f:
        att = __import__("json").load(f)
    assert abs(att["default_flags_on_dumped_frame"][0][0] - e1s[0]) < 1e-9  # the committed attempt table is this oracle's
    assert min(c[0]["worst_rel_miss"][k] for k, c in att["closest"].items()) > 0.05  # ... and records that nothing matched


def test_planar_base_is_the_reduced_problem():
    """galaxea_r1pro: the oracle decouples the three root dofs the model does not have (gmr_blob.h root_dof_mask).  The step it
    takes must be the solution of the reference's 27-dof problem -- H and c without those rows / columns -- and z, roll, pitch,
    the wheels (no task below them) never move."""
    cm = compiled("smplx", "galaxea_r1pro")
    orc = Oracle(cm.blob)
    pos, quat, names, offs, _ = synth.synth_clips(cm, 1, 40, seed=3, hard=True, dtype=np.float64)
    sc = cm.slot_columns(names)
    q, it, _ = orc.ik_solve(pos, quat, sc, make_items(offs, WORK_ITEM_DTYPE))
    assert np.all(q[:, 2] == cm.robot.body_pos[0, 2]) and not q[:, 4:6].any() and not q[:, 7:13].any()
    assert np.abs(np.linalg.norm(q[:, 3:7], axis=1) - 1).max() < 1e-12 and it.min() >= 1 and it.max() <= 11  # one stage only
    # one solve by hand on the reduced system
    tp, tq = orc.prepare_targets(pos[5][sc], quat[5][sc])
    H, c, lo, hi = orc.build_qp(0, q[4], tp, tq)
    keep = np.array([0, 1, 5] + list(range(6, cm.robot.nv)))
    drop = np.array([2, 3, 4])
    assert not H[np.ix_(drop, keep)].any() and not c[drop].any() and np.all(np.diag(H)[drop] > 0)
    dq, _ = box_qp(H, c, lo, hi)
    dq_red, _ = box_qp(np.ascontiguousarray(H[np.ix_(keep, keep)]), c[keep].copy(), lo[keep].copy(), hi[keep].copy())
    assert not dq[drop].any() and np.abs(dq[keep] - dq_red).max() < 1e-12
