"""GPU parity tests: libgmr_amd.so (through the C ABI / Engine) vs the CPU oracle and the golden vectors.

Tolerances: IK qpos is float64 on both sides but assembled by different formulations (composite blocks
vs dense Jacobians) and a data-dependent loop, so we allow 1e-6 (rad / m) on identical inputs -- three
orders below the 1e-3 rad target of BASELINE.json; FK is float32, tol 2e-6 as in test_oracle.
"""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from gmr_amd import synth  # noqa: E402
from gmr_amd.schedule import make_items  # noqa: E402
from oracle.oracle import IKParams as OParams, Oracle  # noqa: E402
from tests.util import CONFIG_ROBOTS, compiled, quat_angle  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    return torch.device("cuda", 0)


def _engine(cm):
    from gmr_amd.engine import Engine
    return Engine(cm, 0)


def _qpos_diff(a, b):
    return max(np.abs(a[:, :3] - b[:, :3]).max(), quat_angle(a[:, 3:7], b[:, 3:7]).max(), np.abs(a[:, 7:] - b[:, 7:]).max())


@pytest.mark.parametrize("robot", ["unitree_g1", "unitree_g1_with_hands", "booster_t1", "stanford_toddy", "fourier_n1"])
def test_fk_golden(robot, golden_dir, dev):
    cm = compiled("smplx", robot)
    eng = _engine(cm)
    g = np.load(os.path.join(golden_dir, f"fk_{robot}.npz"))
    bp, br = eng.fk(torch.from_numpy(g["root_pos"]).to(dev), torch.from_numpy(g["root_rot"]).to(dev), torch.from_numpy(g["dof_pos"]).to(dev))
    assert np.abs(bp.cpu().numpy() - g["body_pos"]).max() < 2e-6 * max(1.0, np.abs(g["body_pos"]).max())
    assert np.abs(br.cpu().numpy() - g["body_rot"]).max() < 2e-6
    T = g["dof_pos"].shape[0]
    ident = torch.tensor([[0, 0, 0, 1.0]], device=dev).repeat(T, 1)
    bp0, none = eng.fk(torch.zeros(T, 3, device=dev), ident, torch.from_numpy(g["dof_pos"]).to(dev), want_rot=False)
    assert none is None
    assert np.abs(bp0.cpu().numpy() - g["local_body_pos"]).max() < 2e-6


@pytest.mark.parametrize("robot", ["unitree_g1", "unitree_g1_with_hands", "booster_t1"])
def test_fk_golden_wide_inputs(robot, golden_dir, dev):
    """Both FK kernels (fk_pos_kernel for positions only, fk_kernel<0> with rotations) against the reference-generated vectors with
    angles of +-7 rad, exact 0 / +-pi / 2 pi, +-50 m root positions and non-unit root quaternions (make_golden_wide.py)."""
    cm = compiled("smplx", robot)
    eng = _engine(cm)
    g = np.load(os.path.join(golden_dir, f"fk_{robot}_wide.npz"))
    rp, rr, dp = (torch.from_numpy(g[k]).to(dev) for k in ("root_pos", "root_rot", "dof_pos"))
    tol_p, tol_r = 2e-6 * max(1.0, np.abs(g["body_pos"]).max()), 2e-6 * max(1.0, np.abs(g["body_rot"]).max())
    bp, br = eng.fk(rp, rr, dp)
    assert np.abs(bp.cpu().numpy() - g["body_pos"]).max() < tol_p and np.abs(br.cpu().numpy() - g["body_rot"]).max() < tol_r
    bp2, _ = eng.fk(rp, rr, dp, want_rot=False)
    assert np.abs(bp2.cpu().numpy() - g["body_pos"]).max() < tol_p
    assert torch.equal(bp, bp2)  # the two kernels run the same chain


@pytest.mark.parametrize("robot", CONFIG_ROBOTS)
@pytest.mark.parametrize("hard", [False, True])
@pytest.mark.parametrize("qp", ["structured", "generic"])
def test_ik_matches_oracle(robot, hard, qp, dev, monkeypatch):
    """Both linear-algebra back ends of the box QP against the oracle; GMR_AMD_GENERIC_QP is read at model creation."""
    monkeypatch.setenv("GMR_AMD_GENERIC_QP", "1" if qp == "generic" else "0")
    cm = compiled("smplx", robot)
    eng, orc = _engine(cm), Oracle(cm.blob)
    assert (eng.info.reserved[0] == 0) == (qp == "generic")  # core size of the structured layout, 0 = dense generic QP
    pos, quat, names, offs, _ = synth.synth_clips(cm, 3, 40, seed=21, hard=hard, dtype=np.float32)
    sc = cm.slot_columns(names)
    items = make_items(offs)
    q_ref, it_ref, _ = orc.ik_solve(pos, quat, sc, items)
    q, it, _ = eng.ik_solve(torch.from_numpy(pos).to(dev), torch.from_numpy(quat).to(dev), sc, items)
    q, it = q.cpu().numpy(), it.cpu().numpy()
    assert not np.isnan(q).any()
    assert (it >> 30).max() == 0, "a QP hit its iteration cap"
    assert _qpos_diff(q, q_ref) < 1e-6, _qpos_diff(q, q_ref)
    assert np.array_equal(it, it_ref)


@pytest.mark.parametrize("robot", ["kuavo_s45", "hightorque_hi", "booster_k1"])
def test_ik_matches_oracle_other_registry_robots(robot, dev):
    """The registry's remaining humanoids with an IK config (13 / 15 / 12 tasks; booster_k1 has a composite that sums more than
    four child blocks, split over two plan entries)."""
    cm = compiled("smplx", robot)
    eng, orc = _engine(cm), Oracle(cm.blob)
    pos, quat, names, offs, _ = synth.synth_clips(cm, 2, 30, seed=3, hard=True, dtype=np.float32, amp=0.2)
    sc = cm.slot_columns(names)
    q_ref, it_ref, _ = orc.ik_solve(pos, quat, sc, make_items(offs))
    q, it, _ = eng.ik_solve(torch.from_numpy(pos).to(dev), torch.from_numpy(quat).to(dev), sc, make_items(offs))
    assert _qpos_diff(q.cpu().numpy(), q_ref) < 1e-6 and np.array_equal(it.cpu().numpy(), it_ref)


@pytest.mark.parametrize("hard", [False, True])
@pytest.mark.parametrize("qp", ["structured", "generic"])
def test_planar_base_robot_matches_oracle(hard, qp, dev, monkeypatch):
    """galaxea_r1pro: slide x + slide y + hinge z base, one IK stage (table 2 switched off), wheels without
    tasks.  The kernel drops z / roll / pitch from the QP (gmr_blob.h root_dof_mask); result == the oracle's reduced problem."""
    monkeypatch.setenv("GMR_AMD_GENERIC_QP", "1" if qp == "generic" else "0")
    cm = compiled("smplx", "galaxea_r1pro")
    eng, orc = _engine(cm), Oracle(cm.blob)
    assert eng.info.n_active_dof == 6 + 4 + 14 and (eng.info.reserved[0] == 0) == (qp == "generic")
    pos, quat, names, offs, _ = synth.synth_clips(cm, 3, 60, seed=8, hard=hard, dtype=np.float32)
    sc = cm.slot_columns(names)
    from gmr_amd._native import INIT_ROOT_TARGET
    for items in (make_items(offs), make_items(offs, clip_init=INIT_ROOT_TARGET)):
        q_ref, it_ref, _ = orc.ik_solve(pos, quat, sc, items)
        q, it, _ = eng.ik_solve(torch.from_numpy(pos).to(dev), torch.from_numpy(quat).to(dev), sc, items)
        q, it = q.cpu().numpy(), it.cpu().numpy()
        assert (it >> 30).max() == 0 and _qpos_diff(q, q_ref) < 1e-6 and np.array_equal(it, it_ref)
        assert np.all(q[:, 2] == cm.robot.body_pos[0, 2]) and not q[:, 4:6].any() and not q[:, 7:13].any()  # z, roll / pitch, wheels


@pytest.mark.parametrize("prm", [dict(damping=0.05), dict(damping=4.0, max_iter=3), dict(tol=1e-5, max_iter=20), dict(limit_gain=0.5, lm_damping=0.1),
                                 dict(max_iter=0)])
def test_solver_constants_are_the_callers(prm, dev):
    """gmr_ik_params: damping is a constructor argument of the reference class (motion_retarget.py:19), max_iter an attribute
    callers set (:56); tol, limit_gain and lm_damping are the values the reference hard-codes.  Every one of them reaches the
    kernel: non-default values against the oracle with the same values, on inputs that reach joint limits."""
    from gmr_amd.engine import IKParams
    cm = compiled("smplx", "unitree_g1")
    eng, orc = _engine(cm), Oracle(cm.blob)
    pos, quat, names, offs, _ = synth.synth_clips(cm, 3, 30, seed=17, hard=True, dtype=np.float32)
    sc = cm.slot_columns(names)
    q_ref, it_ref, _ = orc.ik_solve(pos, quat, sc, make_items(offs), params=OParams(**prm))
    q, it, _ = eng.ik_solve(torch.from_numpy(pos).to(dev), torch.from_numpy(quat).to(dev), sc, make_items(offs), params=IKParams(**prm))
    q, it = q.cpu().numpy(), it.cpu().numpy()
    assert (it >> 30).max() == 0 and _qpos_diff(q, q_ref) < 1e-6 and np.array_equal(it, it_ref)
    q_def, it_def, _ = orc.ik_solve(pos, quat, sc, make_items(offs))
    assert _qpos_diff(q_ref, q_def) > 1e-4 or not np.array_equal(it_ref, it_def)  # the variation is not a no-op
    if "max_iter" in prm:
        assert it.max() <= 2 * (1 + prm["max_iter"])


def _synthetic_robot(tmp_path, limbs, with_tasks_per_limb, jrange="-1.2 1.4", table2_reversed=False):
    """A floating base with `limbs` chains of hinges (list of chain lengths) hanging off it; tasks on the base and on every
    `with_tasks_per_limb`-th link of each chain.  Returns a compiled model."""
    from gmr_amd.ik_config import IKConfig, IKTask
    from gmr_amd.mjcf import load_mjcf
    from gmr_amd.model import compile_model
    axes = ["1 0 0", "0 1 0", "0 0 1"]
    xml = ['<mujoco model="synth"><compiler angle="radian"/><worldbody><body name="base" pos="0 0 1"><freejoint/>']
    tasks = [("base", "h_base")]
    for li, n in enumerate(limbs):
        ang = 2 * np.pi * li / len(limbs)
        for k in range(n):
            pos = f"{0.15 * np.cos(ang):.4f} {0.15 * np.sin(ang):.4f} 0" if k == 0 else "0.02 0.01 -0.12"
            xml.append(f'<body name="l{li}_{k}" pos="{pos}"><joint name="j{li}_{k}" axis="{axes[(k + li) % 3]}" range="{jrange}"/>')
            if (k + 1) % with_tasks_per_limb == 0 or k == n - 1:
                tasks.append((f"l{li}_{k}", f"h{li}_{k}"))
        xml.append("</body>" * n)
    xml.append("</body></worldbody></mujoco>")
    p = tmp_path / "synth.xml"
    p.write_text("".join(xml))
    robot = load_mjcf(str(p))
    t1 = [IKTask(f, h, 0.0 if i % 3 else 50.0, 10.0, [0.0, 0.0, 0.0], [1.0, 0.0, 0.0, 0.0]) for i, (f, h) in enumerate(tasks)]
    t2 = [IKTask(f, h, 10.0, 5.0, [0.0, 0.0, 0.0], [1.0, 0.0, 0.0, 0.0]) for f, h in (tasks[::-1] if table2_reversed else tasks)]
    cfg = IKConfig("base", "h_base", 0.0, 1.8, True, True, {h: 1.0 for _, h in tasks}, t1, t2, source="synthetic")
    return compile_model(robot, cfg)


@pytest.mark.parametrize("limbs,every,expect_nvp,expect_struct", [
    ([9, 9, 8, 8], 3, 40, True),          # 40 dofs: four limbs of <= 9 behind a 6-dof core -> structured, NVP 40
    ([7, 7, 7, 7, 7, 7], 2, 48, False),   # 48 dofs, six limbs: more than four bins -> dense generic QP, NVP 48; 19 tasks (> 16: one lane per task)
    ([14, 14, 14, 14], 4, 64, False),     # 62 dofs: limbs too long for a 16-wide group -> generic, NVP 64, deepest tree
])
def test_synthetic_large_robots(limbs, every, expect_nvp, expect_struct, dev, tmp_path):
    """Kernel variants the registry robots never reach (NVP 40 / 48 / 64, dense QP on big systems, > 16 tasks), against the oracle."""
    cm = _synthetic_robot(tmp_path, limbs, every)
    eng, orc = _engine(cm), Oracle(cm.blob)
    assert eng.info.nv_padded == expect_nvp and (eng.info.reserved[0] > 0) == expect_struct, (eng.info.nv_padded, eng.info.reserved[0])
    pos, quat, names, offs, _ = synth.synth_clips(cm, 2, 16, seed=4, hard=True, dtype=np.float64, amp=0.2)
    sc = cm.slot_columns(names)
    q_ref, it_ref, _ = orc.ik_solve(pos, quat, sc, make_items(offs))
    q, it, _ = eng.ik_solve(torch.from_numpy(pos).to(dev), torch.from_numpy(quat).to(dev), sc, make_items(offs))
    assert (it.cpu().numpy() >> 30).max() == 0
    assert _qpos_diff(q.cpu().numpy(), q_ref) < 1e-6 and np.array_equal(it.cpu().numpy(), it_ref)


def test_joint_angles_beyond_pi(dev, tmp_path):
    """Hinges that turn past +-3.2 rad: the FK's short sin/cos path (no range reduction) must hand over to the general one."""
    cm = _synthetic_robot(tmp_path, [5, 5, 4, 4], 2, jrange="-6.0 6.0")
    eng, orc = _engine(cm), Oracle(cm.blob)
    pos, quat, names, offs, q_true = synth.synth_clips(cm, 2, 24, seed=6, hard=False, dtype=np.float64, amp=0.4)
    sc = cm.slot_columns(names)
    q_ref, it_ref, _ = orc.ik_solve(pos, quat, sc, make_items(offs))
    assert np.abs(q_ref[:, 7:]).max() > 3.3, "the case must actually leave the short path's range"
    q, it, _ = eng.ik_solve(torch.from_numpy(pos).to(dev), torch.from_numpy(quat).to(dev), sc, make_items(offs))
    assert _qpos_diff(q.cpu().numpy(), q_ref) < 1e-6 and np.array_equal(it.cpu().numpy() & 0x3FFFFFFF, it_ref)


def test_tables_with_different_task_order(dev, tmp_path):
    """Stage 2 may inherit stage 1's residual only when both tables list the same (body, target) per task; here table 2 lists
    them in reverse order, so the kernel has to evaluate the residual again at the entry of stage 2."""
    cm = _synthetic_robot(tmp_path, [5, 5, 4, 4], 2, table2_reversed=True)
    eng, orc = _engine(cm), Oracle(cm.blob)
    pos, quat, names, offs, _ = synth.synth_clips(cm, 2, 24, seed=9, hard=True, dtype=np.float64, amp=0.2)
    sc = cm.slot_columns(names)
    q_ref, it_ref, _ = orc.ik_solve(pos, quat, sc, make_items(offs))
    q, it, _ = eng.ik_solve(torch.from_numpy(pos).to(dev), torch.from_numpy(quat).to(dev), sc, make_items(offs))
    assert _qpos_diff(q.cpu().numpy(), q_ref) < 1e-6 and np.array_equal(it.cpu().numpy() & 0x3FFFFFFF, it_ref)


def test_reference_held_frame(dev, golden_dir):
    """GPU twin of tests/test_oracle.py::test_reference_error_logs_plateau: the one IK input frame the reference holds
    (first_frame_debug.json, fbx_to_g1.json, 1.75 m), held still for 40 frames from qpos0 -- far targets (error1 ~ 3, shoulder
    rotation errors of 1.8 rad), the whole 22-solve budget on frame 0, active joint limits.  HIP path == oracle."""
    from tests.test_oracle import _dumped_frame
    cm = compiled("fbx", "unitree_g1", 1.75)
    eng, orc = _engine(cm), Oracle(cm.blob)
    hp, hq = _dumped_frame(golden_dir, cm)
    pos = np.repeat(hp[None], 40, axis=0)
    quat = np.repeat(hq[None], 40, axis=0)
    sc = np.arange(cm.nslot, dtype=np.int32)
    items = make_items(np.array([0, 40]))
    q_ref, it_ref, _ = orc.ik_solve(pos, quat, sc, items)
    q, it, _ = eng.ik_solve(torch.from_numpy(pos).to(dev), torch.from_numpy(quat).to(dev), sc, items)
    q, it = q.cpu().numpy(), it.cpu().numpy()
    assert it_ref[0] == 22 and (it >> 30).max() == 0
    assert _qpos_diff(q, q_ref) < 1e-6 and np.array_equal(it, it_ref)
