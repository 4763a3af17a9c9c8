"""CPU-only tests of the host logic: MJCF/IK-config compilers, packs, blob, scheduling, C-ABI exports."""
import ctypes
import os
import re

import numpy as np
import pytest

from gmr_amd import params
from gmr_amd.ik_config import load_ik_config
from gmr_amd.mjcf import MjcfError, load_mjcf, load_robot
from gmr_amd.model import HEADER_BYTES
from gmr_amd.schedule import make_items, partition_clips
from tests.util import CONFIG_ROBOTS, compiled

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_size_matches_c_struct():
    # 4 u32 + 12 i32 + 23 u32 + 3 u32 = 42 words
    assert HEADER_BYTES == 42 * 4


@pytest.mark.parametrize("robot", CONFIG_ROBOTS)
def test_expected_sizes(robot):
    exp = {"unitree_g1": (38, 36, 35), "unitree_g1_with_hands": (52, 50, 49), "booster_t1": (32, 28, 27),
           "stanford_toddy": (33, 29, 28), "fourier_n1": (29, 30, 29), "engineai_pm01": (29, 31, 30)}[robot]
    rob = compiled("smplx", robot).robot
    assert (rob.nbody, rob.nq, rob.nv) == exp
    assert rob.jnt_limited[rob.hinge_bodies()].all()
    assert np.allclose(np.linalg.norm(rob.body_quat, axis=1), 1.0)
    assert len(compiled("smplx", robot).tasks[0]) == 14


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
@pytest.mark.parametrize("robot", sorted(params._ROBOT_XML_REL))
def test_packs_are_fresh(robot, monkeypatch):
    """The shipped packs equal what the MJCF compiler produces from the reference's XML today."""
    monkeypatch.delenv("GMR_ROOT", raising=False)
    pack = load_robot(params.ROBOT_XML_DICT[robot])
    xml = load_mjcf(os.path.join(REF, "assets", params._ROBOT_XML_REL[robot]), name=robot)
    assert pack.body_names == xml.body_names
    for f in ("parent", "body_pos", "body_quat_raw", "jnt_type", "jnt_axis_raw", "jnt_range", "jnt_limited", "qpos_adr"):
        np.testing.assert_array_equal(getattr(pack, f), getattr(xml, f))
    assert pack.root_dofs == xml.root_dofs and pack.root_jnt_names == xml.root_jnt_names


def test_planar_base_model_and_qpos_layout(tmp_path):
    """galaxea_r1pro (assets/galaxea_r1pro/r1_pro.xml:102-104): slide x + slide y + hinge z on the root body is carried as a
    free-joint root whose z / roll / pitch never move; MuJoCo's own layout [x, y, yaw, hinges] is a view of it."""
    from gmr_amd.mjcf import ROOT_DOFS_PLANAR
    rob = compiled("smplx", "galaxea_r1pro").robot
    assert rob.planar_base and rob.root_dofs == ROOT_DOFS_PLANAR and rob.root_jnt_names == ["base_x", "base_y", "base_yaw"]
    assert (rob.nbody, rob.nq, rob.nv, rob.mj_nq, rob.mj_nv) == (25, 31, 30, 27, 27)
    assert not rob.jnt_limited[rob.body_index("wheel_motor_link1")] and rob.jnt_limited[rob.body_index("steer_motor_link1")]
    cm = compiled("smplx", "galaxea_r1pro")
    assert len(cm.tasks[0]) == 10 and len(cm.tasks[1]) == 0  # table 2 is switched off (and names another robot's links)
    rng = np.random.default_rng(0)
    mj = np.concatenate([rng.normal(size=(50, 2)), rng.uniform(-3.1, 3.1, (50, 1)), rng.normal(size=(50, 24))], axis=1)
    q = rob.from_mj_qpos(mj)
    assert q.shape == (50, 31) and np.all(q[:, 2] == rob.body_pos[0, 2]) and not q[:, 4:6].any()
    np.testing.assert_allclose(rob.to_mj_qpos(q), mj, atol=1e-12)
    import torch
    np.testing.assert_allclose(rob.to_mj_qpos(torch.from_numpy(q)).numpy(), mj, atol=1e-12)
    # the hinge coordinate accumulates: the branch nearest to the previous frame's value
    far = rob.to_mj_qpos(rob.from_mj_qpos(np.array([[0.0, 0.0, 3.0] + [0.0] * 24])), yaw_ref=np.array([3.0 + 4 * np.pi]))
    assert abs(far[0, 2] - (3.0 + 4 * np.pi)) < 1e-12
    # anything else on a root body is still refused
    p = tmp_path / "m.xml"
    p.write_text('<mujoco><compiler angle="radian"/><worldbody><body name="a"><joint type="slide" axis="1 0 0"/>'
                 '<joint type="slide" axis="0 1 0"/><joint type="hinge" axis="0 1 0"/></body></worldbody></mujoco>')
    with pytest.raises(MjcfError):
        load_mjcf(str(p))


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
def test_gmr_root_switches_registry(monkeypatch):
    monkeypatch.setenv("GMR_ROOT", REF)
    assert str(params.ROBOT_XML_DICT["unitree_g1"]).endswith("g1_mocap_29dof.xml")
    assert str(params.IK_CONFIG_DICT["bvh"]["unitree_g1"]).endswith("bvh_to_g1.json")
    a = load_ik_config(params.IK_CONFIG_DICT["smplx"]["engineai_pm01"])
    monkeypatch.delenv("GMR_ROOT")
    assert str(params.ROBOT_XML_DICT["unitree_g1"]).endswith("unitree_g1.json")
    b = load_ik_config(params.IK_CONFIG_DICT["smplx"]["engineai_pm01"])
    assert a.to_dict()["tables"] == b.to_dict()["tables"] and a.human_scale_table == b.human_scale_table


def test_registry_keyerrors():
    with pytest.raises(KeyError):
        params.ROBOT_XML_DICT["no_such_robot"]
    with pytest.raises(KeyError):
        params.IK_CONFIG_DICT["smplx"]["no_such_robot"]
    with pytest.raises(KeyError):
        params.IK_CONFIG_DICT["nope"]


def test_unsupported_model_is_rejected(tmp_path):
    p = tmp_path / "m.xml"
    p.write_text('<mujoco><compiler angle="radian"/><worldbody><body name="a"><freejoint/>'
                 '<body name="b"><joint type="slide" axis="0 0 1"/></body></body></worldbody></mujoco>')
    with pytest.raises(MjcfError):
        load_mjcf(str(p))


def test_mjcf_defaults_degrees_and_comments(tmp_path):
    p = tmp_path / "m.xml"
    p.write_text(
        '<mujoco model="t"><compiler angle="degree"/><default><joint axis="0 1 0"/><default class="k"><joint range="-90 90"/></default></default>'
        '<worldbody><body name="root" pos="0 0 1"><joint type="free"/>'
        '<!-- <joint name="ghost"/> -->'
        '<body name="a" childclass="k" pos="0 0 -0.5" quat="2 0 0 0"><joint name="ja"/>'
        '<body name="b"><joint name="jb" class="main" axis="3 0 0" range="-10 20"/></body></body>'
        '<body name="c"/></body></worldbody><worldbody><light/></worldbody></mujoco>')
    m = load_mjcf(str(p))
    assert m.body_names == ["root", "a", "b", "c"] and m.parent.tolist() == [-1, 0, 1, 0]
    assert (m.nq, m.nv) == (9, 8)
    np.testing.assert_allclose(m.jnt_range[1], np.deg2rad([-90, 90]))
    np.testing.assert_allclose(m.jnt_axis[1], [0, 1, 0])
    np.testing.assert_allclose(m.jnt_axis[2], [1, 0, 0])
    np.testing.assert_allclose(m.jnt_range[2], np.deg2rad([-10, 20]))
    np.testing.assert_allclose(m.body_quat[1], [1, 0, 0, 0])
    np.testing.assert_allclose(m.body_quat_raw[1], [2, 0, 0, 0])
    np.testing.assert_allclose(m.qpos0[:7], [0, 0, 1, 1, 0, 0, 0])


def test_reference_quirks_in_compiled_model():
    cm = compiled("smplx", "engineai_pm01")
    cfg = cm.config
    # quirk 1: table-2 offsets exist in the config but only table-1 offsets reach the blob (motion_retarget.py:121)
    t1 = {t.frame: t for t in cfg.table1}
    t2 = {t.frame: t for t in cfg.table2}
    assert t1["LINK_ELBOW_PITCH_L"].rot_offset != t2["LINK_ELBOW_PITCH_L"].rot_offset
    s = cm.slot_names.index(t1["LINK_ELBOW_PITCH_L"].human)
    q = np.array(t1["LINK_ELBOW_PITCH_L"].rot_offset)
    np.testing.assert_allclose(cm.slot_rot_off[s], q / np.linalg.norm(q))
    # height ratio scales every scale-table entry
    cm2 = compiled("smplx", "engineai_pm01", 1.6)
    np.testing.assert_allclose(cm2.slot_scale, cm.slot_scale * 1.6 / cfg.human_height_assumption)


def test_slot_columns_keyerrors():
    cm = compiled("smplx", "unitree_g1")
    names = list(cm.slot_names)
    cols = cm.slot_columns(names + ["jaw", "left_eye"])
    assert cols.tolist() == list(range(len(names)))
    with pytest.raises(KeyError):
        cm.slot_columns([n for n in names if n != "pelvis"])
    with pytest.raises(KeyError):
        cm.slot_columns([n for n in names if n != "left_wrist"])


def test_make_items_and_partition():
    it = make_items([0, 5, 5, 12])
    assert it["frame_begin"].tolist() == [0, 5] and it["n_out"].tolist() == [5, 7] and it["n_burn"].tolist() == [0, 0]
    it = make_items([0, 10], chunk=4, burn_in=3)
    assert it["frame_begin"].tolist() == [0, 1, 5] and it["n_burn"].tolist() == [0, 3, 3] and it["n_out"].tolist() == [4, 4, 2]
    tr = make_items([0, 10], chunk=4, burn_in=3, track=True)
    assert tr["final_row"].tolist() == [0, 1, 2] and tr["burn_row"].tolist() == [3, 4, 5] and it["burn_row"].tolist() == [-1, -1, -1]
    covered = np.concatenate([np.arange(r["frame_begin"] + r["n_burn"], r["frame_begin"] + r["n_burn"] + r["n_out"]) for r in it])
    assert covered.tolist() == list(range(10))
    with pytest.raises(ValueError):
        make_items([0, 5, 3])
    parts = partition_clips([10, 3, 8, 8, 1, 7], 3)
    assert sorted(sum(parts, [])) == list(range(6))
    loads = [sum([10, 3, 8, 8, 1, 7][i] for i in p) for p in parts]
    assert max(loads) - min(loads) <= 4 and max(loads) <= 15


def test_c_abi_library_exports_every_declared_symbol():
    """libgmr_amd.so loads (hipcc cross-compiled, no GPU needed) and exports every function in include/gmr_amd.h."""
    from gmr_amd import _native
    from gmr_amd.build import build_lib
    build_lib()
    with open(os.path.join(ROOT, "include", "gmr_amd.h")) as f:
        src = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    declared = sorted(set(re.findall(r"\b(gmr_[a-z_]+)\s*\(", src)))
    assert set(declared) == set(_native.EXPORTS)
    lib = _native.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.gmr_abi_version() == _native.ABI_VERSION == 5
    assert ctypes.sizeof(_native.IKParams) == 48 and _native.WORK_ITEM_DTYPE.itemsize == 40


def test_model_create_fails_loudly_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from gmr_amd import _native
    lib = _native.load()
    cm = compiled("smplx", "unitree_g1")
    err = ctypes.create_string_buffer(256)
    h = lib.gmr_model_create(cm.blob, len(cm.blob), 0, err, len(err))
    assert not h and b"device" in err.value.lower()
    bad = bytearray(cm.blob)
    bad[0] ^= 0xFF
    h = lib.gmr_model_create(bytes(bad), len(bad), 0, err, len(err))
    assert not h and b"magic" in err.value
    from gmr_amd.engine import Engine, EngineError
    with pytest.raises(EngineError):
        Engine(cm, 0)


def test_bvh_parser_follows_reference_file_semantics(golden_dir, tmp_path):
    from gmr_amd.bvh import read_bvh
    for name, nj in (("bvh_canonical_40f", 101), ("bvh_lafan_like", 22), ("bvh_pruned_mid_24f", 87)):
        a = read_bvh(os.path.join(golden_dir, name + ".bvh"))
        g = np.load(os.path.join(golden_dir, name + ".npz"))
        assert a.bones == [str(n) for n in g["names"]][:nj] and a.order == (2, 1, 0)  # "Zrotation Yrotation Xrotation"
        assert a.parents[0] == -1 and np.all(a.parents[1:] < np.arange(1, nj)) and a.pos.shape == a.eulers_deg.shape == (len(g["pos"]), nj, 3)
        np.testing.assert_array_equal(a.pos[:, 1:], np.repeat(a.offsets[None, 1:], a.pos.shape[0], axis=0))  # non-root = joint offsets
        # root translation (cm, Y-up) -> golden root position (m, Z-up): (x, y, z) -> (x, -z, y) / 100
        np.testing.assert_allclose(np.stack([a.pos[:, 0, 0], -a.pos[:, 0, 2], a.pos[:, 0, 1]], -1) / 100, g["pos"][:, 0], atol=1e-12)
    # the legacy 9-channel layout (extract.py:152-156): positions-only root, (position, rotation, scale) per joint; the Euler
    # order comes from the first joint that has rotations, the root keeps a zero rotation
    a = read_bvh(os.path.join(golden_dir, "bvh_nine_channel.bvh"))
    g = np.load(os.path.join(golden_dir, "bvh_nine_channel.npz"))
    assert a.bones == [str(n) for n in g["names"]][:22] and a.order == (2, 1, 0) and a.pos.shape == a.eulers_deg.shape == (12, 22, 3)
    assert not a.eulers_deg[:, 0].any() and a.eulers_deg[:, 1:].any() and np.abs(a.pos[:, 1:] - a.offsets[None, 1:]).max() > 0.1
    np.testing.assert_allclose(np.stack([a.pos[:, 0, 0], -a.pos[:, 0, 2], a.pos[:, 0, 1]], -1) / 100, g["pos"][:, 0], atol=1e-12)
    bad = tmp_path / "bad.bvh"
    bad.write_text("HIERARCHY\nROOT a\n{\nOFFSET 0 0 0\nCHANNELS 6 Xposition Yposition Zposition Zrotation Yrotation Xrotation\n}\nMOTION\nFrames: 2\nFrame Time: 0.03\n0 0 0 0 0 0\n")
    with pytest.raises(ValueError):
        read_bvh(str(bad))


def test_bvh_motion_parser_matches_python_float():
    """gmr_bvh_parse_motion (host code of libgmr_amd.so): every token is the double Python's float() gives -- the reference's
    read_bvh does float() per regex group (utils/lafan_vendor/extract.py:140-166) -- including the cases outside the fast path."""
    import ctypes as C
    from gmr_amd import _native
    lib = _native.load()
    rng = np.random.default_rng(0)
    vals = np.concatenate([rng.normal(size=400) * 10.0 ** rng.integers(-8, 9, size=400), [0.0, -0.0, 1e22, 1e23, 5e-324, 1.7976931348623157e308]])
    toks = []
    for k, v in enumerate(vals):
        toks.append([repr(float(v)), f"{v:.6f}", f"{v:.17g}", f"{v:+.3e}"][k % 4])
    toks += ["123456789012345678901234567890", ".5", "-5.", "1E+5", "9007199254740993", "0.1", "1e400", "-1e400", "nan", "007.250"]
    ncol = 8
    toks = toks[: len(toks) // ncol * ncol]
    lines = [" ".join(toks[i:i + ncol]) for i in range(0, len(toks), ncol)]
    text = ("\n\n" + "\r\n".join(lines[:10]) + "\n  \t\n" + "\n".join(lines[10:]) + "\n").encode()
    out = np.empty(len(toks) + 4)
    nl, nc = C.c_int64(), C.c_int64()
    n = lib.gmr_bvh_parse_motion(text, len(text), len(lines), out.ctypes.data, len(out), C.byref(nl), C.byref(nc))
    assert n == len(toks) and nl.value == len(lines) and nc.value == ncol
    exp = np.array([float(t) for t in toks])
    assert np.array_equal(out[:n], exp, equal_nan=True) and np.array_equal(np.signbit(out[:n]), np.signbit(exp))
    # max_lines stops early; ragged rows, garbage and overflow are errors
    assert lib.gmr_bvh_parse_motion(text, len(text), 3, out.ctypes.data, len(out), C.byref(nl), C.byref(nc)) == 3 * ncol and nl.value == 3
    for bad in (b"1 2 3\n4 5\n", b"1 2x 3\n", b"1 - 3\n", b"1 2 3 4 5 6 7 8 9\n" * 2):
        cap = 8 if bad.startswith(b"1 2 3 4") else len(out)
        assert lib.gmr_bvh_parse_motion(bad, len(bad), 10, out.ctypes.data, cap, C.byref(nl), C.byref(nc)) == -1
    # tokens off the fast path go to strtod -- but only those float() reads the same way: strtod alone would take hexadecimal floats
    # and nan(...) (found by tools/fuzz_adapters.py's hostile text: "0x10" came back as 16.0 where the reference raises)
    for tok in ("0x10", "0X1p3", "nan(1)", "1e", "1e+", "+", ".", "1..2", "1e5e5", "infinit", "1_0"):
        bad = f"1.5 {tok} 2\n".encode()
        assert lib.gmr_bvh_parse_motion(bad, len(bad), 10, out.ctypes.data, len(out), C.byref(nl), C.byref(nc)) == -1, tok
        if tok != "1_0":   # (float() takes digit-group underscores; refused here: an error, never another number)
            with pytest.raises(ValueError):
                float(tok)
    for tok in ("inf", "-Infinity", "+INF", "NaN", "-nan", "1e999", "-1e-999", "0." + "0" * 80 + "1", "+.5e+2", "5.E3"):
        good = f"{tok} 1\n".encode()
        assert lib.gmr_bvh_parse_motion(good, len(good), 10, out.ctypes.data, len(out), C.byref(nl), C.byref(nc)) == 2, tok
        assert np.array_equal(out[:1], [float(tok)], equal_nan=True) and np.signbit(out[0]) == np.signbit(float(tok))


def test_bvh_header_tokenizer_grammar(tmp_path):
    """gmr_bvh_parse_header (gmr_amd/csrc/bvh_text.h): the token grammar accepts any line structure, keeps the reference's file
    semantics (joint order, End Sites ignored, Euler order from the first joint with a rotation triple, `\\w+` names) and rejects
    what it cannot lay out -- including truncated and garbage input."""
    from gmr_amd.bvh import read_bvh, _parse_header
    one_line = ("HIERARCHY ROOT Hips { OFFSET 0 1.5 -2e-1 CHANNELS 6 Xposition Yposition Zposition Zrotation Yrotation Xrotation "
                "JOINT mixamorig:Spine { OFFSET 0 10 0 CHANNELS 3 Zrotation Yrotation Xrotation End Site { OFFSET 0 5 0 } } "
                "JOINT Leg_L { OFFSET 1 0 0 CHANNELS 3 Zrotation Yrotation Xrotation JOINT Foot { OFFSET 0 -40 0 CHANNELS 3 Zrotation Yrotation Xrotation "
                "End Site { OFFSET 0 0 10 } } } } MOTION Frames: 2 Frame Time: 0.0333333\n"
                "1 2 3 10 20 30 1 2 3 4 5 6 7 8 9\n-1 -2 -3 0 0 0 0 0 0 0 0 0 0 0 0\n")
    p = tmp_path / "one_line.bvh"
    p.write_text(one_line)
    names, parents, offsets, chan, order, fnum, ftime, moff = _parse_header(p.read_bytes(), str(p))
    assert names == ["Hips", "mixamorig", "Leg_L", "Foot"] and list(parents) == [-1, 0, 0, 2]      # `\\w+` capture, End Sites skipped
    assert list(chan) == [6, 3, 3, 3] and order == (2, 1, 0) and fnum == 2 and abs(ftime - 0.0333333) < 1e-12
    np.testing.assert_array_equal(offsets, [[0, 1.5, -0.2], [0, 10, 0], [1, 0, 0], [0, -40, 0]])
    assert p.read_bytes()[moff:moff + 5] == b"1 2 3"
    a = read_bvh(str(p))
    assert a.pos.shape == (2, 4, 3) and np.array_equal(a.pos[1, 0], [-1, -2, -3]) and np.array_equal(a.eulers_deg[0, 3], [7, 8, 9])
    xyz = one_line.replace("Zrotation Yrotation Xrotation JOINT mixamorig", "Xrotation Yrotation Zrotation JOINT mixamorig", 1)
    p.write_text(xyz)
    assert read_bvh(str(p)).order == (0, 1, 2)                                                       # Euler order follows the first joint
    bad_cases = {
        "truncated": one_line[:200],
        "no hierarchy": one_line.replace("HIERARCHY", "HIERARCHIE"),
        "unbalanced": one_line.replace("} } } } MOTION", "} } } MOTION"),
        "bad offset": one_line.replace("OFFSET 0 10 0", "OFFSET 0 ten 0"),
        "bad channel": one_line.replace("Zrotation Yrotation Xrotation JOINT Foot", "Zrotation Yrotation Wrotation JOINT Foot"),
        "root without rotations": one_line.replace("CHANNELS 6 Xposition Yposition Zposition Zrotation Yrotation Xrotation", "CHANNELS 3 Xposition Yposition Zposition"),
        "no frame time": one_line.replace("Frame Time:", "FrameTime:"),
        "binary": "\x00\x01\x02 HIERARCHY",
    }
    for label, text in bad_cases.items():
        p.write_text(text)
        with pytest.raises((ValueError, NotImplementedError)):
            read_bvh(str(p))
    nine = one_line.replace("CHANNELS 3 Zrotation Yrotation Xrotation JOINT Foot", "CHANNELS 9 Xposition Yposition Zposition Zrotation Yrotation Xrotation Xposition Yposition Zposition JOINT Foot")
    p.write_text(nine)
    with pytest.raises(NotImplementedError):                                                         # mixed channel counts: rejected, not mis-read
        read_bvh(str(p))
    # many joints: the name / joint capacity grows
    deep = "HIERARCHY ROOT j0 { OFFSET 0 0 0 CHANNELS 6 Xposition Yposition Zposition Zrotation Yrotation Xrotation " + \
        "".join(f"JOINT j{i} {{ OFFSET 0 1 0 CHANNELS 3 Zrotation Yrotation Xrotation " for i in range(1, 200)) + "End Site { OFFSET 0 1 0 } " + "} " * 200 + \
        "MOTION\nFrames: 1\nFrame Time: 0.01\n" + " ".join(["0"] * (3 + 3 * 200)) + "\n"
    p.write_text(deep)
    a = read_bvh(str(p))
    assert len(a.bones) == 200 and a.bones[199] == "j199" and list(a.parents[:3]) == [-1, 0, 1]


def test_c_abi_headers_are_plain_c_and_the_c_example_links(tmp_path):
    """include/*.h compile as strict C99 on their own, and examples/c_abi_retarget.c (the boundary used without Python or torch)
    compiles and links against libgmr_amd.so with nothing but the HIP runtime."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    for hdr in ("gmr_blob.h", "gmr_amd.h"):
        src = tmp_path / f"use_{hdr}.c"
        src.write_text(f'#include "{hdr}"\nint main(void) {{ return (int)sizeof(gmr_work_item) - 40; }}\n')
        subprocess.check_call([gcc, "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", f"-I{root}/include", "-c", str(src), "-o", str(tmp_path / "h.o")])
    if not os.path.isdir("/opt/rocm/include"):
        pytest.skip("no ROCm headers")
    from gmr_amd import _native
    _native.load()
    libdir = os.path.dirname(_native.LIB_PATH)
    for name in ("c_abi_retarget", "c_abi_bvh_file"):   # (the second: the BVH file path through the round-3 entry points)
        exe = tmp_path / name
        subprocess.check_call([gcc, "-std=c99", "-Wall", "-Wextra", "-Werror", "-D__HIP_PLATFORM_AMD__", f"-I{root}/include", "-isystem", "/opt/rocm/include",
                               os.path.join(root, "examples", name + ".c"), f"-L{libdir}", "-lgmr_amd", "-L/opt/rocm/lib", "-lamdhip64",
                               f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
        assert exe.exists()


def test_make_items_matches_the_plain_loop():
    """schedule.make_items / plan_walks are vectorised (a LAFAN1-sized chunk plan is ~7 000 rows per call); this is the loop they
    replace, on random clip sets with empty clips, ragged last chunks, burn-in clipped at the clip start and per-clip heights."""
    from gmr_amd.schedule import make_items, plan_walks
    from gmr_amd._native import WORK_ITEM_DTYPE
    rng = np.random.default_rng(5)
    for case in range(40):
        n = int(rng.integers(1, 12))
        lens = rng.integers(0, 200, n)
        lens[rng.integers(0, n)] = 0
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        hs = rng.uniform(0.8, 1.2, n)
        chunk, burn = int(rng.choice([0, 1, 7, 16, 64])), int(rng.choice([0, 3, 24, 100]))
        ci, ki = int(rng.choice([-1, -2])), int(rng.choice([-1, -2]))
        rows = []
        for c in range(n):
            a, b = int(offs[c]), int(offs[c + 1])
            if a == b:
                continue
            if chunk <= 0:
                rows.append((a, 0, b - a, ci, -1, -1, 0, hs[c]))
                continue
            for start in range(a, b, chunk):
                bu = min(burn, start - a)
                rows.append((start - bu, bu, min(chunk, b - start), ci if start == a else ki, -1, -1, 0, hs[c]))
        ref = np.array(rows, dtype=WORK_ITEM_DTYPE) if rows else np.zeros(0, dtype=WORK_ITEM_DTYPE)
        got = make_items(offs, chunk=chunk, burn_in=burn, height_scales=hs, chunk_init=ki, clip_init=ci)
        assert got.dtype == ref.dtype and got.tobytes() == ref.tobytes(), case
        if chunk > 0 and len(ref):
            tr = make_items(offs, chunk=chunk, burn_in=burn, track=True, height_scales=hs, chunk_init=ki, clip_init=ci)
            k = len(tr)
            assert np.array_equal(tr["final_row"], np.arange(k)) and np.array_equal(tr["burn_row"], k + np.arange(k))
            walks = plan_walks(tr, offs, chunk)
            ob = tr["frame_begin"] + tr["n_burn"]
            exp = []
            first = [i for i in range(k) if tr["n_burn"][i] == 0 and ob[i] in set(offs[:-1].tolist())]
            for j, c0 in enumerate(first):
                c1 = first[j + 1] if j + 1 < len(first) else k
                if c1 - c0 > 1:
                    end = int(offs[np.searchsorted(offs, ob[c0], side="right")])
                    exp.append((int(ob[c0 + 1]), 0, end - int(ob[c0 + 1]), c0, c0 + 1, k + c0 + 1, chunk, tr["height_scale"][c0]))
            expw = np.array(exp, dtype=WORK_ITEM_DTYPE) if exp else np.zeros(0, dtype=WORK_ITEM_DTYPE)
            assert walks.tobytes() == expw.tobytes(), case


def test_motion_files_pickle_and_torch_twin(tmp_path):
    """The output wire format (scripts/smplx_to_robot_dataset.py:134-146) and its torch twin (scripts/convert_motion_pkl_to_pt.py):
    same keys and dtypes back through the reference's reader contract (data_loader.py:4-18), existing files are not overwritten."""
    from gmr_amd import dataset
    T = 6
    rng = np.random.default_rng(0)
    m = {"fps": 30, "root_pos": rng.random((T, 3)), "root_rot": rng.random((T, 4)), "dof_pos": rng.random((T, 29)),
         "local_body_pos": rng.random((T, 38, 3)).astype(np.float32), "link_body_list": ["pelvis", "torso"]}
    for ext in ("pkl", "pt"):
        p = str(tmp_path / ("clip." + ext))
        assert dataset.save_motion(p, m) and not dataset.save_motion(p, m) and dataset.save_motion(p, m, override=True)
        d, fps, rp, rr_wxyz, dp, lb, names = dataset.load_robot_motion(p)
        assert fps == 30 and names == ["pelvis", "torso"] and lb.dtype == np.float32 and rp.dtype == np.float64
        assert np.array_equal(rp, m["root_pos"]) and np.array_equal(rr_wxyz, m["root_rot"][:, [3, 0, 1, 2]]) and np.array_equal(dp, m["dof_pos"])
        assert np.array_equal(lb, m["local_body_pos"]) and set(d) == set(m)


def test_motion_writer_pool_is_byte_identical_to_serial_pickle(tmp_path):
    """dataset.save_motions / MotionWriter (row f-3: the overlapped batched writer): every file equals what the reference's
    ``pickle.dump(motion, f)`` (scripts/smplx_to_robot_dataset.py:143-146) writes for the same dict, byte for byte -- clips that are
    row slices of one batch array, arrays below and above the pickler's 64 KiB framing threshold, an empty clip, a fps given as a
    float --; existing files are skipped unless override; .pt files equal the serial torch.save; load_robot_motion round-trips."""
    import pickle
    from gmr_amd import dataset
    assert dataset.fast_pickle_ok()
    rng = np.random.default_rng(3)
    lens = [4000, 1, 0, 2731, 90, 300]
    offs = np.concatenate([[0], np.cumsum(lens)])
    N = int(offs[-1])
    big = [rng.random((N, 3)), rng.random((N, 4)), rng.random((N, 29)), rng.random((N, 38, 3)).astype(np.float32)]
    big[1] /= np.linalg.norm(big[1], axis=1, keepdims=True)
    names = [f"body{i}" for i in range(38)]
    motions = [{"fps": 30 if s % 2 else 29.97, "root_pos": big[0][a:b], "root_rot": big[1][a:b], "dof_pos": big[2][a:b], "local_body_pos": big[3][a:b],
                "link_body_list": names} for s, (a, b) in enumerate(zip(offs[:-1], offs[1:]))]
    paths = [str(tmp_path / "out" / f"c{s}.pkl") for s in range(len(lens))]
    assert dataset.save_motions(motions, paths, workers=4) == len(lens)
    for m, p in zip(motions, paths):
        assert open(p, "rb").read() == pickle.dumps(m)
        d, fps, rp, rr_wxyz, dp, lb, nm = dataset.load_robot_motion(p)
        assert fps == m["fps"] and np.array_equal(rp, m["root_pos"]) and np.array_equal(lb, m["local_body_pos"]) and nm == names
        dataset.validate_motion(d, nq=36) if len(rp) else None
    # skip-if-exists (scripts/smplx_to_robot_dataset.py:219), counted; override rewrites
    open(paths[1], "wb").write(b"x")
    with dataset.MotionWriter(workers=2) as w:
        w.submit(motions, paths)
    assert (w.written, w.skipped) == (0, len(lens)) and open(paths[1], "rb").read() == b"x"
    assert dataset.save_motions(motions[:2], paths[:2], workers=2, override=True) == 2 and open(paths[1], "rb").read() == pickle.dumps(motions[1])
    # the torch twin through the pool == torch.save called serially
    import torch
    pt = [str(tmp_path / f"c{s}.pt") for s in range(2)]
    assert dataset.save_motions(motions[:2], pt, workers=2) == 2
    os.makedirs(tmp_path / "serial")
    for s, (m, p) in enumerate(zip(motions, pt)):
        ref = str(tmp_path / "serial" / f"c{s}.pt")  # (the archive carries the file's stem)
        torch.save({k: torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v for k, v in m.items()}, ref)
        assert open(p, "rb").read() == open(ref, "rb").read()
        assert np.array_equal(dataset.load_robot_motion(p)[4], m["dof_pos"])
    # a worker's error surfaces
    with pytest.raises(Exception):
        dataset.save_motions([{"fps": 30, "bad": (lambda: 0)}], [str(tmp_path / "bad.pkl")], workers=1)
    with pytest.raises(ValueError):
        dataset.save_motions(motions, paths[:2])


def test_auto_chunk_regimes():
    """schedule.auto_chunk: short chunks for a few short clips (the launch lasts one chunk + burn-in), long ones when chunks queue
    for wavefront slots (every burn-in frame is redundant work); always inside [16, 128], burn-in 24; degenerate inputs."""
    from gmr_amd.schedule import auto_chunk, make_items
    assert auto_chunk([0, 3000]) == (24, 24) and auto_chunk(np.arange(25) * 4000) == (32, 24) and auto_chunk(np.arange(5) * 9000) == (48, 24)
    assert auto_chunk(np.arange(78) * 5300)[0] in (96, 104) and auto_chunk(np.arange(513) * 30000) == (128, 24)
    assert auto_chunk(np.arange(8193) * 3000) == (0, 0) and auto_chunk(np.arange(514) * 300) == (0, 0)   # more than a clip per four slots: whole clips
    assert auto_chunk([0]) == (16, 24) and auto_chunk([0, 0, 0]) == (16, 24) and auto_chunk([0, 5]) == (16, 24)
    assert auto_chunk(np.arange(78) * 5300, slots=8 * 2048)[0] == 40      # eight GPUs: the same set no longer fills the slots
    c, b = auto_chunk([0, 100, 100, 7000])
    it = make_items([0, 100, 100, 7000], chunk=c, burn_in=b)
    assert int(it["n_out"].sum()) == 7000 and int(it["n_out"].max()) <= c


def test_bvh_text_entry_points_survive_hostile_input():
    """gmr_bvh_parse_header / gmr_bvh_parse_motion (host code of the library) on mutated files: text that ends at a PROT_NONE page,
    outputs between canaries, capacities around the joint count (tools/fuzz_bvh_text.py; 1.3 M calls in profiles/r03_fuzz_bvh_text.txt)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_bvh_text.py"), "3", "5"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 violations" in r.stdout


def test_motion_stream_templates_are_byte_identical_for_every_clip_length():
    """dataset.motion_stream: a layout's pickle is kept as a template with the numbers taken out (large arrays as views, the small ones --
    below the pickler's 64 KB frame size, i.e. every clip shorter than 2 731 frames has some -- patched into the framed bytes).  Whatever
    the length and however degenerate the data (blocks of zeros also match the zero bytes of a length field: templates are built from a
    random-filled shadow and checked on the second clip), the stream must be pickle.dumps(motion) byte for byte."""
    import pickle
    from gmr_amd import dataset
    assert dataset.fast_pickle_ok()
    rng = np.random.default_rng(3)
    for T in (0, 1, 7, 40, 100, 300, 750, 1366, 1500, 2047, 2049, 2731, 3000):
        ms = [{"fps": 30.0 if T % 2 else 30, "root_pos": rng.random((T, 3)), "root_rot": rng.random((T, 4)), "dof_pos": rng.random((T, 29)),
               "local_body_pos": rng.random((T, 38, 3)).astype(np.float32), "link_body_list": [f"b{i}" for i in range(38)]} for _ in range(8)]
        for i in (0, 1, 4):
            ms[i]["root_pos"][:] = 0
            ms[i]["root_rot"][:] = 0
        ms[5]["dof_pos"][:] = 0
        ms[6]["local_body_pos"][:] = 0
        ms[7]["root_pos"][:] = ms[7]["root_rot"][:, :3]
        for rep in range(2):
            for m in ms:
                got = b"".join(bytes(memoryview(x).cast("B")) for x in dataset.motion_stream(m))
                assert got == pickle.dumps(m), (T, rep)
    # a dict the template cannot express (an object array) still comes out right
    odd = {"fps": 30, "root_pos": np.zeros((500, 3)), "names": np.array(["a", None], dtype=object)}
    assert b"".join(bytes(memoryview(x).cast("B")) for x in dataset.motion_stream(odd)) == pickle.dumps(odd)


def test_dataset_script_folder_walk(tmp_path):
    """gmr_amd.scripts._walk: the folder walk of scripts/bvh_to_robot_dataset.py:59-72 and scripts/smplx_to_robot_dataset.py:205-227 -- sorted
    (natsorted) names per folder, the target is the source path with the folder and the extension replaced, existing targets are skipped
    unless --override, `_stagei` files and the hard-motion lists are left out."""
    from gmr_amd.scripts._walk import hard_motion_names, natural_key, plan_files
    src, tgt = str(tmp_path / "in"), str(tmp_path / "out")
    os.makedirs(os.path.join(src, "sub"))
    for n in ("walk10.bvh", "walk2.bvh", "notes.txt", os.path.join("sub", "run1.bvh")):
        open(os.path.join(src, n), "w").close()
    os.makedirs(tgt)
    open(os.path.join(tgt, "walk2.pkl"), "w").close()
    s, t, skipped = plan_files(src, tgt, lambda n: n.endswith(".bvh"), ".bvh", override=False)
    assert [os.path.relpath(x, src) for x in s] == ["walk10.bvh", os.path.join("sub", "run1.bvh")] and skipped == 1
    assert [os.path.relpath(x, tgt) for x in t] == ["walk10.pkl", os.path.join("sub", "run1.pkl")]
    s, _, skipped = plan_files(src, tgt, lambda n: n.endswith(".bvh"), ".bvh", override=True, natural=True)
    assert [os.path.basename(x) for x in s] == ["walk2.bvh", "walk10.bvh", "run1.bvh"] and skipped == 0
    assert sorted(["a10", "a9", "b1"], key=natural_key) == ["a9", "a10", "b1"]
    lst = tmp_path / "0.txt"
    lst.write_text("header\nMotion: ACCAD/Male2/run_poses.npz, error 3.2\nMotion:  CMU/01/01_01_poses , x\nnothing here\n")
    assert hard_motion_names([str(lst), str(tmp_path / "missing.txt")]) == ["ACCAD/Male2/run_poses", "CMU/01/01_01_poses"]
    # the argument parsers take the reference's flags
    import subprocess, sys
    for mod in ("gmr_amd.scripts.bvh_to_robot_dataset", "gmr_amd.scripts.smplx_to_robot_dataset"):
        r = subprocess.run([sys.executable, "-m", mod, "--src_folder", str(tmp_path / "empty"), "--tgt_folder", tgt, "--robot", "unitree_g1", "--override"],
                           capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0 and "Done." in r.stdout, r.stderr[-500:]
    # --shard_by_rank under torch.distributed.run: rank 1 of 2 takes files[1::2] of the walk -- of the three files that is one
    env = dict(os.environ, RANK="1", WORLD_SIZE="2", LOCAL_RANK="1")
    r = subprocess.run([sys.executable, "-m", "gmr_amd.scripts.bvh_to_robot_dataset", "--src_folder", src, "--tgt_folder", str(tmp_path / "o2"), "--shard_by_rank", "--help"],
                       capture_output=True, text=True, env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "--shard_by_rank" in r.stdout


def test_motion_writer_back_pressure(tmp_path, monkeypatch):
    """MotionWriter.submit waits once ``max_pending`` batches are unwritten (their pinned result arrays stay alive until then): with a
    writer slowed down artificially, the number of batches in flight never exceeds the bound, and everything is written in the end."""
    import threading
    import time
    from gmr_amd import dataset
    in_flight, peak, lock = set(), [0], threading.Lock()
    real = dataset.save_motion

    def slow(path, motion, override=False):
        with lock:
            in_flight.add(motion["batch"])
            peak[0] = max(peak[0], len(in_flight))
        time.sleep(0.01)
        r = real(path, {k: v for k, v in motion.items() if k != "batch"}, override)
        return r
    monkeypatch.setattr(dataset, "save_motion", slow)
    done_batches = []
    with dataset.MotionWriter(workers=2, max_pending=2) as w:
        for b in range(6):
            ms = [{"fps": 30, "root_pos": np.zeros((4, 3)), "batch": b} for _ in range(4)]
            w.submit(ms, [str(tmp_path / f"b{b}_{i}.pkl") for i in range(4)])
            with lock:   # batches older than the two most recent ones must be complete by now
                in_flight.intersection_update({b, b - 1})
            done_batches.append(b)
    assert w.written == 24 and peak[0] <= 3 and len(list(tmp_path.iterdir())) == 24


def test_smoke_test_twin(tmp_path):
    """python -m gmr_amd.scripts.smoke_test: the motion-file checks of scripts/smoke_test.py:19-72 (keys, shapes, hinge count, frames;
    suspect quaternion norms only warn, as there)."""
    import pickle
    from gmr_amd.scripts import smoke_test
    T = 5
    good = {"fps": 30, "root_pos": np.zeros((T, 3)), "root_rot": np.tile([0, 0, 0, 1.0], (T, 1)), "dof_pos": np.zeros((T, 29)), "local_body_pos": None, "link_body_list": None}
    pickle.dump(good, open(tmp_path / "a.pkl", "wb"))
    pickle.dump(dict(good, root_rot=np.zeros((T, 4))), open(tmp_path / "b_norms.pkl", "wb"))
    assert smoke_test.main(["--folder", str(tmp_path), "--robot", "unitree_g1"]) == 0
    pickle.dump({k: v for k, v in good.items() if k != "dof_pos"}, open(tmp_path / "c_missing.pkl", "wb"))
    pickle.dump(dict(good, root_pos=np.zeros((T, 2))), open(tmp_path / "d_shape.pkl", "wb"))
    assert smoke_test.main(["--folder", str(tmp_path), "--robot", "unitree_g1"]) == 1
    assert smoke_test.main(["--folder", str(tmp_path / "nowhere")]) == 0


def test_package_exports_the_reference_package_names(monkeypatch):
    """general_motion_retargeting/__init__.py:2-6 exports the registry dicts, the two classes and load_robot_motion: code written as
    `from general_motion_retargeting import X` finds every X here but the viewer, which says why it is missing."""
    import importlib
    import gmr_amd
    for name in ("IK_CONFIG_ROOT", "ASSET_ROOT", "ROBOT_XML_DICT", "IK_CONFIG_DICT", "ROBOT_BASE_DICT", "VIEWER_CAM_DISTANCE_DICT", "load_robot_motion"):
        assert getattr(gmr_amd, name) is not None, name
    assert set(gmr_amd.VIEWER_CAM_DISTANCE_DICT) == set(gmr_amd.ROBOT_XML_DICT.keys()) == set(gmr_amd.ROBOT_BASE_DICT)
    with pytest.raises(AttributeError, match="viewer"):
        gmr_amd.RobotMotionViewer
    root = os.environ.get("GMR_ROOT") or "/root/reference"
    if os.path.isdir(os.path.join(root, "assets")):   # (the build container: ASSET_ROOT follows GMR_ROOT like the registry paths)
        monkeypatch.setenv("GMR_ROOT", root)
        from gmr_amd import params
        assert str(params.asset_root()) == os.path.join(root, "assets")
