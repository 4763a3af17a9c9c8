"""IK throughput of every registry robot with an SMPL-X config: 8192 clips x 600 frames each (64 distinct clips tiled, half of them hard); `core` != 0: the structured QP back end applies."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gmr_amd import params, synth
from gmr_amd.mjcf import load_robot
from gmr_amd.ik_config import load_ik_config
from gmr_amd.model import compile_model
from gmr_amd.engine import Engine
from gmr_amd.schedule import make_items

S, T, D = 8192, 600, 64
rows = []
ROBOTS = sys.argv[1:] or ["unitree_g1", "unitree_g1_with_hands", "booster_t1", "stanford_toddy", "fourier_n1", "engineai_pm01", "kuavo_s45", "hightorque_hi",
                         "galaxea_r1pro", "booster_k1"]  # (berkeley_humanoid_lite: the registry names smplx_to_bhl.json, which the reference does not ship)
for robot in ROBOTS:
    cm = compile_model(load_robot(params.ROBOT_XML_DICT[robot], name=robot), load_ik_config(params.IK_CONFIG_DICT["smplx"][robot]))
    eng = Engine(cm, 0)
    pe, qe, names, _, _ = synth.synth_clips(cm, D // 2, T, seed=1, hard=False, dtype=np.float32)
    ph, qh, _, _, _ = synth.synth_clips(cm, D // 2, T, seed=2, hard=True, dtype=np.float32)
    pos = torch.from_numpy(np.concatenate([pe, ph])).cuda().repeat(S // D, 1, 1)
    quat = torch.from_numpy(np.concatenate([qe, qh])).cuda().repeat(S // D, 1, 1)
    items = make_items(np.arange(S + 1, dtype=np.int64) * T)
    sc = cm.slot_columns(names)
    out = torch.empty((S * T, eng.nq), dtype=torch.float64, device="cuda")
    eng.ik_solve(pos, quat, sc, items, out=out)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        _, it, _ = eng.ik_solve(pos, quat, sc, items, out=out)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 3
    solves = float((it & 0x3FFFFFFF).double().mean().item())
    rows.append({"robot": robot, "nq": eng.nq, "active_dofs": eng.info.n_active_dof, "nvp": eng.info.nv_padded, "core": eng.info.reserved[0],
                 "lds_bytes": eng.info.lds_bytes, "ms": ms, "frames_per_s": S * T / ms * 1e3, "solves_per_frame": solves, "solves_per_s": S * T * solves / ms * 1e3})
print(json.dumps(rows))
