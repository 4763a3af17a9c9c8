import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from gmr_amd import params, synth
from gmr_amd.mjcf import load_robot
from gmr_amd.ik_config import load_ik_config
from gmr_amd.model import compile_model
from gmr_amd.engine import Engine
from gmr_amd.schedule import make_items
from oracle.oracle import Oracle
ok = bad = 0
for src, d in params.IK_CONFIG_DICT.items():
    for robot in d:
        try:
            cm = compile_model(load_robot(params.ROBOT_XML_DICT[robot], name=robot), load_ik_config(d[robot]))
            eng = Engine(cm, 0)
            pos, quat, names, offs, _ = synth.synth_clips(cm, 2, 20, seed=3, hard=True, dtype=np.float32, amp=0.2)
            sc = cm.slot_columns(names)
            q, it, _ = eng.ik_solve(torch.from_numpy(pos).cuda(), torch.from_numpy(quat).cuda(), sc, make_items(offs))
            q_ref, it_ref, _ = Oracle(cm.blob).ik_solve(pos, quat, sc, make_items(offs))
            err = np.abs(q.cpu().numpy() - q_ref).max()
            same = np.array_equal(it.cpu().numpy() & 0x3FFFFFFF, it_ref)
            print(f"{src:6s} {robot:28s} nq {cm.robot.nq:3d} tasks {len(cm.tasks[0])}/{len(cm.tasks[1])} nvp {eng.info.nv_padded} core {eng.info.reserved[0]} lds {eng.info.lds_bytes}  err {err:.2e} iters_equal {same}")
            ok += 1
        except Exception as e:
            print(f"{src:6s} {robot:28s} FAILED: {type(e).__name__}: {str(e)[:150]}")
            bad += 1
print("ok", ok, "failed", bad)
