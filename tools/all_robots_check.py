import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from gmr_amd import params, synth
from gmr_amd.mjcf import load_robot
from gmr_amd.ik_config import load_ik_config
from gmr_amd.model import compile_model
from gmr_amd.engine import Engine, IKParams
from gmr_amd.schedule import make_items
from oracle.oracle import Oracle, IKParams as OParams
ok = bad = 0
for src, d in params.IK_CONFIG_DICT.items():
    for robot in d:
        try:
            cm = compile_model(load_robot(params.ROBOT_XML_DICT[robot], name=robot), load_ik_config(d[robot]))
            eng = Engine(cm, 0)
            err, same, orc = 0.0, True, Oracle(cm.blob)
            # float32 key-points; float64 key-points with offset_to_ground; per-clip heights and root-target starts
            for dt, otg, hs, init in ((np.float32, 0, None, -1), (np.float64, 1, None, -1), (np.float32, 0, [0.9, 1.15], -2)):
                pos, quat, names, offs, _ = synth.synth_clips(cm, 2, 20, seed=3, hard=True, dtype=dt, amp=0.2)
                sc = cm.slot_columns(names)
                items = make_items(offs, height_scales=hs, clip_init=init)
                q, it, _ = eng.ik_solve(torch.from_numpy(pos).cuda(), torch.from_numpy(quat).cuda(), sc, items, params=IKParams(offset_to_ground=otg))
                q_ref, it_ref, _ = orc.ik_solve(pos, quat, sc, items, params=OParams(offset_to_ground=otg))
                err = max(err, np.abs(q.cpu().numpy() - q_ref).max())
                same = same and np.array_equal(it.cpu().numpy() & 0x3FFFFFFF, it_ref)
            print(f"{src:6s} {robot:28s} nq {cm.robot.nq:3d} tasks {len(cm.tasks[0])}/{len(cm.tasks[1])} nvp {eng.info.nv_padded} core {eng.info.reserved[0]} lds {eng.info.lds_bytes}  err {err:.2e} iters_equal {same}")
            ok += 1
        except Exception as e:
            print(f"{src:6s} {robot:28s} FAILED: {type(e).__name__}: {str(e)[:150]}")
            bad += 1
print("ok", ok, "failed", bad)
