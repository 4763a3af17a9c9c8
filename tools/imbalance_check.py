import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from gmr_amd import params, synth
from gmr_amd.mjcf import load_robot
from gmr_amd.ik_config import load_ik_config
from gmr_amd.model import compile_model
from gmr_amd.engine import Engine
from gmr_amd.schedule import make_items
cm = compile_model(load_robot(params.ROBOT_XML_DICT["unitree_g1"], name="unitree_g1"), load_ik_config(params.IK_CONFIG_DICT["smplx"]["unitree_g1"]))
eng = Engine(cm, 0)
T = 3000
pe, qe, names, _, _ = synth.synth_clips(cm, 32, T, seed=1000, hard=False, dtype=np.float32)
ph, qh, _, _, _ = synth.synth_clips(cm, 32, T, seed=2000, hard=True, dtype=np.float32)
pos = torch.from_numpy(np.concatenate([pe, ph])).cuda(); quat = torch.from_numpy(np.concatenate([qe, qh])).cuda()
sc = cm.slot_columns(names)
def run(reps, Tc):
    S = 64 * reps * (T // Tc)
    p = pos.repeat(reps, 1, 1); q = quat.repeat(reps, 1, 1)
    items = make_items(np.arange(S + 1, dtype=np.int64) * Tc)
    out = torch.empty((p.shape[0], eng.nq), dtype=torch.float64, device="cuda")
    eng.ik_solve(p, q, sc, items, out=out)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); _, it, _ = eng.ik_solve(p, q, sc, items, out=out); b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b)
    per = (it & 0x3FFFFFFF).double().reshape(S, Tc).sum(1)
    print(f"clips {S:6d} x {Tc:5d} frames: {ms:8.2f} ms  {p.shape[0] / ms * 1e3 / 1e6:6.2f} M frames/s   per-clip solves max/mean {float(per.max() / per.mean()):.3f}")
run(32, 3000); run(64, 3000); run(128, 3000); run(256, 3000)
