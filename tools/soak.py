"""Soak: the bench launch repeated for a few minutes, every output compared bitwise with the first (determinism, no races)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gmr_amd import params, synth
from gmr_amd.mjcf import load_robot
from gmr_amd.ik_config import load_ik_config
from gmr_amd.model import compile_model
from gmr_amd.engine import Engine
from gmr_amd.schedule import make_items

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
cm = compile_model(load_robot(params.ROBOT_XML_DICT["unitree_g1"], name="unitree_g1"), load_ik_config(params.IK_CONFIG_DICT["smplx"]["unitree_g1"]))
eng = Engine(cm, 0)
S, T = 4096, 1000
pe, qe, names, _, _ = synth.synth_clips(cm, 16, T, seed=7, hard=False, dtype=np.float32)
ph, qh, _, _, _ = synth.synth_clips(cm, 16, T, seed=8, hard=True, dtype=np.float32)
pos = torch.from_numpy(np.concatenate([pe, ph])).cuda().repeat(S // 32, 1, 1)
quat = torch.from_numpy(np.concatenate([qe, qh])).cuda().repeat(S // 32, 1, 1)
items = make_items(np.arange(S + 1, dtype=np.int64) * T)
sc = cm.slot_columns(names)
ref, it_ref, _ = eng.ik_solve(pos, quat, sc, items)
torch.cuda.synchronize()
t0, n, last = time.time(), 0, time.time()
while time.time() - t0 < seconds:
    out, it, _ = eng.ik_solve(pos, quat, sc, items)
    assert torch.equal(out, ref) and torch.equal(it, it_ref), f"launch {n} differs"
    n += 1
    if time.time() - last > 30:
        print(f"{n} launches identical, {time.time() - t0:.0f} s", flush=True); last = time.time()
print(f"soak ok: {n} launches of {S} x {T} frames bitwise identical in {time.time() - t0:.0f} s ({n * S * T / (time.time() - t0) / 1e6:.1f} M frames/s incl. compares)")
