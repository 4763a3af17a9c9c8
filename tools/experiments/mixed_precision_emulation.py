"""How far would a mixed-precision IK kernel move the results?  (CPU, oracle only.)

The oracle's assembly of H and c is switched to float32 (weighted Jacobian entries rounded to float32, float32 accumulation) while
FK, residuals, LM damping, the box QP and the `curr - next > 1e-3` test stay float64 -- the numerical content of a kernel whose
task blocks / composites / H pairs run on packed v_pk_fma_f32.  Reported against the float64 oracle on the bench's own clips:
max |dq| per hinge / root, and the number of frames whose solve count differs."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gmr_amd import synth
from gmr_amd.schedule import make_items
from oracle import oracle as O
from tests.util import compiled

cm = compiled("smplx", "unitree_g1")
orc = O.Oracle(cm.blob)
T, D = 3000, 32
pe, qe, names, _, _ = synth.synth_clips(cm, D // 2, T, seed=1000, hard=False, dtype=np.float32)
ph, qh, _, _, _ = synth.synth_clips(cm, D // 2, T, seed=2000, hard=True, dtype=np.float32)
pos, quat = np.concatenate([pe, ph]), np.concatenate([qe, qh])
sc = cm.slot_columns(names)
items = make_items(np.arange(D + 1) * T)
q64, it64, _ = orc.ik_solve(pos, quat, sc, items, n_threads=8)
O.set_mixed_assembly(True)
q32, it32, _ = orc.ik_solve(pos, quat, sc, items, n_threads=8)
O.set_mixed_assembly(False)
d = np.abs(q32 - q64)
d[:, 3:7] = np.minimum(d[:, 3:7], np.abs(q32[:, 3:7] + q64[:, 3:7]))
per = d.max(axis=1)
res = {"frames": int(len(per)), "max_abs_dq": float(per.max()), "p999_abs_dq": float(np.quantile(per, 0.999)), "median_abs_dq": float(np.median(per)),
       "frames_over_1e-3": int((per > 1e-3).sum()), "frames_over_1e-6": int((per > 1e-6).sum()),
       "frames_with_different_solve_count": int((it32 != it64).sum()), "mean_solves_f64": float(it64.mean()), "mean_solves_mixed": float(it32.mean())}
for half, sl in (("easy", slice(0, D // 2 * T)), ("hard", slice(D // 2 * T, None))):
    res[half] = {"max_abs_dq": float(per[sl].max()), "frames_with_different_solve_count": int((it32[sl] != it64[sl]).sum())}
print(json.dumps(res, indent=1))
