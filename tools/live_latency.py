"""Latency of one live frame (gmr_session_step), launch per frame vs the resident wavefront (gmr_session_set_persistent), and what
other work of the process sees while the wavefront is resident.

    python tools/live_latency.py [frames]
"""
import sys, time, json
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from gmr_amd import synth
from gmr_amd.engine import Engine, IKParams
from tests.util import compiled

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
cm = compiled("smplx", "unitree_g1")
eng = Engine(cm)
pos, quat, names, offs, _ = synth.synth_clips(cm, 1, n, seed=5, hard=False, dtype=np.float32)
sc = cm.slot_columns(names)
out = {}


def run(sess, label):
    lat, solves = [], []
    for f in range(n):
        t0 = time.perf_counter()
        q, s = sess.step(pos[f], quat[f])
        lat.append(time.perf_counter() - t0)
        solves.append(s & 0x3FFFFFFF)
    lat = np.array(lat[50:]) * 1e6
    out[label] = {"median_us": float(np.median(lat)), "p99_us": float(np.percentile(lat, 99)), "mean_solves": float(np.mean(solves[50:]))}
    return q


a = eng.session(sc, pos.shape[1], IKParams(), dtype=np.float32)
qa = run(a, "launch_per_frame")
b = eng.session(sc, pos.shape[1], IKParams(), dtype=np.float32)
b.set_persistent(200)
qb = run(b, "persistent")
from gmr_amd.schedule import make_items
q_batch, _, _ = eng.ik_solve(torch.from_numpy(pos).cuda(), torch.from_numpy(quat).cuda(), sc, make_items(offs))
q_batch = q_batch.cpu().numpy()
out["last_frame_max_abs_diff"] = {"persistent_vs_launch_per_frame": float(np.abs(qa - qb).max()), "persistent_vs_batch": float(np.abs(qb - q_batch[-1]).max()),
                                  "launch_per_frame_vs_batch": float(np.abs(qa - q_batch[-1]).max())}
# with the wavefront resident: a kernel of another stream of this process, and a second session
b.step(pos[0], quat[0])
x = torch.zeros(1 << 20, device="cuda")
st = torch.cuda.Stream()
t0 = time.perf_counter()
with torch.cuda.stream(st):
    y = x + 1
st.synchronize()
out["other_stream_kernel_while_resident_us"] = (time.perf_counter() - t0) * 1e6
b.step(pos[1], quat[1])
t0 = time.perf_counter()
a.step(pos[0], quat[0])
out["other_session_launch_while_resident_us"] = (time.perf_counter() - t0) * 1e6
b.close(); a.close()
print(json.dumps(out))
