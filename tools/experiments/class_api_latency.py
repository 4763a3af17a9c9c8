"""Per-frame latency of the class the reference scripts call, GeneralMotionRetargeting.retarget(frame_dict), with a profile of the host side
(66 us median on one MI355X: 49 us session step + 17 us of dict / array handling; the reference manages 35-70 frames/s)."""
import sys, time, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from gmr_amd import GeneralMotionRetargeting as GMR, synth
from tests.util import compiled
cm = compiled("smplx", "unitree_g1")
pos, quat, names, _, _ = synth.synth_clips(cm, 1, 600, seed=1, hard=False, dtype=np.float32)
g = GMR(src_human="smplx", tgt_robot="unitree_g1")
frames = [{n: (pos[i, c].astype(np.float64), quat[i, c].astype(np.float64)) for c, n in enumerate(names)} for i in range(600)]
lat = []
for fd in frames:
    t = time.perf_counter(); q = g.retarget(fd); lat.append(time.perf_counter() - t)
lat = np.array(lat[50:]) * 1e6
print("class api median us", np.median(lat), "p99", np.quantile(lat, 0.99), "len(names)", len(names))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for fd in frames[:300]: g.retarget(fd)
pr.disable(); pstats.Stats(pr).sort_stats('tottime').print_stats(14)
