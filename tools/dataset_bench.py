"""End-to-end rate of the dataset path (gmr_amd.dataset.retarget_clips): GPU key-points in -> per-clip motion dicts (numpy) out."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gmr_amd import GeneralMotionRetargeting as GMR, synth
from gmr_amd.dataset import retarget_clips, motions_from_qpos

S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
T = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
g = GMR("smplx", "unitree_g1")
cm = g._cm
pos, quat, names, _, _ = synth.synth_clips(cm, 16, T, seed=5, hard=False, dtype=np.float32)
dev = g.device
tp = torch.from_numpy(pos).to(dev).repeat(S // 16, 1, 1)
tq = torch.from_numpy(quat).to(dev).repeat(S // 16, 1, 1)
offs = np.arange(S + 1, dtype=np.int64) * T
N = S * T
def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    return best, r
t_all, motions = timed(lambda: retarget_clips(g, tp, tq, names, offs))
t_ik, q = timed(lambda: g.retarget_batch(tp, tq, names, seq_offsets=offs))
t_post, _ = timed(lambda: motions_from_qpos(g, q, offs, 30))
print(json.dumps({"clips": S, "frames": N, "dataset_path_frames_per_s": N / t_all, "ik_only_frames_per_s": N / t_ik, "post_only_frames_per_s": N / t_post,
                  "seconds": {"all": t_all, "ik": t_ik, "post": t_post}, "bytes_per_frame_to_host": 24 + 32 + (cm.robot.nq - 7) * 8 + cm.robot.nbody * 12}))
