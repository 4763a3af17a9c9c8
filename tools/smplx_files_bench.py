"""The SMPL-X side of the file path: a folder of joint-array files (what a body-model owner dumps once per AMASS clip,
gmr_amd.smplx_adapter.save_joint_file) -> key-points -> qpos -> one pickle per clip: the loop body of
scripts/smplx_to_robot_dataset.py:63-146 behind the body model.

    python tools/smplx_files_bench.py [n_files] [frames_per_file] [threads] [batch_files]

Synthetic folder on tmpfs: robot-consistent key-points of a smplx_to_g1 clip set stored as float32 joint arrays at 30 fps (55 joints,
1 332 B per frame), half the clips noisy / over-reaching, one height per file.
"""
import json
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def run(n_files=512, T=750, threads=16, device=0, reps=3, batch_files=256, chunk=0):
    from gmr_amd import GeneralMotionRetargeting as GMR, dataset, synth
    from gmr_amd import smplx_adapter as sa
    dev = torch.device("cuda", device)
    tmpd = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    res = {"files": n_files, "frames": n_files * T, "threads": threads, "batch_files": batch_files, "chunk": chunk}
    try:
        g = GMR(src_human="smplx", tgt_robot="unitree_g1")
        cm = g._cm
        lens = np.full(n_files, T)
        heights = list(np.random.default_rng(1).uniform(1.55, 1.9, n_files))
        pos, quat, names, offs = synth.synth_clips_torch(cm, lens, seed=91, device=dev, hard=np.arange(n_files) % 2 == 1, yaw0=1.0, dtype=torch.float64)
        d_in, d_out = os.path.join(tmpd, "in"), os.path.join(tmpd, "out")
        os.makedirs(d_in)
        files = synth.write_smplx_joint_files(d_in, pos, quat, names, offs, fps=30.0, heights=heights)
        res["input_MB"] = sum(os.path.getsize(f) for f in files) / 1e6
        cols = g.ik_columns
        N = n_files * T

        def timed(fn):
            ts = []
            for _ in range(reps):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                r = fn()
                torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            return float(np.median(ts)), r
        t_load, _ = timed(lambda: [len(b) for b in sa.iter_joint_batches(files, batch_files=batch_files, threads=threads, columns=cols)])

        def to_qpos():
            out = []
            for b in sa.iter_joint_batches(files, batch_files=batch_files, threads=threads, columns=cols):
                out.append(g.retarget_batch(b.pos, b.quat, b.body_names, seq_offsets=b.seq_offsets, human_heights=b.human_heights, chunk=chunk))
            return torch.cat(out)
        t_q, q_files = timed(to_qpos)

        def to_pickles():
            k = 0
            with dataset.MotionWriter(workers=max(2, min(16, threads)), override=True) as w:
                for b in sa.iter_joint_batches(files, batch_files=batch_files, threads=threads, columns=cols):
                    motions = dataset.retarget_clips(g, b.pos, b.quat, b.body_names, b.seq_offsets, fps=b.fps, human_heights=b.human_heights, chunk=chunk)
                    w.submit(motions, [os.path.join(d_out, os.path.basename(f)[:-4] + ".pkl") for f in b.files])
                    k += len(b)
            return w.written
        t_p, n_written = timed(to_pickles)
        q_mem = g.retarget_batch(pos, quat, names, seq_offsets=offs, human_heights=heights)
        dq = (q_files - q_mem).abs().amax(dim=1)
        res.update({"frames_beyond_1e-4_of_in_memory": int((dq > 1e-4).sum().item()), "files_to_keypoints_frames_per_s": N / t_load, "input_MB_per_s": res["input_MB"] / t_load, "files_to_qpos_frames_per_s": N / t_q,
                    "files_to_pickles_frames_per_s": N / t_p, "pickles_written": int(n_written),
                    "max_abs_diff_vs_keypoints_in_memory": float((q_files - q_mem).abs().max().item()),
                    "note": "float32 files: the difference to solving the float64 key-points in memory is that rounding (1e-7 m on the targets; a frame that sits on the edge of the loop's 1e-3 stopping rule can take one solve more or less, and its clip differs from there on by a few 1e-3 rad until both runs meet again); a 120 fps source holds 4x the bytes per output frame"})
    finally:
        shutil.rmtree(tmpd, ignore_errors=True)
    return res


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:5]]
    a = a + [512, 750, 16, 256][len(a):]
    chunk = sys.argv[5] if len(sys.argv) > 5 else 0
    print(json.dumps(run(a[0], a[1], a[2], batch_files=a[3], chunk=chunk if chunk == "auto" else int(chunk))))
