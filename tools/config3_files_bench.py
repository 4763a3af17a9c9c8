"""BASELINE config 3 from FILES: a folder of BVH files -> qpos (scripts/bvh_to_robot_dataset.py:59-104 end to end).

    python tools/config3_files_bench.py [n_files] [frames_per_file] [threads]

Two synthetic folders on tmpfs (no LAFAN1 data exists offline):
  * LAFAN1-shaped text (22 bones, 3-channel rows, random joint angles): files -> key-point tensors on the GPU, MOTION blocks parsed
    on the device (`parse="device"`) and on host threads (`parse="host"`, round 2's path).  Its random motion is not something a robot
    can follow, so it measures the loader.
  * robot-consistent key-points of a bvh_to_g1 clip set stored as BVH files (flat 6-channel hierarchy, gmr_amd.synth.
    write_keypoint_files): files -> qpos with per-clip heights and verified parallel-in-time chunks, batches read ahead.
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


def write_files(tmpd, n_files, T, seed=0):  # (kept for the tests that import it)
    from gmr_amd import synth
    return synth.write_lafan_shaped_files(tmpd, n_files, T, seed)


def run(n_files=24, T=4000, threads=16, device=0, reps=3):
    from gmr_amd import GeneralMotionRetargeting as GMR, synth
    from gmr_amd.bvh import iter_lafan1_batches, load_lafan1_files
    dev = torch.device("cuda", device)
    tmpd = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    res = {"files": n_files, "frames": n_files * T, "threads": threads}
    try:
        d1, d2 = os.path.join(tmpd, "shape"), os.path.join(tmpd, "kp")
        os.makedirs(d1); os.makedirs(d2)
        files = synth.write_lafan_shaped_files(d1, n_files, T)
        res["text_MB"] = sum(os.path.getsize(f) for f in files) / 1e6
        g = GMR(src_human="bvh", tgt_robot="unitree_g1")
        cols = list(g._cm.slot_names)

        def timed(fn):
            ts = []
            for _ in range(reps):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                r = fn()
                torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            return float(np.median(ts)), r
        st = {}
        t_dev, b_dev = timed(lambda: load_lafan1_files(files, threads=threads, columns=cols, parse="device", stats=st))
        t_host, b_host = timed(lambda: load_lafan1_files(files, threads=threads, columns=cols, parse="host"))
        N = n_files * T
        res["loader"] = {"files_to_keypoints_frames_per_s": N / t_dev, "text_MB_per_s": res["text_MB"] / t_dev, "host_parse_frames_per_s": N / t_host,
                         "bitwise_equal_to_host_parse": bool(torch.equal(b_dev.pos, b_host.pos) and torch.equal(b_dev.quat, b_host.quat)),
                         "slow_tokens": st.get("slow_tokens"), "files_reparsed_on_host": st.get("files_reparsed_on_host"), "columns": len(cols),
                         "reference_loader": "~600 frames/s (SURVEY f-1: 0.4 s for 250 frames x 101 joints in the reference's Python loader)"}
        del b_dev, b_host
        # robot-consistent clips as files
        cmb = g._cm
        lengths = np.full(n_files, T)
        hard = np.arange(n_files) % 2 == 1
        pos, quat, names, offs = synth.synth_clips_torch(cmb, lengths, seed=33, device=dev, hard=hard, yaw0=1.0, dtype=torch.float64)
        kfiles = synth.write_keypoint_files(d2, pos.cpu().numpy(), quat.cpu().numpy(), names, offs, head_height=cmb.config.human_height_assumption)
        res["keypoint_text_MB"] = sum(os.path.getsize(f) for f in kfiles) / 1e6

        def files_to_qpos(batch_files):
            out, info = [], {"resolved_frames": 0, "heights": []}
            for batch in iter_lafan1_batches(kfiles, batch_files=batch_files, threads=threads, columns=cols):
                q = g.retarget_batch(batch.pos, batch.quat, batch.body_names, seq_offsets=batch.seq_offsets, human_heights=batch.human_heights, chunk="auto")
                info["resolved_frames"] += g.last_chunk_info["resolved_frames"]
                info["heights"] += list(batch.human_heights)
                out.append(q)
            return torch.cat(out), info
        t_q, (q_files, info) = timed(lambda: files_to_qpos(n_files))               # the whole folder as one batch (110 MB of text)
        t_q2, _ = timed(lambda: files_to_qpos(max(1, n_files // 2)))               # two batches, the second read while the first is solved
        # ... and all the way to one pickle per clip (the whole loop body of scripts/bvh_to_robot_dataset.py:59-151): FK for local_body_pos,
        # the reference's schema, files written by the pool
        from gmr_amd import dataset
        d3 = os.path.join(tmpd, "out")

        def files_to_pickles():
            with dataset.MotionWriter(workers=max(2, min(16, threads)), override=True) as w:
                for batch in iter_lafan1_batches(kfiles, batch_files=n_files, threads=threads, columns=cols):
                    motions = dataset.retarget_clips(g, batch.pos, batch.quat, batch.body_names, batch.seq_offsets, fps=30, height_adjust=False,
                                                     root_origin_offset=False, chunk="auto", human_heights=batch.human_heights)
                    w.submit(motions, [os.path.join(d3, os.path.basename(f)[:-4] + ".pkl") for f in batch.files])
            return w.written
        t_p, n_written = timed(files_to_pickles)
        q_direct = g.retarget_batch(pos, quat, names, seq_offsets=offs, human_heights=info["heights"], chunk="auto")
        res["from_files"] = {"files_to_qpos_frames_per_s": N / t_q, "seconds": t_q, "resolved_frames": int(info["resolved_frames"]), "batches": 1,
                             "two_batches_read_ahead_frames_per_s": N / t_q2,
                             "files_to_pickles_frames_per_s": N / t_p, "pickles_written": int(n_written),
                             "max_abs_diff_vs_keypoints_in_memory": float((q_files - q_direct).abs().max().item()),
                             "note": "a folder this small is best taken as one batch (a batch's fixed costs -- two launches, the walk's 63 chunk boundaries per clip -- exceed what read-ahead hides); the files carry the key-points with 6 decimals (1e-8 m, 2e-8 rad): the difference to solving the in-memory key-points is that rounding"}
    finally:
        shutil.rmtree(tmpd, ignore_errors=True)
    return res


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:4]]
    print(json.dumps(run(*(a + [24, 4000, 16][len(a):]))))
