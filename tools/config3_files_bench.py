"""BASELINE config 3 from FILES: a folder of LAFAN1-like BVH files -> qpos (scripts/bvh_to_robot_dataset.py:59-104 end to end).

    python tools/config3_files_bench.py [n_files] [frames_per_file] [threads]

Writes synthetic 22-joint BVH files (LAFAN1 bone names, ZYX Euler channels, cm, Y-up) to tmpfs, then times
load_lafan1_files (threaded native text parse + one gmr_bvh_fk launch) and retarget_batch with per-clip heights and verified
parallel-in-time chunks.  The text parse is host work; this shows where the wall time of the file path goes."""
import os, sys, time, json, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests", "golden"))


def write_files(tmpd, n_files, T, seed=0):
    # reuse the golden generator's skeleton (tests/golden/make_bvh_golden.py) without importing the reference
    LAFAN = [("Hips", -1), ("LeftUpLeg", 0), ("LeftLeg", 1), ("LeftFoot", 2), ("LeftToe", 3), ("RightUpLeg", 0), ("RightLeg", 5),
             ("RightFoot", 6), ("RightToe", 7), ("Spine", 0), ("Spine1", 9), ("Spine2", 10), ("Neck", 11), ("Head", 12),
             ("LeftShoulder", 11), ("LeftArm", 14), ("LeftForeArm", 15), ("LeftHand", 16), ("RightShoulder", 11), ("RightArm", 18),
             ("RightForeArm", 19), ("RightHand", 20)]
    OFFS = {"Hips": (0, 0, 0), "LeftUpLeg": (10, -5, 0), "LeftLeg": (0, -42, 0), "LeftFoot": (0, -40, 0), "LeftToe": (0, -6, 14),
            "RightUpLeg": (-10, -5, 0), "RightLeg": (0, -42, 0), "RightFoot": (0, -40, 0), "RightToe": (0, -6, 14), "Spine": (0, 8, 0),
            "Spine1": (0, 12, 0), "Spine2": (0, 12, 0), "Neck": (0, 22, 0), "Head": (0, 10, 0), "LeftShoulder": (4, 18, 0),
            "LeftArm": (14, 0, 0), "LeftForeArm": (28, 0, 0), "LeftHand": (25, 0, 0), "RightShoulder": (-4, 18, 0), "RightArm": (-14, 0, 0),
            "RightForeArm": (-28, 0, 0), "RightHand": (-25, 0, 0)}
    children = {i: [j for j, (_, p) in enumerate(LAFAN) if p == i] for i in range(len(LAFAN))}
    hdr = ["HIERARCHY"]

    def emit(i, depth):
        name, parent = LAFAN[i]
        ind = "\t" * depth
        hdr.append(f"{ind}{'ROOT' if parent < 0 else 'JOINT'} {name}")
        hdr.append(ind + "{")
        o = OFFS[name]
        hdr.append(f"{ind}\tOFFSET {o[0]:.6f} {o[1]:.6f} {o[2]:.6f}")
        hdr.append(f"{ind}\tCHANNELS 6 Xposition Yposition Zposition Zrotation Yrotation Xrotation" if parent < 0 else f"{ind}\tCHANNELS 3 Zrotation Yrotation Xrotation")
        if not children[i]:
            hdr.extend([f"{ind}\tEnd Site", ind + "\t{", f"{ind}\t\tOFFSET 0.000000 5.000000 0.000000", ind + "\t}"])
        for c in children[i]:
            emit(c, depth + 1)
        hdr.append(ind + "}")
    emit(0, 0)
    rng = np.random.default_rng(seed)
    files = []
    for k in range(n_files):
        t = np.arange(T) / 30.0
        J = len(LAFAN)
        ang = np.zeros((T, J, 3))
        for j in range(J):
            a, f, ph = rng.uniform(2, 25, 3), rng.uniform(0.1, 1.2, 3), rng.uniform(0, 6.28, 3)
            ang[:, j] = a * np.sin(2 * np.pi * f * t[:, None] + ph)
        ang[:, 0, 1] += rng.uniform(-180, 180) + np.cumsum(rng.normal(0, 0.5, T))  # heading (Y-up: yaw is the Y rotation)
        root = np.stack([np.cumsum(rng.normal(0, 1.0, T)), 92 + 2 * np.sin(t), np.cumsum(rng.normal(0, 1.0, T))], -1)
        rows = np.concatenate([root, ang.reshape(T, -1)], axis=1)
        p = os.path.join(tmpd, f"clip{k:03d}.bvh")
        with open(p, "w") as fh:
            fh.write("\n".join(hdr) + f"\nMOTION\nFrames: {T}\nFrame Time: 0.033333\n")
            np.savetxt(fh, rows, fmt="%.6f")
        files.append(p)
    return files


def main():
    n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
    threads = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    from gmr_amd import GeneralMotionRetargeting as GMR
    from gmr_amd.bvh import load_lafan1_files
    tmpd = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        t0 = time.perf_counter()
        files = write_files(tmpd, n_files, T)
        t_write = time.perf_counter() - t0
        nbytes = sum(os.path.getsize(f) for f in files)
        g = GMR(src_human="bvh", tgt_robot="unitree_g1")
        res = {"files": n_files, "frames": n_files * T, "text_MB": nbytes / 1e6, "write_s": t_write, "threads": threads}
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            batch = load_lafan1_files(files, threads=threads)
            torch.cuda.synchronize(); t_load = time.perf_counter() - t0
            t0 = time.perf_counter()
            q = g.retarget_batch(batch.pos, batch.quat, batch.body_names, seq_offsets=batch.seq_offsets, human_heights=batch.human_heights, chunk=64, burn_in=32)
            torch.cuda.synchronize(); t_ik = time.perf_counter() - t0
        N = n_files * T
        res.update({"load_s": t_load, "load_frames_per_s": N / t_load, "parse_MB_per_s": nbytes / 1e6 / t_load, "ik_s": t_ik, "ik_frames_per_s": N / t_ik,
                    "files_to_qpos_frames_per_s": N / (t_load + t_ik), "resolved_frames": g.last_chunk_info["resolved_frames"],
                    "reference_loader": "~600 frames/s (SURVEY f-1: 0.4 s for 250 frames x 101 joints in the reference's Python loader)"})
        print(json.dumps(res))
    finally:
        shutil.rmtree(tmpd, ignore_errors=True)


if __name__ == "__main__":
    main()
