"""Soak of the two folder pipelines (files -> key-points -> qpos -> pickles): the same folders converted over and over for the given number of
seconds; every pass must write byte-identical pickles, and host RSS / device memory must level off (the pinned read buffers and the scratch pool
are grow-only by design, nothing else may accumulate).

    python tools/soak_pipelines.py [seconds]
"""
import hashlib
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import psutil
import torch


def digest(folder):
    h = hashlib.sha256()
    for n in sorted(os.listdir(folder)):
        h.update(n.encode())
        h.update(open(os.path.join(folder, n), "rb").read())
    return h.hexdigest()


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    from gmr_amd import GeneralMotionRetargeting as GMR, dataset, synth
    from gmr_amd import smplx_adapter as sa
    from gmr_amd.bvh import iter_lafan1_batches
    dev = torch.device("cuda", 0)
    tmpd = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    proc = psutil.Process()
    try:
        gs = GMR(src_human="smplx", tgt_robot="unitree_g1")
        gb = GMR(src_human="bvh", tgt_robot="unitree_g1")
        lens = np.random.default_rng(0).integers(100, 900, 384)
        pos, quat, names, offs = synth.synth_clips_torch(gs._cm, lens, seed=5, device=dev, hard=np.arange(len(lens)) % 2 == 1, yaw0=1.0, dtype=torch.float64)
        d_s, d_b = os.path.join(tmpd, "smplx"), os.path.join(tmpd, "bvh")
        os.makedirs(d_s); os.makedirs(d_b)
        sfiles = synth.write_smplx_joint_files(d_s, pos, quat, names, offs, fps=30.0, heights=list(np.linspace(1.5, 1.9, len(lens))))
        bl = np.full(12, 1500)
        bpos, bquat, bnames, boffs = synth.synth_clips_torch(gb._cm, bl, seed=6, device=dev, yaw0=1.0, dtype=torch.float64)
        bfiles = synth.write_keypoint_files(d_b, bpos.cpu().numpy(), bquat.cpu().numpy(), bnames, boffs, head_height=gb._cm.config.human_height_assumption)
        del pos, quat, bpos, bquat
        first, t0, passes, rss, devmem = None, time.time(), 0, [], []
        while time.time() - t0 < seconds:
            out = os.path.join(tmpd, "out")
            shutil.rmtree(out, ignore_errors=True)
            with dataset.MotionWriter(workers=8, override=True) as w:
                for b in sa.iter_joint_batches(sfiles, batch_files=128, columns=gs.ik_columns, threads=8):
                    w.submit(dataset.retarget_clips(gs, b.pos, b.quat, b.body_names, b.seq_offsets, fps=b.fps, human_heights=b.human_heights),
                             [os.path.join(out, "s_" + os.path.basename(f)[:-4] + ".pkl") for f in b.files])
                for b in iter_lafan1_batches(bfiles, batch_files=6, columns=gb.ik_columns, threads=8):
                    w.submit(dataset.retarget_clips(gb, b.pos, b.quat, b.body_names, b.seq_offsets, fps=30, height_adjust=False, root_origin_offset=False,
                                                    chunk="auto", human_heights=b.human_heights),
                             [os.path.join(out, "b_" + os.path.basename(f)[:-4] + ".pkl") for f in b.files])
            d = digest(out)
            first = first or d
            assert d == first, "a pass wrote different bytes"
            passes += 1
            rss.append(proc.memory_info().rss / 2 ** 20)
            devmem.append(torch.cuda.memory_reserved(dev) / 2 ** 20)
            if passes % 10 == 0:
                print(f"{passes} passes, RSS {rss[-1]:.0f} MiB, device reserved {devmem[-1]:.0f} MiB, {time.time() - t0:.0f} s", flush=True)
        k = max(3, passes // 3)
        print(f"pipeline soak ok: {passes} passes over {len(sfiles)} joint files + {len(bfiles)} BVH files, {len(sfiles) + len(bfiles)} pickles byte-identical in every pass; "
              f"host RSS {rss[min(2, passes - 1)]:.0f} MiB after pass 3 -> {rss[-1]:.0f} MiB at the end (max of the last third {max(rss[-k:]):.0f}), "
              f"device memory reserved {devmem[min(2, passes - 1)]:.0f} -> {devmem[-1]:.0f} MiB, {time.time() - t0:.0f} s")
    finally:
        shutil.rmtree(tmpd, ignore_errors=True)


if __name__ == "__main__":
    main()
