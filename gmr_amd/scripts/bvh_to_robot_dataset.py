"""scripts/bvh_to_robot_dataset.py on this engine: a folder of BVH files -> one pickle per clip, same flags, same file layout.

The reference converts file by file, frame by frame (:59-151).  Here the files of a batch are read ahead and parsed on the GPU, solved in one
launch (verified parallel-in-time chunks, one height estimate per file), post-processed and written by a thread pool while the next batch
is on the GPU.  Files that cannot be loaded are reported and skipped like the reference's ``except: print; continue`` (:75-80).
"""
from __future__ import annotations

import argparse
import os


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    ap.add_argument("--src_folder", required=True, type=str, help="Folder containing BVH motion files to load.")
    ap.add_argument("--tgt_folder", default="../../motion_data/LAFAN1_g1_gmr", help="Folder to save the retargeted motion files.")
    ap.add_argument("--robot", default="unitree_g1")
    ap.add_argument("--override", default=False, action="store_true")
    ap.add_argument("--target_fps", default=30, type=int, help="(accepted like the reference, which stores 30 whatever it is given)")
    ap.add_argument("--batch_files", default=64, type=int, help="files per GPU batch (one skeleton per batch)")
    ap.add_argument("--threads", default=8, type=int, help="host threads reading files / writing pickles")
    ap.add_argument("--device", default=None, type=int, help="GPU to use (default: LOCAL_RANK under torch.distributed.run, else 0)")
    ap.add_argument("--clip_start", default="qpos0", choices=["qpos0", "root_target"],
                    help="qpos0: the reference (every clip starts from the model's rest pose); root_target: start with the floating base on the first root target (not the reference's numbers for the first frames; spares clips that face away from qpos0 their slow start)")
    ap.add_argument("--shard_by_rank", default=False, action="store_true", help="under torch.distributed.run: convert files[RANK::WORLD_SIZE] only (no exchange between ranks)")
    args = ap.parse_args(argv)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if args.device is None:
        args.device = int(os.environ.get("LOCAL_RANK", "0"))
    from ._walk import plan_files
    srcs, tgts, skipped = plan_files(args.src_folder, args.tgt_folder, lambda n: n.endswith(".bvh"), ".bvh", args.override)
    print(f"{len(srcs)} files to retarget ({skipped} skipped: target exists)")
    if args.shard_by_rank and world > 1:
        srcs, tgts = srcs[rank::world], tgts[rank::world]
        print(f"rank {rank} of {world}: {len(srcs)} of them")
    if not srcs:
        print("Done. saved to ", args.tgt_folder)
        return 0
    from .. import GeneralMotionRetargeting as GMR, dataset
    from ..bvh import iter_lafan1_batches
    g = GMR(src_human="bvh", tgt_robot=args.robot, device=args.device)
    target_of = dict(zip(srcs, tgts))
    failed = 0
    with dataset.MotionWriter(workers=max(1, args.threads), override=True) as writer:
        todo = srcs
        while todo:  # a batch holds one skeleton (its first readable file's); files of another one wait for the next pass
            again = []
            for batch in iter_lafan1_batches(todo, batch_files=args.batch_files, device=args.device, threads=args.threads, columns=g.ik_columns, skip_errors=True):
                for f, why in batch.skipped:
                    if "skeleton differs" in why:
                        again.append(f)
                    else:
                        print(f"Error loading {f}: {why}")
                        failed += 1
                if not len(batch):
                    continue
                motions = dataset.retarget_clips(g, batch.pos, batch.quat, batch.body_names, batch.seq_offsets, fps=30, height_adjust=False,   # :127-128
                                                 root_origin_offset=False, chunk="auto", human_heights=batch.human_heights, clip_start=args.clip_start)
                writer.submit(motions, [target_of[f] for f in batch.files])
            todo = again if len(again) < len(todo) else []
    print(f"{writer.written} files written, {failed} could not be loaded")
    print("Done. saved to ", args.tgt_folder)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
