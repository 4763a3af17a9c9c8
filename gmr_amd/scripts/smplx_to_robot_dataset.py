"""scripts/smplx_to_robot_dataset.py behind the body model: a folder of joint-array files -> one pickle per clip, same flags, same file layout.

The reference loads each AMASS file, runs the licensed SMPL-X body model, aligns the frame rate, retargets frame by frame and writes a pickle, in
``--num_cpus`` processes (:63-146, 241-242).  The body model stays on the reference side: its outputs are dumped once per clip with
``gmr_amd.smplx_adapter.save_joint_file``; this script does everything behind it on the GPU, batch by batch, with the same
folder walk, the same exclusions (``_stagei`` files, the hard-motion lists, the BMLrub / EKUT / crawl / _lie / stairs names, :193-227) and the
same targets.
"""
from __future__ import annotations

import argparse
import os

EXCLUDE_FILE_CONTENT = ["BMLrub", "EKUT", "crawl", "_lie", "upstairs", "downstairs"]  # smplx_to_robot_dataset.py:218


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    ap.add_argument("--robot", default="unitree_g1")
    ap.add_argument("--src_folder", type=str, required=True, help="folder of joint-array .npz files (smplx_adapter.save_joint_file)")
    ap.add_argument("--tgt_folder", type=str, required=True)
    ap.add_argument("--override", default=False, action="store_true")
    ap.add_argument("--num_cpus", default=4, type=int, help="host threads reading files / writing pickles (the reference's worker processes)")
    ap.add_argument("--hard_motions", nargs="*", default=None, help="lists of motions to leave out (default: $GMR_ROOT/assets/hard_motions/0.txt, 1.txt when present)")
    ap.add_argument("--batch_files", default=1024, type=int)
    ap.add_argument("--device", default=None, type=int, help="GPU to use (default: LOCAL_RANK under torch.distributed.run, else 0)")
    ap.add_argument("--clip_start", default="qpos0", choices=["qpos0", "root_target"],
                    help="qpos0: the reference (every clip starts from the model's rest pose); root_target: start with the floating base on the first root target (not the reference's numbers for the first frames; spares clips that face away from qpos0 their slow start)")
    ap.add_argument("--shard_by_rank", default=False, action="store_true", help="under torch.distributed.run: convert files[RANK::WORLD_SIZE] only (no exchange between ranks)")
    args = ap.parse_args(argv)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if args.device is None:
        args.device = int(os.environ.get("LOCAL_RANK", "0"))
    from ._walk import hard_motion_names, plan_files
    srcs, tgts, skipped = plan_files(args.src_folder, args.tgt_folder, lambda n: n.endswith(".npz") and not n.endswith("_stagei.npz"), ".npz", args.override, natural=True)
    print("full args_list:", len(srcs))
    lists = args.hard_motions
    if lists is None:
        root = os.environ.get("GMR_ROOT", "")
        lists = [os.path.join(root, "assets", "hard_motions", n) for n in ("0.txt", "1.txt")] if root else []
    hard = set(hard_motion_names(lists))
    keep = []
    for s, t in zip(srcs, tgts):
        name = s.split("/")[-1].split(".")[0]
        if name in hard or any(c in name for c in EXCLUDE_FILE_CONTENT):
            continue
        keep.append((s, t))
    print("new args_list:", len(keep))
    print(f"Total number of files to process: {len(keep)}")
    if args.shard_by_rank and world > 1:
        keep = keep[rank::world]
        print(f"rank {rank} of {world}: {len(keep)} of them")
    if not keep:
        print("Done. Saved to ", args.tgt_folder)
        return 0
    from .. import GeneralMotionRetargeting as GMR, dataset
    from ..smplx_adapter import iter_joint_batches
    g = GMR(src_human="smplx", tgt_robot=args.robot, device=args.device)
    target_of = dict(keep)
    failed = 0
    with dataset.MotionWriter(workers=max(1, args.num_cpus), override=True) as writer:
        for batch in iter_joint_batches([s for s, _ in keep], batch_files=args.batch_files, device=args.device, threads=max(1, args.num_cpus), columns=g.ik_columns,
                                        skip_errors=True):
            for f, why in batch.skipped:
                print(f"Error loading {f}: {why}")
                failed += 1
            if not len(batch):
                continue
            motions = dataset.retarget_clips(g, batch.pos, batch.quat, batch.body_names, batch.seq_offsets, fps=batch.fps, human_heights=batch.human_heights, clip_start=args.clip_start)  # :97-141
            writer.submit(motions, [target_of[f] for f in batch.files])
    print(f"{writer.written} files written, {failed} could not be loaded")
    print("Done. Saved to ", args.tgt_folder)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
