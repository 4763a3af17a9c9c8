"""Child process of tests/test_gpu_multi.py: one rank of a 2-GPU RCCL run (started fresh, before any GPU call of its parent).

Each rank owns one GPU.  Checks, on the device and over RCCL: model broadcast, gather_rows of real solver output (equal and
unequal clip lengths), and the chunk-sharded long-clip solve (both exchanges, B-row rewrite detection, packed solve counts)
against the single-GPU verified-chunked solve of the same clips.  Prints one JSON line per rank.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    import torch.distributed as dist
    from gmr_amd import distributed as gdist, params, synth
    from gmr_amd._native import INIT_QPOS0
    from gmr_amd.engine import Engine
    from gmr_amd.ik_config import load_ik_config
    from gmr_amd.mjcf import load_robot
    from gmr_amd.model import compile_model
    from gmr_amd.schedule import make_items

    backend = os.environ.get("GMR_TEST_BACKEND", "nccl")   # "gloo" + GMR_TEST_SHARE_GPU=1: the rehearsal on a one-GPU box
    if os.environ.get("GMR_TEST_SHARE_GPU") == "1":
        os.environ["LOCAL_RANK"] = "0"
    rank, world, local = gdist.init_from_env(backend)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cm = compile_model(load_robot(params.ROBOT_XML_DICT["unitree_g1"], name="unitree_g1"), load_ik_config(params.IK_CONFIG_DICT["smplx"]["unitree_g1"]))
    blob = gdist.broadcast_blob(cm.blob if rank == 0 else None)
    res = {"rank": rank, "world": world, "backend": dist.get_backend(), "blob_ok": blob == cm.blob}
    eng = Engine(cm, local)

    # clip-sharded solve + gather_rows on the device over RCCL, unequal and equal lengths
    for label, lengths in (("unequal", [130, 40, 77, 12, 95, 64, 3]), ("equal", [48] * 9)):
        pos, quat, names, _, _ = synth.synth_clips(cm, 1, sum(lengths), seed=11, hard=True, dtype=np.float32)
        offs = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
        sc = cm.slot_columns(names)
        tp, tq = torch.from_numpy(pos).to(dev), torch.from_numpy(quat).to(dev)
        q_all, _, _ = eng.ik_solve(tp, tq, sc, make_items(offs))                      # every rank: the whole set (the expectation)
        mine = gdist.my_clips(lengths)
        rows = np.concatenate([np.arange(offs[i], offs[i + 1]) for i in mine])
        lo = np.concatenate([[0], np.cumsum([lengths[i] for i in mine])]).astype(np.int64)
        q_mine, _, _ = eng.ik_solve(tp[rows].contiguous(), tq[rows].contiguous(), sc, make_items(lo))
        full = gdist.gather_rows(q_mine, lengths, slab_bytes=world * 288 * 100)          # several slabs
        res[f"gather_{label}_bitwise"] = bool(torch.equal(full, q_all)) and full.device == dev

    # few long clips: chunks of every clip on both ranks, two exchanges; vs the one-GPU verified-chunked solve
    lengths = [700, 333, 520]
    pos, quat, names, offs = synth.synth_clips_torch(cm, np.array(lengths), seed=5, device=dev, hard=np.array([True, False, True]), yaw0=1.0)
    sc = cm.slot_columns(names)
    heights = [1.0, 0.93, 1.05]
    q1, it1, info1 = eng.ik_solve_chunked(pos, quat, sc, offs, chunk=32, burn_in=16, height_scales=heights)
    q2, it2, info2 = eng.ik_solve_chunked_sharded(pos, quat, sc, offs, 32, 16, height_scales=heights)
    res["sharded_vs_single_max_abs_diff"] = float((q1 - q2).abs().max().item())
    res["sharded_equals_single"] = bool(res["sharded_vs_single_max_abs_diff"] < 1e-9 and torch.equal(it1 & 0x3FFFFFFF, it2 & 0x3FFFFFFF))
    res["resolved"] = [info1["resolved_frames"], info2["resolved_frames"], info1.get("resolved_chunks"), info2["resolved_chunks"]]
    res["resolved_equal"] = bool(info1["resolved_frames"] == info2["resolved_frames"])
    # deliberately poor chunk starts (qpos0 at the origin): many chunks re-solved, so exchange 2 carries real rows
    import gmr_amd.schedule as sched
    orig = sched.make_items
    sched.make_items = lambda *a, **k: orig(*a, **{**k, "chunk_init": INIT_QPOS0})
    try:
        q3, it3, info3 = eng.ik_solve_chunked_sharded(pos, quat, sc, offs, 32, 2, height_scales=heights)
    finally:
        sched.make_items = orig
    q_seq, it_seq, _ = eng.ik_solve(pos, quat, sc, make_items(offs, height_scales=heights))
    res["poor_starts_max_abs_diff"] = float((q3 - q_seq).abs().max().item())
    res["poor_starts_iters_equal"] = bool(torch.equal(it3 & 0x3FFFFFFF, it_seq & 0x3FFFFFFF))
    res["poor_starts_resolved_chunks"] = int(info3["resolved_chunks"])
    print("RESULT " + json.dumps(res), flush=True)
    if backend == "nccl":
        dist.barrier(device_ids=[local])
    else:
        dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
