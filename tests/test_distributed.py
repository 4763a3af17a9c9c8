"""world_size-2 gloo tests (CPU) of the multi-GPU plumbing: model broadcast, clip sharding, result gather."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.util import compiled


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, lengths, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from gmr_amd import distributed as gdist
    r, w, _ = gdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    cm = compiled("smplx", "unitree_g1")
    blob = gdist.broadcast_blob(cm.blob if rank == 0 else None)
    ok_blob = blob == cm.blob
    mine = gdist.my_clips(lengths)
    offs = np.concatenate([[0], np.cumsum(lengths)])
    # each rank "solves" its clips: row value = global frame index, so the gathered result must be arange
    local = torch.cat([torch.arange(offs[i], offs[i + 1], dtype=torch.float64)[:, None].repeat(1, 3) for i in mine]) if mine else torch.zeros((0, 3), dtype=torch.float64)
    full = gdist.gather_rows(local, lengths)
    ok_gather = torch.equal(full[:, 0], torch.arange(offs[-1], dtype=torch.float64))
    # the same through several slabs (24 bytes per row: at most two rows per rank and collective), into a caller's buffer,
    # and for equal-length clips (clip-axis placement), with a clip count the ranks do not share evenly
    buf = torch.full((int(offs[-1]), 3), -1.0, dtype=torch.float64)
    ok_gather &= gdist.gather_rows(local, lengths, out=buf, slab_bytes=2 * world * 24) is buf
    ok_gather &= torch.equal(buf[:, 2], torch.arange(offs[-1], dtype=torch.float64))
    for n_eq, slab in ((7, 8 << 30), (7, 3 * 5 * 24 * world), (8, 5 * 24 * world)):
        eq = [5] * n_eq
        mine_eq = gdist.my_clips(eq)
        loc = torch.cat([torch.arange(5 * i, 5 * i + 5, dtype=torch.float64)[:, None].repeat(1, 3) for i in mine_eq])
        ok_gather &= torch.equal(gdist.gather_rows(loc, eq, slab_bytes=slab)[:, 1], torch.arange(5 * n_eq, dtype=torch.float64))
    q.put((rank, ok_blob, mine, bool(ok_gather)))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_shard_gather_world2():
    lengths = [30, 7, 19, 11, 4]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, lengths, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert all(r[1] and r[3] for r in res)
    assert sorted(res[0][2] + res[1][2]) == list(range(len(lengths)))
    loads = [sum(lengths[i] for i in r[2]) for r in res]
    assert abs(loads[0] - loads[1]) <= max(lengths)


def test_broadcast_shard_gather_world4():
    """The same plumbing on four ranks (a rehearsal of more ranks than the two-rank tests: uneven clip counts per rank, a rank
    whose share is a single clip, slabs that some ranks run out of before others)."""
    lengths = [30, 7, 19, 11, 4, 25, 3]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 4, port, lengths, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] and r[3] for r in res)
    assert sorted(i for r in res for i in r[2]) == list(range(len(lengths)))


def test_gather_plan_is_per_clip_and_fast():
    """The host side of gather_rows works on clips, not rows: planning 2 x 8192 clips x 3000 frames (49 M rows) takes milliseconds,
    equal or unequal lengths; slabs cover every local clip exactly once and respect the byte bound."""
    import time
    from gmr_amd.distributed import plan_gather
    rng = np.random.default_rng(1)
    for lengths in (np.full(16384, 3000), rng.integers(1000, 5001, size=16384)):
        t0 = time.perf_counter()
        parts, slabs = plan_gather(lengths, 2, 288, 1 << 30)
        dt = time.perf_counter() - t0
        assert dt < 0.05, dt
        assert sorted(np.concatenate(parts).tolist()) == list(range(16384))
        for r in range(2):
            assert [s["clips"][r][0] for s in slabs][1:] == [s["clips"][r][1] for s in slabs][:-1]
            assert slabs[0]["clips"][r][0] == 0 and slabs[-1]["clips"][r][1] == len(parts[r])
            assert slabs[-1]["rows"][r][1] == int(lengths[parts[r]].sum())
        assert len(slabs) > 1 and all(s["pad"] * 288 * 2 <= (1 << 30) for s in slabs)


def _oracle_solver(orc, pos, quat, sc):
    """The ``solve`` callable of solve_chunked_sharded backed by the CPU oracle's work-item restatement."""
    def solve(items, qinit, qfinal, out, iters, done):
        _, _, _, dn = orc.ik_solve(pos, quat, sc, items, qpos_init=None if qinit is None else qinit.numpy().copy(),
                                   qpos_final=qfinal.numpy(), want_done=True, out=out.numpy(), iters=iters.numpy())
        if done is not None:
            done.numpy()[:] = dn
    return solve


def _chunk_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from gmr_amd import distributed as gdist, synth
    from gmr_amd._native import INIT_QPOS0
    from gmr_amd.schedule import make_items
    from oracle.oracle import Oracle
    gdist.init_from_env(backend="gloo")
    cm = compiled("smplx", "unitree_g1")
    orc = Oracle(cm.blob)
    lengths = [150, 37, 96]
    pos, quat, names, _, _ = synth.synth_clips(cm, 1, sum(lengths), seed=5, hard=True, dtype=np.float32)
    offs = np.concatenate([[0], np.cumsum(lengths)])
    sc = cm.slot_columns(names)
    heights = [1.0, 0.93, 1.05]
    solve = _oracle_solver(orc, pos, quat, sc)
    out, iters, info = gdist.solve_chunked_sharded(solve, int(offs[-1]), orc.nq, offs, 8, 6, torch.device("cpu"), height_scales=heights)
    q_seq, it_seq, _ = orc.ik_solve(pos, quat, sc, make_items(offs, height_scales=heights))
    ok = bool(np.abs(out.numpy() - q_seq).max() < 1e-6 and np.array_equal(iters.numpy(), it_seq))
    # the same from deliberately poor starts: more chunks must travel in the second exchange
    import gmr_amd.schedule as sched
    orig = sched.make_items
    sched.make_items = lambda *a, **k: orig(*a, **{**k, "chunk_init": INIT_QPOS0})
    out2, iters2, info2 = gdist.solve_chunked_sharded(solve, int(offs[-1]), orc.nq, offs, 8, 2, torch.device("cpu"), height_scales=heights)
    sched.make_items = orig
    ok2 = bool(np.abs(out2.numpy() - q_seq).max() < 1e-6 and np.array_equal(iters2.numpy(), it_seq))
    q.put((rank, ok, ok2, info, info2))
    dist.barrier()
    dist.destroy_process_group()


def test_long_clips_chunks_sharded_world2():
    """BASELINE config 3 on N ranks (world 2, gloo, the oracle's work items as the per-rank solver): chunks of every clip spread
    over both ranks, boundary states and rows all-gathered, walks on the clips' owners, re-solved chunks sent back -- every rank
    ends with the sequential result (values and solve counts), also when the speculative starts are poor."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_chunk_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] and r[2] for r in res), res
    assert res[0][3] == res[1][3] and res[0][4] == res[1][4]           # both ranks agree on what was re-solved
    assert res[0][3]["ranks"] == 2 and res[0][4]["resolved_chunks"] > res[0][3]["resolved_chunks"] >= 0
    assert res[0][4]["resolved_frames"] > 0
