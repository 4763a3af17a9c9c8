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
