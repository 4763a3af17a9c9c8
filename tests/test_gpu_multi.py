"""Two ranks on two GPUs over RCCL: skipped where fewer than two devices are visible (the one-GPU box).

The ranks are fresh child processes started before this process makes any GPU call (`torch.cuda.device_count()` does not
initialise HIP on this image), one GPU each; see tests/helpers/rccl_two_rank_worker.py for what they check.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_two_ranks(extra_env):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(2):
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.update(extra_env)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "helpers", "rccl_two_rank_worker.py")], env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=600))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    for so, _ in outs:
        line = [ln for ln in so.splitlines() if ln.startswith("RESULT ")]
        assert len(line) == 1
        d = json.loads(line[0][7:])
        assert d["world"] == 2 and d["backend"] == extra_env.get("GMR_TEST_BACKEND", "nccl") and d["blob_ok"]
        assert d["gather_unequal_bitwise"] and d["gather_equal_bitwise"]
        assert d["sharded_equals_single"] and d["resolved_equal"], d
        assert d["poor_starts_max_abs_diff"] < 1e-6 and d["poor_starts_iters_equal"] and d["poor_starts_resolved_chunks"] > 0, d


def test_two_ranks_two_gpus_over_rccl():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL between devices); the N > 1 path stays 'unmeasured on hardware' where this is skipped")
    _run_two_ranks({})


def test_two_ranks_sharing_one_gpu_gloo_rehearsal():
    """The same worker with CPU collectives and both ranks on cuda:0: everything the RCCL run does on the device -- the kernels'
    B-row rewrite that marks re-solved chunks, in-place range writes, packed solve counts in exchange 2 -- except RCCL itself."""
    _run_two_ranks({"GMR_TEST_BACKEND": "gloo", "GMR_TEST_SHARE_GPU": "1"})
