"""Where the time of a host-fed batch goes (Engine.ik_solve_host): pinned allocation, staging copies, PCIe both ways, kernel."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch

def t(fn, reps=3):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); a = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - a)
    return min(ts)

N = 4_000_000
dev = torch.device("cuda", 0)
res = {}
pos = np.random.default_rng(0).normal(size=(N, 14, 3)).astype(np.float32)      # 168 B/frame
big = np.random.default_rng(0).normal(size=(N // 4, 55, 3)).astype(np.float32)  # 55-joint layout
gb = pos.nbytes / 1e9
res["pin_alloc_GBps"] = (N * 288 / 1e9) / t(lambda: torch.empty((N, 36), dtype=torch.float64, pin_memory=True), reps=2)
res["pageable_alloc_GBps"] = (N * 288 / 1e9) / t(lambda: np.empty((N, 36)), reps=2)
hp = torch.empty((N, 14, 3), dtype=torch.float32, pin_memory=True)
tp = torch.from_numpy(pos)
res["stage_copy_GBps"] = gb / t(lambda: hp.copy_(tp))
cols = torch.arange(14) * 3
hp4 = hp[: N // 4]
tb = torch.from_numpy(big)
res["stage_index_select_GBps_out"] = (hp4.numel() * 4 / 1e9) / t(lambda: torch.index_select(tb, 1, cols, out=hp4))
torch.set_num_threads(16)
res["stage_copy_GBps_16thr"] = gb / t(lambda: hp.copy_(tp))
dp = torch.empty((N, 14, 3), dtype=torch.float32, device=dev)
res["h2d_pinned_GBps"] = gb / t(lambda: dp.copy_(hp, non_blocking=True))
res["h2d_pageable_GBps"] = gb / t(lambda: dp.copy_(tp))
do = torch.empty((N, 36), dtype=torch.float64, device=dev)
ho = torch.empty((N, 36), dtype=torch.float64, pin_memory=True)
res["d2h_pinned_GBps"] = (do.numel() * 8 / 1e9) / t(lambda: ho.copy_(do, non_blocking=True))
po = torch.empty((N, 36), dtype=torch.float64)
res["d2h_pageable_GBps"] = (do.numel() * 8 / 1e9) / t(lambda: po.copy_(do))
# hipHostRegister of a pageable array (pin in place)
rt = torch.cuda.cudart()
arr = np.empty((N, 14, 3), dtype=np.float32); arr[:] = 1
def reg():
    rc = rt.cudaHostRegister(arr.ctypes.data, arr.nbytes, 0); assert int(rc) == 0, rc
    rt.cudaHostUnregister(arr.ctypes.data)
try:
    res["host_register_GBps"] = gb / t(reg, reps=2)
except Exception as e:
    res["host_register_GBps"] = repr(e)
print(json.dumps(res))
