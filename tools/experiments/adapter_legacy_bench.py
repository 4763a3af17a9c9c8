"""Before / after of the adapter kernels through the entry points both libraries export (ABI <= 3: gmr_bvh_fk on split arrays,
gmr_smplx_keypoints): round 2's one-frame-per-lane kernels against round 3's lane-per-joint kernels, same inputs, same box.

    python tools/experiments/adapter_legacy_bench.py <old libgmr_amd.so>
"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gmr_amd import synth  # noqa: E402
from gmr_amd.smplx_adapter import SMPLX_PARENTS  # noqa: E402

vp = C.c_void_p
dev = torch.device("cuda", 0)


def timed(fn, steps=3):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / steps


def run(lib, T=2_000_000, Ts=500_000):
    out = {}
    rows, parents, offsets, order = synth.lafan_rows_torch(T, dev)
    J = len(parents)
    lp = torch.from_numpy(offsets).to(dev)[None].repeat(T, 1, 1).contiguous()
    lp[:, 0] = rows[:, 0:3]
    er = torch.deg2rad(rows[:, 3:]).reshape(T, J, 3).contiguous()
    pos = torch.empty((T, J + 2, 3), dtype=torch.float64, device=dev)
    quat = torch.empty((T, J + 2, 4), dtype=torch.float64, device=dev)
    od, ep, erot = np.asarray(order, np.int32), np.array([3, 7], np.int32), np.array([4, 8], np.int32)
    ms = timed(lambda: lib.gmr_bvh_fk(parents.ctypes.data_as(vp), J, od.ctypes.data_as(vp), ep.ctypes.data_as(vp), erot.ctypes.data_as(vp), 2, vp(lp.data_ptr()),
                                      vp(er.data_ptr()), T, C.c_double(0.01), vp(pos.data_ptr()), vp(quat.data_ptr()), None))
    out["bvh_split_arrays"] = {"frames": T, "ms": ms, "frames_per_s": T / ms * 1e3, "bytes_per_frame": J * 48 + (J + 2) * 56, "GBps": (J * 48 + (J + 2) * 56) * T / ms / 1e6}
    chk = [float(pos.sum().item()), float(quat.abs().sum().item())]
    del lp, er, pos, quat, rows
    par = np.asarray(SMPLX_PARENTS, np.int32)
    for mode, skip in (("resample_120_to_30", 4), ("one_to_one", 1)):
        go, fp, jt = synth.smplx_arrays_torch(Ts * skip, dev, 55, 127)
        pos = torch.empty((Ts, 55, 3), dtype=torch.float64, device=dev)
        quat = torch.empty((Ts, 55, 4), dtype=torch.float64, device=dev)
        ms = timed(lambda: lib.gmr_smplx_keypoints(par.ctypes.data_as(vp), 55, 127, vp(go.data_ptr()), vp(fp.data_ptr()), vp(jt.data_ptr()), Ts * skip, Ts, int(skip > 1),
                                                   vp(pos.data_ptr()), vp(quat.data_ptr()), None))
        bpf = (2 if skip > 1 else 1) * 110 * 24 + 55 * 56
        out["smplx_" + mode] = {"frames": Ts, "ms": ms, "frames_per_s": Ts / ms * 1e3, "bytes_per_frame": bpf, "GBps": bpf * Ts / ms / 1e6}
        chk += [float(pos.sum().item()), float(quat.abs().sum().item())]
        del go, fp, jt, pos, quat
    out["checksums"] = chk
    return out


def load(path):
    lib = C.CDLL(path)
    lib.gmr_bvh_fk.restype = C.c_int
    lib.gmr_bvh_fk.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, C.c_int64, C.c_double, vp, vp, vp]
    lib.gmr_smplx_keypoints.restype = C.c_int
    lib.gmr_smplx_keypoints.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, C.c_int64, C.c_int64, C.c_int, vp, vp, vp]
    return lib


if __name__ == "__main__":
    res = {"round2_kernels": run(load(sys.argv[1])), "round3_kernels": run(load(os.path.join(ROOT, "gmr_amd", "lib", "libgmr_amd.so")))}
    for k in ("bvh_split_arrays", "smplx_resample_120_to_30", "smplx_one_to_one"):
        res.setdefault("speedup", {})[k] = res["round2_kernels"][k]["ms"] / res["round3_kernels"][k]["ms"]
    print(json.dumps(res, indent=1))
