"""Throughput of the KinematicsModel FK kernels (gmr_fk / gmr_fk_min_height) against the HBM roofline."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gmr_amd import params
from gmr_amd.mjcf import load_robot
from gmr_amd.ik_config import load_ik_config
from gmr_amd.model import compile_model
from gmr_amd.engine import Engine

robot = sys.argv[1] if len(sys.argv) > 1 else "unitree_g1"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8_000_000
cm = compile_model(load_robot(params.ROBOT_XML_DICT[robot], name=robot), load_ik_config(params.IK_CONFIG_DICT["smplx"][robot]))
eng = Engine(cm, 0)
dev = eng.device
g = torch.Generator(device=dev).manual_seed(0)
nd = eng.nq - 7
root = torch.randn(N, 3, device=dev, generator=g)
rot = torch.nn.functional.normalize(torch.randn(N, 4, device=dev, generator=g), dim=1)
dof = 0.5 * torch.randn(N, nd, device=dev, generator=g)
out = {}
only = os.environ.get("FK_BENCH_ONLY")
for name, fn, bytes_per in (
    ("fk_pos", lambda: eng.fk(root, rot, dof, want_rot=False), (7 + nd) * 4 + eng.nbody * 12),
    ("fk_pos_rot", lambda: eng.fk(root, rot, dof, want_rot=True), (7 + nd) * 4 + eng.nbody * 28),
    ("fk_min_height", lambda: eng.fk_min_height(root, rot, dof, np.arange(0, N + 1, 4000)), (7 + nd) * 4),
):
    if only and name != only:
        continue
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    out[name] = {"ms": ms, "frames_per_s": N / ms * 1e3, "GBps": N * bytes_per / ms / 1e6, "frac_of_8TBps": N * bytes_per / ms / 1e6 / 8000.0, "bytes_per_frame": bytes_per}
print(json.dumps({"robot": robot, "frames": N, **out}))
