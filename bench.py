#!/usr/bin/env python3
"""Headline benchmark: retargeted frames/s, Unitree G1 29-DoF, SMPL-X mapping (BASELINE.json).

    python bench.py [--gpus N --steps K --warmup W] [--clips S --frames T]

A "step" is one pass of the hot path (two-stage box-constrained IK of every frame, warm start carried
along each clip exactly as the reference's caller loop does) over one batch of synthetic AMASS-shaped
clips already resident in HBM.  N > 1 runs one rank per GPU: under torch.distributed.run (the driver's
launch) the ranks are already there; a bare ``python bench.py --gpus N`` starts them itself (a child
torch.distributed.run, spawned before this process touches the GPU) and relays rank 0's line.  Clips are
independent, so ranks own disjoint clips and there is no collective inside the timed region (weak
scaling: S clips per GPU); rank 0 broadcasts the packed model once before it.

One JSON line on stdout (rank 0).  Besides the driver's contract it carries
  value_unshaped  the same metric on the un-shaped workload: every clip distinct, any initial heading, variable lengths
  roofline        dominant kernel (ik_kernel) vs the bound that binds it, FP64 vector issue: SURVEY 8(d)'s flop per solve x
                  measured solves, over the kernel time from HIP events; `hbm` is the same kernel against the HBM roofline
                  (north_star asks for it; ~0.3 % by construction, not the bound)
  fk (+ kin_ops), adapters, dataset_path (+ from_joint_files), host_fed, single_clip, long_clips (+ from_files), heterogeneous,
  live_session (+ class_api)                                          the other kernels / modes of the path (N = 1), each kernel
                                                                      against its roofline with PMC traffic where a child pass measures it
  collectives, strong, long_clips_sharded                             N > 1: RCCL exchange steps timed on the real outputs
  cpu_baseline    oracle/ (float64 C restatement of the reference algorithm) timed on this host's cores
  parity          max |qpos_gpu - qpos_cpu| on the clips the CPU leg solved (both workloads)
The CPU oracle is used here only as checker / comparator; the timed GPU path never touches it.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VECTOR_PEAK_TF = 78.6  # AMD MI355X datasheet; = 1/2 of the guide's 157.3 TF FP32 (packed) vector peak
ROBOT, SRC = "unitree_g1", "smplx"

# HBM bytes per output frame of ik_kernel measured with rocprofv3 PMC passes (profiles/r01_v17_pmc_*: 2 x FETCH_SIZE
# (gfx950 counts half, MI355X_MICROARCH.md "HBM") + WRITE_SIZE over an 8192 x 600 = 4.9152e6-frame launch; round 2 re-measured):
# 1.93 GB read + 1.44 GB written = 685 B/frame (393 + 292) against 684 algorithmic.  Valid for THIS configuration only
# (unitree_g1 / smplx / float32 key-points); any other launch reports its algorithmic bytes and says so.
MEASURED_TRAFFIC = {"robot": "unitree_g1", "src": "smplx", "in_itemsize": 4,
                    "bytes_per_frame": (2 * 942970.5 * 1024 + 1401604.3 * 1024) / 4915200.0, "source": "profiles/r02_v18_pmc_*"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--clips", type=int, default=8192, help="clips per GPU (AMASS holds >1e4 sequences; several clips per wavefront slot -- 2048 slots on a MI355X -- let the hardware dispatcher balance clips that need more solves than others)")
    ap.add_argument("--frames", type=int, default=3000, help="frames per clip (one AMASS sequence ~3k frames @30fps)")
    ap.add_argument("--distinct", type=int, default=64, help="distinct synthetic clips of the shaped workload, tiled to --clips")
    ap.add_argument("--cpu-clips", type=int, default=0, help="clips solved by the CPU oracle per workload (baseline + parity); 0 = 4 per host core, capped at 512")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-unshaped", action="store_true", help="skip the un-shaped workload (value_unshaped)")
    ap.add_argument("--hot-only", action="store_true", help="only the warmup + timed launches of the headline workload (no other legs): the "
                    "form profiled under rocprofv3 so the kernel's average duration is that of the timed launch")
    ap.add_argument("--hot-adapters", action="store_true", help="with --hot-only: also run the two input-adapter kernels (profiles/r03_adapters_*)")
    ap.add_argument("--hot-fk", action="store_true", help="with --hot-only: also run the timed fk_kernel launches (profiles/r02_fk_*)")
    ap.add_argument("--traffic", choices=["auto", "pmc", "const"], default="auto",
                    help="roofline.traffic: 'pmc' measures HBM bytes of ik_kernel in this run (two child rocprofv3 --pmc passes over a short "
                         "--hot-only launch, FETCH_SIZE x 2 + WRITE_SIZE as MI355X_MICROARCH.md prescribes); 'const' uses the figure of the "
                         "committed profile; 'auto' = pmc when rocprofv3 is available and this is a plain 1-GPU run, else const")
    return ap.parse_args(argv)


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks as a child torch.distributed.run and relay rank 0's line.
    Runs before this process has imported torch or touched the GPU; the parent never initialises HIP (no exec either)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout:
        ln = ln.rstrip("\n")
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        elif ln:
            print(ln, file=sys.stderr)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    return rc if rc != 0 else (0 if line is not None else 1)


ADAPTER_PMC_SIZES = (2_000_000, 500_000)   # (bvh frames, smplx output frames) of the adapters leg inside the PMC child passes


# kernels of the fk leg whose HBM traffic the PMC child passes read (substring of the kernel name -> record)
FK_PMC_KERNELS = ("fk_pos_kernel", "fk_kernel<0", "dof_to_rot_kernel", "rot_to_dof_kernel", "local_to_global_kernel")


def measure_traffic_pmc(frames=600, clips=8192, timeout=300, adapters=True):
    """HBM bytes from PMC counters, measured now: one child `rocprofv3 --pmc C --kernel-trace` run per counter (they do not fit one
    pass) over `bench.py --hot-only [--hot-adapters]` with short clips.  Returns (bytes_per_frame of ik_kernel, detail, adapters)
    -- adapters = {"bvh_fk_kernel": [(FETCH_KB, WRITE_KB) per timed dispatch...], "smplx_keypoints_kernel": [...]} in launch order --
    or (None, reason, None)."""
    import csv
    import glob
    import shutil
    import tempfile
    prof = shutil.which("rocprofv3")
    if prof is None:
        return None, "rocprofv3 not found", None
    if any(k.startswith("ROCPROF") or k.startswith("ROCP_") for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "already under a profiler", None
    vals, ad = {}, {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="gmr_pmc_")
        try:
            cmd = [prof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
                   "--hot-only", "--steps", "1", "--warmup", "1", "--frames", str(frames), "--clips", str(clips), "--traffic", "const"]
            if adapters:
                cmd += ["--hot-adapters", "--hot-fk"]
            r = subprocess.run(cmd, cwd=tempfile.gettempdir(), env=dict(os.environ, TMPDIR=tempfile.gettempdir()), capture_output=True, text=True, timeout=timeout)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, f"{counter} pass failed (rc {r.returncode})", None
            rows = [row for row in csv.DictReader(open(files[0])) if row["Counter_Name"] == counter]
            rows.sort(key=lambda row: int(row.get("Dispatch_Id", 0) or 0))
            got = [float(row["Counter_Value"]) for row in rows if "ik_kernel" in row["Kernel_Name"]]
            if not got:
                return None, f"no ik_kernel rows in the {counter} pass", None
            vals[counter] = got[-1]  # the timed launch (the last dispatch); KB
            for key in ("bvh_fk_kernel", "smplx_keypoints_kernel<double>"):
                seq = [float(row["Counter_Value"]) for row in rows if key in row["Kernel_Name"]]
                ad.setdefault(key, {})[counter] = seq[1::2]  # (warm-up, timed) pairs per configuration: the timed ones
            for key in FK_PMC_KERNELS:  # the fk leg of the same child run: the last dispatch of each kernel
                seq = [float(row["Counter_Value"]) for row in rows if key in row["Kernel_Name"]]
                if seq:
                    ad.setdefault("_fk", {}).setdefault(key, {})[counter] = seq[-1]
        except Exception as ex:  # timeout, parse error: fall back to the committed figure
            return None, f"{counter} pass: {ex!r}", None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    n = frames * clips
    bpf = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0 / n  # gfx950 tallies 128-byte read requests at 64 bytes
    fk_kb = ad.pop("_fk", {})
    adapters_kb = {k: list(zip(v.get("FETCH_SIZE", []), v.get("WRITE_SIZE", []))) for k, v in ad.items()} if adapters else None
    if adapters_kb is not None:
        adapters_kb["_fk"] = {"frames_in_pass": n, "kernels": {k: (v["FETCH_SIZE"], v["WRITE_SIZE"]) for k, v in fk_kb.items() if len(v) == 2}}
    return bpf, {"FETCH_SIZE_KB": vals["FETCH_SIZE"], "WRITE_SIZE_KB": vals["WRITE_SIZE"], "frames_in_pass": n}, adapters_kb


def flops_per_solve(cm) -> float:
    """algorithmic flop per (frame, solve), sparse-Jacobian convention, mean over the two tables."""
    import numpy as np
    rob = cm.robot
    nb, nv = rob.nbody, rob.nv
    out = []
    for tab in range(2):
        T = len(cm.tasks[tab])
        cols = []
        for b in cm.task_body[tab]:
            c = 6
            while b > 0:
                c += int(rob.jnt_type[b] == 1)
                b = rob.parent[b]
            cols.append(c)
        C = sum(cols)
        out.append(100 * (nb - 1) + 150 * T + 40 * C + (200 * T + 72 * C) + sum(6 * c * (c + 1) for c in cols) + 12 * C
                   + (nv ** 3 / 3 + 2 * nv ** 2) + 14 * T)
    return float(np.mean(out))


def host_cores() -> int:
    """CPU threads this process may actually use: cgroup quota, then affinity, then os.cpu_count()."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def bytes_per_frame(cm, in_itemsize=4) -> int:
    """Compulsory HBM traffic of the IK kernel per output frame: key-points in, qpos (f64) + solve count out."""
    return cm.nslot * 7 * in_itemsize + cm.robot.nq * 8 + 4


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    # HBM traffic of the dominant kernel, measured in this run by child profiler passes -- taken first, before this process
    # imports torch or touches the GPU
    live_traffic = (None, "not requested", None)
    if args.traffic == "pmc" or (args.traffic == "auto" and args.gpus == 1 and "WORLD_SIZE" not in os.environ and not args.hot_only):
        live_traffic = measure_traffic_pmc()

    import numpy as np
    import torch

    from gmr_amd import distributed as gdist
    from gmr_amd import GeneralMotionRetargeting, params, synth
    from gmr_amd.ik_config import load_ik_config
    from gmr_amd.mjcf import load_robot
    from gmr_amd.model import compile_model
    from gmr_amd.schedule import make_items

    # Rehearsal of the N > 1 path on a single-GPU box: GMR_BENCH_BACKEND=gloo GMR_BENCH_SHARE_GPU=1 runs every rank on cuda:0
    # with CPU collectives (everything but RCCL itself); the real run uses the defaults (nccl = RCCL, one GPU per rank).
    rank, world, local = gdist.init_from_env(os.environ.get("GMR_BENCH_BACKEND") or None)
    if os.environ.get("GMR_BENCH_SHARE_GPU") == "1":
        local = 0
    on_rccl = world > 1 and torch.distributed.get_backend() == "nccl"
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the engine has no CPU path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    # ---- model: rank 0 compiles, everyone receives the same blob ----
    cm = compile_model(load_robot(params.ROBOT_XML_DICT[ROBOT], name=ROBOT), load_ik_config(params.IK_CONFIG_DICT[SRC][ROBOT]))
    blob = gdist.broadcast_blob(cm.blob if rank == 0 else None)
    assert blob == cm.blob, "packed model differs between ranks"
    gmr = GeneralMotionRetargeting(SRC, ROBOT, device=local)
    eng = gmr._engine
    assert gmr._cm.blob == cm.blob
    kernel_name = f"gmr::ik_kernel<{eng.info.nv_padded}, {'true' if eng.info.reserved[0] else 'false'}>"

    def barrier():
        if world > 1:
            if on_rccl:
                torch.distributed.barrier(device_ids=[local])  # RCCL: the barrier runs on this rank's own GPU
            else:
                torch.distributed.barrier()

    def max_over_ranks(x: float) -> float:
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev if on_rccl else "cpu")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        return float(t.item())

    def timed_steps(fn, steps, warmup):
        """The driver's protocol: W untimed, then EXACTLY K steps between barrier + synchronize, max over ranks; the kernel time
        from HIP events on the launch stream."""
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        barrier()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        t0 = time.perf_counter()
        r = None
        for k in range(steps):
            ev[k][0].record()
            r = fn()
            ev[k][1].record()
        torch.cuda.synchronize()
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        timed_steps.events = ev  # (for callers that record their own events inside fn)
        return elapsed, float(np.mean([a.elapsed_time(b) for a, b in ev])), r

    def solve_stats(iters):
        it = iters & 0x3FFFFFFF
        return (float(it.to(torch.float64).mean().item()),
                torch.bincount(it.flatten().to(torch.int64), minlength=23)[:23].cpu().tolist(), int(((iters >> 30) & 1).ne(0).sum().item()))

    # ---- workload 1 (headline, as in round 1): S clips x T frames per GPU, D distinct clips tiled, initial heading within 1 rad.
    #      Every rank builds the same batch (same seeds): weak scaling with identical work per GPU ----
    S, T, D = args.clips, args.frames, min(args.distinct, args.clips)
    pe, qe, names, _, _ = synth.synth_clips(cm, D // 2, T, seed=1000, hard=False, dtype=np.float32)
    ph, qh, _, _, _ = synth.synth_clips(cm, D - D // 2, T, seed=2000, hard=True, dtype=np.float32)
    base_pos, base_quat = np.concatenate([pe, ph]), np.concatenate([qe, qh])
    reps = (S + D - 1) // D
    pos = torch.from_numpy(base_pos).to(dev).repeat(reps, 1, 1)[: S * T].contiguous()
    quat = torch.from_numpy(base_quat).to(dev).repeat(reps, 1, 1)[: S * T].contiguous()
    offs = np.arange(S + 1, dtype=np.int64) * T
    items = make_items(offs)
    sc = cm.slot_columns(names)
    out = torch.empty((S * T, eng.nq), dtype=torch.float64, device=dev)
    n_frames = S * T

    # One step = the engine's default call: a 32-frame probe of every clip + the device-side order by predicted cost
    # (gmr_ik_plan_order: equal lengths carry no cost information) and the ordered launch (gmr_ik_solve_ordered), all inside the
    # timed region.  The two kernels are also timed on their own (HIP events) for the roofline record.
    planned = eng._order_pays(items)
    mids = []

    def headline_step():  # == eng.ik_solve(pos, quat, sc, items, out=out) with launch_order="auto", an event between its two halves
        order = eng.plan_order(pos, quat, sc, items, probe_frames=eng._probe_frames(items)) if planned else None
        mid = torch.cuda.Event(enable_timing=True)
        mid.record()
        mids.append(mid)
        return eng.ik_solve(pos, quat, sc, items, out=out, launch_order=order)

    elapsed, step_ms, (_, iters, _) = timed_steps(headline_step, args.steps, args.warmup)
    mids = mids[-args.steps:]
    probe_ms = float(np.mean([a.elapsed_time(m_) for (a, _), m_ in zip(timed_steps.events, mids)]))
    kern_ms = float(np.mean([m_.elapsed_time(b) for (_, b), m_ in zip(timed_steps.events, mids)]))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    array_order_ms = None
    if planned and not args.hot_only:  # what the order is worth: the same launch in array order
        e0.record(); eng.ik_solve(pos, quat, sc, items, out=out, launch_order=None); e1.record(); torch.cuda.synchronize()
        array_order_ms = e0.elapsed_time(e1)
    mean_solves, solves_hist, qp_capped = solve_stats(iters)
    if torch.isnan(out).any().item():
        raise SystemExit("non-finite qpos in the benchmark output")

    # ---- workload 2 (un-shaped): every clip distinct, initial heading anywhere in (-pi, pi], lengths U(T/3, 5T/3), half the clips
    #      noisy / over-reaching; generated on the GPU (gmr_amd.synth.synth_clips_torch) ----
    un = None
    if not args.no_unshaped and not args.hot_only:
        rng = np.random.default_rng(7)
        lens = rng.integers(max(1, T // 3), max(2, 5 * T // 3) + 1, size=S)
        hard_mask = np.arange(S) % 2 == 1
        t_gen = time.perf_counter()
        upos, uquat, unames, uoffs = synth.synth_clips_torch(cm, lens, seed=4242, device=dev, hard=hard_mask, yaw0=np.pi)
        torch.cuda.synchronize()
        t_gen = time.perf_counter() - t_gen
        uitems = make_items(uoffs)
        usc = cm.slot_columns(unames)
        uout = torch.empty((int(uoffs[-1]), eng.nq), dtype=torch.float64, device=dev)
        u_el, u_kern_ms, (_, uiters, _) = timed_steps(lambda: eng.ik_solve(upos, uquat, usc, uitems, out=uout), max(1, args.steps - 1), 1)
        u_solves, u_hist, u_capped = solve_stats(uiters)
        if torch.isnan(uout).any().item():
            raise SystemExit("non-finite qpos in the un-shaped benchmark output")
        un = {"frames": int(uoffs[-1]), "elapsed": u_el, "steps": max(1, args.steps - 1), "kern_ms": u_kern_ms, "solves": u_solves, "hist": u_hist,
              "capped": u_capped, "gen_s": t_gen, "lens": lens, "planned": eng._order_pays(uitems)}

    result = None
    if rank == 0:
        total_frames = n_frames * world * args.steps
        value = total_frames / elapsed
        in_sz = pos.element_size()
        bpf, fps_kernel = bytes_per_frame(cm, in_sz), n_frames / (kern_ms * 1e-3)
        ach_gbs = bpf * fps_kernel / 1e9
        fsolve = flops_per_solve(cm)
        ach_tf = fps_kernel * mean_solves * fsolve / 1e12
        measured = (MEASURED_TRAFFIC["robot"], MEASURED_TRAFFIC["src"], MEASURED_TRAFFIC["in_itemsize"]) == (ROBOT, SRC, in_sz)
        traffic = (MEASURED_TRAFFIC["bytes_per_frame"] if measured else bpf) * n_frames
        traffic_source = (MEASURED_TRAFFIC["source"] + " (PMC, this configuration) scaled to this launch") if measured else "algorithmic bytes (no PMC pass for this configuration)"
        if live_traffic[1] != "not requested":
            live, detail = live_traffic[:2]
            if live is not None:
                traffic, traffic_source = live * n_frames, {"measured": "this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes over an 8192 x 600 launch, "
                                                                        "(2 x FETCH + WRITE) per frame scaled to this launch", "bytes_per_frame": live, **detail}
            else:
                traffic_source += f"; live PMC pass not taken: {detail}"
        result = {
            "metric": "retargeted frames/sec (whole node), Unitree G1 29-DoF SMPLX; max qpos err vs CPU",
            "value": value, "n1_equivalent_value": value / world, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": f"AMASS-shaped smplx->unitree_g1 (29 DoF, nq 36): {S} clips x {T} frames @30fps per GPU, equal lengths, "
                            f"{D} distinct clips tiled (half exactly reachable, half 2cm/5deg noise + 1.1x arm reach), initial heading within "
                            "1 rad of the robot's; frames sequential per clip (exact reference warm-start semantics), clips independent.  "
                            "value_unshaped: same size, every clip distinct, heading in (-pi, pi], lengths U(T/3, 5T/3), same easy/hard mix",
                "clips_per_gpu": S, "frames_per_clip": T, "frames_per_step": n_frames * world, "parallelism": f"clip-sharded x{world}",
                "distinct_clips": D, "initial_heading_rad": 1.0,
            },
            "roofline": {"bound": "fp64-vector", "achieved": ach_tf, "peak": FP64_VECTOR_PEAK_TF, "unit": "TFLOP/s", "frac": ach_tf / FP64_VECTOR_PEAK_TF,
                         "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": kernel_name, "kernel_ms": kern_ms, "flop_per_solve": fsolve, "mean_solves_per_frame": mean_solves,
                         "solves_per_frame_histogram": solves_hist,
                         "step_ms": step_ms, "frac_of_step": n_frames / (step_ms * 1e-3) * mean_solves * fsolve / 1e12 / FP64_VECTOR_PEAK_TF,
                         "launch_order": {"planned": bool(planned), "probe_frames": int(eng._probe_frames(items)),
                                          "probe_kernel": kernel_name.replace("ik_kernel", "ik_probe_kernel") + " + gmr::plan_order_kernel",
                                          "probe_ms": probe_ms, "array_order_kernel_ms": array_order_ms,
                                          "note": "equal-length clips differ in cost (solves per frame); every step probes the first frames of "
                                                  "every clip and launches most-expensive-first (inside the timed region, redundant work "
                                                  "not counted as useful flops)"},
                         "note": "achieved = SURVEY 8(d) flop/solve x measured solves/frame x frames / ik_kernel time (HIP events); "
                                 "frac_of_step charges the probe and the sort as well"},
            "hbm": {"bound": "hbm", "non_binding": True, "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_gbs / HBM_PEAK_GBS,
                    "bytes_per_frame": bpf},
            "qp_iteration_caps_hit": qp_capped,
        }
        if un is not None:
            result["value_unshaped"] = un["frames"] * world * un["steps"] / un["elapsed"]
            result["unshaped"] = {
                "frames_per_gpu": un["frames"], "clips_per_gpu": S, "steps": un["steps"], "ms_per_step": 1e3 * un["elapsed"] / un["steps"], "kernel_ms": un["kern_ms"],
                "mean_solves_per_frame": un["solves"], "solves_per_frame_histogram": un["hist"], "qp_iteration_caps_hit": un["capped"],
                "clip_length_min_max": [int(un["lens"].min()), int(un["lens"].max())], "generation_s": un["gen_s"],
                "valu_frac": un["frames"] / (un["kern_ms"] * 1e-3) * un["solves"] * fsolve / 1e12 / FP64_VECTOR_PEAK_TF,
                "launch_order": {"planned": bool(un["planned"]), "note": "the default call: a 32-frame probe of every clip, cost = its solves x the clip's length, most expensive first "
                                                                         "-- probe and sort inside the timed region (729 ms in plain length order, tools/experiments/unshaped_probe_order.py)"},
            }

    # ------------------------------------------------------------------ every N: the CPU path in the same run + parity of rank 0's output
    #      (rank 0 solves a bounded sample of its own clips with the oracle; the other ranks wait at the barrier that follows)
    def cpu_leg():
        from oracle.oracle import Oracle  # checker / comparator only
        orc = Oracle(cm.blob)
        cores = host_cores()
        nc = min(args.cpu_clips if args.cpu_clips > 0 else min(512, max(32, 4 * cores)), S)
        cp, cq = pos[: nc * T].cpu().numpy(), quat[: nc * T].cpu().numpy()
        citems = make_items(offs[: nc + 1])
        one = 4
        t1 = time.perf_counter()
        orc.ik_solve(cp[: one * T], cq[: one * T], sc, make_items(offs[: one + 1]), n_threads=1)
        t_one = time.perf_counter() - t1
        t1 = time.perf_counter()
        q_ref, it_ref, _ = orc.ik_solve(cp, cq, sc, citems, n_threads=cores)
        t_all = time.perf_counter() - t1
        result["cpu_baseline"] = {
            "value": nc * T / t_all, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{nc} of rank 0's headline clips x {T} frames, clip-parallel OpenMP on {cores} threads, float64 C oracle" + (f" (timed while the other {world - 1} ranks wait at a barrier)" if world > 1 else ""),
            "single_core_value": one * T / t_one, "single_core_sample": f"{one} clip(s) x {T} frames",
            "reference_published": "35-70 frames/s single Python process (README.md:617-620, other hardware, config unstated)",
        }

        def parity(q_gpu, it_gpu, q_ref, it_ref):
            d = np.abs(q_gpu - q_ref)
            # root rotation error as the geodesic angle between the two unit quaternions
            dots = np.abs(np.sum(q_gpu[:, 3:7] * q_ref[:, 3:7], axis=1)) / (
                np.linalg.norm(q_gpu[:, 3:7], axis=1) * np.linalg.norm(q_ref[:, 3:7], axis=1))
            geo = 2.0 * np.arccos(np.clip(dots, 0.0, 1.0))
            return {"max_abs_qpos_err_vs_cpu": float(d.max()), "max_abs_hinge_err_rad": float(d[:, 7:].max()),
                    "p999_abs_hinge_err_rad": float(np.quantile(d[:, 7:].max(axis=1), 0.999)),
                    "max_root_pos_err_m": float(np.linalg.norm(q_gpu[:, :3] - q_ref[:, :3], axis=1).max()),
                    "max_root_geodesic_err_rad": float(geo.max()), "frames_compared": int(q_ref.shape[0]),
                    "frames_with_different_solve_count": int((it_gpu != it_ref).sum())}
        result["parity"] = {**parity(out[: nc * T].cpu().numpy(), (iters[: nc * T] & 0x3FFFFFFF).cpu().numpy(), q_ref, it_ref), "tolerance_target_rad": 1e-3}
        if un is not None:
            # the un-shaped clips too: the first nc clips in memory order (lengths differ)
            e = int(uoffs[nc])
            t1 = time.perf_counter()
            uq_ref, uit_ref, _ = orc.ik_solve(upos[:e].cpu().numpy(), uquat[:e].cpu().numpy(), usc, make_items(uoffs[: nc + 1]), n_threads=cores)
            t_u = time.perf_counter() - t1
            result["parity"]["unshaped"] = parity(uout[:e].cpu().numpy(), (uiters[:e] & 0x3FFFFFFF).cpu().numpy(), uq_ref, uit_ref)
            result["cpu_baseline"]["unshaped_value"] = e / t_u

    if rank == 0 and not args.no_cpu and not args.hot_only:
        try:
            cpu_leg()
        except Exception as ex:  # (the other ranks are waiting at the barrier below: never leave them there)
            result["cpu_baseline_error"] = repr(ex)
    barrier()
    if un is not None:  # the un-shaped buffers (37 GB of key-points per GPU at the default size) are not needed any more
        del upos, uquat, uout, uiters
        torch.cuda.empty_cache()

    # ------------------------------------------------------------------ N > 1: the exchange steps, on the real outputs
    if world > 1 and not args.hot_only:
        try:
            cdev = dev if on_rccl else torch.device("cpu")
            ones = torch.ones(1, dtype=torch.float64, device=cdev)
            torch.distributed.all_reduce(ones)
            # all-gather of qpos (north_star: "allgather of qpos over xGMI"): the WHOLE output of every rank -- S clips x T frames x
            # 288 B per rank -- into global clip order on every rank (gather_rows: slabs of <= 8 GiB received per rank and collective,
            # one index_copy_ per slab)
            lengths = np.full(S * world, T, dtype=np.int64)
            mine = gdist.my_clips(lengths)
            local_rows = out[: len(mine) * T]
            full = torch.empty((S * world * T, eng.nq), dtype=out.dtype, device=cdev)
            torch.cuda.synchronize()
            barrier()
            t0 = time.perf_counter()
            gdist.gather_rows(local_rows, lengths, out=full)
            torch.cuda.synchronize()
            barrier()
            t_ag = max_over_ranks(time.perf_counter() - t0)
            # every rank's clips are the same synthetic batch, so global clip i (owner i mod world, its local clip i // world) must
            # equal this rank's own clip i // world: check a strided sample of clips bit for bit
            probe = np.unique(np.linspace(0, S * world - 1, 64).astype(np.int64))
            ok_rows = bool(all(torch.equal(full[i * T:(i + 1) * T].to(out.device), out[(i // world) * T:(i // world + 1) * T]) for i in probe))
            del full
            if on_rccl:
                torch.cuda.empty_cache()
            # strong scaling: the SAME S clips split over the ranks (longest-first greedy; equal lengths -> S / world each)
            s_mine = gdist.my_clips([T] * S)
            s_items = make_items(np.arange(len(s_mine) + 1, dtype=np.int64) * T)
            n_s = len(s_mine) * T
            st_el, _, _ = timed_steps(lambda: eng.ik_solve(pos[:n_s], quat[:n_s], sc, s_items, out=out[:n_s]), args.steps, 1)
            # few long clips (BASELINE config 3): a LAFAN1-sized set, chunks of every clip spread over all ranks
            lc = long_clip_set(cm, synth, dev, yaw0=1.0)
            from gmr_amd.schedule import auto_chunk
            lch, lbi = auto_chunk(lc["offs"], world * 8 * torch.cuda.get_device_properties(dev).multi_processor_count)
            lc_el, _, (q_lc, _, lc_info) = timed_steps(lambda: lc["eng"].ik_solve_chunked_sharded(lc["pos"], lc["quat"], lc["sc"], lc["offs"], lch, lbi), 2, 1)
            if rank == 0:
                result["collectives"] = {
                    "backend": torch.distributed.get_backend(), "rccl_ranks": int(ones.item()),
                    "allgather_qpos_ms": 1e3 * t_ag, "allgather_rows_per_rank": int(local_rows.shape[0]), "allgather_bytes_per_rank": int(local_rows.shape[0]) * eng.nq * 8,
                    "allgather_GBps_received_per_rank": int(local_rows.shape[0]) * eng.nq * 8 * (world - 1) / t_ag / 1e9, "allgather_complete": ok_rows,
                    "allgather_rows_total": int(S * world * T),
                    "note": f"gather_rows of the whole output, {S} clips x {T} frames per rank (288 B/frame), into clip order on every rank; outside the timed region of `value`",
                }
                result["strong"] = {"clips_total": S, "frames_per_step": S * T, "value": S * T * args.steps / st_el, "ms_per_step": 1e3 * st_el / args.steps}
                result["long_clips_sharded"] = {"set": "heading_within_1rad", "clips": len(lc["offs"]) - 1, "frames": int(lc["offs"][-1]), "frames_per_s": 2 * int(lc["offs"][-1]) / lc_el,
                                                "chunk": lch, "burn_in": lbi, **{k: lc_info[k] for k in ("chunks", "resolved_frames", "resolved_chunks", "ranks")}}
        except Exception as ex:  # the headline line must survive a failure of the extra legs
            if rank == 0:
                result["multi_gpu_legs_error"] = repr(ex)

    # ------------------------------------------------------------------ N = 1: the other kernels and modes of the path
    if rank == 0 and world == 1 and (not args.hot_only or args.hot_fk):
        # fk_kernel (KinematicsModel.forward_kinematics): HBM-bound; positions only, as the dataset path calls it
        nf = n_frames
        root_pos32 = out[:, 0:3].to(torch.float32).contiguous()
        root_rot32 = out[:, [4, 5, 6, 3]].to(torch.float32).contiguous()
        dof32 = out[:, 7:].to(torch.float32).contiguous()
        bp_out = torch.empty((nf, eng.nbody, 3), dtype=torch.float32, device=dev)
        fk_el, fk_ms, _ = timed_steps(lambda: eng.fk(root_pos32, root_rot32, dof32, want_rot=False, out_pos=bp_out), 3, 1)
        fk_bytes = (7 + eng.nq - 7) * 4 + eng.nbody * 12
        br_out = torch.empty((nf // 2, eng.nbody, 4), dtype=torch.float32, device=dev)
        fk_el2, fk_ms2, _ = timed_steps(lambda: eng.fk(root_pos32[: nf // 2], root_rot32[: nf // 2], dof32[: nf // 2], want_rot=True,
                                                       out_pos=bp_out[: nf // 2], out_rot=br_out), 3, 1)
        fk_bytes2 = fk_bytes + eng.nbody * 16
        result["fk"] = {
            "kernel": "gmr::fk_pos_kernel<1> (positions; gmr::fk_kernel<0> with rotations)", "frames": nf, "kernel_ms": fk_ms, "frames_per_s": nf / (fk_ms * 1e-3),
            "roofline": {"bound": "hbm", "achieved": fk_bytes * nf / (fk_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": fk_bytes * nf / (fk_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "bytes_per_frame": fk_bytes, "traffic": None},
            "with_rotations": {"frames": nf // 2, "kernel_ms": fk_ms2, "bytes_per_frame": fk_bytes2,
                               "frac": fk_bytes2 * (nf // 2) / (fk_ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS},
        }
        if not args.no_cpu and not args.hot_only:
            # the CPU path beside it: the oracle's float32 restatement of KinematicsModel.forward_kinematics (positions only) on this
            # host's cores, frames split over threads (the ctypes call releases the GIL); a bounded sample of the same frames
            try:
                from concurrent.futures import ThreadPoolExecutor
                from oracle.oracle import Oracle  # checker / comparator only
                orc_fk = Oracle(cm.blob)
                cores = host_cores()
                n_cpu = min(nf, 400_000 * cores)
                h_rp, h_rr, h_d = root_pos32[:n_cpu].cpu().numpy(), root_rot32[:n_cpu].cpu().numpy(), dof32[:n_cpu].cpu().numpy()
                n1 = min(n_cpu, 200_000)
                t1 = time.perf_counter()
                orc_fk.fk_kin(h_rp[:n1], h_rr[:n1], h_d[:n1], want_rot=False)
                t_one = time.perf_counter() - t1
                cuts = np.linspace(0, n_cpu, cores + 1).astype(np.int64)
                t1 = time.perf_counter()
                with ThreadPoolExecutor(max_workers=cores) as ex:
                    parts = list(ex.map(lambda k: orc_fk.fk_kin(h_rp[cuts[k]:cuts[k + 1]], h_rr[cuts[k]:cuts[k + 1]], h_d[cuts[k]:cuts[k + 1]], want_rot=False)[0], range(cores)))
                t_all = time.perf_counter() - t1
                chk = float(np.abs(parts[0][:4096] - bp_out[:4096].cpu().numpy()).max())
                result["fk"]["cpu_baseline"] = {"value": n_cpu / t_all, "unit": "frames/s", "cores": cores, "kind": "port",
                                                "sample": f"{n_cpu} of the same frames, positions only, float32 C oracle, {cores} threads", "single_core_value": n1 / t_one,
                                                "max_abs_diff_gpu_vs_cpu_m": chk,
                                                "reference_measured": "reference torch FK (kinematics_model.py:213-246), G1 38 bodies, 8 CPU threads: 2.1e5 frames/s (SURVEY 6, measured in the build container)"}
                del parts, h_rp, h_rr, h_d
            except Exception as ex:
                result["fk"]["cpu_baseline"] = {"error": repr(ex)}
        try:
            result["fk"]["kin_ops"] = kin_ops_leg(eng, dof32[: min(nf, 8_000_000)])
        except Exception as ex:
            result["fk"]["kin_ops"] = {"error": repr(ex)}
        attach_fk_traffic(result["fk"], live_traffic[2])
        del root_pos32, root_rot32, dof32, bp_out, br_out
    if rank == 0 and world == 1 and (not args.hot_only or args.hot_adapters):
        # the two input-adapter kernels (rows f-1, f-2): HBM-bound by construction (1.9 - 8.4 KB per frame)
        if args.hot_only:   # the PMC child passes: one timed dispatch per configuration, smaller arrays
            adapters_leg(dev, *ADAPTER_PMC_SIZES, steps=1)
        else:
            try:
                result["adapters"] = adapters_leg(dev)
                attach_adapter_traffic(result["adapters"], live_traffic[2])
            except Exception as ex:
                result["adapters"] = {"error": repr(ex)}
    if rank == 0 and world == 1 and not args.hot_only:
        from gmr_amd import dataset

        def timed(fn, reps=5):
            ts = []
            r = None
            for _ in range(reps):
                r = None  # (release the previous result first: pinned result blocks are then reused instead of page-locked anew)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                r = fn()
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t1)
            return float(np.median(ts)), r

        # the dataset path of scripts/smplx_to_robot_dataset.py:79-146 on resident key-points: IK + FK (local) + FK min-height +
        # post-processing, to host arrays ready for pickling (pickle / disk excluded)
        nd = min(S, 2048)
        t_dev, _ = timed(lambda: _dataset_device(gmr, dataset, pos[: nd * T], quat[: nd * T], names, offs[: nd + 1]), reps=3)
        t_all, _ = timed(lambda: dataset.retarget_clips(gmr, pos[: nd * T], quat[: nd * T], names, offs[: nd + 1]), reps=3)
        # ... and all the way from host key-points to pickle files (the scripts' whole process_file body but the SMPL-X forward pass):
        # numpy in -> retarget_clips -> one pickle per clip on tmpfs; 256 clips
        import shutil
        import tempfile
        # batches of one clip per wavefront slot (2048): batch k + 1 is solved while the writer threads put batch k on tmpfs.
        # Host key-points = the 14 columns the IK config consumes (what the adapters' `columns=` deliver), float32
        per = min(S, 2048)
        nbat = max(1, min(S, 8192) // per)
        npk = per * nbat
        sc_t = torch.from_numpy(np.asarray(sc, dtype=np.int64)).to(dev)
        hp_d, hq_d = pos[: npk * T].index_select(1, sc_t).cpu().numpy(), quat[: npk * T].index_select(1, sc_t).cpu().numpy()
        names_d = [names[i] for i in sc]
        tmpd = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        workers = max(2, min(16, host_cores()))

        def to_pickles():
            with dataset.MotionWriter(workers=workers, override=True) as w:
                for b in range(nbat):
                    motions = dataset.retarget_clips(gmr, hp_d[b * per * T:(b + 1) * per * T], hq_d[b * per * T:(b + 1) * per * T], names_d, offs[: per + 1])
                    w.submit(motions, [os.path.join(tmpd, f"{b * per + i}.pkl") for i in range(per)])
        t_pk, _ = timed(to_pickles, reps=3)

        pers = min(per, 512)

        def to_pickles_serial():  # round 2's form on 512 clips: one batch, then one pickle.dump after the other
            motions = dataset.retarget_clips(gmr, hp_d[: pers * T], hq_d[: pers * T], names_d, offs[: pers + 1])
            for i, mo in enumerate(motions):
                with open(os.path.join(tmpd, f"s{i}.pkl"), "wb") as f:
                    pickle.dump(mo, f)
        import pickle
        t_pks, _ = timed(to_pickles_serial, reps=2)
        same_bytes = all(open(os.path.join(tmpd, f"s{i}.pkl"), "rb").read() == open(os.path.join(tmpd, f"{i}.pkl"), "rb").read() for i in range(0, pers, max(1, pers // 8)))
        shutil.rmtree(tmpd, ignore_errors=True)
        result["dataset_path"] = {"clips": nd, "frames": nd * T, "frames_per_s_device": nd * T / t_dev, "frames_per_s_to_host": nd * T / t_all,
                                  "frames_per_s_host_keypoints_to_pickles": npk * T / t_pk, "pickle_clips": npk, "pickle_batches": nbat, "writer_threads": workers,
                                  "frames_per_s_host_keypoints_to_pickles_serial": pers * T / t_pks, "pickles_byte_identical_to_serial": bool(same_bytes),
                                  "pickle_bytes_per_frame": 24 + 32 + (eng.nq - 7) * 8 + eng.nbody * 12,
                                  "includes": "ik_kernel + fk_kernel (local_body_pos) + fk min-height + root adjustments; `to_host` adds the D2H of "
                                              "root_pos / root_rot / dof_pos (f64) + local_body_pos (f32, 456 B/frame) into pinned host arrays and the per-clip dicts"}
        # BASELINE config 2 taken literally: ONE 3000-frame clip on one GPU, parallel-in-time chunks with verified
        # boundaries (Engine.ik_solve_chunked) vs the same clip solved sequentially by one wavefront; an easy and a hard clip
        sclip = {}
        for label, c in (("easy", 0), ("hard", D // 2)):
            one_p, one_q, one_offs = pos[c * T:(c + 1) * T].contiguous(), quat[c * T:(c + 1) * T].contiguous(), offs[:2]
            t_seq, (q_seq, it_seq, _) = timed(lambda: eng.ik_solve(one_p, one_q, sc, make_items(one_offs)), reps=3)
            t_chk, (q_chk, it_chk, info) = timed(lambda: eng.ik_solve_chunked(one_p, one_q, sc, one_offs, chunk=16, burn_in=24))
            sclip[label] = {"sequential_frames_per_s": T / t_seq, "verified_chunked_frames_per_s": T / t_chk, "resolved_frames": info["resolved_frames"],
                            "max_abs_diff_vs_sequential": float((q_chk - q_seq).abs().max().item()),
                            "frames_with_different_solve_count": int((it_chk != it_seq).sum().item())}
        result["single_clip"] = {"frames": T, "chunk": 16, "burn_in": 24, "chunk_start": "root-task target (GMR_INIT_ROOT_TARGET)", **sclip,
                                 "includes": "host scheduling + both launches (chunks, verification walk)"}
        # BASELINE config 3 on one GPU: a LAFAN1-sized set (77 clips of 2000-9000 frames, bvh_to_g1.json), once with every clip's
        # initial heading within 1 rad of the robot's (round 1's set) and once with any heading: there the reference algorithm
        # itself spends long stretches in history-dependent IK basins (its slow far-heading start-up), which no speculative
        # chunk start can reproduce -- the verification walk re-solves those stretches sequentially
        from gmr_amd.schedule import auto_chunk
        result["long_clips"] = {"config": "bvh_to_g1, LAFAN1-sized: 77 clips of 2000-9000 frames, half noisy / over-reaching"}
        for label, yaw0 in (("heading_within_1rad", 1.0), ("any_heading", float(np.pi))):
            lc = long_clip_set(cm, synth, dev, yaw0=yaw0)
            nfr = int(lc["offs"][-1])
            ch, bi = auto_chunk(lc["offs"], 8 * torch.cuda.get_device_properties(dev).multi_processor_count)   # (round 2 used 64 / 32)
            result["long_clips"].update({"chunk": ch, "burn_in": bi, "chunk_choice": "schedule.auto_chunk"})
            t_seq, (q_seq, it_seq, _) = timed(lambda: lc["eng"].ik_solve(lc["pos"], lc["quat"], lc["sc"], make_items(lc["offs"])), reps=2)
            t_chk, (q_chk, it_chk, info) = timed(lambda: lc["eng"].ik_solve_chunked(lc["pos"], lc["quat"], lc["sc"], lc["offs"], chunk=ch, burn_in=bi), reps=3)
            result["long_clips"][label] = {"clips": len(lc["offs"]) - 1, "frames": nfr, "sequential_frames_per_s": nfr / t_seq,
                                           "verified_chunked_frames_per_s": nfr / t_chk, "resolved_frames": info["resolved_frames"],
                                           "max_abs_diff_vs_sequential": float((q_chk - q_seq).abs().max().item()),
                                           "frames_with_different_solve_count": int((it_chk != it_seq).sum().item())}
            if label == "any_heading":  # the opt-in departure from the reference: clips start on their root target (retarget_batch(clip_start=...))
                from gmr_amd._native import INIT_ROOT_TARGET
                t_rt, (q_rt, _, info_rt) = timed(lambda: lc["eng"].ik_solve_chunked(lc["pos"], lc["quat"], lc["sc"], lc["offs"], chunk=ch, burn_in=bi,
                                                                                     clip_init=INIT_ROOT_TARGET), reps=3)
                result["long_clips"]["any_heading_root_target_start"] = {
                    "verified_chunked_frames_per_s": nfr / t_rt, "resolved_frames": info_rt["resolved_frames"],
                    "note": "NOT the reference's semantics: every clip starts with the base on its first root target instead of qpos0"}
            del lc
        # BASELINE config 3 from FILES (scripts/bvh_to_robot_dataset.py:59-104 end to end): a folder of 24 BVH files x 4000 frames on
        # tmpfs -> qpos; MOTION blocks parsed on the device, batches read ahead, verified chunks (tools/config3_files_bench.py)
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import config3_files_bench
            ff = config3_files_bench.run(24, 4000, max(2, min(16, host_cores())), dev.index)
            result["long_clips"]["from_files"] = {"files": ff["files"], "frames": ff["frames"], "text_MB": ff["keypoint_text_MB"], **ff["from_files"],
                                                  "loader_only": {"text_MB": ff["text_MB"], **ff["loader"]}}
        except Exception as ex:
            result["long_clips"]["from_files"] = {"error": repr(ex)}
        # the SMPL-X side of the file path (scripts/smplx_to_robot_dataset.py:63-146 behind the body model): a folder of joint-array files
        # -> adapter -> IK with one height per file -> FK / post-processing -> pickles (tools/smplx_files_bench.py)
        try:
            import smplx_files_bench
            result["dataset_path"]["from_joint_files"] = smplx_files_bench.run(1024, 750, max(2, min(16, host_cores())), dev.index, batch_files=512)
        except Exception as ex:
            if isinstance(result.get("dataset_path"), dict):
                result["dataset_path"]["from_joint_files"] = {"error": repr(ex)}
        # BASELINE config 4: five robots' batches (5 x 64 clips x 1000 frames) as ONE launch (gmr_group_*) and as five launches on
        # five streams (round 1's form)
        try:
            result["heterogeneous"] = heterogeneous_leg(dev, synth, make_items, timed)
        except Exception as ex:
            result["heterogeneous"] = {"error": repr(ex)}
        # live single-sequence mode (gmr_session_*): host frame in -> host qpos out, one launch per frame
        ses = eng.session(sc, int(pos.shape[1]), dtype=np.float32)
        hp, hq = pos[:256].cpu().numpy(), quat[:256].cpu().numpy()
        lat = []
        for i in range(256):
            t1 = time.perf_counter()
            ses.step(hp[i], hq[i])
            lat.append(time.perf_counter() - t1)
        lat = np.array(lat[16:]) * 1e6
        # the same frames through ONE resident wavefront and a pinned mailbox (gmr_session_set_persistent; opt-in)
        ses.reset()
        ses.set_persistent(200)
        plat = []
        for i in range(256):
            t1 = time.perf_counter()
            ses.step(hp[i], hq[i])
            plat.append(time.perf_counter() - t1)
        ses.close()
        plat = np.array(plat[16:]) * 1e6
        result["live_session"] = {"frames": int(lat.size), "median_latency_us": float(np.median(lat)), "p99_latency_us": float(np.quantile(lat, 0.99)),
                                  "frames_per_s": float(1e6 / lat.mean()), "includes": "host staging + launch + kernel + sync, one wavefront",
                                  "persistent": {"median_latency_us": float(np.median(plat)), "p99_latency_us": float(np.quantile(plat, 0.99)),
                                                 "frames_per_s": float(1e6 / plat.mean()),
                                                 "includes": "host staging + mailbox post + resident wavefront's solve + acknowledgement (no launch, no stream sync)"}}
        # ... and through the class the reference's scripts call: `GeneralMotionRetargeting.retarget(frame_dict)` (motion_retarget.py:139-185),
        # a dict of 55 (pos, quat) pairs in, a fresh float64[nq] out -- what a live loop or an unmodified dataset script pays per frame
        try:
            from gmr_amd import GeneralMotionRetargeting as _GMR
            gcls = _GMR(src_human=SRC, tgt_robot=ROBOT)
            frames_d = [{n: (hp[i, c].astype(np.float64), hq[i, c].astype(np.float64)) for c, n in enumerate(names)} for i in range(256)]
            clat = []
            for fd in frames_d:
                t1 = time.perf_counter()
                gcls.retarget(fd)
                clat.append(time.perf_counter() - t1)
            clat = np.array(clat[16:]) * 1e6
            result["live_session"]["class_api"] = {"median_latency_us": float(np.median(clat)), "p99_latency_us": float(np.quantile(clat, 0.99)), "frames_per_s": float(1e6 / clat.mean()),
                                                   "includes": "GeneralMotionRetargeting.retarget(dict): dict -> arrays, session step, qpos copy; the reference's own figure is 35-70 frames/s"}
            del gcls, frames_d
        except Exception as ex:
            result["live_session"]["class_api"] = {"error": repr(ex)}
        # the same path fed from / returned to HOST arrays (what retarget_batch does for numpy callers): PCIe inclusive,
        # pinned double-buffered staging, copies overlapped with the kernel (Engine.ik_solve_host); never part of `value`
        nh = S * T
        hp_all, hq_all = pos[:nh].cpu().numpy(), quat[:nh].cpu().numpy()
        h_offs = offs[: nh // T + 1]
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        q_host, _ = eng.ik_solve_host(hp_all, hq_all, sc, h_offs, want_iters=False)   # first call: page-locks its result array
        t_first = time.perf_counter() - t1
        t_host, _ = timed(lambda: eng.ik_solve_host(hp_all, hq_all, sc, h_offs, want_iters=False, out=q_host), reps=3)
        same = bool(np.array_equal(q_host, out[:nh].cpu().numpy()))

        def serial():
            q_h, _, _ = eng.ik_solve(torch.from_numpy(hp_all[: nh // 4]).to(dev), torch.from_numpy(hq_all[: nh // 4]).to(dev), sc, make_items(h_offs[: nh // 4 // T + 1]), want_iters=False)
            return q_h.cpu().numpy()
        t_ser, _ = timed(serial, reps=2)
        result["host_fed"] = {"frames": nh, "frames_per_s": nh / t_host, "first_call_frames_per_s": nh / t_first, "bitwise_equal_to_resident": same,
                              "serial_pageable_frames_per_s": (nh // 4) / t_ser,
                              "includes": "pageable host key-points read in place by the copy engine (H2D 392 B/frame), a first batch of one clip per wavefront slot "
                                          "(the only exposed copy) and the rest in one cost-ordered batch on the other stream; the kernel writes qpos (288 B/frame) "
                                          "straight into the pinned host result (reused across calls; first_call includes page-locking it): no copy-out; "
                                          "serial_pageable = round 1's "
                                          "copy-in / solve / copy-out into a fresh pageable array, no overlap (a quarter of the frames)"}
        del hp_all, hq_all, q_host
    if rank == 0:
        print(json.dumps(result), flush=True)
    barrier()
    if world > 1:
        torch.distributed.destroy_process_group()


def kin_ops_leg(eng, dof32, steps=3):
    """The other KinematicsModel operators (kinematics_model.py:172-211) against the HBM roofline: dof_to_rot, rot_to_dof and
    convert_local_rot_to_global over the bench's own joint angles (G1: 29 hinges, 38 bodies).  Algorithmic bytes per frame: every
    input and output element once."""
    import numpy as np
    import torch
    T, nb, nd = int(dof32.shape[0]), eng.nbody, eng.nq - 7

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        for a, b in ev:
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        return float(np.mean([a.elapsed_time(b) for a, b in ev]))

    def rec(kernel, ms, bpf):
        gbs = bpf * T / (ms * 1e-3) / 1e9
        return {"kernel": kernel, "frames": T, "kernel_ms": ms, "frames_per_s": T / (ms * 1e-3), "bytes_per_frame": bpf,
                "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None}}
    jr = torch.empty((T, nb - 1, 4), dtype=torch.float32, device=dof32.device)
    out = {"dof_to_rot": rec("gmr::dof_to_rot_kernel", timed(lambda: eng.dof_to_rot(dof32, out=jr)), 4 * nd + 16 * (nb - 1))}
    back = torch.empty((T, nd), dtype=torch.float32, device=dof32.device)
    out["rot_to_dof"] = rec("gmr::rot_to_dof_kernel", timed(lambda: eng.rot_to_dof(jr, out=back)), 16 * (nb - 1) + 4 * nd)
    out["rot_to_dof"]["round_trip_max_abs_diff_rad"] = float((back - dof32).abs().max().item())  # the bench's angles are inside the limits
    del back
    loc = torch.cat([torch.tensor([0.0, 0.0, 0.0, 1.0], device=dof32.device).expand(T, 1, 4), jr], 1).contiguous()
    del jr
    glob = torch.empty_like(loc)
    out["local_rot_to_global"] = rec("gmr::local_to_global_kernel<8>", timed(lambda: eng.local_rot_to_global(loc, out=glob)), 32 * nb)
    return out


def adapters_leg(dev, bvh_frames=4_000_000, smplx_frames_out=1_000_000, steps=3, traffic=None):
    """The two input-adapter kernels (rows f-1, f-2) against the HBM roofline: gmr::bvh_fk_kernel<1> on LAFAN1-shaped motion rows
    (22 joints, 3-channel rows, all 24 columns and the 14 bvh_to_g1.json reads) and gmr::smplx_keypoints_kernel on AMASS-shaped
    SMPL-X arrays (55 of 127 joints, 120 -> 30 fps and 1:1; all 55 columns and the 14 smplx_to_g1.json reads).  Algorithmic bytes
    per output frame = the input values a frame needs + the output values it produces, each once."""
    import ctypes as C
    import numpy as np
    import torch
    from gmr_amd import _native, synth
    from gmr_amd.smplx_adapter import SMPLX_JOINT_NAMES, SMPLX_PARENTS
    lib = _native.load()
    vp = C.c_void_p
    st = vp(torch.cuda.current_stream(dev).cuda_stream)

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        for a, b in ev:
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        return float(np.mean([a.elapsed_time(b) for a, b in ev]))

    def record(kernel, frames, ms, bytes_in, bytes_out, note):
        bpf = bytes_in + bytes_out
        gbs = bpf * frames / (ms * 1e-3) / 1e9
        return {"kernel": kernel, "frames": frames, "kernel_ms": ms, "frames_per_s": frames / (ms * 1e-3), "bytes_per_frame": bpf, "bytes_in": bytes_in, "bytes_out": bytes_out,
                "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None}, "workload": note}

    out = {}
    # ---- f-1: BVH rows -> global poses
    rows, parents, offsets, order = synth.lafan_rows_torch(bvh_frames, dev)
    J = len(parents)
    names = [n for n, _, _ in synth.LAFAN1_BONES] + ["LeftFootMod", "RightFootMod"]
    ep = np.array([names.index("LeftFoot"), names.index("RightFoot")], np.int32)
    er = np.array([names.index("LeftToe"), names.index("RightToe")], np.int32)
    od = np.asarray(order, np.int32)
    d_off = torch.from_numpy(offsets).to(dev)
    ik_cols = ["Hips", "Spine2", "LeftUpLeg", "RightUpLeg", "LeftLeg", "RightLeg", "LeftFootMod", "RightFootMod", "LeftArm", "RightArm", "LeftForeArm", "RightForeArm", "LeftHand", "RightHand"]
    for label, cols in (("all_columns", None), ("ik_columns", np.array([names.index(c) for c in ik_cols], np.int32))):
        B = J + 2 if cols is None else len(cols)
        pos = torch.empty((bvh_frames, B, 3), dtype=torch.float64, device=dev)
        quat = torch.empty((bvh_frames, B, 4), dtype=torch.float64, device=dev)

        def run():
            rc = lib.gmr_bvh_fk_rows(parents.ctypes.data_as(vp), J, od.ctypes.data_as(vp), ep.ctypes.data_as(vp), er.ctypes.data_as(vp), 2, 3, vp(d_off.data_ptr()),
                                     vp(rows.data_ptr()), int(rows.shape[1]), bvh_frames, 0.01, cols.ctypes.data_as(vp) if cols is not None else None, B,
                                     vp(pos.data_ptr()), vp(quat.data_ptr()), st)
            assert rc == 0, rc
        ms = timed(run)
        out.setdefault("bvh", {})[label] = record("gmr::bvh_fk_kernel<1, false>", bvh_frames, ms, (3 + 3 * J) * 8, B * 56,
                                                  f"LAFAN1-shaped: 22 joints, 3-channel rows in degrees, {B} output columns x (3 + 4) float64")
        del pos, quat
    del rows
    # ---- f-2: SMPL-X arrays -> key-points
    par = np.asarray(SMPLX_PARENTS, np.int32)
    Jx, S = len(par), 127
    live_cols = ["pelvis", "left_hip", "right_hip", "left_knee", "right_knee", "spine3", "left_foot", "right_foot", "left_shoulder", "right_shoulder", "left_elbow", "right_elbow", "left_wrist", "right_wrist"]
    for mode, skip in (("resample_120_to_30", 4), ("one_to_one", 1)):
        T_out = smplx_frames_out
        T = T_out * skip
        go, fp, jt = synth.smplx_arrays_torch(T, dev, Jx, S)
        go32 = fp32 = jt32 = None
        for label, cols in (("all_columns", None), ("ik_columns", np.array([SMPLX_JOINT_NAMES.index(c) for c in live_cols], np.int32))):
            B = Jx if cols is None else len(cols)
            if cols is None:
                n_rot, n_pos = Jx, Jx
            else:  # rotations of the emitted joints and their ancestors, positions of the emitted joints
                live = set()
                for c in cols:
                    a = int(c)
                    while a >= 0 and a not in live:
                        live.add(a)
                        a = int(par[a])
                n_rot, n_pos = len(live), len(cols)
            pos = torch.empty((T_out, B, 3), dtype=torch.float64, device=dev)
            quat = torch.empty((T_out, B, 4), dtype=torch.float64, device=dev)

            def run():
                rc = lib.gmr_smplx_keypoints_cols(par.ctypes.data_as(vp), Jx, S, vp(go.data_ptr()), vp(fp.data_ptr()), vp(jt.data_ptr()), T, T_out, int(skip > 1),
                                                  cols.ctypes.data_as(vp) if cols is not None else None, B, vp(pos.data_ptr()), vp(quat.data_ptr()), st)
                assert rc == 0, rc
            ms = timed(run)
            src_rows = 2 if skip > 1 else 1
            rec = record(
                "gmr::smplx_keypoints_kernel<double>", T_out, ms, src_rows * (n_rot + n_pos) * 24, B * 56,
                f"AMASS-shaped: 55 joints of a 127-joint position array, {'two source frames per output frame (slerp / lerp)' if skip > 1 else 'one source frame per output frame'}, "
                f"{n_rot} rotations + {n_pos} positions read, {B} output columns")
            # the same arrays as float32 (a body model's own dtype; gmr_smplx_keypoints_in promotes on load): half the input bytes
            if go32 is None:
                go32, fp32, jt32 = go.float(), fp.float(), jt.float()

            def run32():
                rc = lib.gmr_smplx_keypoints_in(par.ctypes.data_as(vp), Jx, S, vp(go32.data_ptr()), vp(fp32.data_ptr()), vp(jt32.data_ptr()), 0, T, T_out, int(skip > 1),
                                                cols.ctypes.data_as(vp) if cols is not None else None, B, vp(pos.data_ptr()), vp(quat.data_ptr()), st)
                assert rc == 0, rc
            ms32 = timed(run32)
            b32 = src_rows * (n_rot + n_pos) * 12 + B * 56
            rec["float32_input"] = {"kernel": "gmr::smplx_keypoints_kernel<float>", "kernel_ms": ms32, "frames_per_s": T_out / (ms32 * 1e-3), "bytes_per_frame": b32,
                                    "frac": b32 * T_out / (ms32 * 1e-3) / 1e9 / HBM_PEAK_GBS}
            out.setdefault("smplx", {}).setdefault(mode, {})[label] = rec
            del pos, quat
        del go, fp, jt, go32, fp32, jt32
    torch.cuda.empty_cache()
    return out


def attach_fk_traffic(fk_rec, kb):
    """roofline.traffic of the fk records (fk_pos_kernel, fk_kernel<0>, the kin_ops kernels) from the same PMC child passes: the last
    dispatch of each kernel in the child's fk leg, (2 x FETCH_SIZE + WRITE_SIZE) per frame of THAT dispatch, scaled to the timed launch."""
    info = (kb or {}).get("_fk") or {}
    n, ks = info.get("frames_in_pass"), info.get("kernels", {})
    if not n:
        return

    def put(roof, key, frames_in_dispatch, frames, alg):
        if key not in ks:
            return
        fetch_kb, write_kb = ks[key]
        bpf = (2.0 * fetch_kb + write_kb) * 1024.0 / frames_in_dispatch
        roof["traffic"] = bpf * frames
        roof["traffic_bytes_per_frame"] = bpf
        roof["traffic_over_algorithmic"] = bpf / alg
        roof["traffic_source"] = {"FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb, "frames_in_pass": frames_in_dispatch,
                                  "how": "this run: rocprofv3 --pmc child passes, 2 x FETCH_SIZE + WRITE_SIZE"}
    put(fk_rec["roofline"], "fk_pos_kernel", n, fk_rec["frames"], fk_rec["roofline"]["bytes_per_frame"])
    wr = fk_rec.get("with_rotations")
    if wr is not None:
        put(wr, "fk_kernel<0", n // 2, wr["frames"], wr["bytes_per_frame"])
    for name, key in (("dof_to_rot", "dof_to_rot_kernel"), ("rot_to_dof", "rot_to_dof_kernel"), ("local_rot_to_global", "local_to_global_kernel")):
        r = (fk_rec.get("kin_ops") or {}).get(name)
        if isinstance(r, dict) and "roofline" in r:
            put(r["roofline"], key, min(n, 8_000_000), r["frames"], r["bytes_per_frame"])


def attach_adapter_traffic(rec, kb):
    """roofline.traffic of the adapter records from the PMC child passes: (2 x FETCH_SIZE + WRITE_SIZE) per frame of the pass
    (MI355X_MICROARCH.md "HBM": gfx950 tallies 128-byte read requests at 64 bytes), scaled to the timed launch."""
    if not kb:
        return
    order = [("bvh_fk_kernel", [rec["bvh"]["all_columns"], rec["bvh"]["ik_columns"]], ADAPTER_PMC_SIZES[0]),
             ("smplx_keypoints_kernel<double>", [rec["smplx"][m][c] for m in ("resample_120_to_30", "one_to_one") for c in ("all_columns", "ik_columns")], ADAPTER_PMC_SIZES[1])]
    for key, recs, frames_in_pass in order:
        seq = kb.get(key) or []
        if len(seq) != len(recs):
            continue
        for r, (fetch_kb, write_kb) in zip(recs, seq):
            bpf = (2.0 * fetch_kb + write_kb) * 1024.0 / frames_in_pass
            r["roofline"]["traffic"] = bpf * r["frames"]
            r["roofline"]["traffic_bytes_per_frame"] = bpf
            r["roofline"]["traffic_over_algorithmic"] = bpf / r["bytes_per_frame"]
            r["roofline"]["traffic_source"] = {"FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb, "frames_in_pass": frames_in_pass,
                                               "how": "this run: rocprofv3 --pmc child passes, 2 x FETCH_SIZE + WRITE_SIZE"}


def _dataset_device(gmr, dataset, pos, quat, names, offs):
    """IK + the two FK passes + root adjustments with everything left on the device (dataset.motions_from_qpos minus its D2H)."""
    import numpy as np
    import torch
    qpos = gmr.retarget_batch(pos, quat, names, seq_offsets=offs)
    eng = gmr._engine
    N = int(qpos.shape[0])
    root_pos = qpos[:, 0:3].clone()
    root_rot = qpos[:, [4, 5, 6, 3]].contiguous()
    dof32 = qpos[:, 7:].to(torch.float32)
    zeros = torch.zeros((N, 3), dtype=torch.float32, device=qpos.device)
    ident = torch.zeros((N, 4), dtype=torch.float32, device=qpos.device)
    ident[:, 3] = 1.0
    local_body_pos, _ = eng.fk(zeros, ident, dof32, want_rot=False)
    lowest = eng.fk_min_height(root_pos.to(torch.float32), root_rot.to(torch.float32), dof32, offs).to(torch.float64)
    lens = torch.from_numpy(np.diff(offs)).to(qpos.device)
    root_pos[:, 2] -= torch.repeat_interleave(lowest, lens)
    root_pos[:, :2] -= torch.repeat_interleave(root_pos[torch.from_numpy(offs[:-1]).to(qpos.device), :2], lens, dim=0)
    return root_pos, root_rot, local_body_pos


def heterogeneous_leg(dev, synth, make_items, timed):
    import numpy as np
    import torch
    from gmr_amd import params
    from gmr_amd.engine import EngineGroup
    from gmr_amd.ik_config import load_ik_config
    from gmr_amd.mjcf import load_robot
    from gmr_amd.model import compile_model
    robots = ["unitree_g1", "booster_t1", "stanford_toddy", "fourier_n1", "engineai_pm01"]
    cms = [compile_model(load_robot(params.ROBOT_XML_DICT[r], name=r), load_ik_config(params.IK_CONFIG_DICT[SRC][r])) for r in robots]
    grp = EngineGroup(cms, dev.index)
    offs = np.arange(65, dtype=np.int64) * 1000
    batches = []
    for cm in cms:
        pos, quat, names, _, _ = synth.synth_clips(cm, 8, 1000, seed=41, hard=True, dtype=np.float32)
        batches.append((torch.from_numpy(pos).to(dev).repeat(8, 1, 1), torch.from_numpy(quat).to(dev).repeat(8, 1, 1), cm.slot_columns(names), make_items(offs)))
    nfr = 5 * 64 * 1000
    t_one, outs = timed(lambda: grp.ik_solve(batches), reps=5)
    streams = [torch.cuda.Stream(dev) for _ in robots]

    def five():
        r = []
        for e, b, st in zip(grp.engines, batches, streams):
            st.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(st):
                r.append(e.ik_solve(b[0], b[1], b[2], b[3]))
        for st in streams:
            torch.cuda.current_stream(dev).wait_stream(st)
        return r
    t_five, outs5 = timed(five, reps=5)
    same = all(torch.equal(a[0], b[0]) for a, b in zip(outs, outs5))
    grp.close()
    return {"robots": robots, "frames": nfr, "one_launch_frames_per_s": nfr / t_one, "five_launches_frames_per_s": nfr / t_five,
            "kernel_variant_nv_padded": 36, "bitwise_equal": bool(same)}


def long_clip_set(cm_unused, synth, dev, n_clips=77, seed=3, yaw0=None):
    """BASELINE config 3's shape: a LAFAN1-sized set -- 77 clips of 2000..9000 frames, bvh_to_g1.json, half of the base clips
    noisy / over-reaching (tools/config3_bench.py)."""
    import numpy as np
    import torch
    from gmr_amd import params
    from gmr_amd.engine import Engine
    from gmr_amd.ik_config import load_ik_config
    from gmr_amd.mjcf import load_robot
    from gmr_amd.model import compile_model
    cmb = compile_model(load_robot(params.ROBOT_XML_DICT[ROBOT], name=ROBOT), load_ik_config(params.IK_CONFIG_DICT["bvh"][ROBOT]))
    eng = Engine(cmb, dev.index)
    rng = np.random.default_rng(seed)
    lengths = rng.integers(2000, 9001, size=n_clips)
    hard = rng.integers(2, size=n_clips).astype(bool)
    pos, quat, names, offs = synth.synth_clips_torch(cmb, lengths, seed=33, device=dev, hard=hard, yaw0=np.pi if yaw0 is None else yaw0)
    return {"eng": eng, "pos": pos, "quat": quat, "sc": cmb.slot_columns(names), "offs": offs}


if __name__ == "__main__":
    main()
