#!/usr/bin/env python3
"""Registers, spills, scratch and LDS of EVERY kernel the library instantiates (from the code object metadata of a device-only
compile of gmr_amd/csrc/api.hip): the check that no variant touches scratch memory.

    python tools/kernel_resources.py [--strict]      # --strict: exit 1 if any kernel has scratch or spilled VGPRs
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = "/tmp/gmr_kernel_resources.s"


def demangle_short(name):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "")
    except OSError:
        return name


def main():
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{ROOT}/include", "-Wno-unused-value", "-S", "--cuda-device-only", "-o", OUT,
                    f"{ROOT}/gmr_amd/csrc/api.hip"], check=True, stderr=subprocess.DEVNULL)
    s = open(OUT).read()
    rows, bad = [], 0
    for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", s, re.S):
        md = m.group(0)
        g = lambda k: int(re.search(k + r":\s+(\d+)", md).group(1))  # noqa: E731
        name = re.search(r"\.name:\s+(\S+)", md).group(1)
        v, a = g(r"\.vgpr_count"), g(r"\.agpr_count")
        alloc = -(-(v + a) // 8) * 8
        occ = min(8, 512 // max(alloc, 8))
        row = (demangle_short(name), v, a, g(r"\.vgpr_spill_count"), g(r"\.sgpr_spill_count"), g(r"\.private_segment_fixed_size"), g(r"\.group_segment_fixed_size"), occ)
        bad += row[3] > 0 or row[5] > 0
        rows.append(row)
    rows.sort()
    print(f"{'kernel':58s} {'vgpr':>5s} {'agpr':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch':>7s} {'lds':>6s} {'waves/SIMD (regs)':>18s}")
    for r in rows:
        print(f"{r[0][:58]:58s} {r[1]:5d} {r[2]:5d} {r[3]:6d} {r[4]:6d} {r[5]:7d} {r[6]:6d} {r[7]:18d}")
    print(f"{len(rows)} kernels, {bad} with scratch or spilled VGPRs")
    if "--strict" in sys.argv and bad:
        sys.exit(1)


if __name__ == "__main__":
    main()
