"""SGPR spill traffic of ik_kernel<36, true> by phase (the VALU instructions v_writelane / v_readlane that move spilled scalars).

    python tools/spill_report.py [extra hipcc flags]

Builds the kernel with -DGMR_IK_MARKS (phase boundaries as ISA comments, no other change), splits the ISA at the marks and prints,
per phase, the spill reloads / stores and what produced the most reloaded values.  Phases 1-9 run once per solve, 0 / 10 once per
frame (ik_kernel.hip.h GMR_STAMP ids).
"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = "/tmp/gmr_marks.s"
NAMES = {0: "prep", 1: "fk", 2: "residual", 3: "task_block", 4: "screws", 5: "composites", 6: "F_c_limits", 7: "H_assemble", 8: "box_qp",
         9: "integrate", 10: "output"}


def main():
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-DGMR_IK_VARIANTS", "-DGMR_IK_MARKS", "-DGMR_IK_DEV_ONLY36", f"-I{ROOT}/include", "-S",
           "--cuda-device-only", "-o", OUT, f"{ROOT}/gmr_amd/csrc/api.hip"] + sys.argv[1:]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    lines = open(OUT).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN3gmr9ik_kernelILi36ELb1") and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
    body, region, cur = [], [], "pre"
    # a mark closes the phase it names: instructions are attributed to the NEXT mark in layout order
    pending = []
    for l in lines[start:end]:
        t = l.strip()
        m = re.search(r"gmr-mark (\d+)", t)
        if m:
            for p in pending:
                body.append(p); region.append(int(m.group(1)))
            pending = []
            continue
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        pending.append(t)
    for p in pending:
        body.append(p); region.append(99)
    spillv = collections.Counter(re.match(r"v_writelane_b32 (v\d+),", l).group(1) for l in body if l.startswith("v_writelane_b32"))
    spillv = {v for v, c in spillv.items() if c >= 4}
    rd, wr, valu = collections.Counter(), collections.Counter(), collections.Counter()
    prod, rel = collections.defaultdict(str), collections.Counter()
    for i, l in enumerate(body):
        if l.startswith("v_"):
            valu[region[i]] += 1
        m = re.match(r"v_writelane_b32 (v\d+), (s\d+), (\d+)", l)
        if m and m.group(1) in spillv:
            wr[region[i]] += 1
            sg, n = m.group(2), int(m.group(2)[1:])
            for j in range(i - 1, max(0, i - 120), -1):
                ops = body[j].split(None, 1)
                if len(ops) < 2:
                    continue
                dst = ops[1].split(",")[0].strip()
                mm = re.match(r"s\[(\d+):(\d+)\]", dst)
                if dst == sg or (mm and int(mm.group(1)) <= n <= int(mm.group(2))):
                    prod[(m.group(1), int(m.group(3)))] = f"[{region[j]}] {body[j][:70]}"
                    break
        m = re.match(r"v_readlane_b32 (s\d+), (v\d+), (\d+)", l)
        if m and m.group(2) in spillv:
            rd[region[i]] += 1
            if 1 <= region[i] <= 9:
                rel[(m.group(2), int(m.group(3)))] += 1
    print("spill VGPRs:", sorted(spillv))
    tot_r = tot_w = 0
    for r in sorted(set(region)):
        print(f"  phase {r:2d} {NAMES.get(r, ''):12s} valu {valu[r]:5d}  spill reloads {rd[r]:4d}  stores {wr[r]:4d}")
        if 1 <= r <= 9:
            tot_r += rd[r]; tot_w += wr[r]
    print(f"per-solve phases (1-9): {tot_r} reloads + {tot_w} stores (static)")
    for slot, c in rel.most_common(40):
        print(f"    {slot[0]}[{slot[1]:2d}] x{c}  {prod[slot]}")


if __name__ == "__main__":
    main()
