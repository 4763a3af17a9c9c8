"""Static instruction mix of one kernel between s_memtime markers (diagnostic -DGMR_IK_STAMPS build).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -DGMR_IK_VARIANTS -DGMR_IK_STAMPS -Iinclude -S --cuda-device-only -o /tmp/stamps.s gmr_amd/csrc/api.hip
    python tools/isa_regions.py /tmp/stamps.s 'ik_kernelILi36ELb1'

Loops show up as backward branches (listed per region with their body size); straight-line counts are per pass.
"""
import re
import sys
from collections import Counter


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().split(":")[0].endswith("E"))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
    labels = {}
    for i in range(start, end):
        m = re.match(r"^(\.LBB\d+_\d+):", lines[i])
        if m:
            labels[m.group(1)] = i
    region, regions = [], []
    for i in range(start, end):
        l = lines[i].strip()
        if "gmr-mark" in l:
            region.append((i, "mark", l))
            regions.append(region)
            region = []
            continue
        if not l or l.startswith(";") or l.startswith(".") or l.endswith(":"):
            continue
        op = l.split()[0]
        region.append((i, op, l))
        if op == "s_memtime":
            regions.append(region)
            region = []
    regions.append(region)
    for n, reg in enumerate(regions):
        c = Counter()
        f64 = 0
        for _, op, l in reg:
            if op.startswith("v_"):
                c["valu"] += 1
                if "f64" in op:
                    f64 += 1
            elif op.startswith("s_waitcnt"):
                c["wait"] += 1
            elif op.startswith("s_cbranch") or op.startswith("s_branch"):
                c["branch"] += 1
            elif op.startswith("s_load") or op.startswith("s_buffer"):
                c["smem"] += 1
            elif op.startswith("s_"):
                c["salu"] += 1
            elif op.startswith("ds_"):
                c["lds"] += 1
            elif op.startswith("scratch"):
                c["scratch"] += 1
            elif op.startswith("global") or op.startswith("buffer") or op.startswith("flat"):
                c["vmem"] += 1
            else:
                c["other"] += 1
        loops = []
        for i, op, l in reg:
            if op.startswith("s_cbranch") or op.startswith("s_branch"):
                tgt = l.split()[-1]
                if tgt in labels and labels[tgt] < i:
                    body = sum(1 for j, _, _ in reg if labels[tgt] <= j <= i)
                    loops.append(f"{tgt}:{body}")
        first = reg[0][0] + 1 if reg else -1
        tag = reg[-1][2] if reg and reg[-1][1] == "mark" else ""
        print(f"region {n:2d} ends[{tag[-12:]:>12s}] @{first:6d} total {len(reg):5d}  valu {c['valu']:5d} (f64 {f64:5d})  salu {c['salu']:4d} smem {c['smem']:3d} lds {c['lds']:4d} "
              f"vmem {c['vmem']:3d} scratch {c['scratch']:3d} wait {c['wait']:4d} br {c['branch']:3d}  loops {' '.join(loops)}")


if __name__ == "__main__":
    main()
