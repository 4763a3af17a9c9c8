#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in dup0 dup8; do
  export GMR_AMD_LIB=$R/gmr_amd/lib/variants/lib$v.so
  (cd /tmp && timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/gpurun_out/q_$v -- python3 $R/bench.py --steps 1 --warmup 1 --hot-only --frames 300 --clips 2048 > /dev/null 2>$R/gpurun_out/q_$v.err)
  f=$(find $R/gpurun_out/q_$v -name "*counter_collection.csv" | head -1)
  python3 - "$f" $v <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'ik_kernel' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
print(sys.argv[2], {k: f"{v[-1]:.5g}" for k, v in sorted(acc.items())})
PY
done
