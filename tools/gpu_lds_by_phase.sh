#!/bin/bash
# LDS activity / bank conflicts attributed to phases: PMC pass over the phase-duplication variants (tools/build_variant.sh dupN).
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
: > gpurun_out/lds_by_phase.log
for v in dup0 dup1 dup2 dup3 dup4 dup5 dup6 dup7; do
  export GMR_AMD_LIB=$R/gmr_amd/lib/variants/lib$v.so
  (cd /tmp && timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/ldsq_$v -- python3 $R/bench.py --steps 1 --warmup 1 --hot-only --frames 300 --clips 2048 > /dev/null 2>$R/gpurun_out/ldsq_$v.err)
  f=$(find gpurun_out/ldsq_$v -name "*counter_collection.csv" | head -1)
  python3 - "$f" $v <<'PY' | tee -a gpurun_out/lds_by_phase.log
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'ik_kernel' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
print(sys.argv[2], {k: f"{v[-1]:.4g}" for k, v in sorted(acc.items())})
PY
done
