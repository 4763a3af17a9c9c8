#!/bin/bash
# LDS counters of the IK kernel for the library variants named in LDS_VARIANTS (default: the built library).
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
: > gpurun_out/lds_total.log
for v in ${LDS_VARIANTS:-default}; do
  if [ "$v" = default ]; then unset GMR_AMD_LIB; else export GMR_AMD_LIB=$R/gmr_amd/lib/variants/lib$v.so; fi
  rm -rf gpurun_out/ldst_$v
  (cd /tmp && timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/ldst_$v -- python3 $R/bench.py --steps 1 --warmup 1 --hot-only --frames 300 --clips 2048 > /dev/null 2>$R/gpurun_out/ldst_$v.err)
  f=$(find gpurun_out/ldst_$v -name "*counter_collection.csv" | head -1)
  python3 - "$f" $v <<'PY' | tee -a gpurun_out/lds_total.log
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'ik_kernel' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
print(sys.argv[2], {k: f"{v[-1]:.4g}" for k, v in sorted(acc.items())})
PY
done
