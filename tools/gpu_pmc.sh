#!/bin/bash
# PMC passes over the IK kernel (600-frame bench) + occupancy experiment.  Outputs under gpurun_out/.
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_IFETCH" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && timeout -k 10 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmcq_$i -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu --frames 300 > /dev/null 2>$R/gpurun_out/pmcq_$i.err)
  f=$(find gpurun_out/pmcq_$i -name "*counter_collection.csv" | head -1)
  echo "== pass $i $f" | tee -a gpurun_out/pmcq.log
  python3 - "$f" <<'PY' | tee -a gpurun_out/pmcq.log
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'ik_kernel' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    print(f"{k:24s} {v[-1]:.4g}  (dispatches {len(v)})")
PY
done
for S in 256 512 1024 2048 4096; do
  echo "== clips $S" | tee -a gpurun_out/pmcq.log
  timeout -k 10 300 python bench.py --steps 2 --warmup 1 --frames 300 --no-cpu --clips $S 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'])" | tee -a gpurun_out/pmcq.log
done
