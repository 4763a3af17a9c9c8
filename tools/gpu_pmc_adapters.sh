#!/bin/bash
# SQ counter passes over the adapter kernels (tools/adapter_bench.py, one timed dispatch per configuration).  -> gpurun_out/pmc_adapters_sq.log
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
: > gpurun_out/pmc_adapters_sq.log
i=0
while read -r C; do
  [ -z "$C" ] && continue
  i=$((i+1))
  (cd /tmp && timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmcad_$i -- python3 $R/tools/adapter_bench.py --steps 1 > /dev/null 2>$R/gpurun_out/pmcad_$i.err)
  rc=$?
  f=$(find gpurun_out/pmcad_$i -name "*counter_collection.csv" 2>/dev/null | head -1)
  echo "== pass $i rc=$rc [$C]" | tee -a gpurun_out/pmc_adapters_sq.log
  [ -n "$f" ] && python3 - "$f" <<'PY' | tee -a gpurun_out/pmc_adapters_sq.log
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    for key in ("bvh_fk_kernel", "smplx_keypoints_kernel<double>"):
        if key in r["Kernel_Name"]:
            acc[key][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
for key, d in acc.items():
    for k, v in d.items():
        v.sort()
        print(f"{key:24s} {k:26s} " + " ".join(f"{x[1]:.5g}" for x in v[1::2]))   # the timed dispatch of every configuration, launch order
PY
done <<'LIST'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS
SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_BRANCH
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM
LIST
