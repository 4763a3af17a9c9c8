#!/bin/bash
# SQ / traffic counter passes over the kin_ops kernels (tools/kin_ops_bench.py, G1, 4 M frames).  -> gpurun_out/pmc_kin_ops.log
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
: > gpurun_out/pmc_kin_ops.log
i=0
while read -r C; do
  [ -z "$C" ] && continue
  i=$((i+1))
  (cd /tmp && timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmckin_$i -- python3 $R/tools/kin_ops_bench.py 4000000 > /dev/null 2>$R/gpurun_out/pmckin_$i.err)
  rc=$?
  f=$(find gpurun_out/pmckin_$i -name "*counter_collection.csv" 2>/dev/null | head -1)
  echo "== pass $i rc=$rc [$C]" | tee -a gpurun_out/pmc_kin_ops.log
  [ -n "$f" ] && python3 - "$f" <<'PY' | tee -a gpurun_out/pmc_kin_ops.log
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    for key in ("dof_to_rot_kernel", "rot_to_dof_kernel", "local_to_global_kernel"):
        if key in r["Kernel_Name"]:
            acc[key][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
for key, d in acc.items():
    for k, v in d.items():
        v.sort()
        print(f"{key:24s} {k:26s} last dispatch {v[-1][1]:.6g}  ({len(v)} dispatches)")
PY
done <<'LIST'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM
FETCH_SIZE
WRITE_SIZE
LIST
