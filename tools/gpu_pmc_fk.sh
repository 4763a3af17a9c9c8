#!/bin/bash
# PMC passes over the FK kernels (tools/fk_bench.py, 8M frames).  Usage: tools/gpu_pmc_fk.sh [parts]; output gpurun_out/pmcfk_<parts>.log
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P=${1:-1}
export GMR_AMD_FK_PARTS=$P
LOG=gpurun_out/pmcfk_$P.log
: > $LOG
i=0
while read -r C; do
  [ -z "$C" ] && continue
  i=$((i+1))
  (cd /tmp && timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmcfk_${P}_$i -- python3 $R/tools/fk_bench.py unitree_g1 8000000 > /dev/null 2>$R/gpurun_out/pmcfk_${P}_$i.err)
  rc=$?
  f=$(find gpurun_out/pmcfk_${P}_$i -name "*counter_collection.csv" 2>/dev/null | head -1)
  echo "== pass $i rc=$rc [$C]" | tee -a $LOG
  [ -n "$f" ] && python3 - "$f" <<'PY' | tee -a $LOG
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name']
    if 'fk_' in k and 'minkey' not in k:
        name = 'fk_pos_kernel' if 'fk_pos_kernel' in k else ('fk_kernel<1>' if 'ILi1E' in k or '<1>' in k else 'fk_kernel<0>')
        acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
for kn, d in acc.items():
    for k, v in d.items():
        print(f"{kn:16s} {k:28s} last {v[-1]:.6g}  (dispatches {len(v)})")
PY
done <<'LIST'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS
SQ_WAVES SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES GRBM_GUI_ACTIVE SQ_CYCLES
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM
SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT
SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT
FETCH_SIZE
WRITE_SIZE
LIST
