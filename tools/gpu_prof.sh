#!/bin/bash
# stamps (diagnostic build) + PMC passes for HBM traffic of the IK kernel.  Outputs under gpurun_out/.
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== stamps" | tee gpurun_out/prof.log
timeout -k 10 400 python tools/ik_stamps.py 2>&1 | tail -16 | tee -a gpurun_out/prof.log
for C in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $C" | tee -a gpurun_out/prof.log
  (cd /tmp && timeout -k 10 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$C -- python3 $R/bench.py --steps 1 --warmup 1 --hot-only --frames ${PMC_FRAMES:-600} > $R/gpurun_out/pmc_$C.json 2>$R/gpurun_out/pmc_$C.err)
  echo "rc=$?" | tee -a gpurun_out/prof.log
  f=$(find gpurun_out/pmc_$C -name "*counter_collection.csv" | head -1)
  echo $f | tee -a gpurun_out/prof.log
  [ -n "$f" ] && grep ik_kernel $f | head -4 | cut -c1-600 | tee -a gpurun_out/prof.log
done
