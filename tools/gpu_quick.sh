#!/bin/bash
# quick iteration: parity tests + short bench over library variants and clip counts (+ optional stamps).
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -6 | tee gpurun_out/quick.log
for W in ${QUICK_VARIANTS:-default}; do
  if [ "$W" = default ]; then unset GMR_AMD_LIB; else export GMR_AMD_LIB=$GRAFT_REPO_ROOT/gmr_amd/lib/variants/lib$W.so; fi
  for S in ${QUICK_CLIPS:-2048}; do
    echo "== bench frames=${QUICK_FRAMES:-600} clips=$S (variant $W)" | tee -a gpurun_out/quick.log
    timeout -k 10 300 python bench.py --steps 3 --warmup 1 --frames ${QUICK_FRAMES:-600} --clips $S --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'], d['valu']['mean_solves_per_frame'])" | tee -a gpurun_out/quick.log
  done
done
unset GMR_AMD_LIB
if [ -n "$QUICK_STAMPS" ]; then timeout -k 10 400 python tools/ik_stamps.py 2>&1 | tail -14 | tee -a gpurun_out/quick.log; fi
