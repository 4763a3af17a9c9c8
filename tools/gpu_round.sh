#!/bin/bash
# One GPU-box call: parity tests, smoke, bench, kernel-trace profile.  Outputs under gpurun_out/.
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu" | tee gpurun_out/round.log
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee -a gpurun_out/round.log
echo "== smoke" | tee -a gpurun_out/round.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee -a gpurun_out/round.log
echo "== bench" | tee -a gpurun_out/round.log
timeout -k 10 600 python bench.py --steps ${BENCH_STEPS:-3} --warmup 1 ${BENCH_ARGS} 2>gpurun_out/bench.err | tee gpurun_out/bench.json | tee -a gpurun_out/round.log
tail -5 gpurun_out/bench.err | tee -a gpurun_out/round.log
echo "== rocprofv3 kernel trace" | tee -a gpurun_out/round.log
(cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --hot-only ${BENCH_ARGS} > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.json 2>$GRAFT_REPO_ROOT/gpurun_out/prof.err)
echo "rocprof rc=$?" | tee -a gpurun_out/round.log
find gpurun_out/prof -name "*kernel_stats*" | head -3 | tee -a gpurun_out/round.log
for f in $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); do head -8 $f | tee -a gpurun_out/round.log; done
