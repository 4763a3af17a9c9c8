#!/bin/bash
# One GPU-box call for the adapter kernels: timing, rocprofv3 kernel-trace stats, FETCH_SIZE / WRITE_SIZE PMC passes (each its own run).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
TAG=${TAG:-adapters}
echo "== adapter bench" | tee $R/gpurun_out/${TAG}.log
timeout -k 10 300 python3 $R/tools/adapter_bench.py ${ADAPTER_ARGS} > $R/gpurun_out/${TAG}_bench.json 2>$R/gpurun_out/${TAG}_bench.err || { tail -5 $R/gpurun_out/${TAG}_bench.err; exit 1; }
python3 - <<PY | tee -a $R/gpurun_out/${TAG}.log
import json
d = json.load(open("$R/gpurun_out/${TAG}_bench.json"))
def show(name, r):
    print(f"{name:44s} {r['kernel_ms']:8.3f} ms  {r['frames_per_s']:.3e} frames/s  {r['bytes_per_frame']:5d} B/frame  {r['roofline']['achieved']:7.1f} GB/s  frac {r['roofline']['frac']:.3f}")
for k, r in d["bvh"].items(): show("bvh " + k, r)
for m, v in d["smplx"].items():
    for k, r in v.items(): show("smplx " + m + " " + k, r)
PY
[ -n "$QUICK" ] && exit 0
echo "== kernel trace" | tee -a $R/gpurun_out/${TAG}.log
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG} -- python3 $R/tools/adapter_bench.py ${ADAPTER_ARGS} > /dev/null 2>$R/gpurun_out/${TAG}_prof.err)
for f in $(find $R/gpurun_out/prof_${TAG} -name "*kernel_stats.csv" | head -1); do cp $f $R/gpurun_out/${TAG}_kernel_stats.csv; head -6 $f | tee -a $R/gpurun_out/${TAG}.log; done
for C in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $C" | tee -a $R/gpurun_out/${TAG}.log
  (cd /tmp && timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_$C -- python3 $R/tools/adapter_bench.py --steps 1 ${ADAPTER_ARGS} > /dev/null 2>>$R/gpurun_out/${TAG}_prof.err)
  for f in $(find $R/gpurun_out/pmc_${TAG}_$C -name "*counter_collection.csv" | head -1); do
    cp $f $R/gpurun_out/${TAG}_pmc_$C.csv
    python3 - <<PY | tee -a $R/gpurun_out/${TAG}.log
import csv
rows = [r for r in csv.DictReader(open("$f")) if r["Counter_Name"] == "$C" and ("bvh_fk" in r["Kernel_Name"] or "smplx_keypoints" in r["Kernel_Name"])]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
for r in rows: print(r["Dispatch_Id"], r["Kernel_Name"][:40], r["Counter_Value"])
PY
  done
done
