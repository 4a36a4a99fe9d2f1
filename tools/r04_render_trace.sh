#!/bin/bash
# GPU box: rocprofv3 kernel trace of tools/render_only.py (B = 32 forward + backward x 5): per-kernel average durations
#   -> gpurun_out/r04_render_kernels_<tag>.txt     usage: tools/r04_render_trace.sh <tag> [script]
TAG=${1:-a}; SCRIPT=${2:-render_only.py}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/rtrace_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -- python3 $GRAFT_REPO_ROOT/tools/$SCRIPT > $OUT/run.log 2>&1
S=$(find $OUT/raw -name "*kernel_stats.csv" | head -1)
python3 - <<PY > $GRAFT_REPO_ROOT/gpurun_out/r04_render_kernels_$TAG.txt
import csv
for r in csv.DictReader(open("$S")):
    n = r["Name"].split("(")[0]
    if "hifihr" in n: print(f'{n[-70:]:72s} calls {r["Calls"]:>4s} avg_us {float(r["AverageNs"])/1e3:8.1f} min_us {float(r["MinNs"])/1e3:8.1f} max_us {float(r["MaxNs"])/1e3:8.1f}')
PY
rm -rf $OUT/raw
cat $GRAFT_REPO_ROOT/gpurun_out/r04_render_kernels_$TAG.txt
