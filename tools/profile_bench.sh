#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace + stats of the default bench, summaries under gpurun_out/prof/.
# usage: tools/profile_bench.sh <tag> [bench args]
set -u
TAG=${1:-r01}; shift || true
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $OUT/bench_under_prof.log 2>&1
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
ls -la $OUT $OUT/trace 2>/dev/null | head -30
head -40 $OUT/kernel_stats.csv
