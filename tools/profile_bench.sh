#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel trace + stats of the default bench command, summarised per training step by
# tools/trace_summary.py.  usage: tools/profile_bench.sh <tag> [bench args]
#   -> gpurun_out/prof_<tag>/{kernel_stats.csv, steady.md, bench_under_prof.log}
set -u
TAG=${1:-r02}; shift || true
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-rooflines "$@" > $OUT/bench_under_prof.log 2>&1
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
T=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/trace_summary.py "$T" 5 $OUT/timeline.txt > $OUT/steady.md 2>&1
rm -rf $OUT/trace          # the raw trace is large; the two summaries are what gets committed under profiles/
head -70 $OUT/steady.md
