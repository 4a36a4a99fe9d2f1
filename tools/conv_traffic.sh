#!/bin/bash
# HBM traffic of the convolution kernel inside the training step (BASELINE configs[1]) from the PMC counters, as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (the TCC block
# cannot hold both), eager step so that every dispatch is counted on its own; gfx950 correction (FETCH_SIZE x 2) applied by
# tools/conv_traffic_summary.py.  usage (GPU box): bash tools/conv_traffic.sh <tag>
set -u
TAG=${1:-r01}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_conv_$TAG
mkdir -p $OUT
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/$C -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 2 --graph 0 --no-cpu-baseline > $OUT/$C.log 2>&1
  find $OUT/$C -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $OUT/$C.csv
done
python3 $GRAFT_REPO_ROOT/tools/conv_traffic_summary.py $OUT > $OUT/summary.json
cat $OUT/summary.json
