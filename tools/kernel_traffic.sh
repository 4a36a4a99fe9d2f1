#!/bin/bash
# HBM traffic per launch of every hand-written kernel inside the training step (BASELINE configs[1]) from the PMC counters, as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (the TCC block cannot
# hold both), eager step so that every dispatch is counted on its own; gfx950 correction (FETCH_SIZE x 2) applied by
# tools/kernel_traffic_summary.py, which also stamps the digest of the kernel sources so that bench.py only uses a measurement of
# THIS tree.  usage (GPU box, through gpurun): bash tools/kernel_traffic.sh   -> gpurun_out/${ROUND_TAG:-r03}_kernel_traffic.json
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_traffic
mkdir -p $OUT
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/$C -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --graph 0 --no-cpu-baseline --no-rooflines > $OUT/$C.log 2>&1
  find $OUT/$C -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $OUT/$C.csv
  rm -rf $OUT/$C
done
python3 $GRAFT_REPO_ROOT/tools/kernel_traffic_summary.py $OUT > $GRAFT_REPO_ROOT/gpurun_out/${ROUND_TAG:-r03}_kernel_traffic.json
rm -f $OUT/FETCH_SIZE.csv $OUT/WRITE_SIZE.csv
head -c 1500 $GRAFT_REPO_ROOT/gpurun_out/${ROUND_TAG:-r03}_kernel_traffic.json
