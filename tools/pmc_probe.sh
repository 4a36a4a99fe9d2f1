#!/bin/bash
# usage: tools/pmc_probe.sh <tag> "<counters>" <script.py> ; writes gpurun_out/pmc_<tag>/
set -u
TAG=$1; CTRS=$2; SCRIPT=$3
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT/tools
rocprofv3 --pmc $CTRS --output-format csv -d $OUT/raw -- python3 $SCRIPT > $OUT/run.log 2>&1
find $OUT/raw -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $OUT/counters.csv
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("$OUT/counters.csv")))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"].split("(")[0][-60:]
    if "hifihr" not in r["Kernel_Name"]: continue
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in d.items():
        v = sorted(v)
        print(f"   {c:28s} n={len(v):3d} median={v[len(v)//2]:.4g} max={v[-1]:.4g}")
PY
