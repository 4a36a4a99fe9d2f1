#!/bin/bash
# GPU box: rocprofv3 steady-state table of this tree and of _r03tree in one call; tools/diff_steady.py lists what moved
cd $GRAFT_REPO_ROOT
bash tools/profile_bench.sh r04_cmp_new > /dev/null 2>&1
(cd _r03tree && GRAFT_REPO_ROOT=$GRAFT_REPO_ROOT/_r03tree bash tools/profile_bench.sh r03_cmp_old > /dev/null 2>&1)
cp _r03tree/gpurun_out/prof_r03_cmp_old/steady.md gpurun_out/prof_r04_cmp_new/steady_r03tree.md 2>/dev/null
python3 tools/diff_steady.py gpurun_out/prof_r04_cmp_new/steady_r03tree.md gpurun_out/prof_r04_cmp_new/steady.md
