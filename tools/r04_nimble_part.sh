#!/bin/bash
# GPU box: render_fwd3_kernel on the NIMBLE-shaped mesh inside the config-3 step, by faces per part (HIFIHR_RENDER_PART)
cd $GRAFT_REPO_ROOT
for p in 128 256 512 1024 4096; do
  HIFIHR_RENDER_PART=$p bash tools/profile_bench.sh r04_np_$p --config 3 --hand nimble-synthetic-uv > /dev/null 2>&1
  echo "part $p: $(grep -h 'render_fwd3' gpurun_out/prof_r04_np_$p/steady.md | cut -c1-110)"
done
