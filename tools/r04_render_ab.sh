#!/bin/bash
# GPU box: the rasteriser's forward, third form (persistent work queue) against the second (HIFIHR_RENDER_FWD3=0), MANO mesh at B = 32
# and the NIMBLE-sized mesh through rocprofv3 for per-kernel times.   -> gpurun_out/r04_render_ab.txt
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_render_ab.txt; : > $O
for v in 1 0; do
  for w in 16 8 32; do
    [ $v = 0 ] && [ $w != 16 ] && continue
    echo "== HIFIHR_RENDER_FWD3=$v HIFIHR_RENDER_WGS=$w" >> $O
    HIFIHR_RENDER_FWD3=$v HIFIHR_RENDER_WGS=$w python3 tools/time_render.py 2>&1 | grep -v amdgpu.ids >> $O
  done
done
cat $O
