#!/bin/bash
# Everything profiles/r04_* is made of, in one GPU-box call (through gpurun): bash tools/collect_r04.sh [quick]
#   -> gpurun_out/r04/*  (copied into profiles/ by hand after a look)
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
last() { tail -1 "$1" > "$2"; }
python3 bench.py > $O/bench_default.log 2>&1; last $O/bench_default.log $O/r04_bench_default.json
python3 bench.py --config 3 --steps 15 --warmup 3 > $O/b3.log 2>&1; last $O/b3.log $O/r04_bench_cfg3.json
python3 bench.py --config 5 --steps 15 --warmup 3 > $O/b5.log 2>&1; last $O/b5.log $O/r04_bench_cfg5.json
python3 bench.py --encoder res50 --steps 15 --warmup 3 --no-cpu-baseline > $O/b50.log 2>&1; last $O/b50.log $O/r04_bench_res50.json
python3 bench.py --config 3 --hand nimble-synthetic-uv --steps 15 --warmup 3 > $O/b3n.log 2>&1; last $O/b3n.log $O/r04_bench_cfg3_nimble_uv.json
python3 bench.py --config 5 --hand nimble-synthetic-uv --steps 15 --warmup 3 > $O/b5n.log 2>&1; last $O/b5n.log $O/r04_bench_cfg5_nimble_uv.json
python3 bench.py --config 5 --hand nimble-synthetic-uv --aa 1 --steps 15 --warmup 3 > $O/b5n1.log 2>&1; last $O/b5n1.log $O/r04_bench_cfg5_nimble_uv_aa1.json
for spec in "res18:" "cfg3:--config 3" "cfg5:--config 5" "res50:--encoder res50" "cfg3_nimble_uv:--config 3 --hand nimble-synthetic-uv"; do
  tag=${spec%%:*}; args=${spec#*:}
  bash tools/profile_bench.sh r04_$tag $args > /dev/null 2>&1
  cp gpurun_out/prof_r04_$tag/steady.md $O/r04_steady_state_$tag.md
  cp gpurun_out/prof_r04_$tag/kernel_stats.csv $O/r04_kernel_stats_$tag.csv
  [ "$tag" = res18 ] && cp gpurun_out/prof_r04_$tag/timeline.txt $O/r04_step_timeline_res18.txt
done
ROUND_TAG=r04 bash tools/kernel_traffic.sh > /dev/null 2>&1; cp gpurun_out/r04_kernel_traffic.json $O/ 2>/dev/null
bash tools/pmc_probe.sh render "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" render_only.py > $O/r04_pmc_render_sq_counters.txt 2>&1
python3 tools/time_render.py 2>&1 | grep -v amdgpu.ids > $O/r04_time_render.txt
# the NIMBLE-shaped mesh with TexturesUV (B = 48): entry-point times, and the tile kernels' SQ counters (review item 3c: "never had a counter pass")
python3 tools/render_only_nimble.py 2>&1 | grep "NIMBLE" > $O/r04_time_render_nimble.txt
bash tools/pmc_probe.sh nimble_sq "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" render_only_nimble.py > $O/r04_pmc_render_nimble_sq_counters.txt 2>&1
bash tools/r04_pmc_wino_bn.sh > /dev/null 2>&1; cp gpurun_out/r04_pmc_wino_bn.txt $O/ 2>/dev/null
python3 tools/time_wino_bn.py 2>&1 | grep "H =" > $O/r04_time_wino_bn.txt
python3 tools/time_conv_wino2.py 2>&1 | grep -v amdgpu.ids > $O/r04_time_conv_wino2.txt
if [ -f tools/_probe/libhifihr_halo_stamp.so ]; then python3 tools/wino2_stamp.py 2>&1 | grep -v amdgpu.ids > $O/r04_wino2_stamps.txt; fi
if [ -f tools/_probe/libhifihr_render_stamp2.so ]; then python3 tools/render_stamp2.py 2>&1 | grep -v amdgpu.ids > $O/r04_render_fwd_phase_stamps.txt; fi
python3 tools/time_dwconv.py 2>/dev/null > $O/r04_time_dwconv.txt
# trunk gradient error vs the reference by dispatch (README "Precision of the default dispatch")
python3 -m pytest tests/test_gpu_conv.py -q -s -k "precision_reference_knob or gradient_error_by_dispatch" 2>&1 | grep -E "fixture|HIFIHR_|passed|failed" > $O/r04_precision_by_dispatch.txt
if [ "${1:-}" != quick ]; then bash tools/ablation.sh > $O/r04_ablation.txt 2>&1; fi
ls -la $O | tail -40
