#!/bin/bash
# Everything profiles/r06_* is made of, in one GPU-box call (through gpurun): bash tools/collect_r06.sh [quick]
#   -> gpurun_out/r06/*  (copied into profiles/ by hand after a look)
# The probe libraries (tools/_probe/*.so: stamped builds of single translation units) are REBUILT here from the current sources -- a stale
# one made round 4 file a Python traceback as evidence -- and the script fails when any collected file holds a traceback.
set -u
cd $GRAFT_REPO_ROOT
for b in tools/build_render_probe2.sh tools/build_halo_probe.sh tools/build_split_bf16_probe.sh tools/build_gemm_probe.sh; do [ -f $b ] && bash $b > /dev/null 2>&1; done
O=gpurun_out/r06; mkdir -p $O
export TMPDIR=/tmp
last() { tail -1 "$1" > "$2"; }
python3 bench.py > $O/bench_default.log 2>&1; last $O/bench_default.log $O/r06_bench_default.json
python3 bench.py --config 3 --steps 15 --warmup 3 > $O/b3.log 2>&1; last $O/b3.log $O/r06_bench_cfg3.json
python3 bench.py --config 5 --steps 15 --warmup 3 > $O/b5.log 2>&1; last $O/b5.log $O/r06_bench_cfg5.json
python3 bench.py --encoder res50 --steps 15 --warmup 3 --no-cpu-baseline > $O/b50.log 2>&1; last $O/b50.log $O/r06_bench_res50.json
python3 bench.py --config 3 --hand nimble-synthetic-uv --steps 15 --warmup 3 > $O/b3n.log 2>&1; last $O/b3n.log $O/r06_bench_cfg3_nimble_uv.json
python3 bench.py --config 5 --hand nimble-synthetic-uv --steps 15 --warmup 3 > $O/b5n.log 2>&1; last $O/b5n.log $O/r06_bench_cfg5_nimble_uv.json
python3 bench.py --config 5 --hand nimble-synthetic-uv --aa 1 --steps 15 --warmup 3 > $O/b5n1.log 2>&1; last $O/b5n1.log $O/r06_bench_cfg5_nimble_uv_aa1.json
for spec in "res18:" "cfg3:--config 3" "cfg5:--config 5" "res50:--encoder res50" "cfg3_nimble_uv:--config 3 --hand nimble-synthetic-uv"; do
  tag=${spec%%:*}; args=${spec#*:}
  bash tools/profile_bench.sh r06_$tag $args > /dev/null 2>&1
  cp gpurun_out/prof_r06_$tag/steady.md $O/r06_steady_state_$tag.md
  cp gpurun_out/prof_r06_$tag/kernel_stats.csv $O/r06_kernel_stats_$tag.csv
  [ "$tag" = res18 ] && cp gpurun_out/prof_r06_$tag/timeline.txt $O/r06_step_timeline_res18.txt
done
ROUND_TAG=r06 bash tools/kernel_traffic.sh > /dev/null 2>&1; cp gpurun_out/r06_kernel_traffic.json $O/ 2>/dev/null
bash tools/pmc_probe.sh render "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" render_only.py > $O/r06_pmc_render_sq_counters.txt 2>&1
python3 tools/time_render.py 2>&1 | grep -v amdgpu.ids > $O/r06_time_render.txt
# the NIMBLE-shaped mesh with TexturesUV (B = 48): entry-point times, and the tile kernels' SQ counters (review item 3c: "never had a counter pass")
python3 tools/render_only_nimble.py 2>&1 | grep "NIMBLE" > $O/r06_time_render_nimble.txt
bash tools/pmc_probe.sh nimble_sq "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" render_only_nimble.py > $O/r06_pmc_render_nimble_sq_counters.txt 2>&1
python3 tools/time_wino_bn.py 2>&1 | grep "H =" > $O/r06_time_wino_bn.txt
python3 tools/time_conv_wino2.py 2>&1 | grep -v amdgpu.ids > $O/r06_time_conv_wino2.txt
if [ -f tools/_probe/libhifihr_halo_stamp.so ]; then python3 tools/wino2_stamp.py 2>&1 | grep -v amdgpu.ids > $O/r06_wino2_stamps.txt; fi
# (render_stamp2.py exits 1 when a phase or a counter reads zero or the phases do not add up to the items' own clock: VERDICT r05 14a)
if [ -f tools/_probe/libhifihr_render_stamp2.so ]; then
  python3 tools/render_stamp2.py > $O/_stamps.txt 2>&1; STAMPS_RC=$?
  grep -v amdgpu.ids $O/_stamps.txt > $O/r06_render_fwd_phase_stamps.txt; rm -f $O/_stamps.txt
else STAMPS_RC=1; fi
# one Winograd product on split-bf16 MFMA (measurement only)
if [ -f tools/_probe/libhifihr_split_bf16.so ]; then
  (python3 tools/split_bf16_probe.py 2>&1 | grep -v amdgpu.ids; echo; echo "--- standalone plain LDS-tiled kernel (tools/split_bf16_probe.hip), bf16x3 and bf16x6 ---"; tools/_probe/split_bf16_probe hifihr_amd/libhifihr.so) > $O/r06_split_bf16_probe.txt 2>&1
fi
python3 tools/time_dwconv.py 2>/dev/null > $O/r06_time_dwconv.txt
# where a workgroup of the row-share GEMM spends its life at the step's F(4x4) shapes (stamped build), and what a store costs by pattern
if [ -f tools/_probe/libhifihr_gemm_stamp.so ]; then python3 tools/gemm_stamp4.py 2>&1 | grep -v amdgpu.ids > $O/r06_gemm_stamp4.txt; fi
[ -x tools/_probe/store_pattern ] && tools/_probe/store_pattern > $O/r06_store_pattern.txt 2>&1
# trunk gradient error vs the reference by dispatch (README "Precision of the default dispatch")
python3 -m pytest tests/test_gpu_conv.py -q -s -k "precision_knob or gradient_error_by_dispatch" 2>&1 | grep -E "fixture|HIFIHR_|passed|failed" > $O/r06_precision_by_dispatch.txt
if [ "${1:-}" != quick ]; then bash tools/ablation.sh > $O/r06_ablation.txt 2>&1; fi
# the headline line once more, now that THIS tree's counter file exists (bench.py reads profiles/r*_kernel_traffic.json by source digest): the
# line with `traffic` populated is the one to commit as r06_bench_default.json; the second run shows the box's drift
cp $O/r06_kernel_traffic.json profiles/r06_kernel_traffic.json 2>/dev/null
python3 bench.py > $O/bench_default_a.log 2>&1; last $O/bench_default_a.log $O/r06_bench_default.json
python3 bench.py > $O/bench_default_b.log 2>&1; last $O/bench_default_b.log $O/r06_bench_default_second_run.json
ls -la $O | tail -40
if [ "${STAMPS_RC:-1}" != 0 ]; then echo "collect_r06: the render phase stamps are incomplete (tools/render_stamp2.py): NOT evidence" >&2; exit 1; fi
if grep -l "Traceback (most recent call last)" $O/* 2>/dev/null; then echo "collect_r06: the files above hold a Python traceback: NOT evidence" >&2; exit 1; fi
