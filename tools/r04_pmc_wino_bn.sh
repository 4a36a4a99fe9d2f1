#!/bin/bash
# GPU box: what bounds the fused batch-norm / Winograd transform kernels (review item 5a).  Separate --pmc passes over tools/time_wino_bn.py:
# L2 hit / miss, fabric-side bytes (FETCH_SIZE x 2 on gfx950, WRITE_SIZE), wave occupancy and stall counters.  -> gpurun_out/r04_pmc_wino_bn.txt
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_pmc_wino_bn.txt; : > $O
for set in "l2:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "fetch:FETCH_SIZE" "write:WRITE_SIZE" "sq:SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
  tag=${set%%:*}; ctrs=${set#*:}
  echo "==== pass $tag: $ctrs" >> $O
  bash tools/pmc_probe.sh wbn_$tag "$ctrs" time_wino_bn.py 2>&1 | grep -A12 "wino4\|bn_bwd_apply\|bn_act" | grep -v "^--" >> $O
done
cat $O | head -150
