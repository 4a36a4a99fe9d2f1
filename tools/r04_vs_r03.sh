#!/bin/bash
# GPU box: the headline bench of this tree against the round-3 tree (git archive of b1a6464 under _r03tree/, built here) on the SAME box,
# alternating, three runs each: boxes of the pool differ by +-1.5 % in ms/step, a comparison across gpurun calls does not resolve 1 %.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_vs_r03.txt; : > $O
for i in 1 2 3; do
  python3 bench.py --no-cpu-baseline --steps 40 2>/dev/null | tail -1 | python3 -c "import json,sys; p=json.loads(sys.stdin.read()); print('round 4 tree  ms/step %.4f  resident %.4f' % (p['ms_per_step'], p['resident_batch']['ms_per_step']))" >> $O
  (cd _r03tree && python3 bench.py --no-cpu-baseline --steps 40 2>/dev/null | tail -1 | python3 -c "import json,sys; p=json.loads(sys.stdin.read()); print('round 3 tree  ms/step %.4f  resident %.4f' % (p['ms_per_step'], p['resident_batch']['ms_per_step']))") >> $O
done
cat $O
