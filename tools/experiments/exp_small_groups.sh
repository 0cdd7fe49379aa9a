#!/bin/bash
# resident windows/s of small batches against the number of stream groups (LLD_BA_GROUPS): does a 32-window batch - the per-GPU share of the
# strong-scaling form - gain from more dependent chains in flight?   bash tools/experiments/exp_small_groups.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
for n in 16 32 64 128; do
  for g in 1 2 3 4 6 8; do
    LLD_BA_GROUPS=$g python3 $R/bench.py --windows-per-gpu $n --steps 12 --warmup 3 --no-secondary --no-e2e --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('windows %4d groups %d  %8.1f windows/s  %7.3f ms/step' % ($n, $g, d['value'], d['ms_per_step']))"
  done
done
