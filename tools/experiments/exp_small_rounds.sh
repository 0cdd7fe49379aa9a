#!/bin/bash
# resident windows/s of small batches against the tasks-per-wavefront of the landmark kernels (LLD_BA_ROUNDS, experiments build) and the
# number of stream groups: where do the wide kernels of a 16- / 32-window batch lose their efficiency?   bash tools/experiments/exp_small_rounds.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
export LLD_AMD_LIB=$R/lld_slam_amd/csrc/liblld_amd_exp.so
for n in ${NS:-16 32}; do
  for g in ${GS:-1 2 4}; do
    for r in ${RS:-1,1,1,1 2,1,2,1 4,2,4,1 8,2,4,1 16,4,16,1}; do
      LLD_BA_GROUPS=$g LLD_BA_ROUNDS=$r python3 $R/bench.py --windows-per-gpu $n --steps 12 --warmup 3 --no-secondary --no-e2e --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('windows %4d groups %d rounds %-10s %8.1f windows/s  %7.3f ms/step  %s' % ($n, $g, '$r', d['value'], d['ms_per_step'], d['roofline']['phase_ms_single_stream_step']))"
    done
  done
done
