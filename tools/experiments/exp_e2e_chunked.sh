#!/bin/bash
# host-buffer pipeline (3 lanes of 256-window batches) with large groups queueing several super-steps per poll (experiments build, LLD_BA_CHUNK_FROM)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export LLD_AMD_LIB=$R/lld_slam_amd/csrc/liblld_amd_exp.so
for cf in default 1000; do
  if [ $cf = default ]; then unset LLD_BA_CHUNK_FROM; else export LLD_BA_CHUNK_FROM=$cf; fi
  echo "LLD_BA_CHUNK_FROM $cf"
  python3 $R/tools/experiments/exp_e2e_lanes.py 256 8 3 2>&1 | grep -v amdgpu.ids
  python3 $R/bench.py --windows-per-gpu 256 --steps 10 --warmup 3 --no-secondary --no-e2e --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('  resident %8.1f windows/s  %7.3f ms/step' % (d['value'], d['ms_per_step']))"
done
