#!/bin/bash
# Round 5, after the per-phase events left the solve: stream groups x (queued super-steps / fused pair kernels from how many windows per group) for
# small and medium batches.  Experiments build (LLD_BA_GROUPS, LLD_BA_CHUNK_FROM, LLD_BA_FUSE_BELOW).   bash tools/experiments/exp_small_sweep2.sh [windows ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
export LLD_AMD_LIB=$R/lld_slam_amd/csrc/liblld_amd_exp.so
for n in ${@:-32 64 128}; do
  for cf in ${CFS:-24 40 72}; do
    for g in ${GROUPS_LIST:-1 2 3 4}; do
      LLD_BA_GROUPS=$g LLD_BA_CHUNK_FROM=$cf LLD_BA_FUSE_BELOW=$cf python3 $R/bench.py --windows-per-gpu $n --steps 12 --warmup 3 --no-secondary --no-e2e --no-cpu-baseline --no-rccl-check 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('windows %4d  chunk/fuse below %3d  groups %d  %8.1f windows/s  %7.3f ms/solve' % ($n, $cf, $g, d['value'], d['ms_per_step']))"
    done
  done
done
