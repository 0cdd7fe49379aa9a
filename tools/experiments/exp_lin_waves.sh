#!/bin/bash
# wavefronts per point / line linearise workgroup (experiments build, LLD_BA_LIN_WAVES="pt,ln"; bit-reproducible mode: = accumulator copies):
# resident rate of 256 LBA-B windows and the linearise phase of one stream.   bash tools/experiments/exp_lin_waves.sh [extra bench flags]
R=${GRAFT_REPO_ROOT:-$(pwd)}
export LLD_AMD_LIB=$R/lld_slam_amd/csrc/liblld_amd_exp.so
for lw in default 4,4 8,4 6,4 8,8 4,2 6,2 8,2; do
  if [ $lw = default ]; then unset LLD_BA_LIN_WAVES; else export LLD_BA_LIN_WAVES=$lw; fi
  python3 $R/bench.py --windows-per-gpu 256 --steps 10 --warmup 3 --no-secondary --no-e2e --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('lin waves $lw  %8.1f windows/s  %7.3f ms/step  phases(1 stream) %s' % (d['value'], d['ms_per_step'], d['roofline']['phase_ms_single_stream_step']))"
done
