#!/bin/bash
# windows/s of resident batches of 32 / 64 / 128 / 256 LBA-B windows on ONE GPU: the 32-window figure over the 256-window one predicts
# the strong-scaling efficiency of 256 windows on 8 GPUs.   bash tools/experiments/exp_small_batches.sh [groups-override]
R=${GRAFT_REPO_ROOT:-$(pwd)}
for n in 32 64 128 256; do
  python3 $R/bench.py --windows-per-gpu $n --steps 12 --warmup 3 --no-secondary --no-e2e --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('windows %4d  %8.1f windows/s  %7.3f ms/step  phases(1 stream) %s' % ($n, d['value'], d['ms_per_step'], d['roofline']['phase_ms_single_stream_step']))"
done
