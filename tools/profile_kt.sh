#!/bin/bash
# kernel-trace summary of the bench:  bash tools/profile_kt.sh <tag> [windows] [groups]
TAG=${1:-kt}; NW=${2:-256}; GROUPS_=${3:-1}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/kt_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/db -o kt -- python3 $R/bench.py --steps 1 --warmup 1 --windows-per-gpu $NW --no-cpu-baseline --no-secondary --no-e2e --no-rccl-check --ramp-seconds 0 --gen-workers 1 --groups $GROUPS_ > $OUT/kt.log 2>&1
python3 $R/tools/rocpd_summary.py $(find $OUT/db -name "*_results.db" | head -1) > $OUT/kt.txt 2>&1
rm -rf $OUT/db; head -20 $OUT/kt.txt | cut -c1-150
