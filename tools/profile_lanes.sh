#!/bin/bash
# kernel + memory-copy trace of the host-buffer pipeline:  bash tools/profile_lanes.sh <lanes> [batches]
L=${1:-2}; NB=${2:-4}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/lanes_$L; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp LLD_GEN_WORKERS=1 LLD_EXP_SOLVE_LOCK=${3:-0}
timeout 500 rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/db -o lanes -- python3 $R/tools/exp_e2e_lanes.py 256 $NB $L > $OUT/run.log 2>&1
tail -3 $OUT/run.log
python3 $R/tools/rocpd_overlap.py $(find $OUT/db -name "*_results.db" | head -1) 200 2>&1 | grep -v "^tables\|^\[" | head -40
rm -rf $OUT/db
