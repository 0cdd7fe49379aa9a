#!/bin/bash
# PMC passes for the local-BA kernels (each counter group in its OWN rocprofv3 run, with --kernel-trace only, as the
# MI355X guide prescribes).  Run on the GPU box:  bash tools/profile_pmc.sh <tag> [windows] [groups]
set -u
TAG=${1:-r01}; NW=${2:-64}; GROUPS_=${3:-1}   # one stream by default: what bench.py's roofline pass times
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 1 --warmup 1 --windows-per-gpu $NW --no-cpu-baseline --no-secondary --no-e2e --no-rccl-check --ramp-seconds 0 --gen-workers 1 --groups $GROUPS_"
run() { # name counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o $name -- $CMD > $OUT/$name.log 2>&1
  python3 $R/tools/rocpd_summary.py $(find $OUT/$name -name "*_results.db" | head -1) > $OUT/$name.txt 2>&1
}
run sq_a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
run sq_b SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_LDS_ATOMIC SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VMEM
run tcc_fetch FETCH_SIZE
run tcc_write WRITE_SIZE
rm -rf $OUT/*/  # keep the text summaries only (the .db files are large)
ls -la $OUT
