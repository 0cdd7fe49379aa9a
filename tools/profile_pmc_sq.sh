#!/bin/bash
# two SQ counter passes only (see tools/profile_pmc.sh):  bash tools/profile_pmc_sq.sh <tag> [windows]
set -u
TAG=${1:-x}; NW=${2:-256}; GROUPS_=1
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 1 --warmup 1 --windows-per-gpu $NW --no-cpu-baseline --no-secondary --no-e2e --no-rccl-check --ramp-seconds 0 --gen-workers 1 --groups $GROUPS_"
run() { local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o $name -- $CMD > $OUT/$name.log 2>&1
  python3 $R/tools/rocpd_summary.py $(find $OUT/$name -name "*_results.db" | head -1) > $OUT/$name.txt 2>&1
}
run sq_a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
run sq_b SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_LDS_ATOMIC SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VMEM
run sq_c SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU
rm -rf $OUT/*/
