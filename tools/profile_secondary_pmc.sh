#!/bin/bash
# kernel trace + PMC passes (one rocprofv3 run per counter group, --kernel-trace only) of the secondary configs:
#   bash tools/profile_secondary_pmc.sh <tag>
set -u
TAG=${1:-sec}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/sec_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp LLD_GEN_WORKERS=1
CMD="python3 $R/tools/run_secondary_kernels.py"
rocprofv3 --kernel-trace -d $OUT/kt -o kt -- $CMD > $OUT/kt.log 2>&1
python3 $R/tools/rocpd_summary.py $(find $OUT/kt -name "*_results.db" | head -1) > $OUT/kt.txt 2>&1
run() { local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o $name -- $CMD > $OUT/$name.log 2>&1
  python3 $R/tools/rocpd_summary.py $(find $OUT/$name -name "*_results.db" | head -1) > $OUT/$name.txt 2>&1
}
run sq_a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
run sq_b SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_LDS_ATOMIC SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VMEM
run tcc_fetch FETCH_SIZE
run tcc_write WRITE_SIZE
rm -rf $OUT/*/
tail -1 $OUT/kt.log | cut -c1-300; head -8 $OUT/kt.txt | cut -c1-150
