#!/bin/bash
# VALU / LDS instruction counts of ba_schur_items_both for two builds of the library (separate --pmc pass each, --kernel-trace only):
#   bash tools/profile_pmc_schur_ab.sh <tag> <label=lib-or-"default"> ...
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_schur_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  L=${spec%%=*}; P=${spec#*=}
  if [ "$P" = default ]; then unset LLD_AMD_LIB; else export LLD_AMD_LIB=$R/$P; fi
  for pass in "sq_b SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU"; do
    set -- $pass; name=$1; shift
    rocprofv3 --kernel-trace --pmc "$@" -d $OUT/${L}_$name -o $name -- python3 $R/bench.py --steps 1 --warmup 1 --windows-per-gpu 256 --no-cpu-baseline --no-secondary --no-e2e --no-rccl-check --ramp-seconds 0 --gen-workers 1 --groups 1 > $OUT/${L}_$name.log 2>&1
    python3 $R/tools/rocpd_summary.py $(find $OUT/${L}_$name -name "*_results.db" | head -1) > $OUT/${L}_$name.txt 2>&1
  done
done
rm -rf $OUT/*/
ls -la $OUT
