#!/bin/bash
# Every profile the round's documents cite, on ONE box and build:  bash tools/profile_round.sh <tag>
#   kernel trace of the bench (one stream group), SQ passes a / b / c, TCC fetch / write passes (each its own rocprofv3 --kernel-trace --pmc run),
#   the secondary kernels (trace + passes), the Tracking-thread chain (trace), then the plain bench line.
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
bash $R/tools/profile_kt.sh ${TAG}_256 256 1 > /dev/null 2>&1
bash $R/tools/profile_pmc_sq.sh $TAG 256 > /dev/null 2>&1
( OUT=$R/gpurun_out/pmc_$TAG; cd /tmp && export TMPDIR=/tmp
  CMD="python3 $R/bench.py --steps 1 --warmup 1 --windows-per-gpu 256 --no-cpu-baseline --no-secondary --no-e2e --no-rccl-check --ramp-seconds 0 --gen-workers 1 --groups 1"
  for pass in "tcc_fetch FETCH_SIZE" "tcc_write WRITE_SIZE"; do set -- $pass; name=$1; shift
    rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o $name -- $CMD > $OUT/$name.log 2>&1
    python3 $R/tools/rocpd_summary.py $(find $OUT/$name -name "*_results.db" | head -1) > $OUT/$name.txt 2>&1
  done; rm -rf $OUT/*/ )
bash $R/tools/profile_secondary_pmc.sh $TAG > /dev/null 2>&1
bash $R/tools/profile_track_chain.sh $TAG > /dev/null 2>&1
cd $R && python3 bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
ls gpurun_out/pmc_$TAG gpurun_out/sec_$TAG gpurun_out/kt_${TAG}_256 gpurun_out/kt_$TAG; tail -c 600 gpurun_out/bench_$TAG.json
