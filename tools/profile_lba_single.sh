#!/bin/bash
# kernel trace of one lld_local_ba call per LBA-B window:  bash tools/profile_lba_single.sh <tag>
TAG=${1:-lba1}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/kt_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/lba1 -o kt -- python3 $R/tools/time_lba_single.py > $OUT/lba1.log 2>&1
python3 $R/tools/rocpd_summary.py $(find $OUT/lba1 -name "*_results.db" | head -1) > $OUT/lba1.txt 2>&1
rm -rf $OUT/lba1
head -8 $OUT/lba1.txt | cut -c1-150; tail -1 $OUT/lba1.log
