#!/bin/bash
# kernel-trace summaries of the secondary paths (pose optimisation, guided ORB searches incl. ComputeStereoMatches):
#   bash tools/profile_secondary.sh <tag>
TAG=${1:-sec}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/kt_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/pose -o kt -- python3 $R/tools/time_pose.py > $OUT/pose.log 2>&1
python3 $R/tools/rocpd_summary.py $(find $OUT/pose -name "*_results.db" | head -1) > $OUT/pose.txt 2>&1
rocprofv3 --kernel-trace -d $OUT/orb -o kt -- python3 $R/tools/exp_orb_search.py > $OUT/orb.log 2>&1
python3 $R/tools/rocpd_summary.py $(find $OUT/orb -name "*_results.db" | head -1) > $OUT/orb.txt 2>&1
rm -rf $OUT/pose $OUT/orb
head -12 $OUT/pose.txt | cut -c1-160; head -16 $OUT/orb.txt | cut -c1-160; tail -3 $OUT/pose.log; tail -8 $OUT/orb.log
