#!/bin/bash
# kernel-trace summaries of the secondary paths (pose optimisation, guided ORB searches incl. ComputeStereoMatches, one
# lld_local_ba call on an LBA-B window, the essential graph with both solvers):
#   bash tools/profile_secondary.sh <tag>
TAG=${1:-sec}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/kt_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/pose -o kt -- python3 $R/tools/time_pose.py > $OUT/pose.log 2>&1
python3 $R/tools/rocpd_summary.py $(find $OUT/pose -name "*_results.db" | head -1) > $OUT/pose.txt 2>&1
rocprofv3 --kernel-trace -d $OUT/orb -o kt -- python3 $R/tools/exp_orb_search.py > $OUT/orb.log 2>&1
python3 $R/tools/rocpd_summary.py $(find $OUT/orb -name "*_results.db" | head -1) > $OUT/orb.txt 2>&1
rocprofv3 --kernel-trace -d $OUT/lba1 -o kt -- python3 $R/tools/time_lba_single.py > $OUT/lba1.log 2>&1
python3 $R/tools/rocpd_summary.py $(find $OUT/lba1 -name "*_results.db" | head -1) > $OUT/lba1.txt 2>&1
rocprofv3 --kernel-trace -d $OUT/eg -o kt -- python3 $R/tools/time_essential_graph.py 300 1000 > $OUT/eg.log 2>&1
python3 $R/tools/rocpd_summary.py $(find $OUT/eg -name "*_results.db" | head -1) > $OUT/eg.txt 2>&1
rm -rf $OUT/pose $OUT/orb $OUT/lba1 $OUT/eg
head -12 $OUT/pose.txt | cut -c1-160; head -16 $OUT/orb.txt | cut -c1-160; tail -3 $OUT/pose.log; tail -8 $OUT/orb.log
head -20 $OUT/lba1.txt | cut -c1-160; head -16 $OUT/eg.txt | cut -c1-160; tail -2 $OUT/lba1.log; tail -4 $OUT/eg.log
