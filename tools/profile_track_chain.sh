#!/bin/bash
# kernel trace of the Tracking-thread chain driven from C++ (examples/harness track, 40 frames on one handle):  bash tools/profile_track_chain.sh <tag>
TAG=${1:-track}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/kt_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 -c "
import sys; sys.path.insert(0, '$R')
from lld_slam_amd import synth, tracking
tracking.write_harness_scene('/tmp/track_in.bin', synth.make_tracking_scene(0), repeats=40)"
rocprofv3 --kernel-trace -d $OUT/db -o kt -- $R/examples/harness track /tmp/track_in.bin /tmp/track_out.bin > $OUT/kt.log 2>&1
python3 $R/tools/rocpd_summary.py $(find $OUT/db -name "*_results.db" | head -1) > $OUT/kt.txt 2>&1
rm -rf $OUT/db; head -30 $OUT/kt.txt | cut -c1-150
