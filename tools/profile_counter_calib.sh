#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on known byte counts:  bash tools/profile_counter_calib.sh   -> gpurun_out/r06_counter_calibration.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/calib; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $OUT/$c -o c -- $R/tools/microbench/counter_calib > $OUT/$c.log 2>&1
  python3 $R/tools/rocpd_summary.py $(find $OUT/$c -name "*_results.db" | head -1) > $OUT/$c.txt 2>&1
done
python3 $R/tools/counter_calib_summary.py $OUT/FETCH_SIZE.txt $OUT/WRITE_SIZE.txt > $R/gpurun_out/r06_counter_calibration.txt
grep -A12 "^kernel" $OUT/FETCH_SIZE.txt | head -12 >> $R/gpurun_out/r06_counter_calibration.txt
rm -rf $OUT/FETCH_SIZE $OUT/WRITE_SIZE
cat $R/gpurun_out/r06_counter_calibration.txt
