#!/bin/bash
# launch-ordered timeline of one bench solve (start offset, duration, stream/queue, kernel) to see the tail of the super-steps:
#   bash tools/profile_timeline.sh <tag> [windows] [groups]
TAG=${1:-tl}; NW=${2:-256}; GROUPS_=${3:-0}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/tl_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/db -o kt -- python3 $R/bench.py --steps 1 --warmup 1 --windows-per-gpu $NW --no-cpu-baseline --no-secondary --no-e2e --no-rccl-check --ramp-seconds 0 --gen-workers 1 --groups $GROUPS_ > $OUT/kt.log 2>&1
python3 - $(find $OUT/db -name "*_results.db" | head -1) > $OUT/timeline.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
sel = f"select {name_col}, start, end" + (f", {qcol}" if qcol else ", 0") + " from kernels order by start"
rows = list(cur.execute(sel))
t0 = rows[0][1]
for n, s, e, q in rows:
    print("%10.1f %8.1f q%-4s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n.split("(")[0][-40:]))
PY
rm -rf $OUT/db; wc -l $OUT/timeline.txt
