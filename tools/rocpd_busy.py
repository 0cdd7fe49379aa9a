#!/usr/bin/env python3
"""GPU busy fraction from a rocprofv3 rocpd database: union of all kernel intervals / (last end - first start), plus the
average number of kernels in flight.  Usage: python tools/rocpd_busy.py <db> [skip_first_ms [window_ms]]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = sorted(cur.execute("select start, end from kernels").fetchall())
skip = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 0.0
t0 = rows[0][0] + skip
t1 = t0 + float(sys.argv[3]) * 1e6 if len(sys.argv) > 3 else None
rows = [(max(s, t0), e if t1 is None else min(e, t1)) for s, e in rows if e > t0 and (t1 is None or s < t1)]
busy = 0; cs, ce = rows[0]
tot = sum(e - s for s, e in rows)
for s, e in rows[1:]:
    if s > ce: busy += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
busy += ce - cs
wall = max(e for _, e in rows) - rows[0][0]
print("wall %.2f ms  busy %.2f ms (%.1f%%)  kernel-time %.2f ms  avg in flight %.2f" % (wall / 1e6, busy / 1e6, 100.0 * busy / wall, tot / 1e6, tot / wall))
