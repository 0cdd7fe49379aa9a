#!/usr/bin/env python3
"""List every kernel dispatch of a rocprofv3 rocpd database in launch order: index, duration (us), kernel name."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
for i, (n, s, e) in enumerate(cur.execute(f"select {name_col}, start, end from kernels order by start")):
    print(i, "%.1f" % ((e - s) / 1e3), n.split("(")[0][-50:])
