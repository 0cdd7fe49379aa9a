#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (what `rocprofv3 --kernel-trace --stats` writes on this image) as text:
per-kernel calls / total / average / min / max duration, and per-kernel PMC counter sums when a --pmc pass was recorded.

    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/r01_kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by {name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# {path}")
    print(f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for n, c, t, a, mn, mx in rows:
        short = n.replace("(anonymous namespace)::", "").split("(")[0][-70:]
        print(f"{short:70s} {c:7d} {t / 1e6:10.3f} {a / 1e3:10.2f} {mn / 1e3:10.2f} {mx / 1e3:10.2f} {100.0 * t / total:6.2f}")
    try:
        pm = cur.execute("select name from sqlite_master where name='counters_collection'").fetchall()
        if pm:
            ccols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
            kn = "kernel_name" if "kernel_name" in ccols else "name"
            cn = "counter_name" if "counter_name" in ccols else "pmc_name"
            vn = "value" if "value" in ccols else "counter_value"
            rows = cur.execute(f"select {kn}, {cn}, count(*), sum({vn}) from counters_collection group by {kn}, {cn} order by 1, 2").fetchall()
            if rows:
                print("\n# PMC counters: kernel, counter, dispatches, sum, per-dispatch")
                for k, c, n, v in rows:
                    print(f"{k.replace('(anonymous namespace)::', '').split('(')[0][-60:]:60s} {c:28s} {n:7d} {v:18.1f} {v / max(n, 1):16.1f}")
    except sqlite3.Error as e:
        print("# no counters:", e)


if __name__ == "__main__":
    main(sys.argv[1])
