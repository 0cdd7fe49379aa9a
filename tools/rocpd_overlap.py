#!/usr/bin/env python3
"""Kernel / memory-copy overlap from a rocprofv3 rocpd database (--kernel-trace --memory-copy-trace):
   python tools/rocpd_overlap.py <db> [skip_first_ms]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kern = sorted(cur.execute("select start, end from kernels").fetchall())
mc_tab = [t for t in tabs if "memory_cop" in t.lower()]
print("tables with copies:", mc_tab)
cols = [r[1] for r in cur.execute(f"pragma table_info({mc_tab[0]})")]
print(cols)
size_col = "size" if "size" in cols else None
name_col = "name" if "name" in cols else None
q = f"select start, end{', ' + size_col if size_col else ''}{', ' + name_col if name_col else ''} from {mc_tab[0]}"
cop = sorted(cur.execute(q).fetchall())
skip = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 0.0
t0 = min(kern[0][0], cop[0][0]) + skip
def union(iv):
    out = []; 
    for s, e in sorted(iv):
        if e <= t0: continue
        s = max(s, t0)
        if out and s <= out[-1][1]: out[-1][1] = max(out[-1][1], e)
        else: out.append([s, e])
    return out
def total(u): return sum(e - s for s, e in u)
def inter(a, b):
    i = j = 0; t = 0
    while i < len(a) and j < len(b):
        s = max(a[i][0], b[j][0]); e = min(a[i][1], b[j][1])
        if e > s: t += e - s
        if a[i][1] < b[j][1]: i += 1
        else: j += 1
    return t
uk = union([(s, e) for s, e in kern]); big = [c for c in cop if (c[1] - c[0]) > 1e6]
uc = union([(c[0], c[1]) for c in cop]); ub = union([(c[0], c[1]) for c in big])
wall = max(uk[-1][1], uc[-1][1]) - t0
print("wall %.1f ms | kernels busy %.1f ms | copies busy %.1f ms (copies > 1 ms: %d, busy %.1f ms) | kernel-and-copy overlap %.1f ms | neither %.1f ms" %
      (wall / 1e6, total(uk) / 1e6, total(uc) / 1e6, len(big), total(ub) / 1e6, inter(uk, uc) / 1e6, (wall - total(uk) - total(uc) + inter(uk, uc)) / 1e6))
for c in big[:16]:
    if c[1] <= t0: continue
    ov = inter(uk, [[max(c[0], t0), c[1]]])
    print("  copy at %8.1f ms, %.2f ms" % ((c[0] - t0) / 1e6, (c[1] - c[0]) / 1e6), c[2:] if len(c) > 2 else "", "GB/s %.1f" % (c[2] / (c[1] - c[0])) if size_col else "",
          "| kernels running during %.2f ms of it" % (ov / 1e6))
# idle gaps between kernels, and the kernels' own times
gaps = sorted(((uk[i + 1][0] - uk[i][1]) / 1e3 for i in range(len(uk) - 1)), reverse=True)
print("idle gaps: n %d, total %.1f ms, > 1 ms: %d (%.1f ms), 0.1..1 ms: %d (%.1f ms), largest %s us" %
      (len(gaps), sum(gaps) / 1e3, sum(g > 1000 for g in gaps), sum(g for g in gaps if g > 1000) / 1e3, sum(100 < g <= 1000 for g in gaps),
       sum(g for g in gaps if 100 < g <= 1000) / 1e3, [int(g) for g in gaps[:8]]))
kc = [r[1] for r in cur.execute("pragma table_info(kernels)")]
nm = "name" if "name" in kc else "kernel_name"
agg = {}
for n, s, e in cur.execute(f"select {nm}, start, end from kernels"):
    if e <= t0: continue
    k = n.split("(")[0].split("::")[-1][:40]; a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:10]:
    print("  %-40s calls %6d  total %9.1f ms  avg %8.1f us" % (k, n, t / 1e3, t / n))
