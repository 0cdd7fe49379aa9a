"""Instruction classes per basic block of one kernel of build/<file>.s (tools/kernel_regs.sh writes it):
   python tools/kernel_blocks.py lld_ba ba_linearize_pt_kernel"""
import sys, re
from collections import Counter
f, name = sys.argv[1], sys.argv[2]
s = open(f"build/{f}.s").read()
i = s.index("\n_ZN5lldba%d%sE" % (len(name), name)); i = s.index(":\n", i); j = s.index("s_endpgm", i)
blocks = []; cur = ("entry", Counter(), [])
for line in s[i:j].split("\n"):
    t = line.strip()
    if not t or t.startswith((";", ".")) and not t.startswith(".LBB"): continue
    if t.startswith(".LBB") and t.split()[0].endswith(":"):
        blocks.append(cur); cur = (t.split()[0], Counter(), []); continue
    op = t.split()[0]
    c = cur[1]
    if "dpp" in t: c["dpp"] += 1
    elif op.startswith("ds_"): c[op] += 1
    elif op.startswith("v_") and "f64" in op: c["f64"] += 1
    elif op.startswith("v_"): c["v32:" + op.split("_")[1]] += 1
    elif op.startswith(("global_", "buffer_", "flat_")): c["vmem"] += 1
    elif op.startswith("s_waitcnt"): c["waitcnt"] += 1
    elif op.startswith(("s_cbranch", "s_branch")): c["branch"] += 1; cur[2].append(t)
    elif op.startswith("s_"): c["salu"] += 1
blocks.append(cur)
for lbl, c, br in blocks:
    n = sum(c.values())
    if n >= int(sys.argv[3]) if len(sys.argv) > 3 else 25:
        v32 = sum(v for k, v in c.items() if k.startswith("v32:"))
        top = ", ".join(f"{k[4:]} {v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1]) if k.startswith("v32:"))[:150]
        rest = {k: v for k, v in c.items() if not k.startswith("v32:")}
        print(f"{lbl:12s} n={n:4d} v32={v32:3d} {rest}  [{top}]  -> {' | '.join(b.split()[-1] for b in br)}")
