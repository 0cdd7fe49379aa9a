import csv, glob, sys, collections
for d in sys.argv[1:]:
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    if not f: print(d, "no trace"); continue
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    seen = collections.OrderedDict()
    for r in rows:
        n = r["Kernel_Name"].split("(")[0]
        if "linearize" in n or "backsub" in n:
            seen.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(d)
    for n, v in seen.items(): print("   %-60s first launches (us): %s" % (n[-60:], " ".join("%.0f" % x for x in v[:4])))
