#!/usr/bin/env python3
"""Builds profiles/roofline_issue.json from the SQ passes of tools/profile_pmc.sh: per kernel family and launch (= super-step), the VALU and LDS
instructions issued and the LDS pipe / bank-conflict cycles - the counters behind bench.py's compute ruler for EVERY family (round 6):
   valu_issue = SQ_INSTS_VALU x 4 clocks / (1024 SIMDs x launch time x 2.4 GHz)     (one VALU instruction per 4 clocks and SIMD is the issue peak)
   lds_busy   = SQ_LDS_IDX_ACTIVE / (256 CUs x launch time x 2.4 GHz)

    python tools/make_roofline_issue.py gpurun_out/pmc_<tag> ["what the passes were taken on"] > profiles/roofline_issue.json
"""
import json, re, sys
sys.path.insert(0, __file__.rsplit("/", 1)[0])
from make_roofline_traffic import FAMILIES

COUNTERS = ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_LDS_ATOMIC", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_ADDR_CONFLICT",
            "SQ_LDS_BANK_CONFLICT", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAVES")


def sums(path):
    out = {}
    for line in open(path):
        m = re.match(r"^(.*?)\s+(SQ_[A-Z_]+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s*$", line.rstrip())
        if m: out[(m.group(1).strip(), m.group(2))] = (int(m.group(3)), float(m.group(4)))
    return out


def main(d):
    s = {}
    for name in ("sq_a", "sq_b"):
        s.update(sums(f"{d}/{name}.txt"))
    n_steps = [n for (k, c), (n, _) in s.items() if "ba_schur_reduce" in k and c == "SQ_INSTS_VALU"][0]
    res = {"_note": "per launch (= per super-step) of each kernel family: sums over the family's kernels and dispatches of rocprofv3 --pmc SQ passes divided by the "
                    "number of super-steps; bench.py --windows-per-gpu 256, one stream group. Used by bench.py: valu_issue = SQ_INSTS_VALU x 4 / (1024 SIMDs x "
                    "launch time x 2.4 GHz), lds_busy = SQ_LDS_IDX_ACTIVE / (256 CUs x launch time x 2.4 GHz).", "_super_steps": n_steps}
    for fam, kernels in FAMILIES.items():
        res[fam] = {c: round(sum(v for (k, cc), (_, v) in s.items() if cc == c and any(x in k for x in kernels)) / n_steps, 1) for c in COUNTERS}
    if len(sys.argv) > 2: res["_taken_on"] = sys.argv[2]
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
