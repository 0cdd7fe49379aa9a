#!/usr/bin/env python3
"""Builds profiles/roofline_traffic.json from the FETCH_SIZE / WRITE_SIZE summaries of tools/profile_pmc.sh:
HBM-side bytes per launch of each kernel family = sum over its kernels of the per-dispatch average.

    python tools/make_roofline_traffic.py gpurun_out/pmc_<tag> ["commit / date the passes were taken on"] > profiles/roofline_traffic.json
"""
import json, re, sys

FAMILIES = {"ba_linearize": ("ba_linearize_", "ba_hpp_reduce"),
            "ba_schur": ("ba_schur_items", "ba_schur_reduce", "ba_symmetrize"),
            "ba_solve": ("ba_chol_mfma", "ba_chol_sparse", "ba_chol_kernel", "ba_pcg"),
            "ba_backsub": ("ba_backsub_",),
            "ba_control": ("ba_control",)}


def sums(path, counter):
    """kernel -> (dispatches, sum over its dispatches)"""
    out = {}
    for line in open(path):
        m = re.match(r"^(.*?)\s+" + counter + r"\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s*$", line.rstrip())
        if m:
            out[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)))
    return out


def main(d):
    fetch = sums(d + "/tcc_fetch.txt", "FETCH_SIZE"); write = sums(d + "/tcc_write.txt", "WRITE_SIZE")
    # one family launch per super-step: ba_schur_reduce runs in every one of them (the point / line kernels of a pair share a launch in the
    # tail of a solve, when few windows are left: per-kernel averages would count such a super-step twice)
    n_steps = [n for k, (n, _) in fetch.items() if "ba_schur_reduce" in k][0]
    res = {"_note": "HBM-side bytes per launch (= per super-step) of each kernel family: the sum over the family's kernels and dispatches divided by "
                    "the number of super-steps, from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, bench.py --windows-per-gpu 256, one stream (what "
                    "bench.py's roofline pass times). Counters are in KiB; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (it "
                    "tallies 128-B requests at 64 B); WRITE_SIZE is uncalibrated.", "_super_steps": n_steps, "_raw_KiB_per_super_step": {}}
    for fam, kernels in FAMILIES.items():
        f = sum(v for k, (_, v) in fetch.items() if any(x in k for x in kernels)) / n_steps
        w = sum(v for k, (_, v) in write.items() if any(x in k for x in kernels)) / n_steps
        res["_raw_KiB_per_super_step"][fam] = {"FETCH_SIZE": round(f, 1), "WRITE_SIZE": round(w, 1)}
        res[fam] = int((2.0 * f + w) * 1024)
    # everything else the solves launch (init, classification, read-back), per SOLVE: the finalize kernel runs once per solve and group
    fam_all = tuple(x for v in FAMILIES.values() for x in v)
    n_solves = max(1, [n for k, (n, _) in fetch.items() if "ba_finalize" in k][0])
    fo = sum(v for k, (_, v) in fetch.items() if k.strip().startswith("ba_") or "lldba" in k if not any(x in k for x in fam_all))
    wo = sum(v for k, (_, v) in write.items() if k.strip().startswith("ba_") or "lldba" in k if not any(x in k for x in fam_all))
    res["_other_per_solve"] = int((2.0 * fo + wo) * 1024 / n_solves)
    res["_solves"] = n_solves
    if len(sys.argv) > 2:
        res["_taken_on"] = sys.argv[2]
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
