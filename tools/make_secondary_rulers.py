#!/usr/bin/env python3
"""Counter-backed rulers of the secondary kernels from tools/profile_secondary_pmc.sh (kernel trace + SQ / TCC passes of tools/run_secondary_kernels.py):
per kernel and dispatch the VALU instructions, LDS pipe cycles and HBM-side bytes the counters saw, over the kernel-trace duration.
    python tools/make_secondary_rulers.py gpurun_out/sec_<tag> ["taken on ..."] > profiles/secondary_rulers.json
valu_issue: SQ_INSTS_VALU x 4 clocks / (SIMDs the launch can use x duration x 2.4 GHz) - the chip's 1024 SIMDs for the batched kernels, the SIMDs of
the compute units its workgroups occupy for a launch smaller than the chip (one frame's pose_opt_kernel / orb_search_kernel: ONE workgroup, 4 SIMDs)."""
import json, re, sys


def load(path):
    kt, pm = {}, {}
    for line in open(path):
        m = re.match(r"^(.*?)\s+((?:SQ_|FETCH|WRITE)[A-Z_]+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s*$", line.rstrip())
        if m: pm[(m.group(1).strip(), m.group(2))] = float(m.group(5)); continue
        m = re.match(r"^(.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line.rstrip())
        if m: kt[m.group(1).strip()] = (int(m.group(2)), float(m.group(4)))
    return kt, pm


def main(d):
    kt, _ = load(d + "/kt.txt")
    pm = {}
    for n in ("sq_a", "sq_b", "tcc_fetch", "tcc_write"): pm.update(load(f"{d}/{n}.txt")[1])
    names = sorted(set(k for k, _ in pm))
    res = {"_note": "per dispatch, averages over the dispatches of tools/run_secondary_kernels.py; durations from the kernel trace of the same program; "
                    "HBM bytes = FETCH_SIZE x 2 + WRITE_SIZE (factors calibrated on known byte counts: profiles/r06_counter_calibration.txt)"}
    for k, (n, avg_us) in kt.items():
        if k.startswith("lldba::") or k.startswith("__amd"): continue
        q = next((x for x in names if x[:48] == k[:48]), None)
        if q is None: continue
        g = lambda c: pm.get((q, c), 0.0)
        t = avg_us * 1e-6
        waves = g("SQ_WAVES")
        simds = 1024 if waves >= 4096 else max(4, 4 * min(256, int(round(waves / 16.0 + 0.49)))) if waves > 16 else 4     # small launches: the CUs their workgroups occupy
        hbm = g("FETCH_SIZE") * 2048 + g("WRITE_SIZE") * 1024
        res[k] = {"dispatches": n, "avg_us": round(avg_us, 2), "waves": int(waves), "simds_available": simds, "valu_insts": int(g("SQ_INSTS_VALU")),
                  "valu_issue_frac": round(g("SQ_INSTS_VALU") * 4 / (simds * t * 2.4e9), 4), "lds_pipe_busy_frac": round(g("SQ_LDS_IDX_ACTIVE") / (simds / 4 * t * 2.4e9), 4),
                  "hbm_bytes": int(hbm), "hbm_TBps": round(hbm / t / 1e12, 4), "hbm_frac_of_8TBps": round(hbm / t / 8e12, 4)}
    if len(sys.argv) > 2: res["_taken_on"] = sys.argv[2]
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
