#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE of tools/microbench/counter_calib against the bytes each pattern must move.
    python tools/counter_calib_summary.py <fetch summary .txt> <write summary .txt>      (summaries of tools/rocpd_summary.py)"""
import re, sys
GiB = float(1 << 30)
n8 = 2 * GiB / 8
# kernel -> (bytes that must be read, bytes that must be written)
KNOWN = {"read_stream16": (2 * GiB, 0), "read_stream8": (2 * GiB, 0), "read_stream4": (2 * GiB, 0),
         "read_gather8_blocks": (2 * GiB + n8 * 4, 0),                          # every 128-B block of doubles once + the index list
         "read_gather8_random": ((n8 / 8) * 8 + (n8 / 8) * 4, 0),               # USEFUL bytes only: 8 B of each fetched sector + its index
         "write_stream16": (0, 2 * GiB), "write_stream8": (0, 2 * GiB), "write_pairs16_blocks": ((2 * GiB / 16) * 4, 2 * GiB)}


def per_dispatch(path, counter):
    out = {}
    for line in open(path):
        m = re.match(r"^(.*?)\s+" + counter + r"\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s*$", line.rstrip())
        if m: out[m.group(1).strip()] = float(m.group(4))
    return out


f = per_dispatch(sys.argv[1], "FETCH_SIZE"); w = per_dispatch(sys.argv[2], "WRITE_SIZE")
print("# rocprofv3 FETCH_SIZE / WRITE_SIZE (KiB per dispatch) of tools/microbench/counter_calib against the bytes the pattern must move; 2 GiB per kernel (8x the Infinity Cache)")
print(f"{'kernel':26s} {'FETCH KiB':>12s} {'read bytes':>14s} {'bytes / FETCH':>14s}   {'WRITE KiB':>12s} {'write bytes':>14s} {'bytes / WRITE':>14s}")
for k, (rb, wb) in KNOWN.items():
    fk = next((v for n, v in f.items() if k in n), 0.0); wk = next((v for n, v in w.items() if k in n), 0.0)
    rf = rb / (fk * 1024) if fk > 0 and rb > 0 else float("nan"); wf = wb / (wk * 1024) if wk > 0 and wb > 0 else float("nan")
    print(f"{k:26s} {fk:12.0f} {rb:14.0f} {rf:14.3f}   {wk:12.0f} {wb:14.0f} {wf:14.3f}")
print("# bytes / FETCH = the factor a FETCH_SIZE reading must be multiplied by to give bytes (2.0: the guide's figure for 16-B streams); for")
print("# read_gather8_random the 'bytes' are the USEFUL ones, so 1 / factor is the inflation of a fully random 8-byte gather")
