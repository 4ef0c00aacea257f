#!/usr/bin/env python3
"""HBM traffic per kernel from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, both in KiB; they do not fit into
one pass: 3 + 2 of the 4 TCC counters) joined with the kernel durations of a --kernel-trace pass.

    python tools/hbm_summary.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <kernel_trace.csv> [out.txt]

Per kernel name: calls, mean duration (us, from the trace pass: counter collection slows kernels down), GB read and
written per call, and the rate that traffic means at the traced duration.

Calibration: on this stack (ROCm 7.2, gfx950) FETCH_SIZE reports HALF of the bytes kernels of known traffic read - the
LayerNorm forward reads one 16-bit [tokens, H] tensor and writes one (0.134 GB reported against 0.268 GB read, 0.269 GB
written), AdamW reads four fp32 arrays and writes three plus the 16-bit copy (2.67 GB reported against 5.36 GB read,
4.67 GB written as expected) - so the read column is the counter times FETCH_SCALE = 2.
"""
import csv
import glob
import sys
from collections import defaultdict


def key_of(r):
    """A call site: kernel name + grid size (tells the GEMM shapes of one kernel apart when they are launched with one
    workgroup per tile)."""
    return r["Kernel_Name"], int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0)


def counters(path, name):
    f = glob.glob(path + "/**/*_counter_collection.csv", recursive=True)[0]
    tot, calls = defaultdict(float), defaultdict(set)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name:
            tot[key_of(r)] += float(r["Counter_Value"])
            calls[key_of(r)].add(r["Dispatch_Id"])
    return {k: tot[k] / max(1, len(calls[k])) for k in tot}


FETCH_SCALE = 2.0


def main(fetch_dir, write_dir, trace, out=None):
    rd, wr = counters(fetch_dir, "FETCH_SIZE"), counters(write_dir, "WRITE_SIZE")
    rd = {k: v * FETCH_SCALE for k, v in rd.items()}
    dur, n = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(trace)):
        dur[key_of(r)] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        n[key_of(r)] += 1
    rows = []
    for k in dur:
        if k in rd or k in wr:
            us = dur[k] / n[k] / 1e3
            gb_r, gb_w = rd.get(k, 0.0) * 1024 / 1e9, wr.get(k, 0.0) * 1024 / 1e9
            rows.append((dur[k], k, n[k], us, gb_r, gb_w, (gb_r + gb_w) / (us * 1e-6) / 1e3 if us > 0 else 0.0))
    rows.sort(reverse=True)
    lines = ["# HBM traffic per kernel: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (two passes) + kernel trace durations",
             f"# (reads = FETCH_SIZE x {FETCH_SCALE:g}: the counter reports half the bytes of kernels with known traffic, see tools/hbm_summary.py)", "",
             f"{'calls':>6} {'avg us':>9} {'GB read':>9} {'GB written':>11} {'TB/s':>6} {'threads':>9}  kernel"]
    for _, k, c, us, r, w, rate in rows[:24]:
        lines.append(f"{c:6d} {us:9.1f} {r:9.3f} {w:11.3f} {rate:6.2f} {k[1]:9d}  {k[0].replace('pdt::', '')[:100]}")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else None)
