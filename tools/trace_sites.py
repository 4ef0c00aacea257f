#!/usr/bin/env python3
"""Per-call-site view of a rocprofv3 kernel trace: one optimizer step, grouped by (kernel, grid, block).

    python tools/trace_sites.py gpurun_out/prof/vm/*_kernel_trace.csv [--step -2]

A step is the run of dispatches between two consecutive AdamW kernels.  GEMM call sites are told apart by
their grid size (tiles x split-K slices), which is what makes this more useful than `--stats` for finding
the slow shapes.  Also prints the wall time of the step and the sum of kernel times (the difference is idle
GPU time: launch gaps).
"""
from __future__ import annotations

import argparse
import csv
import re
from collections import OrderedDict


def short(name: str) -> str:
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"^void ", "", name)
    name = name.replace("pdt::", "").replace("(anonymous namespace)::", "")
    return name[:110]


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--step", type=int, default=-2, help="which step to show (index into the list of steps)")
    ap.add_argument("--order", action="store_true", help="print the dispatches of the step in launch order")
    ap.add_argument("--gaps", type=int, default=0,
                    help="also list the N longest spans of the step in which no kernel ran, with the kernels around them, "
                         "and the busy time per hardware queue")
    args = ap.parse_args()
    rows = []
    with open(args.trace) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                         int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]),
                         int(r["Workgroup_Size_X"]), int(r["LDS_Block_Size"]), int(r["VGPR_Count"]),
                         int(r["Accum_VGPR_Count"]), r.get("Queue_Id", "?")))
    rows.sort()
    cuts = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
    if len(cuts) < 3:
        raise SystemExit("need at least 3 optimizer steps in the trace")
    steps = [rows[cuts[i] + 1: cuts[i + 1] + 1] for i in range(len(cuts) - 1)]
    st = steps[args.step]
    wall = (st[-1][1] - st[0][0]) / 1e6
    busy = sum(r[1] - r[0] for r in st) / 1e6
    print(f"step with {len(st)} dispatches: wall {wall:.3f} ms, kernel time {busy:.3f} ms, idle {wall - busy:.3f} ms")
    if args.gaps:
        queues: "OrderedDict[str, list]" = OrderedDict()
        for r in st:
            queues.setdefault(r[8], []).append(r)
        for q, rs in queues.items():
            print(f"queue {q}: {len(rs)} dispatches, kernel time {sum(r[1] - r[0] for r in rs) / 1e6:.3f} ms")
        gaps, end, last = [], st[0][1], st[0]
        for r in st[1:]:
            if r[0] > end:
                gaps.append((r[0] - end, last, r))
            if r[1] > end:
                end, last = r[1], r
        print(f"{len(gaps)} idle spans, {sum(g[0] for g in gaps) / 1e6:.3f} ms in total; the longest:")
        for g in sorted(gaps, key=lambda g: -g[0])[:args.gaps]:
            print(f"{g[0] / 1e3:9.1f} us after {short(g[1][2])[:60]} (queue {g[1][8]}) before {short(g[2][2])[:60]} (queue {g[2][8]})")
    if args.order:
        for r in st:
            print(f"{(r[1] - r[0]) / 1e3:9.1f} us  grid {r[3] // r[4]:6d} x {r[4]:4d}  lds {r[5]:6d}  {short(r[2])}")
        return
    sites: "OrderedDict[tuple, list]" = OrderedDict()
    for r in st:
        sites.setdefault((short(r[2]), r[3] // r[4], r[4], r[5], r[6] + r[7]), []).append(r[1] - r[0])
    print(f"{'total ms':>9} {'calls':>6} {'avg us':>9}  {'blocks':>7} {'thr':>4} {'lds':>6} {'regs':>4}  kernel")
    for k, v in sorted(sites.items(), key=lambda kv: -sum(kv[1])):
        print(f"{sum(v) / 1e6:9.3f} {len(v):6d} {sum(v) / len(v) / 1e3:9.1f}  {k[1]:7d} {k[2]:4d} {k[3]:6d} {k[4]:4d}  {k[0]}")


if __name__ == "__main__":
    main()
