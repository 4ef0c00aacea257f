#!/usr/bin/env python3
"""Idle gaps of the GPU in a rocprofv3 kernel trace: for one optimizer step (between the last two AdamW launches)
every interval longer than a threshold during which no kernel ran, with the kernels before and after it, and the
position of the gap inside the step.

    python tools/trace_gaps.py <..._kernel_trace.csv> [min_gap_us=30]"""
import csv
import sys


def main(path, min_gap_us=30.0):
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))]
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
    seg = rows[marks[-2]:marks[-1] + 1]
    t0 = seg[0][0]
    wall = seg[-1][1] - t0
    end, last, gaps, idle = seg[0][1], seg[0][2], [], 0
    for s, e, n in seg[1:]:
        if s > end:
            idle += s - end
            if (s - end) / 1e3 >= min_gap_us:
                gaps.append(((end - t0) / 1e6, (s - end) / 1e3, last, n))
        if e > end:
            end, last = e, n
    print(f"step wall {wall / 1e6:.2f} ms, idle {idle / 1e6:.2f} ms in total; gaps >= {min_gap_us} us:")
    short = lambda n: n.split("(")[0][-48:]
    for at, g, a, b in gaps:
        print(f"  at {at:7.2f} ms: {g:7.1f} us   after {short(a)}   before {short(b)}")
    print(f"  ({len(gaps)} gaps, {sum(g for _, g, _, _ in gaps) / 1e3:.2f} ms)")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 30.0)
