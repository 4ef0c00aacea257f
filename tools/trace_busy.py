#!/usr/bin/env python3
"""GPU occupancy over time from a rocprofv3 kernel trace: per optimizer step (dispatches between two AdamW kernels)
the wall time, the sum of kernel durations, the time during which at least one kernel was running (union) and during
which at least two were (overlap of streams).  wall - union is time the GPU sat idle waiting for the host.

    python tools/trace_busy.py <..._kernel_trace.csv>"""
import csv
import sys


def main(path):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
    for a, b in zip(marks[:-1], marks[1:]):
        seg = rows[a + 1:b + 1]
        t0, t1 = seg[0][0], max(e for _, e, _ in seg)
        ev = sorted([(s, 1) for s, _, _ in seg] + [(e, -1) for _, e, _ in seg])
        depth, last, union, multi = 0, t0, 0, 0
        for t, d in ev:
            if depth >= 1:
                union += t - last
            if depth >= 2:
                multi += t - last
            depth += d
            last = t
        print(f"step: {len(seg)} dispatches, wall {(t1 - t0) / 1e6:.2f} ms, kernel sum {sum(e - s for s, e, _ in seg) / 1e6:.2f} ms, "
              f"busy {union / 1e6:.2f} ms ({100 * union / (t1 - t0):.1f} %), >=2 kernels {multi / 1e6:.2f} ms")


if __name__ == "__main__":
    main(sys.argv[1])
