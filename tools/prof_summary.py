#!/usr/bin/env python3
"""Condense a rocprofv3 kernel_stats.csv into a short per-kernel table (ms per step, share)."""
import csv
import glob
import sys


def main(path, steps, out=None):
    f = glob.glob(path + "/**/*_kernel_stats.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    lines = [f"# rocprofv3 --kernel-trace --stats  ({f})", f"total kernel time {tot / 1e6:.2f} ms over {steps} steps = {tot / 1e6 / steps:.2f} ms/step", "",
             f"{'ms/step':>9} {'share':>7} {'calls/step':>10} {'avg us':>9}  kernel"]
    for r in rows[:30]:
        n = r["Name"].replace("pdt::", "")
        for a, b in (("_ZN3pdt", ""), ("IDF16_", "<f16,"), ("IDF16b", "<bf16,")):
            n = n.replace(a, b)
        lines.append(f"{float(r['TotalDurationNs']) / 1e6 / steps:9.3f} {float(r['Percentage']):6.2f}% {int(r['Calls']) / steps:10.1f} {float(r['AverageNs']) / 1e3:9.1f}  {n[:100]}")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), sys.argv[3] if len(sys.argv) > 3 else None)
