#!/usr/bin/env python3
"""Median device time per (kernel, grid) of a rocprofv3 kernel trace: python tools/gemm_trace_summary.py <dir>"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[(r["Kernel_Name"][:72], r["Grid_Size_X"], r["Workgroup_Size_X"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(d.items()):
    v.sort()
    print(f"{len(v):5d} calls  median {v[len(v) // 2] / 1000:8.1f} us  min {v[0] / 1000:8.1f}  wgs {int(k[1]) // int(k[2]):>5} x {k[2]:>4}  {k[0]}")
