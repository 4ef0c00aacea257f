#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel: MFMA utilisation, LDS bank-conflict rate, wait share.

  MFMA util       = SQ_VALU_MFMA_BUSY_CYCLES / (128 * GRBM_GUI_ACTIVE) (matrix-pipe busy cycles summed over the 1024 SIMDs,
                                                                        over the kernel's cycles: GRBM_GUI_ACTIVE is summed
                                                                        over the 8 XCDs, so cycles * 1024 = GUI_ACTIVE * 128).
                                                                        Calibration: a GEMM running at 1000 TFLOP/s of the
                                                                        2500 TFLOP/s dense fp16 peak reads 40 %.
  LDS conflicts   = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE           (extra LDS cycles per active LDS cycle)
  wait share      = SQ_WAIT_ANY / SQ_WAVE_CYCLES                       (wave-cycles parked on s_waitcnt / barrier)
"""
import csv
import glob
import sys
from collections import defaultdict


def main(path, out=None):
    f = glob.glob(path + "/**/*_counter_collection.csv", recursive=True)[0]
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k].add(r["Dispatch_Id"])
    rows = []
    for k, c in agg.items():
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        rows.append((c.get("SQ_WAVE_CYCLES", 0.0), k, len(calls[k]),
                     c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (128 * gui) if gui else float("nan"),
                     c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"] if c.get("SQ_LDS_IDX_ACTIVE") else 0.0,
                     c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else 0.0,
                     c.get("SQ_ACTIVE_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else 0.0))
    rows.sort(reverse=True)
    lines = [f"# rocprofv3 --pmc  ({f})", __doc__.split("\n\n", 1)[1].rstrip(), "",
             f"{'calls':>6} {'MFMA util':>10} {'LDS confl':>10} {'wait':>7} {'issue':>7}  kernel"]
    for wc, k, n, mf, lds, wait, act in rows[:24]:
        name = k.replace("pdt::", "")[:96]
        lines.append(f"{n:6d} {100 * mf:9.1f}% {100 * lds:9.2f}% {100 * wait:6.1f}% {100 * act:6.1f}%  {name}")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
