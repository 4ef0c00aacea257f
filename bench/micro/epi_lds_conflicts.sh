#!/bin/bash
# LDS bank conflicts of the ping-pong GEMM per epilogue kind, straight-line epilogue (fast=1) against the general one (0):
# one bounded rocprofv3 counter pass per run of the stamped harness binaries in bench/micro/bin.
export TMPDIR=/tmp
root=$(cd "$(dirname "$0")/../.." && pwd); out=$root/gpurun_out; mkdir -p "$out"
cd "$root/bench/micro/bin" || exit 1
for bin in ${BINS:-g8_e0 g8_e2 g8_e3}; do
  for fast in 1 0; do
    d=$out/pmc_epi_${bin}_$fast; rm -rf "$d"
    timeout -s KILL 90 rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS \
        -d "$d" -- ./$bin ${SHAPE:-32768 4096 1024} 256 $fast > "$out/pmc_epi_run.log" 2>&1 || { echo "pass $bin $fast failed"; tail -5 "$out/pmc_epi_run.log"; exit 1; }
    python3 - "$d" "$bin fast=$fast" <<'PY'
import csv, glob, sys, collections
d, label = sys.argv[1], sys.argv[2]
tot = collections.Counter()
for f in glob.glob(d + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "gemm8" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
a = tot.get("SQ_LDS_IDX_ACTIVE", 0.0)
print(f"{label:16s} LDS active {a:.3e}  bank conflicts {tot.get('SQ_LDS_BANK_CONFLICT', 0):.3e} ({100 * tot.get('SQ_LDS_BANK_CONFLICT', 0) / max(a, 1):.2f} %)  addr conflicts {tot.get('SQ_LDS_ADDR_CONFLICT', 0):.3e}  LDS insts {tot.get('SQ_INSTS_LDS', 0):.3e}")
PY
    rm -rf "$d"
  done
done
