#!/bin/bash
# The README performance table, measured back to back on one GPU: bash bench/perf_table.sh > table.jsonl
cd "$(dirname "$0")/.."
run() { timeout -k 10 600 python bench.py --steps 15 --warmup 4 "$@" | tail -1 | python -c '
import json,sys
d=json.loads(sys.stdin.read()); c=d["config"]
print(json.dumps({"args": sys.argv[1:], "samples_per_s": d["value"], "ms_per_step": d["ms_per_step"], "dtype": d["dtype"], "batch": c["per_gpu_batch"], "micro_batches": c["micro_batches"], "seq": c["seq_len"], "peak_hbm_gb": c["peak_hbm_gb"]}))' -- "$@" || exit 1; }
run
run
run --batch 512
run --batch 256
run --batch 128
run --dtype bf16
run --dtype bf16 --seq-len 512
run --batch 96 --accum 12
PDT_OVERLAP_MICRO=0 run --batch 96 --accum 12
run --batch 8
run --model roberta-large --dtype bf16 --seq-len 512
for b in 1024 256; do
  timeout -k 10 600 python bench/eager_baseline.py --hf --sdpa --batch $b --steps 6 --warmup 2 2>/dev/null | tail -1 || exit 1
  timeout -k 10 600 python bench/eager_baseline.py --hf --batch $b --steps 6 --warmup 2 2>/dev/null | tail -1 || exit 1
done
timeout -k 10 600 python bench/eager_baseline.py --hf --sdpa --batch 8 --steps 20 --warmup 5 2>/dev/null | tail -1
