#!/usr/bin/env bash
# One GPU: what the data-parallel gradient path costs besides the wire.  bench.py with the machinery rehearsed in a
# 1-rank RCCL group (every bucket all-reduce, overflow check and per-bucket AdamW really issued on the communication
# stream) against the same run without it, for each transport variant.  One JSON line per variant on stdout.
#   bash tools/rehearse_dp_matrix.sh [steps] [warmup] > profiles/rehearsal_dp_1gpu.jsonl
set -euo pipefail
cd "$(dirname "$0")/.."
STEPS=${1:-20}; WARMUP=${2:-5}
run() { # run <env assignments...> -- <bench flags...>
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  echo "# ${envs[*]} bench.py $*"
  env "${envs[@]}" timeout -k 10 240 python bench.py --steps "$STEPS" --warmup "$WARMUP" "$@" | tail -1
}
for batch in 1024 96; do
  run PDT_X=0 -- --batch $batch
  run PDT_X=0 -- --batch $batch --rehearse-dp
  run PDT_X=0 -- --batch $batch --rehearse-dp --gemm-persist 0
  run PDT_MAX_BUCKET_MB=100000 -- --batch $batch --rehearse-dp
  run PDT_X=0 -- --batch $batch --rehearse-dp --comm c10d
  run PDT_X=0 -- --batch $batch --rehearse-dp --comm-algo rs_ag
  run PDT_X=0 -- --batch $batch --rehearse-dp --comm-dtype bf16
  run PDT_X=0 -- --batch $batch --dtype bf16
  run PDT_X=0 -- --batch $batch --dtype bf16 --rehearse-dp
  run PDT_ADAMW_PER_BUCKET=0 -- --batch $batch --dtype bf16 --rehearse-dp
done
