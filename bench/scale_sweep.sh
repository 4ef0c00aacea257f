#!/usr/bin/env bash
# 1/2/4/8-GPU weak-scaling sweep of the headline benchmark on ONE node (one process per GPU, RCCL over xGMI).
# Prints one JSON line per GPU count and the scaling efficiency relative to the 1-GPU run.
#   bash bench/scale_sweep.sh [steps] [warmup] [extra bench.py flags...]
# With PDT_SWEEP_MP=1 the data-parallel x 2-stage-pipeline layout follows on all visible GPUs: bench.py --pp 2
# (BERT-large seq 128 fp16, RoBERTa-large seq 512 bf16) and python train_mp.py --fp16=True --max-steps 20.
set -euo pipefail
cd "$(dirname "$0")/.."
STEPS=${1:-20}; WARMUP=${2:-5}; shift $(( $# > 2 ? 2 : $# )) || true
NGPU=$(python -c 'import torch; print(torch.cuda.device_count())')
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
base=""
for n in 1 2 4 8; do
  [ "$n" -le "$NGPU" ] || break
  if [ "$n" -eq 1 ]; then
    line=$(python bench.py --gpus 1 --steps "$STEPS" --warmup "$WARMUP" "$@")
  else
    line=$(python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 \
           --master-port $((29600 + n)) bench.py --gpus "$n" --steps "$STEPS" --warmup "$WARMUP" "$@" | tail -1)
  fi
  echo "$line"
  v=$(python -c 'import json,sys; print(json.loads(sys.argv[1])["value"])' "$line")
  [ -n "$base" ] || base=$v
  python -c 'import sys; n,v,b=int(sys.argv[1]),float(sys.argv[2]),float(sys.argv[3]); print(f"# {n} GPU(s): {v:.1f} samples/s, scaling efficiency {100*v/(n*b):.1f} %")' "$n" "$v" "$base"
done
if [ "${PDT_SWEEP_MP:-0}" = "1" ] && [ "$NGPU" -ge 2 ]; then
  # data parallel x 2-stage pipeline on all GPUs (GPU pairs form replicas): throughput of the same model, then the
  # seq-512 RoBERTa-large configuration, then the training recipe (loss curve)
  n=$(( NGPU / 2 * 2 ))
  for extra in "" "--model roberta-large --seq-len 512 --dtype bf16"; do
    python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port 29650 \
      bench.py --gpus "$n" --pp 2 --micro-batches 8 --steps "$STEPS" --warmup "$WARMUP" $extra | tail -1
  done
  python train_mp.py --fp16=True --max-steps 20 --log-every 1
fi
