#!/usr/bin/env bash
# 1/2/4/8-GPU weak-scaling sweep of the headline benchmark on ONE node (one process per GPU, RCCL over xGMI).
# Prints one JSON line per GPU count and the scaling efficiency relative to the 1-GPU run.
#   bash bench/scale_sweep.sh [steps] [warmup] [extra bench.py flags...]
# With PDT_SWEEP_MP=1 the data-parallel x 2-stage-pipeline layout follows on all visible GPUs: bench.py --pp 2
# (BERT-large seq 128 fp16, RoBERTa-large seq 512 bf16) and python train_mp.py --fp16=True --max-steps 20.
# The reference's own batch follows for 1 GPU and all GPUs (96 samples per process and optimizer step,
# /root/reference/test_data_parallelism.py:174, as one micro-batch and as the literal 12 x 8 accumulation layout): a step
# of ~25 ms instead of ~260 ms, the configuration where bucket sizes, wire dtype and where the optimizer starts decide the
# scaling efficiency (PDT_SWEEP_REF=0 skips it).
# With PDT_SWEEP_COMM=1 a gradient-transport matrix follows on all visible GPUs (communicator, bucket algorithm,
# wire dtype, persistent GEMMs, RCCL channel count).
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
if [ "${PDT_SWEEP_REF:-1}" = "1" ] && [ "$NGPU" -ge 2 ]; then
  for cfg in "--batch 96" "--batch 96 --accum 12" "--batch 96 --comm-dtype bf16"; do
    echo "# reference batch: $cfg"
    one=$(python bench.py --gpus 1 --steps "$STEPS" --warmup "$WARMUP" ${cfg/--comm-dtype bf16/} "$@")
    all=$(python -m torch.distributed.run --nnodes=1 --nproc-per-node "$NGPU" --master-addr 127.0.0.1 --master-port 29640 \
          bench.py --gpus "$NGPU" --steps "$STEPS" --warmup "$WARMUP" $cfg "$@" | tail -1)
    echo "$one"; echo "$all"
    python -c 'import json,sys; a,b,n=json.loads(sys.argv[1]),json.loads(sys.argv[2]),int(sys.argv[3]); print(f"# {n} GPUs: {b[\"value\"]:.1f} vs {a[\"value\"]:.1f} samples/s on one: scaling efficiency {100*b[\"value\"]/(n*a[\"value\"]):.1f} %, comm exposed {b[\"comm_exposed_ms_per_step\"]} ms of {b[\"ms_per_step\"]} ms")' "$one" "$all" "$NGPU"
  done
fi
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
if [ "${PDT_SWEEP_COMM:-0}" = "1" ] && [ "$NGPU" -ge 2 ]; then
  # gradient-transport matrix on all GPUs: who drives RCCL, how a bucket is reduced, wire dtype, persistent GEMMs
  # next to the communication kernels, RCCL channel count.  Compare "value" and "comm_exposed_ms_per_step".
  # (default batch: the persistent launches only exist from 1024 output tiles per GEMM, i.e. not at 96 samples per GPU)
  run() { # run <label> <env assignments...> -- <bench flags...>
    local label=$1; shift; local envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
    echo "# $label"
    env "${envs[@]}" python -m torch.distributed.run --nnodes=1 --nproc-per-node "$NGPU" --master-addr 127.0.0.1 \
      --master-port 29670 bench.py --gpus "$NGPU" --steps "$STEPS" --warmup "$WARMUP" "$@" | tail -1
  }
  run "own communicator, all-reduce buckets (default)" PDT_X=0 --
  run "torch process group" PDT_X=0 -- --comm c10d
  run "reduce-scatter + all-gather buckets" PDT_X=0 -- --comm-algo rs_ag
  run "bf16 on the wire" PDT_X=0 -- --comm-dtype bf16
  run "one workgroup per GEMM tile instead of persistent launches next to the collectives" PDT_X=0 -- --gemm-persist 0
  run "AdamW per bucket as the buckets land (bf16)" PDT_X=0 -- --dtype bf16
  run "AdamW after the last bucket (bf16)" PDT_ADAMW_PER_BUCKET=0 -- --dtype bf16
  for ch in 8 16 32; do run "RCCL channels pinned to $ch" NCCL_MIN_NCHANNELS=$ch NCCL_MAX_NCHANNELS=$ch --; done
fi
