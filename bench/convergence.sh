#!/usr/bin/env bash
# Convergence evidence on one GPU: a BERT-base-shaped model trained from random init with the native fp16 path on the
# synthetic paraphrase rule (data/synthetic.py) until it generalises to held-out pairs.
#   bash bench/convergence.sh <tag> [extra train_dp.py flags...]
set -euo pipefail
cd "$(dirname "$0")/.."
TAG=$1; shift
mkdir -p gpurun_out
python -u train_dp.py --model bert-base-cased --fp16=True --train-examples 16384 --eval-examples 1024 \
  --batch-size 128 --max-gpu-batch-size 128 --num-warmup-steps 100 --log-every 20 --seed 42 "$@" \
  2>&1 | grep -v -E "amdgpu.ids|hostname of the client" | tee gpurun_out/convergence_$TAG.log
