#!/bin/bash
# A/B of two builds of the native extension on the same GPU, interleaved: bench/ab.sh <variant> [rounds] [bench.py args...]
# (variant built with `python build_ext.py --variant <variant> -D...`); prints ms/step of every run.
set -e
v=$1; rounds=${2:-3}; shift; shift || true
cd "$(dirname "$0")/.."
for r in $(seq 1 "$rounds"); do
  a=$(timeout -k 10 300 python bench.py --steps 20 --warmup 5 "$@" | python -c 'import json,sys; print(json.loads(sys.stdin.readlines()[-1])["ms_per_step"])')
  b=$(PDT_NATIVE_LIB=$PWD/pytorch_distributed_training_amd/_C_$v.so timeout -k 10 300 python bench.py --steps 20 --warmup 5 "$@" | python -c 'import json,sys; print(json.loads(sys.stdin.readlines()[-1])["ms_per_step"])')
  echo "round $r: default $a ms   $v $b ms"
done
