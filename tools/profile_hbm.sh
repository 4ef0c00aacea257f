#!/bin/bash
# HBM bytes per kernel of one bench.py configuration: bash tools/profile_hbm.sh <name> [bench.py flags...]
# -> gpurun_out/<name>_hbm.txt  (three bounded rocprofv3 runs: kernel trace, FETCH_SIZE, WRITE_SIZE)
set -u
# (a multi-rank bench.py would have to start its ranks from a process the profiler has already attached to the GPU)
for a in "$@"; do
  if [ "${prev:-}" = "--gpus" ] && [ "$a" != "1" ]; then echo "this script profiles single-GPU runs only (--gpus 1)"; exit 2; fi
  case "$a" in --gpus=1) ;; --gpus=*) echo "this script profiles single-GPU runs only (--gpus 1)"; exit 2;; esac
  prev=$a
done
name=$1; shift
export TMPDIR=/tmp
out=$(pwd)/gpurun_out
mkdir -p "$out"
rm -rf "$out/hbm_t_$name" "$out/hbm_r_$name" "$out/hbm_w_$name"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$out/hbm_t_$name" -- python3 bench.py --steps 2 --warmup 1 "$@" > "$out/${name}_hbm_run.log" 2>&1 || { echo "trace pass failed"; exit 1; }
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d "$out/hbm_r_$name" -- python3 bench.py --steps 1 --warmup 1 "$@" >> "$out/${name}_hbm_run.log" 2>&1 || { echo "FETCH_SIZE pass failed"; exit 1; }
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d "$out/hbm_w_$name" -- python3 bench.py --steps 1 --warmup 1 "$@" >> "$out/${name}_hbm_run.log" 2>&1 || { echo "WRITE_SIZE pass failed"; exit 1; }
python3 tools/hbm_summary.py "$out/hbm_r_$name" "$out/hbm_w_$name" "$(ls $out/hbm_t_$name/*/*_kernel_trace.csv | head -1)" "$out/${name}_hbm.txt" > /dev/null
rm -rf "$out/hbm_t_$name" "$out/hbm_r_$name" "$out/hbm_w_$name"
echo "hbm profile $name done"
