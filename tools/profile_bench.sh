#!/bin/bash
# Kernel statistics, call-site view and PMC summary of one bench.py configuration.
#   bash tools/profile_bench.sh <name> <timed steps> [bench.py flags...]
# writes gpurun_out/<name>_{kernel_stats,call_sites,pmc}.txt; each rocprofv3 pass is its own bounded run.
set -u
# (a multi-rank bench.py would have to start its ranks from a process the profiler has already attached to the GPU)
for a in "$@"; do
  if [ "${prev:-}" = "--gpus" ] && [ "$a" != "1" ]; then echo "this script profiles single-GPU runs only (--gpus 1)"; exit 2; fi
  case "$a" in --gpus=1) ;; --gpus=*) echo "this script profiles single-GPU runs only (--gpus 1)"; exit 2;; esac
  prev=$a
done
name=$1; steps=$2; shift 2
export TMPDIR=/tmp
root=$(pwd)
out=$root/gpurun_out
mkdir -p "$out"
rm -rf "$out/prof_$name" "$out/pmc_$name"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_$name" -- \
    python3 bench.py --steps "$steps" --warmup 2 "$@" > "$out/${name}_trace_run.log" 2>&1 || { echo "trace pass failed"; tail -5 "$out/${name}_trace_run.log"; exit 1; }
python3 tools/prof_summary.py "$out/prof_$name" $((steps + 2)) "$out/${name}_kernel_stats.txt" > /dev/null
python3 tools/trace_sites.py "$(ls $out/prof_$name/*/*_kernel_trace.csv | head -1)" > "$out/${name}_call_sites.txt"
timeout -s KILL 400 rocprofv3 --kernel-trace --output-format csv \
    --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY \
    -d "$out/pmc_$name" -- python3 bench.py --steps 1 --warmup 1 "$@" > "$out/${name}_pmc_run.log" 2>&1 || { echo "pmc pass failed"; tail -5 "$out/${name}_pmc_run.log"; exit 1; }
python3 tools/pmc_summary.py "$out/pmc_$name" "$out/${name}_pmc.txt" > /dev/null
# the raw traces are large: keep the summaries only
rm -rf "$out/prof_$name" "$out/pmc_$name"
echo "profiled $name"
