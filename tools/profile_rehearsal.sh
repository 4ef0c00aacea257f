#!/bin/bash
# Call-site view (tools/trace_sites.py) of the 1-rank data-parallel rehearsal at the reference's batch, per transport.
#   bash tools/profile_rehearsal.sh   -> gpurun_out/rehearse96_<variant>_call_sites.txt
set -u
export TMPDIR=/tmp
root=$(pwd); out=$root/gpurun_out; mkdir -p "$out"
for v in ${VARIANTS:-none rccl c10d}; do
  flags="--batch 96"
  [ "$v" = rccl ] && flags="$flags --rehearse-dp"
  [ "$v" = c10d ] && flags="$flags --rehearse-dp --comm c10d"
  rm -rf "$out/prof_r96_$v"
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$out/prof_r96_$v" -- \
      python3 bench.py --steps 6 --warmup 2 $flags > "$out/rehearse96_${v}_run.log" 2>&1 || { echo "trace pass $v failed"; tail -5 "$out/rehearse96_${v}_run.log"; exit 1; }
  python3 tools/trace_sites.py "$(ls $out/prof_r96_$v/*/*_kernel_trace.csv | head -1)" --gaps 12 > "$out/rehearse96_${v}_call_sites.txt"
  rm -rf "$out/prof_r96_$v"
  grep "^step\|^queue\|idle spans" "$out/rehearse96_${v}_call_sites.txt"
done
