#!/bin/bash
# Run the GPU kernel tests group by group, each in its own process with its own time limit.
# A group that only has assertion failures (pytest exit 1) lets the next group run; anything else
# (abort, segfault, timeout: the GPU may be faulted) stops the script there.
mkdir -p gpurun_out
for grp in "gemm_nt" "gemm_nn" "gemm_tn or gemm_dropout" "layernorm" "attention" "embeddings or head or adamw"; do
  name=$(echo "$grp" | tr ' ' '_')
  timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py -q -k "$grp" --timeout 120 --timeout-method thread \
      > "gpurun_out/kt_${name}.log" 2>&1
  rc=$?
  echo "group [$grp] exit=$rc: $(tail -1 gpurun_out/kt_${name}.log)"
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "stopping: abnormal exit"; exit $rc; fi
done
# the remaining GPU test files, one process each (model path, prefetching loader, multi-process rehearsals)
for f in tests/test_model_gpu.py tests/test_loader_gpu.py tests/test_distributed_gpu.py; do
  name=$(basename "$f" .py)
  timeout -k 10 600 python -u -m pytest "$f" -q --timeout 300 --timeout-method thread > "gpurun_out/kt_${name}.log" 2>&1
  rc=$?
  echo "file [$f] exit=$rc: $(tail -1 gpurun_out/kt_${name}.log)"
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "stopping: abnormal exit"; exit $rc; fi
done
