set -u
for q in 4 8 16; do
  for flags in "--batch 96" "--batch 96 --rehearse-dp" "--batch 96 --rehearse-dp --comm c10d" "--batch 96 --accum 12" "--batch 1024"; do
    echo "# GPU_MAX_HW_QUEUES=$q $flags"
    GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python bench.py --steps 20 --warmup 5 $flags | tail -1 | cut -c1-200 || exit 1
  done
done
