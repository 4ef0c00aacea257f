set -u
for b in 96 1024; do
for c in 1 0; do
  echo "# PDT_BUCKET_OVERFLOW_CHECK=$c --batch $b --rehearse-dp"
  PDT_BUCKET_OVERFLOW_CHECK=$c timeout -k 10 200 python bench.py --steps 30 --warmup 5 --batch $b --rehearse-dp | tail -1 | cut -c1-200 || exit 1
done
done
echo "# --batch 96 --rehearse-dp --comm c10d"
timeout -k 10 200 python bench.py --steps 30 --warmup 5 --batch 96 --rehearse-dp --comm c10d | tail -1 | cut -c1-200
