#!/bin/bash
# Diagnostic: radix bits per pass of the per-field LDS sort (run on the GPU box).
cd "$(dirname "$0")/.." || exit 1
for rb in 4 5 6 8; do
  touch satrans_amd/csrc/embed_adam.hip
  SATRANS_EXTRA_FLAGS="-DSATRANS_SORT_RADIX_BITS=$rb" bash satrans_amd/csrc/build.sh > /dev/null 2>&1
  echo "== radix bits per pass: $rb"
  python -m pytest tests/test_gpu_parity.py -m gpu -q -k per_field_sort 2>&1 | tail -1
  for rep in 1 2; do
    python bench.py --steps 20 --warmup 5 --train-only 2>/dev/null > /tmp/sw.json
    python tools/show_bench.py /tmp/sw.json | grep -E "samples/s|embed_sort" | tr '\n' ' '; echo
  done
done
