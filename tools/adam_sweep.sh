#!/bin/bash
# Sweep of the streaming-Adam kernel variants (serial mode so that the kernel runs alone).
for v in 0 4 5 6 7; do
  for b in 256 384 512 768; do
    SATRANS_OVERLAP=0 SATRANS_ADAM_VARIANT=$v SATRANS_ADAM_BLOCKS=$b python bench.py --steps 10 --warmup 3 --cpu-steps 0 2>/dev/null | \
      python -c "import json,sys; d=json.load(sys.stdin); k=d['kernels']['adam_untouched']; print('variant $v blocks $b:', k['ms_per_launch'], 'ms', k['achieved'], 'GB/s', d['ms_per_step'])"
  done
done
