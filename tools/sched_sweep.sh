#!/bin/bash
# Diagnostic: the training bench with the library rebuilt under different AMDGPU scheduling strategies (run on the GPU box).
cd "$(dirname "$0")/.." || exit 1
for fl in "" "-mllvm -amdgpu-sched-strategy=max-ilp" "-mllvm -amdgpu-sched-strategy=iterative-ilp" "-mllvm -amdgpu-sched-strategy=max-memory-clause" "-mllvm -amdgpu-schedule-metric-bias=0"; do
  touch satrans_amd/csrc/layer_fused_common.h satrans_amd/csrc/common.h
  SATRANS_EXTRA_FLAGS="$fl" bash satrans_amd/csrc/build.sh > /dev/null 2>&1
  echo "== flags: '$fl'"
  for rep in 1 2; do
    python bench.py --steps 20 --warmup 5 --train-only 2>/dev/null > /tmp/sw.json
    python tools/show_bench.py /tmp/sw.json | grep -E "samples/s|layer_|lazy_flush" | tr '\n' ' '; echo
  done
done
