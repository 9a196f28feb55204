#!/bin/bash
# What the phases of layer_fwd_bf16_kernel cost, by REMOVAL (as for the fp32 backward: library variants whose attention / MetaNet
# bodies never run - tools/experiments/build_variant.sh with a sed that adds an always-false runtime condition), alternating with
# the shipped library on one box: ms per 32,768-sample evaluation forward (3 layer launches + gather + head).
cd "$(dirname "$0")/../.." || exit 1
for round in 1 2; do
  for lib in "" tools/experiments/_variants/lib_bf16_*.so; do
    r=$(SATRANS_LIB_PATH=${lib:+$PWD/$lib} python bench.py --cpu-steps 0 --no-other-configs --sustained-steps 0 --fit-batches 0 --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['forward_only_bf16']['ms_per_batch'], d['forward_only']['ms_per_batch'])")
    echo "round $round ${lib:-shipped}: bf16 / fp32 forward ms per 32768: $r"
  done
done
