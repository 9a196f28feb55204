#!/bin/bash
# bench --train-only with every library variant under tools/experiments/_variants (kernel experiments; SATRANS_LIB_PATH)
cd "$(dirname "$0")/../.." || exit 1
for lib in "" tools/experiments/_variants/lib_*.so; do
  for m in f32; do
    r=$(SATRANS_LIB_PATH=${lib:+$PWD/$lib} python bench.py --steps 20 --warmup 5 --train-only 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']; print(d['ms_per_step'], 'fwd', k['layer_fwd']['ms_per_launch'], 'bwd', k['layer_bwd']['ms_per_launch'])")
    echo "${lib:-shipped} [$m]: $r"
  done
done
