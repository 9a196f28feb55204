#!/bin/bash
# A/B on ONE box: bench --train-only alternating between the shipped library and every variant, ROUNDS times (default 3);
# prints ms/step and the phases named in PHASES (default: the layer and optimizer phases).  Box-to-box noise is +-2 %: only
# differences inside one run of this script mean anything.
cd "$(dirname "$0")/../.." || exit 1
for round in $(seq 1 ${ROUNDS:-3}); do
  for lib in "" tools/experiments/_variants/lib_*.so; do
    [ -n "$lib" ] && [ ! -f "$lib" ] && continue
    r=$(SATRANS_LIB_PATH=${lib:+$PWD/$lib} python bench.py ${BENCH_ARGS:-} --steps ${STEPS:-40} --warmup 5 --train-only 2>/dev/null | tail -1 | python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); k=d['kernels']
names=os.environ.get('PHASES','layer_fwd,layer_bwd,layer_bwd_head,adam_touched,lazy_flush,lazy_replay,layer_bwd_reduce').split(',')
print(d['ms_per_step'], ' '.join(f\"{n}={k[n]['ms_per_launch']}\" for n in names if n in k))")
    echo "round $round ${lib:-shipped}: $r"
  done
done
