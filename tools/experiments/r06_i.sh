#!/bin/bash
# round 6, run I: GPU suite (register-direct hand-over, one-launch reductions), A/B against the previous build, the configs[4] line
cd "$(dirname "$0")/../.." || exit 1
o=gpurun_out/r06; mkdir -p $o
python -m pytest tests -m gpu -x -q > $o/pytest_j.log 2>&1; tail -3 $o/pytest_j.log
ROUNDS=3 PHASES=layer_fwd,layer_bwd,layer_bwd_head bash tools/experiments/ab.sh | tee $o/ab_regh.txt
BENCH_ARGS="--config alimama" ROUNDS=2 PHASES=layer_fwd,layer_bwd,layer_bwd_head bash tools/experiments/ab.sh | tee $o/ab_regh_alimama.txt
python bench.py --config c5 --train-only --steps 20 --warmup 5 --no-other-configs > $o/bench_c5b.json 2> $o/bench_c5b.err
python -c "
import json; d=json.load(open('$o/bench_c5b.json')); print(d['ms_per_step'], d['roofline']['frac'], {k: v['ms_per_launch'] for k, v in d['kernels'].items()})"
