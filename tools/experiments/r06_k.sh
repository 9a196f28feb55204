#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
o=gpurun_out/r06; mkdir -p $o
python -m pytest tests -m gpu -x -q > $o/pytest_k.log 2>&1; tail -3 $o/pytest_k.log
ROUNDS=3 PHASES=layer_fwd_gather,layer_fwd,layer_bwd,layer_bwd_head bash tools/experiments/ab.sh | tee $o/ab_prow.txt
BENCH_ARGS="--config alimama" ROUNDS=2 PHASES=layer_fwd,layer_bwd,layer_bwd_head bash tools/experiments/ab.sh | tee -a $o/ab_prow.txt
