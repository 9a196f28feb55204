#!/bin/bash
# round 6, run H: the general path after the one-launch reductions: its parity tests, the configs[4] line and kernel stats
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
o=gpurun_out/r06; mkdir -p $o
python -m pytest tests -m gpu -x -q -k "configs4 or general or sibling or generic or gate or bilinear or selfatt or metanet" > $o/pytest_i.log 2>&1; tail -3 $o/pytest_i.log
python bench.py --config c5 --train-only --steps 20 --warmup 5 --with-gather --no-other-configs > $o/bench_c5.json 2> $o/bench_c5.err
python -c "
import json; d=json.load(open('$o/bench_c5.json')); print(d['ms_per_step'], d['roofline']['frac'], {k: v['ms_per_launch'] for k, v in d['kernels'].items()})"
rocprofv3 --kernel-trace --stats --output-format csv -d $o/ks5 -o p -- python3 bench.py --config c5 --train-only --steps 10 --warmup 2 --no-phase-timing --no-other-configs > $o/ks5.log 2>&1
python tools/kernel_stats.py $(find $o/ks5 -name "*kernel_stats.csv" | head -1) 30 > $o/c5_kernel_stats.md
rm -rf $o/ks5
head -32 $o/c5_kernel_stats.md
