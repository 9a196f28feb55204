#!/bin/bash
# round 6, run E: GPU suite, gate / bilinear lines, fit() after the speculative order draw, kernel stats + one step's timeline
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
o=gpurun_out/r06; mkdir -p $o
python -m pytest tests -m gpu -x -q > $o/pytest_g.log 2>&1; tail -3 $o/pytest_g.log
for f in sota-gate sota-bilinear; do
  python bench.py --flag $f --train-only --steps 40 --warmup 10 > $o/bench_$f.json 2>/dev/null
  python -c "
import json; d=json.load(open('$o/bench_$f.json')); print('$f', d['ms_per_step'], d['roofline']['frac'], d['roofline']['launch_ms'], {k: v['ms_per_launch'] for k, v in d['kernels'].items()})"
done
python tools/fit_time.py 2>/dev/null | grep "^fit" | tee $o/fit_time2.txt
python bench.py --no-other-configs --cpu-steps 0 --sustained-steps 0 > $o/bench_fit.json 2> /dev/null
python -c "
import json; d=json.load(open('$o/bench_fit.json')); print(d['ms_per_step'], d['fit'])"
rocprofv3 --kernel-trace --stats --output-format csv -d $o/ks -o p -- python3 bench.py --train-only --steps 25 --warmup 5 --no-phase-timing > $o/ks.log 2>&1
python tools/kernel_stats.py $(find $o/ks -name "*kernel_stats.csv" | head -1) 40 > $o/kernel_stats.md
python tools/step_timeline.py $(find $o/ks -name "*kernel_trace.csv" | head -1) 12 > $o/timeline.txt
rm -rf $o/ks
cat $o/timeline.txt | head -80
