#!/bin/bash
# round 6, run G: GPU suite after the reader-stream change, fit() per epoch again, the fit leg of the bench
cd "$(dirname "$0")/../.." || exit 1
o=gpurun_out/r06; mkdir -p $o
python -m pytest tests -m gpu -x -q > $o/pytest_h.log 2>&1; tail -3 $o/pytest_h.log
python tools/experiments/r06_fit_epoch.py 2>/dev/null | tee $o/fit_epoch2.txt
python bench.py --no-other-configs --cpu-steps 0 > $o/bench_fit2.json 2> /dev/null
python -c "
import json; d=json.load(open('$o/bench_fit2.json')); print(d['ms_per_step'], d['sustained_ms_per_step'], d['fit'])"
