#!/bin/bash
# round 6, run C: GPU suite, the Adam arithmetic experiment, the full default bench line (with other_configs)
cd "$(dirname "$0")/../.." || exit 1
o=gpurun_out/r06; mkdir -p $o
python -m pytest tests -m gpu -x -q > $o/pytest_e.log 2>&1; tail -3 $o/pytest_e.log
python tools/experiments/r06_adam_drift.py > $o/adam_drift.txt 2> $o/adam_drift.err; cat $o/adam_drift.txt; tail -3 $o/adam_drift.err
( time python bench.py > $o/bench_default.json 2> $o/bench_default.err ) 2>&1 | tail -3
python - <<P
import json
d = json.load(open("$o/bench_default.json"))
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["launch_ms"], d["sustained_ms_per_step"], d["fit_samples_per_s"])
print(d["cpu_baseline"])
print(json.dumps(d["other_configs"], indent=1)[:6000])
print(d["forward_only"], d["forward_only_bf16"])
P
