#!/bin/bash
# round 6, run A: GPU suite + the train-only bench line of the current sources
cd "$(dirname "$0")/../.." || exit 1
o=gpurun_out/r06; mkdir -p $o
tag=${1:-a}
python -m pytest tests -m gpu -x -q > $o/pytest_$tag.log 2>&1; tail -5 $o/pytest_$tag.log
python bench.py --train-only --steps 40 --warmup 10 > $o/bench_$tag.json 2> $o/bench_$tag.err
python - <<P
import json
d = json.load(open("$o/bench_$tag.json"))
print(d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["launch_ms"])
print({k: v["ms_per_launch"] for k, v in d["kernels"].items()})
P
