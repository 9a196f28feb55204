#!/bin/bash
# round 6, run B: GPU suite + train-only bench + phase stamps of the current sources
cd "$(dirname "$0")/../.." || exit 1
o=gpurun_out/r06; mkdir -p $o
tag=${1:-b}
bash tools/experiments/r06_a.sh $tag
python bench.py --train-only --steps 40 --warmup 10 --config alimama > $o/bench_${tag}_alimama.json 2> /dev/null
python - <<P
import json
d = json.load(open("$o/bench_${tag}_alimama.json"))
print("alimama", d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["launch_ms"])
print({k: v["ms_per_launch"] for k, v in d["kernels"].items()})
P
python tools/stamps.py > $o/stamps_$tag.txt 2>&1; cat $o/stamps_$tag.txt
