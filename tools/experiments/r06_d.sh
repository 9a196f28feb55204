#!/bin/bash
# round 6, run D: noise floor of the Adam arithmetic question, GPU suite (engine split, bf16 attention arm), forward legs, fit() cost
cd "$(dirname "$0")/../.." || exit 1
o=gpurun_out/r06; mkdir -p $o
{
python tools/experiments/r06_adam_noise.py exact
python tools/experiments/r06_adam_noise.py fast
SATRANS_LIB_PATH=$PWD/tools/experiments/_variants/lib_lane.so python tools/experiments/r06_adam_noise.py exact
SATRANS_LIB_PATH=$PWD/tools/experiments/_variants/lib_lane.so python tools/experiments/r06_adam_noise.py fast
} 2>/dev/null | tee $o/adam_noise.txt
python -m pytest tests -m gpu -x -q > $o/pytest_f.log 2>&1; tail -3 $o/pytest_f.log
python bench.py --no-other-configs --cpu-steps 0 --sustained-steps 0 --fit-batches 0 > $o/bench_fwd.json 2> $o/bench_fwd.err
python - <<P
import json
d = json.load(open("$o/bench_fwd.json"))
print(d["ms_per_step"], d["forward_only"], d["forward_only_bf16"])
P
python tools/fit_time.py 2>/dev/null | tee $o/fit_time.txt
python tools/fit_profile.py 1 2>&1 | tail -45 | tee $o/fit_profile.txt
