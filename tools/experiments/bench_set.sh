#!/bin/bash
# The bench records of a round: default line, Alimama shape, one rank through RCCL, flags gate / bilinear (train-only legs)
cd "$(dirname "$0")/../.." || exit 1
o=gpurun_out/benchset; mkdir -p $o
python bench.py > $o/bench_default.json 2> $o/bench_default.err; echo "default rc=$?"
python bench.py --config alimama --train-only > $o/bench_alimama.json 2>/dev/null
SATRANS_FORCE_EXCHANGE=1 python bench.py --train-only > $o/bench_owner.json 2>/dev/null
python bench.py --flag sota-gate --train-only > $o/bench_gate.json 2>/dev/null
python bench.py --flag sota-bilinear --train-only > $o/bench_bilinear.json 2>/dev/null
for f in $o/*.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["value"])
PY
done
