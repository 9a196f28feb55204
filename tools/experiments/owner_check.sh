#!/bin/bash
# Owner-form (row-ownership data parallel) checks on ONE GPU: the multi-rank parity tests, one rank through RCCL, the N-rank compute emulation.
cd "$(dirname "$0")/../.." || exit 1
o=${1:-gpurun_out/owner}; mkdir -p $o
python -m pytest tests -m gpu -q -x -k "rank or rccl or hint or merge_of" > $o/pytest_ranks.log 2>&1; echo "pytest(ranks) rc=$?"; tail -2 $o/pytest_ranks.log
python bench.py --train-only > $o/bench_local.json 2> $o/bench_local.err; echo "local rc=$?"
SATRANS_FORCE_EXCHANGE=1 python bench.py --train-only > $o/bench_owner.json 2> $o/bench_owner.err; echo "owner rc=$?"
python tools/fake_world.py 1 2 4 8 > $o/fake_world.txt 2> $o/fake_world.err; echo "fake_world rc=$?"
python - <<PY
import json
for n in ("local", "owner"):
    try:
        d = json.loads(open("$o/bench_%s.json" % n).read().strip().splitlines()[-1])
        print(n, d["ms_per_step"], {k: v.get("ms_per_launch") for k, v in d["kernels"].items()})
    except Exception as e:
        print(n, "failed", e)
PY
cat $o/fake_world.txt
SATRANS_FORCE_EXCHANGE=1 python tools/host_time.py 2>&1 | head -30 > $o/host_time_owner.txt; head -3 $o/host_time_owner.txt
