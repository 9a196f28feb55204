#!/bin/bash
# HBM traffic of BASELINE configs[4] (the general layer path: a layer is a chain of ~25 / ~45 launches) from rocprofv3 PMC passes:
# FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md, rocprofv3 PMC slots), kernel-trace only.
#   bash tools/pmc_c5.sh [table rows, default 100000000]   ->  gpurun_out/pmc_c5/{fetch,write}/...; summarise with tools/pmc_c5_summary.py
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
rows=${1:-100000000}
out=gpurun_out/pmc_c5
mkdir -p $out
run() {
  name=$1; shift
  timeout 900 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $out/$name -o p -- python3 bench.py --config c5 --table-rows $rows --steps 3 --warmup 1 --train-only --no-phase-timing --no-other-configs < /dev/null > $out/$name.log 2>&1
  echo "$name rc=$?"
}
run fetch FETCH_SIZE
run write WRITE_SIZE
