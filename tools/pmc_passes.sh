#!/bin/bash
# PMC passes over a short bench run (one counter group per pass, kernel-trace only; MI355X_MICROARCH.md "rocprofv3 PMC slots").
# Output: gpurun_out/pmc/<pass>/... counter_collection.csv ; summarise with tools/pmc_summary.py
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=${PMC_OUT:-gpurun_out/pmc}          # PMC_BENCH_ARGS: extra bench.py arguments (e.g. "--config alimama"; PMC_ONLY_TRAFFIC=1: two passes)
mkdir -p $out
run() {  # name, counters...
  name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $out/$name -o p -- python3 bench.py --steps 4 --warmup 2 --train-only --no-phase-timing ${PMC_BENCH_ARGS:-} < /dev/null > $out/$name.log 2>&1
  echo "$name rc=$?"
}
run fetch FETCH_SIZE
run write WRITE_SIZE
[ -n "${PMC_ONLY_TRAFFIC:-}" ] && exit 0
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_ACTIVE_INST_MISC
find $out -name "*.csv" | head -20
