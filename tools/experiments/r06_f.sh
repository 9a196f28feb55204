#!/bin/bash
# round 6, run F: fit() per-epoch costs; first pass of the configs[4] PMC tooling
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
o=gpurun_out/r06; mkdir -p $o
python tools/experiments/r06_fit_epoch.py 2>/dev/null | tee $o/fit_epoch.txt
bash tools/pmc_c5.sh
python tools/pmc_c5_summary.py gpurun_out/pmc_c5 6 > $o/c5_pmc_summary.json 2> $o/c5_pmc_summary.err
head -c 1500 $o/c5_pmc_summary.json; tail -3 $o/c5_pmc_summary.err
tail -3 gpurun_out/pmc_c5/fetch.log
rm -rf gpurun_out/pmc_c5
