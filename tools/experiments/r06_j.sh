#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
o=gpurun_out/r06; mkdir -p $o
for m in nosave save; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $o/ksn -o p -- python3 tools/experiments/r06_nosave_stats.py $m > $o/ksn.log 2>&1
  echo "== $m"; python tools/kernel_stats.py $(find $o/ksn -name "*kernel_stats.csv" | head -1) 6
  rm -rf $o/ksn
done
python tools/experiments/r05_attr_ab.py save_attention 2>/dev/null
