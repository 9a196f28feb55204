#!/bin/bash
# rocprofv3 kernel trace of one rank through RCCL (owner form) and of the local step: one step each as a timeline
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
o=${1:-gpurun_out/owner_tl}; mkdir -p $o
export SATRANS_FORCE_EXCHANGE=1
rocprofv3 --kernel-trace --output-format csv -d $o/ko -o p -- python3 bench.py --train-only --steps 25 --warmup 5 --no-phase-timing > $o/ko.log 2>&1
python tools/step_timeline.py $(find $o/ko -name "*kernel_trace.csv" | head -1) 12 > $o/timeline_owner.txt
unset SATRANS_FORCE_EXCHANGE
rocprofv3 --kernel-trace --output-format csv -d $o/kl -o p -- python3 bench.py --train-only --steps 25 --warmup 5 --no-phase-timing > $o/kl.log 2>&1
python tools/step_timeline.py $(find $o/kl -name "*kernel_trace.csv" | head -1) 12 > $o/timeline_local.txt
rm -rf $o/ko $o/kl
python tools/host_time.py 2>&1 | grep "host enqueue" > $o/host_time_local.txt
cat $o/timeline_owner.txt; cat $o/host_time_local.txt
