#!/bin/bash
# The GPU parity suite under every switch the engine reads from the environment (each path that can be turned off must stay green).
cd "$(dirname "$0")/../.." || exit 1
o=gpurun_out/flags; mkdir -p $o
run() { name=$1; shift; env "$@" python -m pytest tests -m gpu -q -x > $o/$name.log 2>&1; echo "$name rc=$? $(tail -1 $o/$name.log)"; }
run save_off SATRANS_SAVE_ATTENTION=0
run split SATRANS_PRODUCTS=split
run split_save SATRANS_PRODUCTS=split SATRANS_SAVE_ATTENTION=1
run no_fuse_head SATRANS_FUSE_HEAD=0
run no_prefetch SATRANS_PREFETCH=0
run no_defer SATRANS_DEFER_REDUCE=0
run no_side_tail SATRANS_SIDE_TAIL=0
