#!/bin/bash
# Cost of the attention phases of the fused backward, by removal (round 6): the shipped library against variants without phase D,
# without phase E, without both, and against the wavefront arm (-DSATRANS_ATTN_LANE) with and without them, alternating on ONE box.
# Build the variants first (CPU):  for v in "mfa_noD -DSATRANS_DIAG_SKIP=1" ...: see below.
cd "$(dirname "$0")/../.." || exit 1
if [ "$1" = build ]; then
  rm -f tools/experiments/_variants/lib_*.so
  SATRANS_EXTRA_FLAGS="-DSATRANS_DIAG_SKIP=1" tools/experiments/build_variant.sh mfa_noD 's/^$//' layer_fused.hip
  SATRANS_EXTRA_FLAGS="-DSATRANS_DIAG_SKIP=2" tools/experiments/build_variant.sh mfa_noE 's/^$//' layer_fused.hip
  SATRANS_EXTRA_FLAGS="-DSATRANS_DIAG_SKIP=3" tools/experiments/build_variant.sh mfa_noDE 's/^$//' layer_fused.hip
  SATRANS_EXTRA_FLAGS="-DSATRANS_ATTN_LANE" tools/experiments/build_variant.sh lane 's/^$//' layer_fused.hip
  SATRANS_EXTRA_FLAGS="-DSATRANS_ATTN_LANE -DSATRANS_DIAG_SKIP=3" tools/experiments/build_variant.sh lane_noDE 's/^$//' layer_fused.hip
  exit 0
fi
mkdir -p gpurun_out/r06
ROUNDS=${ROUNDS:-2} PHASES=layer_fwd,layer_bwd,layer_bwd_head bash tools/experiments/ab.sh | tee gpurun_out/r06/phase_cost.txt
