#!/bin/bash
# Build libsatrans_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU present.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../libsatrans_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-result ${SATRANS_EXTRA_FLAGS:-}"
objs=()
mkdir -p "$here/build"
pids=()
for src in "$here"/*.hip; do
  obj="$here/build/$(basename "${src%.hip}").o"
  objs+=("$obj")
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ -n "$(find "$here" "$here/../../include" -name '*.h' -newer "$obj" 2>/dev/null)" ]; then
    $HIPCC $FLAGS -c "$src" -o "$obj" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$out" "${objs[@]}"
echo "built $out"
