#!/bin/bash
# Build a kernel-experiment variant of the library without touching the shipped sources:
#   tools/experiments/build_variant.sh NAME 'sed-expression' [file.hip ...]
# copies satrans_amd/csrc to a temporary directory, applies the sed expression to the named files (default: all), builds for
# gfx950 and leaves tools/experiments/_variants/lib_NAME.so (git-ignored; picked up by run_variants.sh / ab.sh via SATRANS_LIB_PATH).
set -euo pipefail
root="$(cd "$(dirname "$0")/../.." && pwd)"
name="$1"; expr="$2"; shift 2
tmp="$(mktemp -d)"
mkdir -p "$tmp/satrans_amd" "$tmp/include" "$root/tools/experiments/_variants"
cp -r "$root/satrans_amd/csrc" "$tmp/satrans_amd/csrc"
cp "$root/include/"*.h "$tmp/include/"
rm -rf "$tmp/satrans_amd/csrc/build"
files=("$@"); [ ${#files[@]} -eq 0 ] && files=($(cd "$tmp/satrans_amd/csrc" && ls *.hip *.h))
for f in "${files[@]}"; do sed -i -E "$expr" "$tmp/satrans_amd/csrc/$f"; done
bash "$tmp/satrans_amd/csrc/build.sh" > /dev/null
cp "$tmp/satrans_amd/libsatrans_hip.so" "$root/tools/experiments/_variants/lib_$name.so"
rm -rf "$tmp"
echo "built tools/experiments/_variants/lib_$name.so"
