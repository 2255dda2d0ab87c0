#!/bin/bash
# build_variant.sh <name> [sed expressions...]: compiles a patched copy of the (flattened) library source into build_variants/<name>.so
set -e
name=$1; shift
mkdir -p build_variants
args=()
for e in "$@"; do args+=(-e "$e"); done
python3 scripts/flatten_source.py > build_variants/$name.flat
if [ ${#args[@]} -gt 0 ]; then sed "${args[@]}" build_variants/$name.flat > build_variants/$name.hip; else cp build_variants/$name.flat build_variants/$name.hip; fi
rm build_variants/$name.flat
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Iinclude -shared -o build_variants/$name.so build_variants/$name.hip
