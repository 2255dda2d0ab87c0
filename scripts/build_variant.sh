#!/bin/bash
# build_variant.sh <name> [sed expressions...]: compiles a patched copy of the kernel source into build_variants/<name>.so
set -e
name=$1; shift
mkdir -p build_variants
args=()
for e in "$@"; do args+=(-e "$e"); done
if [ ${#args[@]} -gt 0 ]; then sed "${args[@]}" ascii-renderer_amd/csrc/ascii_rast.hip > build_variants/$name.hip; else cp ascii-renderer_amd/csrc/ascii_rast.hip build_variants/$name.hip; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Iinclude -shared -o build_variants/$name.so build_variants/$name.hip
