#!/bin/bash
# Build a variant of the library with extra compiler flags for ONE translation unit:
#   bash tools/build_variant.sh <name> <source.hip> <flags...>   ->  variants/libbsgl_<name>.so
# (the other objects are the tree's; run `make -C baby-sglang_amd/csrc` first)
set -eu
name=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
csrc=$root/baby-sglang_amd/csrc
mkdir -p "$root/variants"
extra=""
case $src in paged_prefill_mfma*.hip) extra="-fno-honor-nans";; esac
obj=$root/variants/${src%.hip}_$name.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-gpu-rdc $extra "$@" -c "$csrc/$src" -o "$obj"
others=$(cd "$csrc" && ls *.o | grep -v "^${src%.hip}.o$" | sed "s|^|$csrc/|")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/variants/libbsgl_$name.so" "$obj" $others -ldl
rm -f "$obj"
echo "built variants/libbsgl_$name.so"
