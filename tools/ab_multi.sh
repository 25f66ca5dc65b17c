#!/bin/bash
# A/B/C... of library variants on the extend timing, interleaved rounds in one session:
#   bash tools/ab_multi.sh <rounds> <profile 0|1> "<bench_prefill args>" tree x1 x2 ...
# ("tree" = the tree's library, other names = variants/libbsgl_<name>.so).  Runs on the GPU box from
# the repo root; restores the tree's library at the end.
set -u
rounds=$1; prof=$2; args=$3; shift 3
lib=baby-sglang_amd/libbsgl_attn.so
cp "$lib" /tmp/bsgl_tree.so
trap 'cp /tmp/bsgl_tree.so "$lib"' EXIT
for r in $(seq $rounds); do
  for which in "$@"; do
    if [ $which = tree ]; then cp /tmp/bsgl_tree.so "$lib"; else cp "variants/libbsgl_$which.so" "$lib"; fi
    extra=""; [ "$prof" = 1 ] && [ $r = 1 ] && extra="--profile"
    echo "== $which (round $r): "
    timeout -k 10 150 python -u tools/bench_prefill.py --warmup 30 --reps 40 $args $extra 2>&1 | grep -v "^$" | cut -c1-220 || exit 1
  done
done
