#!/bin/bash
# A/B of two builds of the library on the extend timing (sustained rate: 30 untimed launches first),
# interleaved rounds in one session:
#   bash tools/ab_prefill.sh <other_lib.so> [rounds]
# Runs on the GPU box from the repo root; restores the tree's library at the end.
set -u
other=$1; rounds=${2:-2}
lib=baby-sglang_amd/libbsgl_attn.so
cp "$lib" /tmp/bsgl_tree.so
trap 'cp /tmp/bsgl_tree.so "$lib"' EXIT
shapes=("" "--hq 12 --hkv 2 --batch 64" "--hq 8 --hkv 8 --batch 64" "--batch 16 --ext 4096 --prefix 0" "--batch 256 --ext 24 --prefix 2000" "--dtype f16")
for s in "${shapes[@]}"; do
  echo "== shape: ${s:-default (7B chunk)}"
  for r in $(seq $rounds); do
    for which in other tree; do
      if [ $which = other ]; then cp "$other" "$lib"; else cp /tmp/bsgl_tree.so "$lib"; fi
      echo -n "$which: "; timeout -k 10 120 python -u tools/bench_prefill.py --warmup 30 --reps 40 $s 2>/dev/null | grep -o "[0-9.]* ms/layer *[0-9.]* TFLOP/s" || exit 1
    done
  done
done
