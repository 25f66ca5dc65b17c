#!/bin/bash
# A/B of two builds of the library on the decode bench, interleaved rounds in one session:
#   bash tools/ab_decode.sh <other_lib.so> [rounds] [env assignments for the tree's runs, e.g. BSGL_DECODE_SPAN=0]
# Runs on the GPU box from the repo root; restores the tree's library at the end.
set -u
other=$1; rounds=${2:-2}; treeenv=${3:-}
lib=baby-sglang_amd/libbsgl_attn.so
cp "$lib" /tmp/bsgl_tree.so
trap 'cp /tmp/bsgl_tree.so "$lib"' EXIT
one() { timeout -k 10 200 env "$@" python -u bench.py --steps 40 --warmup 5 --no-extras --cpu-seconds 0 --table "$t" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4f ms/step  %.2f us/launch' % (d['ms_per_step'], d['roofline']['avg_launch_us']))" || exit 1; }
for t in perm radix ragged; do
  echo "== table $t"
  for r in $(seq $rounds); do
    cp "$other" "$lib"; echo -n "other: "; one X=1
    cp /tmp/bsgl_tree.so "$lib"; echo -n "tree:  "; one X=1
    if [ -n "$treeenv" ]; then echo -n "tree ($treeenv): "; one $treeenv; fi
  done
done
