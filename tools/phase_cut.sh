#!/bin/bash
# What each phase of the headline decode launch is worth: the bench with the kernel cut short
# after its streaming loop, or after its partial rows (no ticket, no merge).  Results of the cut
# kernels are wrong; only their timing is read.  Run on the GPU box from the repo root:
#   bash tools/phase_cut.sh
set -u
csrc=baby-sglang_amd/csrc
lib=baby-sglang_amd/libbsgl_attn.so
make -C $csrc -j16 > /dev/null 2>&1 || exit 1  # (object files do not travel to the box: build them here first)
cp "$lib" /tmp/bsgl_whole.so
trap 'cp /tmp/bsgl_whole.so "$lib"' EXIT
B="python -u bench.py --steps 40 --warmup 5 --no-extras --cpu-seconds 0"
run() { timeout -k 10 120 $B | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ms/step', round(d['ms_per_step'], 4), ' us/launch', round(d['roofline']['avg_launch_us'], 2))" || exit 1; }
echo "whole kernel"; run
# (CUTS="KNOWN_LEN=2048": the batch's lengths known without a trip to memory -- what a piece list
# handed in from outside could save at most; headline batch only)
for cut in ${CUTS:-AFTER_STREAM AFTER_PARTIALS NO_DRAIN KNOWN_LEN=2048}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -w -fno-gpu-rdc -DBSGL_CUT_$cut -c $csrc/paged_decode_mfma.hip -o /tmp/bsgl_cut.o || exit 1
  objs=$(ls $csrc/*.o | grep -v "paged_decode_mfma.o\|gfx950")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$lib" $objs /tmp/bsgl_cut.o -ldl || exit 1
  echo "cut $cut"; run
done
