#!/bin/bash
# rocprofv3 kernel trace of the extend benchmark; prints the per-kernel stats of our kernels.
# Usage: bash tools/prof_prefill.sh <tag> [bench_prefill args...]
set -u
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -- python3 "$GRAFT_REPO_ROOT/tools/bench_prefill.py" "$@" > "$out/bench.txt" 2> "$out/prof.err"
rc=$?
echo "rocprof rc=$rc"
cat "$out/bench.txt" | cut -c1-400
f=$(find "$out/prof" -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" "$out/kernel_stats.csv"; grep -E "^\"Name\"|bsgl" "$f" | cut -c1-260; fi
exit $rc
