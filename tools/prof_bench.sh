#!/bin/bash
# rocprofv3 kernel trace of a short bench run; prints the per-kernel stats of our kernels.
# Usage: bash tools/prof_bench.sh <tag> [bench args...]
set -u
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 2 --no-extras --cpu-seconds 0 --mode eager "$@" > "$out/bench.json" 2> "$out/prof.err"
rc=$?
echo "rocprof rc=$rc"
cat "$out/bench.json" | cut -c1-1500
f=$(find "$out/prof" -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then grep -E "^\"Name\"|bsgl" "$f" | cut -c1-260; fi
exit $rc
