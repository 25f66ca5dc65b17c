#!/bin/bash
# One GPU-box visit: parity tests, smoke, a short bench and a kernel-trace profile.
# Usage (from the repo root on the box): bash tools/gpu_check.sh <tag> [pytest-args...]
# Every GPU step runs under its own time limit; after a crash/timeout nothing else starts.
set -u
tag=${1:-run}; shift || true
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp

timeout -k 10 900 python -u -m pytest tests -m gpu -q --maxfail=10 --timeout 120 --timeout-method thread "$@" > "$out/pytest.log" 2>&1
rc=$?
tail -n 25 "$out/pytest.log"
echo "pytest rc=$rc"
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "stopping: pytest ended abnormally"; exit $rc; fi

timeout -k 10 300 python -u -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1
rc=$?; tail -n 3 "$out/smoke.log"; echo "smoke rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi

timeout -k 10 600 python -u bench.py --steps 20 --warmup 3 > "$out/bench.json" 2> "$out/bench.err"
rc=$?; cat "$out/bench.json"; tail -n 5 "$out/bench.err"; echo "bench rc=$rc"
if [ $rc -ne 0 ]; then exit $rc; fi

( cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/prof" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 2 --no-extras --cpu-seconds 0 --mode eager > "$GRAFT_REPO_ROOT/$out/prof_bench.json" 2> "$GRAFT_REPO_ROOT/$out/prof.err" )
rc=$?; echo "rocprof rc=$rc"
f=$(find "$out/prof" -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then grep -E "^\"Name\"|bsgl" "$f" | cut -c1-260; fi
exit $rc
