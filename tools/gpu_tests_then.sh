#!/bin/bash
# Parity tests first; if they pass (or fail only by assertion), run the given command.
# Usage: bash tools/gpu_tests_then.sh <tag> <pytest -k expr or ""> <command...>
set -u
tag=$1; kexpr=$2; shift 2
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
if [ -n "$kexpr" ]; then
  timeout -k 10 900 python -u -m pytest tests -m gpu -q --maxfail=10 --timeout 120 --timeout-method thread -k "$kexpr" > "$out/pytest.log" 2>&1
else
  timeout -k 10 900 python -u -m pytest tests -m gpu -q --maxfail=10 --timeout 120 --timeout-method thread > "$out/pytest.log" 2>&1
fi
rc=$?
tail -n 30 "$out/pytest.log" | cut -c1-300
echo "pytest rc=$rc"
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "stopping: pytest ended abnormally"; exit $rc; fi
if [ $# -gt 0 ]; then
  timeout -k 10 900 "$@" > "$out/cmd.log" 2>&1
  rc=$?
  tail -n 120 "$out/cmd.log" | cut -c1-400
  echo "cmd rc=$rc"
fi
exit $rc
