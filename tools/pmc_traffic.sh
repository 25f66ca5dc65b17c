#!/bin/bash
# HBM traffic of the decode kernel from rocprofv3 PMC counters, one counter per pass
# (FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2).  Usage: bash tools/pmc_traffic.sh <tag>
set -u
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 240 rocprofv3 --pmc $ctr --output-format csv -d "$out/$ctr" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-extras --cpu-seconds 0 --mode eager "$@" > "$out/$ctr.json" 2> "$out/$ctr.err"
  rc=$?
  echo "$ctr pass rc=$rc"
  if [ $rc -ne 0 ]; then tail -5 "$out/$ctr.err"; exit $rc; fi
done
python3 "$GRAFT_REPO_ROOT/tools/pmc_summary.py" "$out"
