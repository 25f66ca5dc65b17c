#!/bin/bash
# Rehearse bench.py's multi-rank path on ONE GPU: N ranks share cuda:0 and talk over gloo
# (RCCL needs one GPU per rank).  Checks the sharding / exchange / merge code on real GPU
# tensors; the numbers mean nothing.  Usage: [EXTRAS=1] bash tools/rehearse_tp.sh <ranks> [bench args]
# (EXTRAS=1 also runs the side measurements, e.g. replicas x tp2 beside the one-group line)
set -u
n=$1; shift
out=gpurun_out/rehearse_tp$n
mkdir -p "$out"
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29617 WORLD_SIZE=$n BABYSGL_TP_BACKEND=gloo LOCAL_RANK=0
noextras=--no-extras; [ "${EXTRAS:-0}" = 1 ] && noextras=
pids=()
for r in $(seq 0 $((n - 1))); do
  RANK=$r timeout -k 10 150 python tools/run_with_dump.py 60 bench.py --gpus "$n" --steps 3 --warmup 1 --cpu-seconds 0 $noextras "$@" > "$out/rank$r.log" 2>&1 &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait "$p" || rc=$?; done
for r in $(seq 0 $((n - 1))); do echo "--- rank $r"; tail -n 5 "$out/rank$r.log" | cut -c1-600; done
exit $rc
