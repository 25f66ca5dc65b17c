#!/bin/bash
# The tensor-parallel MODEL across processes on ONE GPU: <ranks> fresh processes share cuda:0, set up
# their group with init_tp() and talk over gloo (RCCL wants one GPU per rank); every rank checks its
# logits against the reference model's.  tests/test_gpu_tp_procs.py runs the same two ranks under pytest.
# Usage: bash tools/rehearse_tp_model.sh [ranks=2]
set -u
n=${1:-2}
out=gpurun_out/rehearse_tp_model$n
mkdir -p "$out"
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29627 WORLD_SIZE=$n BABYSGL_TP_BACKEND=gloo LOCAL_RANK=0
pids=()
for r in $(seq 0 $((n - 1))); do
  RANK=$r timeout -k 10 200 python tools/run_with_dump.py 120 tools/tp_model_procs.py > "$out/rank$r.log" 2>&1 &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait "$p" || rc=$?; done
for r in $(seq 0 $((n - 1))); do echo "--- rank $r"; grep -v amdgpu.ids "$out/rank$r.log" | tail -n 6 | cut -c1-400; done
exit $rc
