#!/bin/bash
# The one-shot all-reduce between <ranks> fresh processes on ONE GPU (tools/p2p_procs.py): windows
# mapped through IPC handles, results checked bit for bit, then the time of a call.  The same ranks
# run under pytest as tests/test_gpu_tp_procs.py.
# Usage: bash tools/rehearse_p2p.sh [ranks=4]
set -u
n=${1:-4}
out=gpurun_out/rehearse_p2p$n
mkdir -p "$out"
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29641 WORLD_SIZE=$n BABYSGL_TP_BACKEND=gloo LOCAL_RANK=0
pids=()
for r in $(seq 0 $((n - 1))); do
  RANK=$r timeout -k 10 200 python tools/run_with_dump.py 120 tools/p2p_procs.py > "$out/rank$r.log" 2>&1 &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait "$p" || rc=$?; done
for r in $(seq 0 $((n - 1))); do echo "--- rank $r"; grep -v amdgpu.ids "$out/rank$r.log" | tail -n 6 | cut -c1-400; done
exit $rc
