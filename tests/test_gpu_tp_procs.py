"""The tensor-parallel model as two PROCESSES on the one GPU: each rank sets its group up with
init_tp() from the torchrun-style environment (gloo as the transport: RCCL wants a GPU per rank),
shards the reference model's weights (column-parallel qkv / gate_up, row-parallel o_proj / down_proj),
sums the row-parallel outputs with tp_all_reduce_sum and compares its logits with the reference
model's unsharded ones (reference: tp_runtime.py:10-33, qwen2.py:179-197).  The ranks are
tools/tp_model_procs.py; tests/conftest.py starts them when the session starts, before this process
has touched the GPU, and this test waits for their verdicts."""
import subprocess

import pytest

from conftest import TP_PROCS

pytestmark = pytest.mark.gpu


def test_tp_model_as_two_processes(gpu):
    if not TP_PROCS.get("procs"):
        pytest.skip("the ranks are only started when this file (or its directory) is run with -m gpu")
    codes = []
    for p in TP_PROCS["procs"]:
        try:
            codes.append(p.wait(timeout=180))
        except subprocess.TimeoutExpired:
            p.kill()
            codes.append("timeout")
    logs = []
    for path in TP_PROCS["logs"]:
        with open(path, errors="replace") as f:
            logs.append(f.read())
    report = "\n".join(f"--- rank {r} (exit {c})\n{log[-3000:]}" for r, (c, log) in enumerate(zip(codes, logs)))
    assert codes == [0] * len(codes), report
    for r, log in enumerate(logs):
        for dt in ("f32", "bf16"):
            assert f"rank {r}/{len(logs)} {dt}:" in log and "equal the reference model's logits" in log, report
