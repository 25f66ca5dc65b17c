"""Tensor-parallel ranks as PROCESSES that share the one GPU (gloo is the store and, where nothing
else is said, the transport: RCCL wants a GPU per rank).  The rank programs are under tools/;
tests/conftest.py starts them when the session starts, before this process has touched the GPU, and
the tests here wait for their exit codes and verdict lines.

* tools/tp_model_procs.py, two ranks: init_tp() from the torchrun-style environment, the reference
  model's weights sharded (column-parallel qkv / gate_up, row-parallel o_proj / down_proj), the
  row-parallel outputs summed with tp_all_reduce_sum, logits against the reference model's unsharded
  ones (reference: tp_runtime.py:10-33, qwen2.py:179-197) -- once with gloo's sums, once with the
  library's one-shot all-reduce through windows mapped by IPC handles;
* tools/p2p_procs.py, four ranks: that all-reduce by itself over dtypes and sizes, bit for bit;
* bench.py typed as ``python bench.py --gpus 2`` with no launcher around it: it starts its two ranks
  itself (over gloo here, one GPU) and prints ONE line for one tensor-parallel group of two."""
import json
import subprocess

import pytest

from conftest import TP_PROCS

pytestmark = pytest.mark.gpu


def _verdicts(name):
    job = TP_PROCS.get(name)
    if not job:
        pytest.skip("the ranks are only started when this file (or its directory) is run with -m gpu")
    codes = []
    for p in job["procs"]:
        try:
            codes.append(p.wait(timeout=180))
        except subprocess.TimeoutExpired:
            p.kill()
            codes.append("timeout")
    logs = []
    for path in job["logs"]:
        with open(path, errors="replace") as f:
            logs.append(f.read())
    report = "\n".join(f"--- rank {r} (exit {c})\n{log[-3000:]}" for r, (c, log) in enumerate(zip(codes, logs)))
    assert codes == [0] * len(codes), report
    return logs, report


@pytest.mark.parametrize("job", ["tp_model", "tp_model_one_shot"])
def test_tp_model_as_two_processes(gpu, job):
    logs, report = _verdicts(job)
    for r, log in enumerate(logs):
        for dt in ("f32", "bf16"):
            assert f"rank {r}/{len(logs)} {dt}:" in log and "equal the reference model's logits" in log, report
        if job == "tp_model_one_shot":
            assert "were one-shot all-reduces through IPC windows" in log, report


def test_one_shot_all_reduce_between_four_processes(gpu):
    logs, report = _verdicts("one_shot_all_reduce")
    for r, log in enumerate(logs):
        assert f"rank {r}/{len(logs)}:" in log and "equal the rank-ordered fp32 sum bit for bit" in log, report


def test_bench_starts_its_own_ranks(gpu):
    job = TP_PROCS.get("bench_two_ranks")
    if not job:
        pytest.skip("the job is only started when this file (or its directory) is run with -m gpu")
    p = job["procs"][0]
    try:
        code = p.wait(timeout=200)
    except subprocess.TimeoutExpired:
        p.kill()
        code = "timeout"
    with open(job["logs"][0], errors="replace") as f:
        out = f.read()
    with open(job["logs"][1], errors="replace") as f:
        err = f.read()
    assert code == 0, f"exit {code}\n{err[-3000:]}"
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1, out
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["value"] > 0
    assert "ONE tensor-parallel group" in line["config"]["parallelism"] and "tp2" in line["config"]["parallelism"]


def test_no_workspace_decode_refuses_a_first_call_under_capture(gpu):
    """tools/static_capture_first_call.py in a process of its own (in this one the entry point has
    run already, or will have by the time a capture could be tried)."""
    job = TP_PROCS.get("static_first_call")
    if not job:
        pytest.skip("the job is only started when this file (or its directory) is run with -m gpu")
    p = job["procs"][0]
    try:
        code = p.wait(timeout=200)
    except subprocess.TimeoutExpired:
        p.kill()
        code = "timeout"
    with open(job["logs"][0], errors="replace") as f:
        out = f.read()
    with open(job["logs"][1], errors="replace") as f:
        err = f.read()
    assert code == 0 and "first call refused, later capture replays right" in out, f"exit {code}\n{out[-1000:]}\n{err[-3000:]}"
