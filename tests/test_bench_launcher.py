"""bench.py typed as ``python bench.py --gpus N`` (no launcher, WORLD_SIZE unset) starts its N ranks
itself: the environment each rank gets is the one init_tp reads (reference: tp_runtime.py:10-25),
rank 0's output is forwarded, and a failing or stuck rank ends the job with a non-zero code within
the launcher's time limit.  CPU only: the ranks here are small stand-in programs."""
import io
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def _py(code):
    return [sys.executable, "-c", code]


def test_every_rank_gets_the_launcher_environment_and_rank_zero_is_forwarded():
    out = io.StringIO()
    code = ("import os, json; print(json.dumps({k: os.environ.get(k) for k in "
            "('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}))")
    rc = bench.self_launch(3, _py(code), timeout_s=60, out=out)
    assert rc == 0
    lines = out.getvalue().splitlines()
    assert len(lines) == 1, lines  # rank 0's line only
    env = json.loads(lines[0])
    assert env["RANK"] == "0" and env["LOCAL_RANK"] == "0" and env["WORLD_SIZE"] == "3"
    assert env["MASTER_ADDR"] == "127.0.0.1" and int(env["MASTER_PORT"]) > 0


def test_local_ranks_can_share_a_device():
    out = io.StringIO()
    code = "import os; print(os.environ['LOCAL_RANK']) if os.environ['RANK'] == '0' else None"
    assert bench.self_launch(2, _py(code), timeout_s=60, out=out, local_ranks=[0, 0]) == 0
    assert out.getvalue().strip() == "0"


def test_a_rank_that_fails_ends_the_job_with_its_code():
    t0 = time.monotonic()
    code = "import os, sys, time\nif os.environ['RANK'] == '1': sys.exit(3)\ntime.sleep(120)"
    rc = bench.self_launch(2, _py(code), timeout_s=100, out=io.StringIO())
    assert rc == 3
    assert time.monotonic() - t0 < 30  # the surviving rank was ended, not waited for


def test_ranks_that_never_finish_are_ended_by_the_watchdog():
    t0 = time.monotonic()
    rc = bench.self_launch(2, _py("import time; time.sleep(120)"), timeout_s=1.0, out=io.StringIO())
    assert rc == 124
    assert time.monotonic() - t0 < 30


def test_typed_without_a_launcher_and_without_the_gpus_it_says_so():
    """No GPU here (or fewer than asked for): the command ends at once with a message, it does not hang."""
    import torch

    if torch.cuda.device_count() >= 2:
        import pytest

        pytest.skip("two GPUs are visible: the command would run the benchmark")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BABYSGL_TP_BACKEND")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0
    assert "GPU(s) visible" in p.stderr and p.stdout == ""
