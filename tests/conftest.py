"""Shared pytest setup: repo root on sys.path, the `gpu` marker, golden loader."""
import glob
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


TP_PROCS_TEST = "test_gpu_tp_procs.py"
# job name -> (program under tools/, ranks, extra environment)
TP_PROC_JOBS = {
    "tp_model": ("tp_model_procs.py", 2, {}),
    "tp_model_one_shot": ("tp_model_procs.py", 2, {"BABYSGL_P2P_ALLREDUCE": "1"}),
    "one_shot_all_reduce": ("p2p_procs.py", 4, {}),
}
# programs that start their own ranks: job name -> (argv after the interpreter, extra environment); no rank environment
SELF_LAUNCH_JOBS = {
    "bench_two_ranks": (["bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0", "--no-extras",
                         "--launch-timeout", "150"], {"BABYSGL_TP_BACKEND": "gloo"}),
    # needs a process in which the no-workspace decode entry point has never run
    "static_first_call": (["tools/static_capture_first_call.py"], {}),
}
TP_PROCS = {}  # filled by pytest_sessionstart: job name -> {"procs": [Popen...], "logs": [path...]}


def _wants_tp_procs(config):
    expr = config.getoption("markexpr", "") or ""
    if "gpu" not in expr or "not gpu" in expr:
        return False
    for arg in config.args:
        path = arg.split("::")[0]
        if os.path.isdir(path) or os.path.basename(path) == TP_PROCS_TEST:
            return True
    return False


def pytest_sessionstart(session):
    """The ranks of tests/test_gpu_tp_procs.py are started HERE, before any test has touched the GPU:
    a process that has initialised the GPU should not start other programs on the GPU boxes, and
    counting devices does not initialise it.  The tests themselves only wait for them."""
    if not _wants_tp_procs(session.config) or torch.cuda.device_count() == 0:
        return
    import socket
    import subprocess
    import tempfile

    tmp = tempfile.mkdtemp(prefix="bsgl_tp_procs_")
    # one rendezvous port per job, all picked before any is given back: no two jobs get the same one
    socks = [socket.socket() for _ in TP_PROC_JOBS]
    for s in socks:
        s.bind(("127.0.0.1", 0))
    ports = [s.getsockname()[1] for s in socks]
    for s in socks:
        s.close()
    for port, (name, (program, world, extra)) in zip(ports, TP_PROC_JOBS.items()):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world),
                   LOCAL_RANK="0", BABYSGL_TP_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", **extra)
        job = TP_PROCS[name] = {"procs": [], "logs": []}
        for rank in range(world):
            log = os.path.join(tmp, f"{name}.rank{rank}.log")
            with open(log, "wb") as f:
                p = subprocess.Popen(
                    [sys.executable, os.path.join(ROOT, "tools", "run_with_dump.py"), "170", os.path.join(ROOT, "tools", program)],
                    env=dict(env, RANK=str(rank)), stdout=f, stderr=subprocess.STDOUT, cwd=ROOT)
            job["procs"].append(p)
            job["logs"].append(log)


    for name, (argv, extra) in SELF_LAUNCH_JOBS.items():
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
        env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", **extra)
        log, err = os.path.join(tmp, f"{name}.out"), os.path.join(tmp, f"{name}.err")
        with open(log, "wb") as f, open(err, "wb") as e:
            p = subprocess.Popen([sys.executable, os.path.join(ROOT, argv[0])] + argv[1:], env=env, stdout=f, stderr=e, cwd=ROOT)
        TP_PROCS[name] = {"procs": [p], "logs": [log, err]}


def pytest_sessionfinish(session, exitstatus):
    for job in TP_PROCS.values():
        for p in job["procs"]:
            if p.poll() is None:
                p.terminate()  # (a program that started ranks of its own ends them on SIGTERM)
                try:
                    p.wait(timeout=15)
                except Exception:  # noqa: BLE001
                    p.kill()
                    p.wait()


def golden_names(prefix):
    return sorted(
        os.path.splitext(os.path.basename(p))[0]
        for p in glob.glob(os.path.join(GOLDEN_DIR, prefix + "*.pt"))
    )


def load_golden(name):
    return torch.load(os.path.join(GOLDEN_DIR, name + ".pt"), weights_only=True)


def golden_new_rows(case):
    """q, k, v exactly as the reference received them: v is a strided column
    slice of the merged qkv row (qwen2.py:288)."""
    hkv, dim = case["n_kv_heads"], case["head_dim"]
    qkv = case["qkv"]
    v = qkv[:, case["v_col0"] :].view(qkv.shape[0], hkv, dim)
    return case["q"], case["k"], v


@pytest.fixture(scope="session")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture
def knobs():
    """``knobs(NAME, value)`` sets one of the library's tuning knobs for the test (the library
    reads the environment only once per process) and puts the old value back afterwards."""
    from baby_sglang_amd import _hip

    saved = []

    def set_(name, value):
        saved.append((name, _hip.set_knob(name, value)))

    yield set_
    for name, old in reversed(saved):
        _hip.set_knob(name, old)
