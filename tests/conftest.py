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
