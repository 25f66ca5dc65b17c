#!/usr/bin/env python3
"""Distance of the GPU path's outputs to the REFERENCE's own outputs on every golden case
(tests/golden/decode_*.pt, prefill_*.pt: produced by running the reference in the build container),
and to the float64 definition; worst absolute difference per case.  For the tolerance table."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import golden_names, load_golden  # noqa: E402
from test_gpu_parity import oracle_exact, product_forward  # noqa: E402

dev = torch.device("cuda:0")
worst = {}
for name in golden_names("decode_") + golden_names("prefill_"):
    case = load_golden(name)
    out, _, _ = product_forward(case, dev)
    want, _, _ = oracle_exact(case)
    got = out.double().cpu()
    d_ref = (got - case["out"].double()).abs().max().item() if got.numel() else 0.0
    d_def = (got - want).abs().max().item() if got.numel() else 0.0
    r_def = (case["out"].double() - want).abs().max().item() if got.numel() else 0.0
    key = (case["kind"], str(case["q"].dtype).replace("torch.", ""))
    worst[key] = max(worst.get(key, 0.0), d_ref)
    print(f"{name:34s} {key[1]:9s} |gpu - reference| {d_ref:.3e}   |gpu - float64| {d_def:.3e}   |reference - float64| {r_def:.3e}")
print("worst |gpu - reference| per (kind, dtype):", {f"{k[0]} {k[1]}": f"{v:.3e}" for k, v in sorted(worst.items())})
