"""Full-size cases against outputs of the REFERENCE itself: decode batches with requests of up to 4095
tokens, extends of 512 / 1024 new tokens over 1536 / 0 cached ones, on the 1.5B and 7B head shapes.

``tests/golden/full_*.pt`` hold rows of what the reference computed on the CPU
(``tests/golden/gen_fullsize_golden.py``, run in the build container) and checksums of the inputs;
the inputs themselves are rebuilt from their seed (``tests/golden/fullsize_inputs.py``).  On the CPU the
oracle's step-by-step restatement must give those rows (bit for bit on a host like the one that made
them, within one unit in the last place elsewhere); on the GPU the product path must be within one unit in the last place of them (both sides are rounded to the dtype), and write the same
K/V rows bit for bit."""
import os
import sys

import pytest
import torch

from conftest import GOLDEN_DIR, load_golden
from oracle import attn_oracle as orc

sys.path.insert(0, GOLDEN_DIR)
from fullsize_inputs import CASES, build, checksums, host_fingerprint  # noqa: E402


def _case(name):
    if not os.path.exists(os.path.join(GOLDEN_DIR, name + ".pt")):
        pytest.skip("fixture not generated")
    stored = load_golden(name)
    case = build(name)
    # the inputs are the ones the reference saw (same torch, same generator stream)
    assert torch.equal(checksums(case), stored["checksums"]), "the seeded inputs differ from the generator's"
    assert torch.equal(case["rows_kept"], stored["rows_kept"])
    return case, stored


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_restatement_is_bit_exact_at_full_size(name):
    case, stored = _case(name)
    hkv, dim = case["n_kv_heads"], case["head_dim"]
    q, k = case["q"], case["k"]
    v = case["qkv"][:, case["v_col0"] :].view(q.shape[0], hkv, dim)
    k_slab, v_slab = case["k_slab"].clone(), case["v_slab"].clone()
    orc.kv_scatter_ref(k_slab, v_slab, case["out_cache_loc"], k, v)
    args = (q, k_slab, v_slab, case["req_to_token"], case["req_pool_indices"], case["seq_lens"])
    if case["kind"] == "prefill":
        out = orc.prefill_ref(*args, case["prefix_lens"], case["extended_lens"])
    else:
        out = orc.decode_ref(*args)
    keep = stored["rows_kept"]
    assert list(out.shape) == stored["out_shape"].tolist() and out.dtype == stored["out_rows"].dtype
    # Bit for bit on the host that made the fixture.  Reductions of this length are cut by torch's CPU
    # kernels according to the host's cores and vector width, so another host may round a sum the other way:
    # there, at most one unit in the last place, and nearly all elements identical.
    got, ref = out[keep].double(), stored["out_rows"].double()
    ulp = {torch.float16: 2.0**-10, torch.bfloat16: 2.0**-7}[out.dtype]
    assert bool(((got - ref).abs() <= ulp * ref.abs().clamp_min(1.0)).all())
    if stored["host"] == host_fingerprint():
        assert torch.equal(out[keep], stored["out_rows"])
    loc = case["out_cache_loc"].long()
    assert torch.equal(k_slab[loc][keep], stored["k_rows_kept"]) and torch.equal(v_slab[loc][keep], stored["v_rows_kept"])


@pytest.mark.gpu
@pytest.mark.parametrize("private_slots", [False, True])
@pytest.mark.parametrize("name", sorted(CASES))
def test_product_path_against_the_references_rows_at_full_size(gpu, name, private_slots):
    from test_gpu_parity import product_forward

    case, stored = _case(name)
    if case["kind"] == "prefill" and private_slots:
        pytest.skip("the promise of fresh slots only matters to decode steps")
    out, k_slab, v_slab = product_forward(case, gpu, private_slots=private_slots)
    keep = stored["rows_kept"]
    assert list(out.shape) == stored["out_shape"].tolist() and out.dtype == stored["out_rows"].dtype
    ulp = {torch.float16: 2.0**-10, torch.bfloat16: 2.0**-7}[out.dtype]
    ref = stored["out_rows"].double()
    err = (out.cpu()[keep].double() - ref).abs()
    assert bool((err <= ulp * ref.abs().clamp_min(1.0)).all()), f"{name}: max |gpu - reference| {err.max().item():.3e}"
    loc = case["out_cache_loc"].long()
    assert torch.equal(k_slab.cpu()[loc][keep], stored["k_rows_kept"])
    assert torch.equal(v_slab.cpu()[loc][keep], stored["v_rows_kept"])
