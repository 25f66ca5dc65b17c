"""The oracle against the vectors captured from the reference (CPU only).

`*_ref` must reproduce the reference's output bit for bit (same torch ops in the
same dtype); `*_exact` (float64) must sit within the reference's own rounding.
"""
import pytest
import torch

from conftest import golden_names, golden_new_rows, load_golden
from oracle import attn_oracle as orc

# |reference(dtype) - float64 definition| on N(0,1) data; outputs are O(0.1-1)
EXACT_TOL = {torch.float32: 2e-6, torch.float16: 1.5e-3, torch.bfloat16: 1.2e-2}


def run_oracle(case, exact=False):
    q, k, v = golden_new_rows(case)
    k_slab, v_slab = case["k_slab"].clone(), case["v_slab"].clone()
    if case["save_kv_cache"]:
        orc.kv_scatter_ref(k_slab, v_slab, case["out_cache_loc"], k, v)
    args = (q, k_slab, v_slab, case["req_to_token"], case["req_pool_indices"], case["seq_lens"])
    if case["kind"] == "prefill":
        args += (case["prefix_lens"], case["extended_lens"])
        out = (orc.prefill_exact if exact else orc.prefill_ref)(*args)
    else:
        out = (orc.decode_exact if exact else orc.decode_ref)(*args)
    return out, k_slab, v_slab


@pytest.mark.parametrize("name", golden_names("decode_") + golden_names("prefill_"))
def test_ref_restatement_is_bit_exact(name):
    case = load_golden(name)
    out, k_slab, v_slab = run_oracle(case)
    assert out.dtype == case["out"].dtype and out.shape == case["out"].shape
    assert torch.equal(out, case["out"])
    loc = case["out_cache_loc"].long()
    if case["save_kv_cache"]:
        assert torch.equal(k_slab[loc], case["k_rows_after"])
        assert torch.equal(v_slab[loc], case["v_rows_after"])
    assert case["rest_unchanged"]
    keep = torch.ones(k_slab.shape[0], dtype=torch.bool)
    keep[loc] = False
    assert torch.equal(k_slab[keep], case["k_slab"][keep])
    assert torch.equal(v_slab[keep], case["v_slab"][keep])


@pytest.mark.parametrize("name", golden_names("decode_") + golden_names("prefill_"))
def test_exact_definition_matches_reference(name):
    case = load_golden(name)
    out, _, _ = run_oracle(case, exact=True)
    tol = EXACT_TOL[case["out"].dtype]
    err = (out - case["out"].double()).abs().max().item()
    assert err <= tol, f"{name}: max abs err {err:.3e} > {tol:.1e}"


def test_gather_indices_bit_exact():
    """The slot vector of a request is the page-table prefix, widened to int64."""
    case = load_golden("decode_q15_bf16")
    for i in range(len(case["seq_lens"])):
        r, n = int(case["req_pool_indices"][i]), int(case["seq_lens"][i])
        slots, k_i, v_i = orc.gather_request_kv(case["k_slab"], case["v_slab"], case["req_to_token"], r, n)
        assert slots.dtype == torch.int64 and slots.tolist() == case["req_to_token"][r, :n].tolist()
        assert torch.equal(k_i, case["k_slab"][slots]) and torch.equal(v_i, case["v_slab"][slots])


def test_scheduler_trace_replays():
    """Every ForwardBatch the reference scheduler produced (prefix reuse, two
    admission waves, recycled slots) replays through the oracle bit for bit."""
    tr = load_golden("sched_trace")
    assert any(b["is_prefill"] and int(b["prefix_lens"].max()) > 0 for b in tr["batches"])
    for b in tr["batches"]:
        k_slab, v_slab = b["k_slab_after"].clone(), b["v_slab_after"].clone()
        out = orc.forward_ref(
            b["q"], b["k"], b["v"], k_slab, v_slab, b["req_to_token"], b["req_pool_indices"],
            b["seq_lens"], b["out_cache_loc"], b["is_prefill"], b["prefix_lens"], b["extended_lens"],
        )
        assert torch.equal(out, b["out"])
        assert torch.equal(k_slab, b["k_slab_after"]) and torch.equal(v_slab, b["v_slab_after"])


def test_merge_states_exact_is_consistent():
    """Attention over a token set == merge of attention over two halves."""
    case = load_golden("decode_q15_f32")
    q, k, v = golden_new_rows(case)
    full, lse = orc.decode_exact(q, case["k_slab"], case["v_slab"], case["req_to_token"],
                                 case["req_pool_indices"], case["seq_lens"], return_lse=True)
    table = case["req_to_token"]
    rows = case["req_pool_indices"]
    seq = case["seq_lens"]
    parts_o, parts_lse = [], []
    for half in range(2):
        t = torch.zeros_like(table)
        lens = []
        for i in range(len(seq)):
            sel = table[rows[i], : int(seq[i])][half::2]
            t[rows[i], : len(sel)] = sel
            lens.append(len(sel))
        o, l = orc.decode_exact(q, case["k_slab"], case["v_slab"], t, rows, torch.tensor(lens), return_lse=True)
        parts_o.append(o.view(len(seq), case["n_q_heads"], -1))
        parts_lse.append(l)
    merged, mlse = orc.merge_states_exact(torch.stack(parts_o), torch.stack(parts_lse))
    assert torch.allclose(merged, full, atol=1e-12)
    assert torch.allclose(mlse, lse, atol=1e-12)
