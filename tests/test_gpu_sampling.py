"""Next-token selection ON THE DEVICE (the serving loop samples there) against the distributions
captured from the reference's sample.py on the CPU (tests/golden/sampling.pt).

What is exact on the device: the greedy ids, and WHICH tokens survive temperature / top-k /
top-p / repetition penalty (the support of the distribution).  The probabilities themselves go
through the device's exp and its sum order, which may differ from the CPU's in the last bits,
so they are held to rtol 2e-6 -- "parity unpinned" to the
bit for the device softmax, which no file of the reference covers; the CPU test
(tests/test_sampling.py) holds the same code to the reference bit for bit.
"""
import pytest
import torch

from conftest import load_golden
from baby_sglang_amd import sampling as smp

pytestmark = pytest.mark.gpu

CASES = load_golden("sampling")["cases"]


def test_device_distribution_has_the_reference_s_support_and_values(gpu):
    for c in CASES:
        logits = smp.apply_repetition_penalty_(c["logits"].to(gpu), c["prev_ids"], c["repetition_penalty"])
        probs = smp.next_token_probs(logits, c["temperature"], c["top_k"], c["top_p"])
        assert probs.is_cuda
        got, want = probs.cpu(), c["probs"]
        what = (c["temperature"], c["top_k"], c["top_p"], c["repetition_penalty"])
        # a token may only change sides of the nucleus cut if its cumulative mass sits on the cut
        flipped = (got > 0) != (want > 0)
        if bool(flipped.any()):
            assert c["top_p"] < 1.0 and int(flipped.sum()) <= got.shape[0], what
            assert float(want[flipped].max()) < 1e-3 and float(got[flipped].max()) < 1e-3, what
        same = ~flipped
        assert torch.allclose(got[same], want[same], rtol=2e-6, atol=1e-9), what
        assert torch.allclose(got.sum(-1), torch.ones(got.shape[0]), atol=1e-5)


def test_device_greedy_ids_and_caller_tensor_untouched(gpu):
    for c in CASES:
        logits = c["logits"].to(gpu)
        before = logits.clone()
        ids = smp.sample_next_ids(logits, False, c["temperature"], c["top_k"], c["top_p"], c["prev_ids"], c["repetition_penalty"])
        assert ids.is_cuda and torch.equal(ids.cpu(), c["greedy_ids"])
        assert torch.equal(logits, before)


def test_device_draws_stay_inside_the_kept_set_and_follow_the_generator(gpu):
    c = next(c for c in CASES if c["top_k"] == 10)
    logits = c["logits"].to(gpu)
    g = torch.Generator(device=gpu).manual_seed(1)
    a = torch.stack([smp.sample_next_ids(logits, True, c["temperature"], c["top_k"], c["top_p"], generator=g) for _ in range(50)])
    kept = (c["probs"] > 0).to(gpu)
    assert bool(kept.gather(1, a.t()).all())
    g2 = torch.Generator(device=gpu).manual_seed(1)
    b = torch.stack([smp.sample_next_ids(logits, True, c["temperature"], c["top_k"], c["top_p"], generator=g2) for _ in range(50)])
    assert torch.equal(a, b)


def test_a_page_table_that_names_another_requests_new_slot(gpu):
    """Reference order: every new row is written before any request reads the cache
    (attn_backend.py:20-23), so a request whose page-table row names ANOTHER request's new slot sees
    that request's new row.  With default settings the backend gives exactly that (the new rows are
    a launch of their own ahead of attention).  A producer that wrongly vouches for fresh slots
    (``private_decode_slots``) is caught by the opt-in debug check."""
    import types

    import baby_sglang_amd as pkg
    from oracle import attn_oracle as orc
    from test_gpu_parity import assert_close, random_decode_problem

    cpu, r2t, kv = random_decode_problem(gpu, torch.bfloat16, 12, 2, 128, [20, 30, 9], seed=2)
    rows, seq = cpu["req_pool_indices"].to(gpu), cpu["seq_lens"].to(gpu)
    loc = r2t.req_to_token[rows, seq - 1].contiguous()
    r2t.req_to_token[rows[1], 3] = loc[0]  # request 1 also names request 0's new slot
    fb = pkg.SimplifiedForwardBatch.create_decode_batch(torch.zeros(3, dtype=torch.int64, device=gpu), rows, seq, loc, r2t, kv)
    assert fb.private_decode_slots is False
    layer = types.SimpleNamespace(layer_id=0, scaling=128**-0.5)
    gen = torch.Generator().manual_seed(3)
    q = cpu["q"].to(gpu)
    k = (4 * torch.randn((3, 2, 128), generator=gen)).to(torch.bfloat16)
    v = (4 * torch.randn((3, 2, 128), generator=gen)).to(torch.bfloat16)
    out = pkg.SimpleAttentionBackend.forward(q, k.to(gpu), v.to(gpu), layer, fb)
    k_ref, v_ref = cpu["k_slab"].clone(), cpu["v_slab"].clone()
    orc.kv_scatter_ref(k_ref, v_ref, loc.cpu(), k, v)
    table = r2t.req_to_token.cpu()
    want = orc.decode_exact(cpu["q"], k_ref, v_ref, table, cpu["req_pool_indices"], cpu["seq_lens"])
    assert_close(out, want, torch.bfloat16, "aliased page table, default settings")
    # (and the new row does matter to request 1: without it the answer is another one)
    stale = orc.decode_exact(cpu["q"], cpu["k_slab"], cpu["v_slab"], table, cpu["req_pool_indices"], cpu["seq_lens"])
    assert (want[1] - stale[1]).abs().max().item() > 0.05


def test_fused_kv_write_invariant_check(gpu):
    """The opt-in debug check of the decode backend: a batch that says its new slots are fresh
    while one request's page-table row names another request's new slot is refused; the
    scheduler-shaped batch passes."""
    import types

    import baby_sglang_amd as pkg
    from test_gpu_parity import random_decode_problem

    cpu, r2t, kv = random_decode_problem(gpu, torch.bfloat16, 12, 2, 128, [20, 30, 9], seed=2)
    rows, seq = cpu["req_pool_indices"].to(gpu), cpu["seq_lens"].to(gpu)
    loc = r2t.req_to_token[rows, seq - 1].contiguous()
    fb = pkg.SimplifiedForwardBatch.create_decode_batch(torch.zeros(3, dtype=torch.int64, device=gpu), rows, seq, loc, r2t, kv)
    fb.private_decode_slots = True
    layer = types.SimpleNamespace(layer_id=0, scaling=128**-0.5)
    q = cpu["q"].to(gpu)
    k = torch.randn((3, 2, 128), device=gpu).to(torch.bfloat16)
    old = pkg.SimpleAttentionBackend.check_fused_kv_invariant
    pkg.SimpleAttentionBackend.check_fused_kv_invariant = True
    try:
        pkg.SimpleAttentionBackend.forward(q, k, k, layer, fb)  # fresh slots: fine
        r2t.req_to_token[rows[1], 3] = loc[0]  # request 1 now also names request 0's new slot
        with pytest.raises(ValueError, match="named by another request"):
            pkg.SimpleAttentionBackend.forward(q, k, k, layer, fb)
    finally:
        pkg.SimpleAttentionBackend.check_fused_kv_invariant = old
