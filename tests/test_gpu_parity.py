"""Parity of the HIP path with the reference (goldens) and the float64 oracle.

Everything here runs on a real MI355X and goes through the product API
(pools -> SimplifiedForwardBatch -> SimpleAttentionBackend.forward), i.e. through
the C ABI of libbsgl_attn.so.  Tolerances (kernel output in dtype vs float64
oracle on the same rounded inputs):
  fp32  rtol 1e-4 atol 1e-5
  fp16  rtol 1e-3 atol 1e-3   (the bar the task states)
  bf16  rtol 8e-3 atol 2e-3   (bf16 output rounding is 2^-8 relative; the matrix-core
                               kernels feed the probabilities as hi + lo bf16 pairs so
                               that their rounding does not add to it)
KV rows written by the scatter kernel are compared bit for bit.
"""
import math
import types

import pytest
import torch

from conftest import golden_names, golden_new_rows, load_golden
from oracle import attn_oracle as orc

pytestmark = pytest.mark.gpu

TOL = {
    torch.float32: dict(rtol=1e-4, atol=1e-5),
    torch.float16: dict(rtol=1e-3, atol=1e-3),
    torch.bfloat16: dict(rtol=8e-3, atol=2e-3),
}


def assert_close(got, want64, dtype, what=""):
    got64 = got.detach().double().cpu()
    tol = TOL[dtype]
    ok = torch.isclose(got64, want64, **tol)
    if not bool(ok.all()):
        err = (got64 - want64).abs()
        bad = int((~ok).sum())
        raise AssertionError(
            f"{what}: {bad}/{ok.numel()} elements outside rtol={tol['rtol']} atol={tol['atol']}; "
            f"max abs err {err.max().item():.3e} at {int(err.argmax())}"
        )


def product_forward(case, dev, layout="interleaved", variant=None, private_slots=False):
    """Run one golden case through the product backend on the GPU (``private_slots``: the batch says
    that its new slots are fresh, so that the decode kernel stores the new rows itself)."""
    import baby_sglang_amd as pkg

    hkv, dim = case["n_kv_heads"], case["head_dim"]
    dtype = case["q"].dtype
    n_slots = case["k_slab"].shape[0]
    r2t = pkg.ReqToTokenPool(case["req_to_token"].shape[0], case["req_to_token"].shape[1], str(dev))
    r2t.req_to_token.copy_(case["req_to_token"])
    kv = pkg.MHATokenToKVPool(n_slots - 1, dtype, hkv, dim, layer_num=3, device=str(dev), kv_layout=layout)
    layer_id = 2
    kv.k_buffer[layer_id].copy_(case["k_slab"])
    kv.v_buffer[layer_id].copy_(case["v_slab"])
    q, k, v = (t.to(dev) for t in (case["q"], case["k"], case["qkv"]))
    v = v[:, case["v_col0"] :].view(v.shape[0], hkv, dim)  # strided view, as in the model
    common = dict(
        input_ids=torch.zeros(q.shape[0], dtype=torch.int64, device=dev),
        req_pool_indices=case["req_pool_indices"].to(dev), seq_lens=case["seq_lens"].to(dev),
        out_cache_loc=case["out_cache_loc"].to(dev), req_to_token_pool=r2t, token_to_kv_pool=kv,
    )
    if case["kind"] == "prefill":
        fb = pkg.SimplifiedForwardBatch.create_prefill_batch(prefix_lens=case["prefix_lens"].to(dev), **common)
    else:
        fb = pkg.SimplifiedForwardBatch.create_decode_batch(**common)
        fb.private_decode_slots = private_slots
    layer = types.SimpleNamespace(layer_id=layer_id, scaling=dim**-0.5)
    old = pkg.SimpleAttentionBackend.decode_variant
    if variant is not None:
        pkg.SimpleAttentionBackend.decode_variant = variant
    try:
        out = pkg.SimpleAttentionBackend.forward(q, k, v, layer, fb, save_kv_cache=case["save_kv_cache"])
    finally:
        pkg.SimpleAttentionBackend.decode_variant = old
    torch.cuda.synchronize()
    return out, kv.k_buffer[layer_id], kv.v_buffer[layer_id]


def oracle_exact(case):
    q, k, v = golden_new_rows(case)
    k_slab, v_slab = case["k_slab"].clone(), case["v_slab"].clone()
    if case["save_kv_cache"]:
        orc.kv_scatter_ref(k_slab, v_slab, case["out_cache_loc"], k, v)
    args = (q, k_slab, v_slab, case["req_to_token"], case["req_pool_indices"], case["seq_lens"])
    if case["kind"] == "prefill":
        return orc.prefill_exact(*args, case["prefix_lens"], case["extended_lens"]), k_slab, v_slab
    return orc.decode_exact(*args), k_slab, v_slab


@pytest.mark.parametrize("layout", ["interleaved", "split"])
@pytest.mark.parametrize("name,private_slots", [(n, False) for n in golden_names("decode_") + golden_names("prefill_")]
                         + [(n, True) for n in golden_names("decode_")])
def test_golden_cases(gpu, name, layout, private_slots):
    """(decode cases twice: new rows written by a launch of their own -- the default, right for any
    page table -- and by the decode kernel, as for a producer that vouches for fresh slots)"""
    case = load_golden(name)
    dtype = case["q"].dtype
    out, k_slab, v_slab = product_forward(case, gpu, layout, private_slots=private_slots)
    want, k_want, v_want = oracle_exact(case)
    assert out.dtype == dtype and tuple(out.shape) == tuple(case["out"].shape)
    assert_close(out, want, dtype, f"{name} vs float64 oracle")
    # and directly against what the reference produced (its own rounding included): both sides are
    # rounded to the dtype, so they may differ by ONE unit in the last place of a value of that size
    # (observed on these cases, tools/golden_distance.py: bf16 7.8e-3 = 2^-7, f16 9.8e-4 = 2^-10, f32 9.5e-7)
    ulp = {torch.float32: 2e-6, torch.float16: 2.0**-10, torch.bfloat16: 2.0**-7}[dtype]
    ref = case["out"].double()
    assert bool(((out.double().cpu() - ref).abs() <= ulp * ref.abs().clamp_min(1.0)).all())
    # KV write: bit exact, untouched slots untouched
    assert torch.equal(k_slab.cpu(), k_want) and torch.equal(v_slab.cpu(), v_want)


@pytest.mark.parametrize("private_slots", [False, True])
def test_scheduler_trace_on_gpu(gpu, private_slots):
    """Replay the reference scheduler's ForwardBatch sequence (prefix reuse,
    recycled slots) through the product backend, one persistent pool; as the bare drop-in
    (new rows of a decode step written ahead of attention) and with the word that the reference's
    scheduler could give but its descriptor cannot carry (fresh slots: the kernel stores them)."""
    import baby_sglang_amd as pkg

    tr = load_golden("sched_trace")
    hkv, dim = tr["n_kv_heads"], tr["head_dim"]
    first = tr["batches"][0]
    n_slots = first["k_slab_after"].shape[0]
    r2t = pkg.ReqToTokenPool(first["req_to_token"].shape[0], first["req_to_token"].shape[1], str(gpu))
    kv = pkg.MHATokenToKVPool(n_slots - 1, first["q"].dtype, hkv, dim, layer_num=1, device=str(gpu))
    kv.k_buffer[0].zero_()
    kv.v_buffer[0].zero_()
    layer = types.SimpleNamespace(layer_id=0, scaling=dim**-0.5)
    for step, b in enumerate(tr["batches"]):
        r2t.req_to_token.copy_(b["req_to_token"])
        common = dict(
            input_ids=torch.zeros(b["q"].shape[0], dtype=torch.int64, device=gpu),
            req_pool_indices=b["req_pool_indices"].to(gpu), seq_lens=b["seq_lens"].to(gpu),
            out_cache_loc=b["out_cache_loc"].to(gpu), req_to_token_pool=r2t, token_to_kv_pool=kv,
        )
        if b["is_prefill"]:
            fb = pkg.SimplifiedForwardBatch.create_prefill_batch(prefix_lens=b["prefix_lens"].to(gpu), **common)
        else:
            fb = pkg.SimplifiedForwardBatch.create_decode_batch(**common)
            fb.private_decode_slots = private_slots
        out = pkg.SimpleAttentionBackend.forward(b["q"].to(gpu), b["k"].to(gpu), b["v"].to(gpu), layer, fb)
        assert (out.double().cpu() - b["out"].double()).abs().max().item() <= 1e-5, f"batch {step}"
        # every slot the scheduler has handed out so far holds the reference's bytes
        live = b["req_to_token"][b["req_pool_indices"]].flatten().long().unique()
        live = live[live > 0]  # 0 marks unused page-table entries; slot 0 is never written
        assert torch.equal(kv.k_buffer[0].cpu()[live], b["k_slab_after"][live]), f"batch {step}"
        assert torch.equal(kv.v_buffer[0].cpu()[live], b["v_slab_after"][live]), f"batch {step}"


# ----------------------------------------------------------------------------- seeded random cases
def random_decode_problem(dev, dtype, hq, hkv, dim, seq_lens, seed, layout="interleaved", slack=16):
    import baby_sglang_amd as pkg

    gen = torch.Generator().manual_seed(seed)
    batch = len(seq_lens)
    size = sum(seq_lens) + slack
    max_ctx = max(max(seq_lens), 1) + 5
    table = torch.zeros((batch + 1, max_ctx), dtype=torch.int32)
    perm = (torch.randperm(size, generator=gen) + 1).to(torch.int32)
    rows = torch.randperm(batch + 1, generator=gen)[:batch]
    used = 0
    for i, n in enumerate(seq_lens):
        table[rows[i], :n] = perm[used : used + n]
        used += n
    k_slab = torch.randn((size + 1, hkv, dim), generator=gen).to(dtype)
    v_slab = torch.randn((size + 1, hkv, dim), generator=gen).to(dtype)
    q = torch.randn((batch, hq, dim), generator=gen).to(dtype)
    r2t = pkg.ReqToTokenPool(batch + 1, max_ctx, str(dev))
    r2t.req_to_token.copy_(table)
    kv = pkg.MHATokenToKVPool(size, dtype, hkv, dim, layer_num=1, device=str(dev), kv_layout=layout)
    kv.k_buffer[0].copy_(k_slab)
    kv.v_buffer[0].copy_(v_slab)
    cpu = dict(q=q, k_slab=k_slab, v_slab=v_slab, req_to_token=table, req_pool_indices=rows.long(),
               seq_lens=torch.tensor(seq_lens, dtype=torch.int64))
    return cpu, r2t, kv


@pytest.mark.parametrize("variant", [1, 0])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("hq,hkv,dim", [(12, 2, 128), (14, 2, 64), (28, 4, 128), (4, 4, 64), (8, 1, 128), (6, 1, 128)])
def test_decode_random_ragged(gpu, dtype, hq, hkv, dim, variant):
    from baby_sglang_amd import _hip

    seq_lens = [1, 2, 63, 64, 65, 300, 1025, 7, 128, 129, 511]
    cpu, r2t, kv = random_decode_problem(gpu, dtype, hq, hkv, dim, seq_lens, seed=hq * 1000 + dim)
    want, want_lse = orc.decode_exact(cpu["q"], cpu["k_slab"], cpu["v_slab"], cpu["req_to_token"],
                                      cpu["req_pool_indices"], cpu["seq_lens"], return_lse=True)
    k_slab, v_slab = kv.get_kv_buffer(0)
    for n_splits in (1, 3, 16):
        lse = torch.empty((len(seq_lens), hq), dtype=torch.float32, device=gpu)
        out = _hip.paged_decode(cpu["q"].to(gpu), k_slab, v_slab, r2t.req_to_token,
                                cpu["req_pool_indices"].to(gpu), cpu["seq_lens"].to(gpu),
                                n_splits=n_splits, variant=variant, lse=lse)
        torch.cuda.synchronize()
        assert_close(out, want, dtype, f"decode splits={n_splits}")
        assert torch.allclose(lse.double().cpu(), want_lse, rtol=1e-4, atol=2e-3 if dtype != torch.float32 else 1e-4)


@pytest.mark.parametrize("waves", [1, 2, 4])
@pytest.mark.parametrize("dtype,hq,hkv,dim", [(torch.bfloat16, 12, 2, 128), (torch.float16, 28, 4, 128), (torch.bfloat16, 14, 2, 64)])
def test_pieces_shared_by_the_waves_of_a_workgroup(gpu, dtype, hq, hkv, dim, waves, knobs):
    """Small launches give a piece to two or four waves (tiles in turn, states added up in LDS):
    every width against the oracle, on lengths that leave some waves without a tile (1, 16, 17
    tokens), with a block border inside a piece (600, 1100 tokens in one piece) and with splits;
    output, log-sum-exp and the fused write of the new token."""
    from baby_sglang_amd import _hip

    knobs("BSGL_MFMA_WAVES", waves)
    seq_lens = [1, 16, 17, 33, 600, 1100, 64, 2047]
    cpu, r2t, kv = random_decode_problem(gpu, dtype, hq, hkv, dim, seq_lens, seed=11 * hq + waves)
    gen = torch.Generator().manual_seed(8)
    k_new = torch.randn((len(seq_lens), hkv, dim), generator=gen).to(dtype).to(gpu)
    v_new = torch.randn((len(seq_lens), hkv, dim), generator=gen).to(dtype).to(gpu)
    rows, seq = cpu["req_pool_indices"].to(gpu), cpu["seq_lens"].to(gpu)
    loc = r2t.req_to_token[rows, seq - 1].contiguous()
    k_ref, v_ref = cpu["k_slab"].clone(), cpu["v_slab"].clone()
    orc.kv_scatter_ref(k_ref, v_ref, loc.cpu(), k_new.cpu(), v_new.cpu())
    want, want_lse = orc.decode_exact(cpu["q"], k_ref, v_ref, cpu["req_to_token"], cpu["req_pool_indices"], cpu["seq_lens"],
                                      return_lse=True)
    for n_splits in (1, 4, 16):
        kv.k_buffer[0].copy_(cpu["k_slab"])
        kv.v_buffer[0].copy_(cpu["v_slab"])
        k_slab, v_slab = kv.get_kv_buffer(0)
        lse = torch.empty((len(seq_lens), hq), dtype=torch.float32, device=gpu)
        out = _hip.paged_decode(cpu["q"].to(gpu), k_slab, v_slab, r2t.req_to_token, rows, seq, n_splits=n_splits, lse=lse,
                                k_new=k_new, v_new=v_new, out_cache_loc=loc, fuse_kv_write=True)
        torch.cuda.synchronize()
        assert_close(out, want, dtype, f"waves={waves} splits={n_splits}")
        assert torch.allclose(lse.double().cpu(), want_lse, rtol=1e-4, atol=2e-3)
        assert torch.equal(k_slab.cpu(), k_ref) and torch.equal(v_slab.cpu(), v_ref)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("hq", [14, 4, 16])
def test_two_kv_heads_of_64_on_the_matrix_core_kernel(gpu, dtype, hq):
    """Two kv heads of 64 values (Qwen2.5-0.5B: 14 q heads) streamed as one 128-wide kv head:
    forced matrix-core variant (and the automatic choice, which folds too) against the
    float64 oracle, with the new token's rows written by the kernel, f32 output + lse, every
    split count incl. the unsplit direct path, and a batch large enough for the automatic choice."""
    from baby_sglang_amd import _hip

    hkv, dim = 2, 64
    seq_lens = [1, 2, 63, 64, 65, 300, 1025, 7, 128, 129, 511]
    cpu, r2t, kv = random_decode_problem(gpu, dtype, hq, hkv, dim, seq_lens, seed=hq * 100 + dim)
    rows, seq = cpu["req_pool_indices"].to(gpu), cpu["seq_lens"].to(gpu)
    gen = torch.Generator().manual_seed(8)
    k_new = torch.randn((len(seq_lens), hkv, dim), generator=gen).to(dtype)
    v_new = torch.randn((len(seq_lens), hkv, dim), generator=gen).to(dtype)
    loc = r2t.req_to_token[rows, seq - 1].contiguous()
    k_ref, v_ref = cpu["k_slab"].clone(), cpu["v_slab"].clone()
    orc.kv_scatter_ref(k_ref, v_ref, loc.cpu(), k_new, v_new)
    want, want_lse = orc.decode_exact(cpu["q"], k_ref, v_ref, cpu["req_to_token"], cpu["req_pool_indices"],
                                      cpu["seq_lens"], return_lse=True)
    for n_splits in (1, 3, 16):
        kv.k_buffer[0].copy_(cpu["k_slab"])
        kv.v_buffer[0].copy_(cpu["v_slab"])
        k_slab, v_slab = kv.get_kv_buffer(0)
        out = _hip.paged_decode(cpu["q"].to(gpu), k_slab, v_slab, r2t.req_to_token, rows, seq, n_splits=n_splits, variant=2,
                                k_new=k_new.to(gpu), v_new=v_new.to(gpu), out_cache_loc=loc, fuse_kv_write=True)
        assert_close(out, want, dtype, f"folded heads, splits={n_splits}")
        assert torch.equal(k_slab.cpu(), k_ref) and torch.equal(v_slab.cpu(), v_ref)
        lse = torch.empty((len(seq_lens), hq), dtype=torch.float32, device=gpu)
        out32 = _hip.paged_decode(cpu["q"].to(gpu), k_slab, v_slab, r2t.req_to_token, rows, seq, n_splits=n_splits, variant=2,
                                  lse=lse, out_dtype=torch.float32)
        assert out32.dtype == torch.float32
        assert_close(out32, want, dtype, f"folded heads f32 out, splits={n_splits}")
        assert torch.allclose(lse.double().cpu(), want_lse, rtol=1e-4, atol=2e-3)
    # 40 requests: the automatic choice folds as well; same answer as the generic kernel's
    seq_lens = [17 + 13 * i for i in range(40)]
    cpu, r2t, kv = random_decode_problem(gpu, dtype, hq, hkv, dim, seq_lens, seed=3)
    want = orc.decode_exact(cpu["q"], cpu["k_slab"], cpu["v_slab"], cpu["req_to_token"], cpu["req_pool_indices"], cpu["seq_lens"])
    k_slab, v_slab = kv.get_kv_buffer(0)
    args = (cpu["q"].to(gpu), k_slab, v_slab, r2t.req_to_token, cpu["req_pool_indices"].to(gpu), cpu["seq_lens"].to(gpu))
    assert_close(_hip.paged_decode(*args), want, dtype, "folded heads, automatic")
    assert_close(_hip.paged_decode(*args, variant=1), want, dtype, "generic kernel")


@pytest.mark.parametrize("dtype,hq,hkv", [(torch.bfloat16, 12, 2), (torch.float16, 28, 4), (torch.bfloat16, 6, 1)])
def test_fused_kv_write_and_merge(gpu, dtype, hq, hkv, knobs):
    """The decode kernel that stores the new token itself and merges its own
    splits must agree bit for bit with the three-launch form (scatter kernel,
    decode, merge kernel) on output, lse and slab contents -- twice in a row on
    one workspace, the second call with fewer requests.  (Equal pieces per
    request in both forms: the three-launch form cannot cut by length.)"""
    from baby_sglang_amd import _hip

    dim = 128
    seq_lens = [1, 16, 17, 333, 64, 1500, 129]
    cpu, r2t, kv = random_decode_problem(gpu, dtype, hq, hkv, dim, seq_lens, seed=99 + hq)
    gen = torch.Generator().manual_seed(5)
    k_new = torch.randn((len(seq_lens), hkv, dim), generator=gen).to(dtype).to(gpu)
    qkv = torch.randn((len(seq_lens), (hq + 2 * hkv) * dim), generator=gen).to(dtype).to(gpu)
    v_new = qkv[:, (hq + hkv) * dim :].view(len(seq_lens), hkv, dim)  # strided rows
    rows, seq = cpu["req_pool_indices"].to(gpu), cpu["seq_lens"].to(gpu)
    loc = r2t.req_to_token[rows, seq - 1].contiguous()
    q = cpu["q"].to(gpu)
    ws = _hip.new_decode_workspace(_hip.decode_workspace_bytes(len(seq_lens), hq, dim, 7), gpu)
    for n_req in (len(seq_lens), 3):
        results = []
        for fused in (True, False):
            kv.k_buffer[0].copy_(cpu["k_slab"])
            kv.v_buffer[0].copy_(cpu["v_slab"])
            knobs("BSGL_NO_FUSED_MERGE", 0 if fused else 1)
            k_slab, v_slab = kv.get_kv_buffer(0)
            lse = torch.empty((n_req, hq), dtype=torch.float32, device=gpu)
            out = _hip.paged_decode(q[:n_req], k_slab, v_slab, r2t.req_to_token, rows[:n_req], seq[:n_req],
                                    n_splits=7, variant=2, lse=lse, workspace=ws,
                                    k_new=k_new[:n_req], v_new=v_new[:n_req], out_cache_loc=loc[:n_req],
                                    fuse_kv_write=fused, uniform_splits=True)
            torch.cuda.synchronize()
            results.append((out.clone(), lse.clone(), k_slab.clone(), v_slab.clone()))
        for a, b in zip(*results):
            assert torch.equal(a, b)
        # tickets are back at zero for the next call
        assert int(ws[: _hip.lib.bsgl_decode_counter_bytes()].max()) == 0
        # and the result is right
        k_ref, v_ref = cpu["k_slab"].clone(), cpu["v_slab"].clone()
        orc.kv_scatter_ref(k_ref, v_ref, loc[:n_req].cpu(), k_new[:n_req].cpu(), v_new[:n_req].cpu())
        want = orc.decode_exact(cpu["q"][:n_req], k_ref, v_ref, cpu["req_to_token"], rows[:n_req].cpu(), seq[:n_req].cpu())
        assert_close(results[0][0], want, dtype, "fused decode")
        assert torch.equal(results[0][2].cpu(), k_ref) and torch.equal(results[0][3].cpu(), v_ref)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_length_balanced_splits(gpu, dtype):
    """Pieces of one batch-wide length: a request gets between 1 and n_splits of
    them.  One very long request among short ones, zero-length requests, and a
    second call with other lengths on the same workspace (tickets back at zero)."""
    from baby_sglang_amd import _hip

    hq, hkv, dim = 12, 2, 128
    ws = _hip.new_decode_workspace(_hip.decode_workspace_bytes(9, hq, dim, 16), gpu)
    for seed, seq_lens in enumerate(([3000, 5, 17, 0, 200, 64, 1, 900, 31], [40, 40, 0, 2100], [1, 1, 1])):
        cpu, r2t, kv = random_decode_problem(gpu, dtype, hq, hkv, dim, seq_lens, seed=31 + seed)
        want, want_lse = orc.decode_exact(cpu["q"], cpu["k_slab"], cpu["v_slab"], cpu["req_to_token"],
                                          cpu["req_pool_indices"], cpu["seq_lens"], return_lse=True)
        k_slab, v_slab = kv.get_kv_buffer(0)
        for n_splits in (16, 5, 2):
            lse = torch.empty((len(seq_lens), hq), dtype=torch.float32, device=gpu)
            out = _hip.paged_decode(cpu["q"].to(gpu), k_slab, v_slab, r2t.req_to_token,
                                    cpu["req_pool_indices"].to(gpu), cpu["seq_lens"].to(gpu),
                                    n_splits=n_splits, variant=2, lse=lse, workspace=ws)
            torch.cuda.synchronize()
            assert_close(out, want, dtype, f"balanced splits={n_splits} lens={seq_lens}")
            live = torch.tensor(seq_lens) > 0
            assert torch.allclose(lse.double().cpu()[live], want_lse[live], rtol=1e-4, atol=2e-3)
            assert int(ws[: _hip.lib.bsgl_decode_counter_bytes()].max()) == 0


@pytest.mark.parametrize("length", [1, 15, 16, 777, 2048])
def test_equal_lengths_take_the_closed_form_of_the_piece_search(gpu, length, knobs):
    """A batch of at most 64 requests of ONE length finds its pieces in closed form instead of the
    wave-wide search.  Both must cut a request the same way: the first rows of a 65-request batch
    of that length (searched: more than 64 rows) and the same rows as a batch of their own
    (closed form) give the same bits, for every split count, and agree with the oracle."""
    from baby_sglang_amd import _hip

    # (one wave per piece whatever the batch: small launches otherwise share a piece among
    # several waves, which changes the order of the sums, not the pieces)
    knobs("BSGL_MFMA_WAVES", 1)
    hq, hkv, dim, dtype = 12, 2, 128, torch.bfloat16
    cpu, r2t, kv = random_decode_problem(gpu, dtype, hq, hkv, dim, [length] * 65, seed=77 + length)
    want = orc.decode_exact(cpu["q"], cpu["k_slab"], cpu["v_slab"], cpu["req_to_token"], cpu["req_pool_indices"], cpu["seq_lens"])
    k_slab, v_slab = kv.get_kv_buffer(0)
    q, rows, seq = cpu["q"].to(gpu), cpu["req_pool_indices"].to(gpu), cpu["seq_lens"].to(gpu)
    for n_splits in (16, 7, 2):
        # (the number of pieces wanted per request pinned: small batches are otherwise allowed finer cuts)
        old = _hip.set_knob("BSGL_DECODE_TARGET", (n_splits + 1) // 2)
        try:
            searched = _hip.paged_decode(q, k_slab, v_slab, r2t.req_to_token, rows, seq, n_splits=n_splits, variant=2)
            assert_close(searched, want, dtype, f"65 rows of {length}, splits {n_splits}")
            for n in (1, 9, 64):
                closed = _hip.paged_decode(q[:n], k_slab, v_slab, r2t.req_to_token, rows[:n], seq[:n], n_splits=n_splits, variant=2)
                assert torch.equal(closed, searched[:n]), f"{n} rows of {length}, splits {n_splits}"
        finally:
            _hip.set_knob("BSGL_DECODE_TARGET", old)
        # the cuts a small batch gets by itself: same sums in another order
        for n in (1, 2, 3, 8):
            closed = _hip.paged_decode(q[:n], k_slab, v_slab, r2t.req_to_token, rows[:n], seq[:n], n_splits=n_splits, variant=2)
            assert_close(closed, want[:n], dtype, f"{n} rows of {length}, splits {n_splits}")


@pytest.mark.parametrize("seq_lens", [[4000, 300, 2500, 17, 3300], [3900, 4090, 4001], [4095] * 8, [2047] * 7 + [16],
                                      [260, 255, 257, 3000], [4095], [4095, 1]])
@pytest.mark.parametrize("hq,hkv", [(12, 2), (28, 4)])
def test_small_batches_of_long_requests(gpu, seq_lens, hq, hkv):
    """Up to sixteen (request, kv head) units: the host allows the kernel finer cuts than for a larger
    batch (pieces of no less than 256 tokens, up to twice as many per request), decided on the device
    from the lengths -- equal lengths in closed form, ragged ones by the search.  Against the oracle,
    with the default number of splits for the batch and with the new token's K/V rows written by the
    kernel."""
    from baby_sglang_amd import _hip

    dim, dtype = 128, torch.bfloat16
    cpu, r2t, kv = random_decode_problem(gpu, dtype, hq, hkv, dim, seq_lens, seed=sum(seq_lens) + hq)
    batch = len(seq_lens)
    gen = torch.Generator().manual_seed(5)
    k_new = torch.randn((batch, hkv, dim), generator=gen).to(dtype)
    v_new = torch.randn((batch, hkv, dim), generator=gen).to(dtype)
    loc = cpu["req_to_token"][cpu["req_pool_indices"], cpu["seq_lens"] - 1].to(torch.int32)
    orc.kv_scatter_ref(cpu["k_slab"], cpu["v_slab"], loc, k_new, v_new)
    want, want_lse = orc.decode_exact(cpu["q"], cpu["k_slab"], cpu["v_slab"], cpu["req_to_token"],
                                      cpu["req_pool_indices"], cpu["seq_lens"], return_lse=True)
    k_slab, v_slab = kv.get_kv_buffer(0)
    n_splits = _hip.decode_num_splits(batch, hkv, r2t.req_to_token.shape[1])
    lse = torch.empty((batch, hq), dtype=torch.float32, device=gpu)
    out = _hip.paged_decode(cpu["q"].to(gpu), k_slab, v_slab, r2t.req_to_token, cpu["req_pool_indices"].to(gpu),
                            cpu["seq_lens"].to(gpu), n_splits=n_splits, lse=lse, k_new=k_new.to(gpu), v_new=v_new.to(gpu),
                            out_cache_loc=loc.to(gpu), fuse_kv_write=True)
    assert_close(out, want, dtype, f"{seq_lens} x {hq}/{hkv}")
    assert (lse.cpu().double() - want_lse).abs().max().item() < 2e-3
    assert torch.equal(k_slab.cpu(), cpu["k_slab"]) and torch.equal(v_slab.cpu(), cpu["v_slab"])


@pytest.mark.parametrize("case", [
    ("64 short, exact hint", 64, 256, 256, 12, 2, 128),
    ("128 short, exact hint", 128, 200, 200, 12, 2, 128),
    ("300 short: two waves per piece", 300, 100, 100, 12, 2, 128),
    ("700 short: one wave per piece", 700, 60, 60, 12, 2, 128),
    ("7B shapes", 40, 256, 256, 28, 4, 128),
    ("two kv heads of 64 values, folded", 64, 256, 256, 14, 2, 64),
    ("folded, more requests", 200, 180, 180, 14, 2, 64),
    ("folded, a hint that is too small", 130, 1500, 100, 14, 2, 64),
    ("hint larger than 256: no effect", 64, 700, 700, 12, 2, 128),
    ("a hint that is too small", 48, 2000, 64, 12, 2, 128),
    ("a hint that is far too small", 3, 4000, 1, 12, 2, 128),
])
def test_length_hint_shapes_the_launch_only(gpu, case):
    """``max_seq_len`` (bsgl_decode_call) is a caller's word that no request is longer; batches of
    short requests are then launched as what they are (every request one piece, four or two waves
    per piece).  Results must not depend on it: ragged batches against the oracle with a true hint,
    and with hints that are wrong."""
    from baby_sglang_amd import _hip

    name, batch, longest, hint, hq, hkv, dim = case
    dtype = torch.bfloat16
    gen = torch.Generator().manual_seed(batch + longest)
    seq_lens = torch.randint(1, longest + 1, (batch,), generator=gen).tolist()
    seq_lens[0] = longest
    cpu, r2t, kv = random_decode_problem(gpu, dtype, hq, hkv, dim, seq_lens, seed=batch * 7 + hint)
    k_new = torch.randn((batch, hkv, dim), generator=gen).to(dtype)
    v_new = torch.randn((batch, hkv, dim), generator=gen).to(dtype)
    loc = cpu["req_to_token"][cpu["req_pool_indices"], cpu["seq_lens"] - 1].to(torch.int32)
    orc.kv_scatter_ref(cpu["k_slab"], cpu["v_slab"], loc, k_new, v_new)
    want, want_lse = orc.decode_exact(cpu["q"], cpu["k_slab"], cpu["v_slab"], cpu["req_to_token"],
                                      cpu["req_pool_indices"], cpu["seq_lens"], return_lse=True)
    k_slab, v_slab = kv.get_kv_buffer(0)
    lse = torch.empty((batch, hq), dtype=torch.float32, device=gpu)
    out = _hip.paged_decode(cpu["q"].to(gpu), k_slab, v_slab, r2t.req_to_token, cpu["req_pool_indices"].to(gpu),
                            cpu["seq_lens"].to(gpu), lse=lse, k_new=k_new.to(gpu), v_new=v_new.to(gpu),
                            out_cache_loc=loc.to(gpu), fuse_kv_write=True, max_seq_len=hint)
    assert_close(out, want, dtype, name)
    assert (lse.cpu().double() - want_lse).abs().max().item() < 2e-3, name
    assert torch.equal(k_slab.cpu(), cpu["k_slab"]) and torch.equal(v_slab.cpu(), cpu["v_slab"]), name


def test_decode_empty_and_zero_length(gpu):
    from baby_sglang_amd import _hip

    cpu, r2t, kv = random_decode_problem(gpu, torch.bfloat16, 12, 2, 128, [5, 0, 9], seed=7)
    k_slab, v_slab = kv.get_kv_buffer(0)
    out = _hip.paged_decode(cpu["q"].to(gpu), k_slab, v_slab, r2t.req_to_token,
                            cpu["req_pool_indices"].to(gpu), cpu["seq_lens"].to(gpu))
    want = orc.decode_exact(cpu["q"], cpu["k_slab"], cpu["v_slab"], cpu["req_to_token"],
                            cpu["req_pool_indices"], cpu["seq_lens"])
    assert_close(out, want, torch.bfloat16, "zero-length request")
    assert float(out[1].abs().max()) == 0.0
    empty = _hip.paged_decode(cpu["q"][:0].to(gpu), k_slab, v_slab, r2t.req_to_token,
                              cpu["req_pool_indices"][:0].to(gpu), cpu["seq_lens"][:0].to(gpu))
    assert tuple(empty.shape) == (0, 12 * 128)


def test_decode_softmax_spike(gpu):
    """A key that dominates late in the sequence forces the running-max rescale
    path in every split and wave (large positive score after many small ones)."""
    from baby_sglang_amd import _hip

    seq = 700
    cpu, r2t, kv = random_decode_problem(gpu, torch.bfloat16, 12, 2, 128, [seq, seq], seed=11)
    for pos in (seq - 1, 333):
        slot = int(cpu["req_to_token"][cpu["req_pool_indices"][0], pos])
        cpu["k_slab"][slot] = (cpu["q"][0, ::6] * 3.0).to(torch.bfloat16)  # aligned with q heads 0 and 6
    kv.k_buffer[0].copy_(cpu["k_slab"])
    want = orc.decode_exact(cpu["q"], cpu["k_slab"], cpu["v_slab"], cpu["req_to_token"],
                            cpu["req_pool_indices"], cpu["seq_lens"])
    k_slab, v_slab = kv.get_kv_buffer(0)
    for variant in (1, 0):
        for n_splits in (1, 5):
            out = _hip.paged_decode(cpu["q"].to(gpu), k_slab, v_slab, r2t.req_to_token,
                                    cpu["req_pool_indices"].to(gpu), cpu["seq_lens"].to(gpu),
                                    n_splits=n_splits, variant=variant)
            assert_close(out, want, torch.bfloat16, f"spike variant={variant} splits={n_splits}")


def test_bad_metadata_never_faults(gpu):
    """Out-of-range slots, rows and lengths are clamped: wrong numbers, no fault."""
    from baby_sglang_amd import _hip

    cpu, r2t, kv = random_decode_problem(gpu, torch.float16, 12, 2, 128, [40, 40], seed=3)
    r2t.req_to_token[:, 3] = 2**30
    r2t.req_to_token[:, 4] = -5
    k_slab, v_slab = kv.get_kv_buffer(0)
    out = _hip.paged_decode(cpu["q"].to(gpu), k_slab, v_slab, r2t.req_to_token,
                            torch.tensor([99, -3], device=gpu), torch.tensor([10**6, 40], device=gpu))
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out.float()).all())
    loc = torch.tensor([1, 10**6, -1], dtype=torch.int32, device=gpu)
    rows = torch.ones((3, 2, 128), dtype=torch.float16, device=gpu)
    before = k_slab.clone()
    _hip.kv_scatter(k_slab, v_slab, loc, rows, rows)
    torch.cuda.synchronize()
    assert float(k_slab[1].float().min()) == 1.0
    assert torch.equal(k_slab[2:], before[2:]) and torch.equal(k_slab[0], before[0])


@pytest.mark.parametrize("variant", [2, 1])
@pytest.mark.parametrize("hq,hkv,n_q", [(12, 2, 70), (12, 2, 9), (4, 4, 70)])
def test_bad_extend_metadata_never_leaves_q_and_out(gpu, variant, hq, hkv, n_q):
    """extended_lens that promise more rows than q / out have (one request far too long, the sum
    too long, a negative one): every extend kernel -- 64- and 16-position blocks, the 32x32x16 and
    the 16x16x32 forms, the generic rows -- must stay inside the n_q rows it was given.  `out` is
    the head of a larger buffer whose tail must keep its fill."""
    from baby_sglang_amd import _hip

    dim, dtype = 128, torch.float16
    cpu, r2t, kv = random_decode_problem(gpu, dtype, hq, hkv, dim, [300, 200, 150, 90], seed=13)
    k_slab, v_slab = kv.get_kv_buffer(0)
    rows, seq = cpu["req_pool_indices"].to(gpu), cpu["seq_lens"].to(gpu)
    q = torch.randn((n_q, hq, dim), device=gpu).to(dtype)
    guard_rows = 256
    for ext, pre in (([40, 500, 30, 20], [0, 0, 0, 0]), ([10**7, 5, 5, 5], [10, 0, 0, 0]), ([30, -4, 10**12, 2], [0, 7, 3, 0])):
        buf = torch.full((n_q + guard_rows, hq * dim), 7.0, dtype=dtype, device=gpu)
        out = _hip.paged_prefill(q, k_slab, v_slab, r2t.req_to_token, rows, seq, torch.tensor(pre, device=gpu),
                                 torch.tensor(ext, device=gpu), variant=variant, out=buf[:n_q])
        torch.cuda.synchronize()
        assert out.data_ptr() == buf.data_ptr()
        assert bool((buf[n_q:] == 7.0).all()), f"rows past the end of out were written (ext {ext})"


def test_errors_map_to_python_exceptions(gpu):
    from baby_sglang_amd import _hip

    cpu, r2t, kv = random_decode_problem(gpu, torch.float16, 12, 2, 128, [8], seed=5)
    k_slab, v_slab = kv.get_kv_buffer(0)
    with pytest.raises(NotImplementedError):  # head_dim 96 has no kernel
        _hip.paged_decode(torch.zeros((1, 2, 96), dtype=torch.float16, device=gpu),
                          torch.zeros((4, 2, 96), dtype=torch.float16, device=gpu),
                          torch.zeros((4, 2, 96), dtype=torch.float16, device=gpu),
                          r2t.req_to_token, torch.zeros(1, dtype=torch.int64, device=gpu),
                          torch.ones(1, dtype=torch.int64, device=gpu))
    with pytest.raises(ValueError):  # dtype mismatch between q and slabs
        _hip.paged_decode(cpu["q"].to(gpu).float(), k_slab, v_slab, r2t.req_to_token,
                          cpu["req_pool_indices"].to(gpu), cpu["seq_lens"].to(gpu))
    with pytest.raises(RuntimeError, match="MI355X only"):
        _hip.paged_decode(cpu["q"], k_slab, v_slab, r2t.req_to_token,
                          cpu["req_pool_indices"].to(gpu), cpu["seq_lens"].to(gpu))


def test_merge_attn_states(gpu):
    from baby_sglang_amd import _hip

    gen = torch.Generator().manual_seed(0)
    parts_o = torch.randn((3, 5, 12, 128), generator=gen)
    parts_lse = torch.randn((3, 5, 12), generator=gen) * 4
    parts_lse[1, 2] = float("-inf")  # an empty part
    parts_lse[:, 4, 3] = float("-inf")  # nothing anywhere -> zero row
    want, want_lse = orc.merge_states_exact(parts_o, parts_lse)
    for dtype in (torch.bfloat16, torch.float16, torch.float32):
        lse = torch.empty((5, 12), dtype=torch.float32, device=gpu)
        out = _hip.merge_attn_states(parts_o.to(gpu), parts_lse.to(gpu), dtype, lse_out=lse)
        assert_close(out, want, dtype, "merge")
        assert torch.allclose(lse.double().cpu(), want_lse, rtol=1e-5, atol=1e-5)


# ----------------------------------------------------------------------------- full-size properties
def test_full_size_properties(gpu):
    """Qwen2.5-1.5B shapes, batch 64, seq 2048, random-permutation page table:
    split-count invariance, page-permutation invariance, constant-V identity,
    half/half merge, and four requests checked against the float64 oracle."""
    from baby_sglang_amd import _hip

    B, S, hq, hkv, dim = 64, 2048, 12, 2, 128
    dtype = torch.bfloat16
    gen = torch.Generator(device=gpu).manual_seed(0)
    n_slots = B * S + 1
    k_slab = torch.randn((n_slots, hkv, dim), generator=gen, device=gpu, dtype=torch.float32).to(dtype)
    v_slab = torch.randn((n_slots, hkv, dim), generator=gen, device=gpu, dtype=torch.float32).to(dtype)
    q = torch.randn((B, hq, dim), generator=gen, device=gpu, dtype=torch.float32).to(dtype)
    perm = torch.randperm(B * S, generator=gen, device=gpu).to(torch.int32) + 1
    table = torch.zeros((B, 4096), dtype=torch.int32, device=gpu)
    table[:, :S] = perm.view(B, S)
    rows = torch.arange(B, device=gpu)
    seq = torch.full((B,), S, dtype=torch.int64, device=gpu)

    base = _hip.paged_decode(q, k_slab, v_slab, table, rows, seq, n_splits=1, variant=1).float()
    # (a) any split count / kernel variant gives the same rows (fp32 partials, one final rounding)
    for variant in (1, 0):
        for n_splits in (4, 16, 37):
            other = _hip.paged_decode(q, k_slab, v_slab, table, rows, seq, n_splits=n_splits, variant=variant).float()
            assert torch.allclose(other, base, rtol=8e-3, atol=2e-3), (variant, n_splits)
    # (b) attention is invariant under a permutation of a request's pages
    shuffled = table.clone()
    shuffled[:, :S] = table[:, torch.randperm(S, generator=gen, device=gpu)]
    other = _hip.paged_decode(q, k_slab, v_slab, shuffled, rows, seq).float()
    assert torch.allclose(other, base, rtol=8e-3, atol=2e-3)
    # (c) constant V rows -> the output is that constant, whatever the scores are
    const = _hip.paged_decode(q, k_slab, torch.full_like(v_slab, 0.75), table, rows, seq).float()
    assert torch.allclose(const, torch.full_like(const, 0.75), rtol=0, atol=4e-3)
    # (d) two halves merged == whole
    parts_o, parts_lse = [], []
    for half in range(2):
        t = torch.zeros_like(table)
        t[:, : S // 2] = table[:, half * (S // 2) : (half + 1) * (S // 2)]
        lse = torch.empty((B, hq), dtype=torch.float32, device=gpu)
        o = _hip.paged_decode(q, k_slab, v_slab, t, rows, seq // 2, lse=lse, out_dtype=torch.float32)
        parts_o.append(o.view(B, hq, dim))
        parts_lse.append(lse)
    merged = _hip.merge_attn_states(torch.stack(parts_o), torch.stack(parts_lse), dtype).float()
    assert torch.allclose(merged, base, rtol=8e-3, atol=2e-3)
    # (e) float64 oracle on four requests
    pick = [0, 17, 42, 63]
    want = orc.decode_exact(q[pick].cpu(), k_slab.cpu(), v_slab.cpu(), table.cpu(), rows[pick].cpu(), seq[pick].cpu())
    got = _hip.paged_decode(q, k_slab, v_slab, table, rows, seq)
    assert_close(got[pick], want, dtype, "full size vs oracle")


def test_prefill_random(gpu):
    """Extend batches: ragged lengths, prefixes, an ext==0 request, 7B head shape."""
    import baby_sglang_amd as pkg
    from baby_sglang_amd import _hip

    for dtype, (hq, hkv, dim) in [(torch.bfloat16, (12, 2, 128)), (torch.float16, (28, 4, 128)), (torch.float32, (14, 2, 64))]:
        seq_lens = [150, 64, 33, 257, 1]
        prefix_lens = [20, 0, 33, 128, 0]
        cpu, r2t, kv = random_decode_problem(gpu, dtype, hq, hkv, dim, seq_lens, seed=dim + hq)
        ext = [s - p for s, p in zip(seq_lens, prefix_lens)]
        gen = torch.Generator().manual_seed(1)
        q = torch.randn((sum(ext), hq, dim), generator=gen).to(dtype)
        pl, el = torch.tensor(prefix_lens), torch.tensor(ext)
        want = orc.prefill_exact(q, cpu["k_slab"], cpu["v_slab"], cpu["req_to_token"],
                                 cpu["req_pool_indices"], cpu["seq_lens"], pl, el)
        k_slab, v_slab = kv.get_kv_buffer(0)
        out = _hip.paged_prefill(q.to(gpu), k_slab, v_slab, r2t.req_to_token, cpu["req_pool_indices"].to(gpu),
                                 cpu["seq_lens"].to(gpu), pl.to(gpu), el.to(gpu))
        assert_close(out, want, dtype, f"prefill {dtype}")


@pytest.mark.parametrize("dtype,hq,hkv", [(torch.bfloat16, 12, 2), (torch.float16, 28, 4), (torch.bfloat16, 8, 1), (torch.float16, 5, 5),
                                          (torch.bfloat16, 14, 2), (torch.float16, 14, 2), (torch.float16, 4, 2)])
def test_prefill_kernels_agree(gpu, dtype, hq, hkv):
    """Matrix-core extend kernel vs the generic one vs the float64 oracle: block
    boundaries (15/16/17 rows), long prefixes, several slot blocks, chunked prefill
    (a chunk is an extend over the request's own earlier chunks)."""
    from baby_sglang_amd import _hip

    dim = 64 if (hq, hkv) in ((14, 2), (4, 2)) else 128  # (two kv heads of 64: folded into one of 128)
    seq_lens = [16, 15, 17, 700, 1100, 33, 1]
    prefix_lens = [0, 0, 0, 512, 1024, 32, 0]
    cpu, r2t, kv = random_decode_problem(gpu, dtype, hq, hkv, dim, seq_lens, seed=hq + 7)
    ext = [s - p for s, p in zip(seq_lens, prefix_lens)]
    gen = torch.Generator().manual_seed(2)
    q = torch.randn((sum(ext), hq, dim), generator=gen).to(dtype)
    pl, el = torch.tensor(prefix_lens), torch.tensor(ext)
    want = orc.prefill_exact(q, cpu["k_slab"], cpu["v_slab"], cpu["req_to_token"],
                             cpu["req_pool_indices"], cpu["seq_lens"], pl, el)
    k_slab, v_slab = kv.get_kv_buffer(0)
    args = (q.to(gpu), k_slab, v_slab, r2t.req_to_token, cpu["req_pool_indices"].to(gpu),
            cpu["seq_lens"].to(gpu), pl.to(gpu), el.to(gpu))
    got_mfma = _hip.paged_prefill(*args, variant=2)
    got_valu = _hip.paged_prefill(*args, variant=1)
    assert_close(got_valu, want, dtype, "generic extend")
    if dtype == torch.bfloat16:
        # The matrix-core kernel rounds each probability to bf16 once before P V (as torch's
        # SDPA, i.e. the reference, does; its own distance to the float64 definition on the golden
        # cases is up to 1.2e-2).  That is a relative error of at most 2^-8 per term of
        # sum_t p_t v_t, so the output may move by 2^-8 * sum_t p_t |v_t| -- the same attention
        # over |V| -- on top of the rounding of the result itself.
        spread = orc.prefill_exact(q, cpu["k_slab"], cpu["v_slab"].abs(), cpu["req_to_token"],
                                   cpu["req_pool_indices"], cpu["seq_lens"], pl, el)
        err = (got_mfma.double().cpu() - want).abs()
        bound = 2.0**-8 * want.abs() + 2.0**-8 * spread + 1e-4
        assert bool((err <= bound).all()), f"matrix-core extend: worst excess {(err - bound).max().item():.3e}"
        assert err.mean().item() < 1.5e-3
    else:
        assert_close(got_mfma, want, dtype, "matrix-core extend")


@pytest.mark.parametrize("dtype,hq,hkv,dim", [(torch.bfloat16, 28, 4, 128), (torch.float16, 12, 2, 128), (torch.float16, 6, 2, 128),
                                              (torch.bfloat16, 14, 2, 64)])
def test_extend_over_many_page_table_blocks(gpu, dtype, hq, hkv, dim):
    """Sequences of several page-table blocks (512 entries each) in one stream: a chunk deep
    inside a long request, a long pure prefill whose causal edge crosses block borders, lengths
    one short of / at / one past a block border, and an odd number of q heads per kv head (the
    last pair of heads of a group is half empty)."""
    from baby_sglang_amd import _hip

    seq_lens = [2600, 1500, 1024, 1025, 1023, 90]
    prefix_lens = [2100, 0, 960, 1000, 1000, 26]
    cpu, r2t, kv = random_decode_problem(gpu, dtype, hq, hkv, dim, seq_lens, seed=hq + dim)
    ext = [s - p for s, p in zip(seq_lens, prefix_lens)]
    gen = torch.Generator().manual_seed(4)
    q = torch.randn((sum(ext), hq, dim), generator=gen).to(dtype)
    pl, el = torch.tensor(prefix_lens), torch.tensor(ext)
    want = orc.prefill_exact(q, cpu["k_slab"], cpu["v_slab"], cpu["req_to_token"], cpu["req_pool_indices"], cpu["seq_lens"], pl, el)
    spread = orc.prefill_exact(q, cpu["k_slab"], cpu["v_slab"].abs(), cpu["req_to_token"], cpu["req_pool_indices"], cpu["seq_lens"], pl, el)
    k_slab, v_slab = kv.get_kv_buffer(0)
    got = _hip.paged_prefill(q.to(gpu), k_slab, v_slab, r2t.req_to_token, cpu["req_pool_indices"].to(gpu),
                             cpu["seq_lens"].to(gpu), pl.to(gpu), el.to(gpu), variant=2)
    ulp = 2.0**-8 if dtype == torch.bfloat16 else 2.0**-11
    err = (got.double().cpu() - want).abs()
    bound = ulp * want.abs() + ulp * spread + 2e-4
    assert bool(torch.isfinite(got).all()) and bool((err <= bound).all()), f"worst excess {(err - bound).max().item():.3e}"
    if dtype == torch.float16:
        assert_close(got, want, dtype, "f16 extend against the stated bar (rtol 1e-3, atol 1e-3)")


@pytest.mark.parametrize("dtype,hq,hkv,dim", [(torch.bfloat16, 28, 4, 128), (torch.float16, 12, 2, 128), (torch.bfloat16, 16, 2, 128),
                                              (torch.float16, 14, 2, 64), (torch.bfloat16, 4, 2, 128)])
def test_short_extends_share_one_stream(gpu, dtype, hq, hkv, dim):
    """Batches of short extends (a few new tokens over long cached prefixes, as after prefix-cache
    hits): 16-position workgroups that serve up to 8 q heads from one K/V stream.  Lengths of
    1, 15, 16, 17 and 33 new tokens (one, one full, two and three blocks), an empty request, and a
    mean below the switch to 64-position blocks."""
    from baby_sglang_amd import _hip

    ext = [1, 15, 16, 17, 33, 0, 8, 3, 16, 2]
    prefix_lens = [0, 700, 1030, 5, 520, 40, 1500, 64, 0, 2047]
    seq_lens = [p + e for p, e in zip(prefix_lens, ext)]
    cpu, r2t, kv = random_decode_problem(gpu, dtype, hq, hkv, dim, seq_lens, seed=3 * hq + dim)
    gen = torch.Generator().manual_seed(6)
    q = torch.randn((sum(ext), hq, dim), generator=gen).to(dtype)
    pl, el = torch.tensor(prefix_lens), torch.tensor(ext)
    want = orc.prefill_exact(q, cpu["k_slab"], cpu["v_slab"], cpu["req_to_token"], cpu["req_pool_indices"], cpu["seq_lens"], pl, el)
    spread = orc.prefill_exact(q, cpu["k_slab"], cpu["v_slab"].abs(), cpu["req_to_token"], cpu["req_pool_indices"], cpu["seq_lens"], pl, el)
    k_slab, v_slab = kv.get_kv_buffer(0)
    got = _hip.paged_prefill(q.to(gpu), k_slab, v_slab, r2t.req_to_token, cpu["req_pool_indices"].to(gpu),
                             cpu["seq_lens"].to(gpu), pl.to(gpu), el.to(gpu), variant=2)
    ulp = 2.0**-8 if dtype == torch.bfloat16 else 2.0**-11
    err = (got.double().cpu() - want).abs()
    bound = ulp * want.abs() + ulp * spread + 2e-4
    assert bool(torch.isfinite(got).all()) and bool((err <= bound).all()), f"worst excess {(err - bound).max().item():.3e}"
    if dtype == torch.float16:
        assert_close(got, want, dtype, "f16 extend against the stated bar (rtol 1e-3, atol 1e-3)")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("slope", [0.02, 0.4])
def test_extend_scores_rising_along_the_sequence(gpu, dtype, slope):
    """Scores that keep growing with the token position: the extend kernel's running reference
    value (renewed only when a score exceeds it by a margin) must be renewed again and again --
    slowly (several steps per renewal) and fast (every step) -- and late tokens dominate."""
    from baby_sglang_amd import _hip

    hq, hkv, dim = 4, 2, 128
    seq_lens, prefix_lens = [900, 330, 64], [512, 0, 63]
    cpu, r2t, kv = random_decode_problem(gpu, dtype, hq, hkv, dim, seq_lens, seed=5)
    for i, n in enumerate(seq_lens):
        slots = cpu["req_to_token"][i, :n].long()
        cpu["k_slab"][slots, :, 0] = (slope * torch.arange(n, dtype=torch.float32))[:, None].to(dtype)
    kv.k_buffer[0].copy_(cpu["k_slab"])
    ext = [s - p for s, p in zip(seq_lens, prefix_lens)]
    gen = torch.Generator().manual_seed(3)
    q = torch.randn((sum(ext), hq, dim), generator=gen)
    q[:, :, 0] = 8.0
    q = q.to(dtype)
    pl, el = torch.tensor(prefix_lens), torch.tensor(ext)
    want = orc.prefill_exact(q, cpu["k_slab"], cpu["v_slab"], cpu["req_to_token"], cpu["req_pool_indices"], cpu["seq_lens"], pl, el)
    spread = orc.prefill_exact(q, cpu["k_slab"], cpu["v_slab"].abs(), cpu["req_to_token"], cpu["req_pool_indices"], cpu["seq_lens"], pl, el)
    k_slab, v_slab = kv.get_kv_buffer(0)
    got = _hip.paged_prefill(q.to(gpu), k_slab, v_slab, r2t.req_to_token, cpu["req_pool_indices"].to(gpu),
                             cpu["seq_lens"].to(gpu), pl.to(gpu), el.to(gpu), variant=2)
    ulp = 2.0**-8 if dtype == torch.bfloat16 else 2.0**-11
    err = (got.double().cpu() - want).abs()
    bound = ulp * want.abs() + ulp * spread + 2e-4
    assert bool(torch.isfinite(got).all()) and bool((err <= bound).all()), f"worst excess {(err - bound).max().item():.3e}"
    if dtype == torch.float16:
        assert_close(got, want, dtype, "f16 extend against the stated bar (rtol 1e-3, atol 1e-3)")


@pytest.mark.parametrize("seed", range(16))
def test_random_shapes_against_the_oracle(gpu, seed):
    """Random model shapes, dtypes, batch sizes, lengths, prefixes, split counts and KV layouts:
    decode (new rows written by the call) and extend through the automatic kernel choice, against
    the float64 oracle; KV rows bit exact."""
    from baby_sglang_amd import _hip

    gen = torch.Generator().manual_seed(900 + seed)

    def pick(options):
        return options[int(torch.randint(0, len(options), (1,), generator=gen))]

    hq, hkv, dim = pick([(12, 2, 128), (14, 2, 64), (28, 4, 128), (4, 4, 64), (8, 1, 128), (16, 2, 64), (7, 1, 128), (16, 2, 128)])
    dtype = pick([torch.bfloat16, torch.float16, torch.float32])
    layout = pick(["interleaved", "split"])
    batch = pick([1, 2, 5, 33, 70])
    seq_lens = [int(n) for n in torch.randint(1, pick([40, 300, 1400]), (batch,), generator=gen)]
    cpu, r2t, kv = random_decode_problem(gpu, dtype, hq, hkv, dim, seq_lens, seed=seed, layout=layout)
    rows, seq = cpu["req_pool_indices"].to(gpu), cpu["seq_lens"].to(gpu)
    k_slab, v_slab = kv.get_kv_buffer(0)
    # decode with the step's new rows
    k_new = torch.randn((batch, hkv, dim), generator=gen).to(dtype)
    v_new = torch.randn((batch, hkv, dim), generator=gen).to(dtype)
    loc = r2t.req_to_token[rows, seq - 1].contiguous()
    k_ref, v_ref = cpu["k_slab"].clone(), cpu["v_slab"].clone()
    orc.kv_scatter_ref(k_ref, v_ref, loc.cpu(), k_new, v_new)
    want = orc.decode_exact(cpu["q"], k_ref, v_ref, cpu["req_to_token"], cpu["req_pool_indices"], cpu["seq_lens"])
    n_splits = pick([None, 1, 2, 7, 32])
    out = _hip.paged_decode(cpu["q"].to(gpu), k_slab, v_slab, r2t.req_to_token, rows, seq, n_splits=n_splits,
                            k_new=k_new.to(gpu), v_new=v_new.to(gpu), out_cache_loc=loc, fuse_kv_write=pick([True, False]))
    assert_close(out, want, dtype, f"decode {hq}/{hkv}/{dim} {dtype} batch {batch} splits {n_splits} {layout}")
    assert torch.equal(k_slab.cpu(), k_ref) and torch.equal(v_slab.cpu(), v_ref)
    # the same call from a caller that names its longest request -- truthfully or not
    hint = [max(seq_lens), 1, 64, 256, 5000][seed % 5]
    out = _hip.paged_decode(cpu["q"].to(gpu), k_slab, v_slab, r2t.req_to_token, rows, seq, n_splits=n_splits, max_seq_len=hint,
                            k_new=k_new.to(gpu), v_new=v_new.to(gpu), out_cache_loc=loc, fuse_kv_write=True)
    assert_close(out, want, dtype, f"decode {hq}/{hkv}/{dim} {dtype} batch {batch} splits {n_splits} {layout} hint {hint}")
    assert torch.equal(k_slab.cpu(), k_ref) and torch.equal(v_slab.cpu(), v_ref)
    # extend of the same requests over random prefixes
    prefix = [int(torch.randint(0, n, (1,), generator=gen)) if n > 1 and pick([True, False]) else 0 for n in seq_lens]
    ext = [n - p for n, p in zip(seq_lens, prefix)]
    q = torch.randn((sum(ext), hq, dim), generator=gen).to(dtype)
    pl, el = torch.tensor(prefix), torch.tensor(ext)
    want = orc.prefill_exact(q, k_ref, v_ref, cpu["req_to_token"], cpu["req_pool_indices"], cpu["seq_lens"], pl, el)
    got = _hip.paged_prefill(q.to(gpu), k_slab, v_slab, r2t.req_to_token, rows, seq, pl.to(gpu), el.to(gpu))
    if dtype == torch.bfloat16:  # (probabilities rounded once: bound by the rounding of the terms, as above)
        spread = orc.prefill_exact(q, k_ref, v_ref.abs(), cpu["req_to_token"], cpu["req_pool_indices"], cpu["seq_lens"], pl, el)
        err = (got.double().cpu() - want).abs()
        assert bool((err <= 2.0**-8 * want.abs() + 2.0**-8 * spread + 1e-4).all()), f"extend {hq}/{hkv}/{dim} batch {batch}"
    else:
        assert_close(got, want, dtype, f"extend {hq}/{hkv}/{dim} {dtype} batch {batch} {layout}")


def test_side_stream_and_thread(gpu):
    """The reference calls its backend from a worker thread inside a side stream
    (overlap_worker.py:104-127); results must not depend on either."""
    import threading

    from baby_sglang_amd import _hip

    cpu, r2t, kv = random_decode_problem(gpu, torch.bfloat16, 12, 2, 128, [200, 31, 77], seed=21)
    k_slab, v_slab = kv.get_kv_buffer(0)
    args = (cpu["q"].to(gpu), k_slab, v_slab, r2t.req_to_token, cpu["req_pool_indices"].to(gpu), cpu["seq_lens"].to(gpu))
    base = _hip.paged_decode(*args)
    torch.cuda.synchronize()
    box = {}

    def work():
        s = torch.cuda.Stream(device=gpu)
        with torch.cuda.stream(s):
            box["out"] = _hip.paged_decode(*args)
        s.synchronize()

    t = threading.Thread(target=work)
    t.start()
    t.join()
    assert torch.equal(box["out"], base)
