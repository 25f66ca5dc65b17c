"""KV pools of the size a 288 GB card is bought for (BASELINE config 5: 7B shapes, "288 GB KV sizing").

Every kernel forms ``slot x row stride`` products; the other tests run on pools of at most some 40 000
tokens, where those products fit 32 bits whatever their type.  Here the arena of a 28-layer 7B-shaped
pool is more than 200 GB (``torch.empty``: only the rows that are used are ever written), a layer's
slab is 7.4 GB, and every request's slots sit in the top 1 % of the slab, so every row address is
beyond 4 GiB inside its slab and beyond 190 GB inside the arena on the last layer.  On layers 0 and 27:
the KV write (``set_kv_buffer``, reference: memory_pool.py:152-171) read back bit for bit, a decode
step through the matrix-core kernel and through the generic one (with the new token written by the
kernel), and a 512-over-1536 extend, against the float64 oracle on the same rows.  Then a one-layer
pool whose slab exceeds 64 GiB: the matrix-core kernels address rows as 32-bit multiples of 16 bytes
and must refuse it, and the automatic choice must fall back to the generic kernels and still be right.
"""
import pytest
import torch

from oracle import attn_oracle as orc
from test_gpu_parity import assert_close

pytestmark = pytest.mark.gpu

HQ, HKV, DIM, DTYPE = 28, 4, 128, torch.bfloat16
SEQ = [2048, 1301, 2048]  # request 0 also runs as an extend of 512 over 1536


def _need_free(gpu, n_bytes):
    torch.cuda.empty_cache()
    free, _total = torch.cuda.mem_get_info(gpu)
    if free < n_bytes + (4 << 30):
        pytest.skip(f"needs {n_bytes / 2**30:.0f} GiB of free device memory, {free / 2**30:.0f} GiB are free")


def _problem(gpu, pool, layer, first_slot, seed):
    """Requests whose slots are a random choice among [first_slot, pool.size]; their K/V rows are written
    with the pool's own set_kv_buffer.  Returns the device side and a COMPACT CPU copy (slot i of the
    compact slabs = the i-th used slot) for the oracle."""
    import baby_sglang_amd as pkg

    gen = torch.Generator().manual_seed(seed)
    n_used = sum(SEQ) + len(SEQ)  # the cached tokens and one new token per request
    span = pool.size - first_slot + 1
    assert span >= n_used
    slots = (torch.randperm(span, generator=gen)[:n_used] + first_slot).to(torch.int32)
    k_rows = torch.randn((n_used, HKV, DIM), generator=gen).to(DTYPE)
    v_rows = torch.randn((n_used, HKV, DIM), generator=gen).to(DTYPE)
    n_old = sum(SEQ)
    # cached rows: the pool's KV write (kv_scatter through the C ABI), then read back through the slab views
    pool.set_kv_buffer(layer, slots[:n_old].to(gpu), k_rows[:n_old].to(gpu), v_rows[:n_old].to(gpu))
    k_slab, v_slab = pool.get_kv_buffer(layer)
    idx = slots[:n_old].long().to(gpu)
    assert torch.equal(k_slab[idx].cpu(), k_rows[:n_old]) and torch.equal(v_slab[idx].cpu(), v_rows[:n_old]), "KV write read back"
    # page tables: the real one names the real slots, the compact one the positions in k_rows / v_rows (+ 1: row 0 is the pad row)
    max_ctx = 4096
    r2t = pkg.ReqToTokenPool(len(SEQ), max_ctx, str(gpu))
    table = torch.zeros((len(SEQ), max_ctx), dtype=torch.int32)
    compact = torch.zeros((len(SEQ), max_ctx), dtype=torch.int32)
    used = 0
    for i, n in enumerate(SEQ):
        table[i, :n] = slots[used:used + n]
        compact[i, :n] = torch.arange(used, used + n, dtype=torch.int32) + 1
        used += n
        table[i, n] = slots[n_old + i]  # the token the decode step appends
        compact[i, n] = n_old + i + 1
    r2t.req_to_token.copy_(table.to(gpu))
    pad = torch.zeros((1, HKV, DIM), dtype=DTYPE)
    k_cpu, v_cpu = torch.cat([pad, k_rows]), torch.cat([pad, v_rows])
    return dict(r2t=r2t, k_slab=k_slab, v_slab=v_slab, table=table, compact=compact, k_cpu=k_cpu, v_cpu=v_cpu,
                slots=slots, n_old=n_old, gen=gen)


def _decode_and_extend(gpu, p, variants, extend=True):
    from baby_sglang_amd import _hip

    gen = p["gen"]
    rows = torch.arange(len(SEQ), dtype=torch.int64)
    seq_new = torch.tensor([n + 1 for n in SEQ], dtype=torch.int64)
    q = torch.randn((len(SEQ), HQ, DIM), generator=gen).to(DTYPE)
    n_old = p["n_old"]
    k_new, v_new = p["k_cpu"][1 + n_old:], p["v_cpu"][1 + n_old:]  # (the compact slabs already hold the new rows)
    loc = p["slots"][n_old:].to(gpu)
    want = orc.decode_exact(q, p["k_cpu"], p["v_cpu"], p["compact"], rows, seq_new)
    for variant in variants:
        got = _hip.paged_decode(q.to(gpu), p["k_slab"], p["v_slab"], p["r2t"].req_to_token, rows.to(gpu), seq_new.to(gpu),
                                variant=variant, k_new=k_new.to(gpu), v_new=v_new.to(gpu), out_cache_loc=loc, fuse_kv_write=True)
        assert_close(got, want, DTYPE, f"decode, variant {variant}")
        idx = p["slots"][n_old:].long().to(gpu)
        assert torch.equal(p["k_slab"][idx].cpu(), k_new) and torch.equal(p["v_slab"][idx].cpu(), v_new), "new rows"
    if not extend:
        return
    # request 0: its last 512 tokens over the 1536 before them
    pre, ext = torch.tensor([1536]), torch.tensor([512])
    qe = torch.randn((512, HQ, DIM), generator=gen).to(DTYPE)
    got = _hip.paged_prefill(qe.to(gpu), p["k_slab"], p["v_slab"], p["r2t"].req_to_token, rows[:1].to(gpu),
                             torch.tensor([SEQ[0]]).to(gpu), pre.to(gpu), ext.to(gpu))
    args = (p["compact"], rows[:1], torch.tensor([SEQ[0]]), pre, ext)
    want = orc.prefill_exact(qe, p["k_cpu"], p["v_cpu"], *args)
    spread = orc.prefill_exact(qe, p["k_cpu"], p["v_cpu"].abs(), *args)
    err = (got.double().cpu() - want).abs()
    bound = 2.0**-8 * want.abs() + 2.0**-8 * spread + 1e-4  # (the bf16 extend bound of test_gpu_parity.py)
    assert bool((err <= bound).all()), f"extend: worst excess {(err - bound).max().item():.3e}"


def test_seven_b_pool_of_two_hundred_gigabytes(gpu):
    import baby_sglang_amd as pkg
    from baby_sglang_amd import _hip

    size, layers = 3_600_000, 28
    arena_bytes = layers * (size + 1) * 2 * HKV * DIM * 2
    assert arena_bytes > 200e9
    _need_free(gpu, arena_bytes)
    pool = pkg.MHATokenToKVPool(size, DTYPE, HKV, DIM, layers, str(gpu))
    try:
        assert pool.arena.numel() * 2 == arena_bytes and pool.available_size() == size
        first = size - size // 100  # the top 1 % of the slab
        slab_bytes = (size + 1) * 2 * HKV * DIM * 2
        assert first * 2 * HKV * DIM * 2 > (4 << 30) and slab_bytes < (64 << 30)
        for layer in (0, layers - 1):
            p = _problem(gpu, pool, layer, first, seed=100 + layer)
            _decode_and_extend(gpu, p, (_hip.DECODE_MFMA, _hip.DECODE_VALU, _hip.DECODE_AUTO))
        torch.cuda.synchronize()
    finally:
        del pool
        torch.cuda.empty_cache()


def test_a_slab_beyond_64_gib_is_refused_by_the_matrix_core_kernels_and_served_by_the_generic_ones(gpu):
    import baby_sglang_amd as pkg
    from baby_sglang_amd import _hip

    size = 34_000_000  # one layer: 34 M rows of 2 KiB = 69.6 GB > 64 GiB
    arena_bytes = (size + 1) * 2 * HKV * DIM * 2
    assert arena_bytes > (64 << 30)
    _need_free(gpu, arena_bytes)
    pool = pkg.MHATokenToKVPool(size, DTYPE, HKV, DIM, 1, str(gpu))
    try:
        p = _problem(gpu, pool, 0, size - size // 100, seed=7)
        rows = torch.arange(len(SEQ), dtype=torch.int64, device=gpu)
        seq = torch.tensor(SEQ, dtype=torch.int64, device=gpu)
        q = torch.randn((len(SEQ), HQ, DIM), generator=p["gen"]).to(DTYPE).to(gpu)
        with pytest.raises(NotImplementedError, match="64 GiB"):
            _hip.paged_decode(q, p["k_slab"], p["v_slab"], p["r2t"].req_to_token, rows, seq, variant=_hip.DECODE_MFMA)
        _decode_and_extend(gpu, p, (_hip.DECODE_AUTO, _hip.DECODE_VALU))
        torch.cuda.synchronize()
    finally:
        del pool
        torch.cuda.empty_cache()
