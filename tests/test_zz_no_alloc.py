"""bsgl_paged_decode_static: the decode entry point that allocates nothing.

Its workspace is an array inside the library's code object (16 MiB per GPU) instead of caller
memory.  Called through the C ABI and compared bit for bit with bsgl_paged_decode (same kernels,
same split plan) on two calls in a row, the second with a smaller input -- so the second call
runs on whatever the first left in the array --, with the new token's cache rows written by the
kernel, for the bf16 output and for the fp32 output + log-sum-exp form; sizes beyond the array
are refused with a clear error before anything is launched.
"""
import pytest
import torch

from oracle import attn_oracle as orc
from test_gpu_parity import assert_close, random_decode_problem

pytestmark = pytest.mark.gpu


def _both(gpu, hip, cpu, r2t, kv, n_req, n_splits, k_new, v_new, out_dtype, variant=0):
    """(static-scratch result, workspace result), each as (out, lse, k_slab, v_slab) clones."""
    rows, seq = cpu["req_pool_indices"][:n_req].to(gpu), cpu["seq_lens"][:n_req].to(gpu)
    loc = r2t.req_to_token[rows, seq - 1].contiguous()
    q = cpu["q"][:n_req].to(gpu)
    hq, dim = q.shape[1], q.shape[2]
    results = []
    for static in (True, False):
        kv.k_buffer[0].copy_(cpu["k_slab"])
        kv.v_buffer[0].copy_(cpu["v_slab"])
        k_slab, v_slab = kv.get_kv_buffer(0)
        out = torch.full((n_req, hq * dim), float("nan"), dtype=out_dtype, device=gpu)
        lse = torch.full((n_req, hq), float("nan"), dtype=torch.float32, device=gpu)
        hip.paged_decode(q, k_slab, v_slab, r2t.req_to_token, rows, seq, n_splits=n_splits, variant=variant, out=out, lse=lse,
                         out_dtype=out_dtype, k_new=k_new[:n_req], v_new=v_new[:n_req], out_cache_loc=loc,
                         fuse_kv_write=True, static_scratch=static)
        torch.cuda.synchronize()
        results.append((out.clone(), lse.clone(), k_slab.clone(), v_slab.clone()))
    return results, loc


@pytest.mark.parametrize("dtype,hq,hkv,dim", [(torch.bfloat16, 12, 2, 128), (torch.float16, 28, 4, 128), (torch.bfloat16, 14, 2, 64),
                                              (torch.float32, 4, 4, 64)])
def test_static_scratch_equals_the_workspace_entry_point(gpu, dtype, hq, hkv, dim):
    from baby_sglang_amd import _hip

    seq_lens = [1, 16, 17, 333, 64, 1500, 129, 2047, 900]
    cpu, r2t, kv = random_decode_problem(gpu, dtype, hq, hkv, dim, seq_lens, seed=5 + hq)
    gen = torch.Generator().manual_seed(6)
    k_new = torch.randn((len(seq_lens), hkv, dim), generator=gen).to(dtype).to(gpu)
    v_new = torch.randn((len(seq_lens), hkv, dim), generator=gen).to(dtype).to(gpu)
    # two calls in a row on the library's array, the second smaller; then the fp32 + lse form
    for n_req, n_splits, out_dtype in ((len(seq_lens), 16, dtype), (4, 7, dtype), (len(seq_lens), 5, torch.float32), (2, 1, dtype)):
        (st, ws), loc = _both(gpu, _hip, cpu, r2t, kv, n_req, n_splits, k_new, v_new, out_dtype)
        for a, b in zip(st, ws):
            assert torch.equal(a, b), f"{n_req} requests, {n_splits} splits, {out_dtype}"
        assert bool(torch.isfinite(st[0].float()).all())
        # and it is right: against the float64 oracle, cache rows bit exact
        k_ref, v_ref = cpu["k_slab"].clone(), cpu["v_slab"].clone()
        orc.kv_scatter_ref(k_ref, v_ref, loc.cpu(), k_new[:n_req].cpu(), v_new[:n_req].cpu())
        want, want_lse = orc.decode_exact(cpu["q"][:n_req], k_ref, v_ref, cpu["req_to_token"], cpu["req_pool_indices"][:n_req],
                                          cpu["seq_lens"][:n_req], return_lse=True)
        assert_close(st[0], want, dtype, "static scratch vs the float64 oracle")
        assert torch.allclose(st[1].double().cpu(), want_lse, rtol=1e-4, atol=2e-3 if dtype != torch.float32 else 1e-4)
        assert torch.equal(st[2].cpu(), k_ref) and torch.equal(st[3].cpu(), v_ref)


def test_static_scratch_headline_batch_then_a_smaller_one(gpu):
    """64 x 2048 (Qwen2.5-1.5B shapes: 6.4 MiB of the array), then 5 short requests."""
    from baby_sglang_amd import _hip

    dtype, hq, hkv, dim = torch.bfloat16, 12, 2, 128
    for seq_lens in ([2048] * 64, [300, 1, 77, 1024, 5]):
        cpu, r2t, kv = random_decode_problem(gpu, dtype, hq, hkv, dim, seq_lens, seed=len(seq_lens))
        gen = torch.Generator().manual_seed(2)
        k_new = torch.randn((len(seq_lens), hkv, dim), generator=gen).to(dtype).to(gpu)
        v_new = torch.randn((len(seq_lens), hkv, dim), generator=gen).to(dtype).to(gpu)
        n_splits = _hip.decode_num_splits(len(seq_lens), hkv, 4096)
        (st, ws), _ = _both(gpu, _hip, cpu, r2t, kv, len(seq_lens), n_splits, k_new, v_new, dtype)
        for a, b in zip(st, ws):
            assert torch.equal(a, b)


def test_static_scratch_refuses_what_does_not_fit(gpu):
    from baby_sglang_amd import _hip

    dtype, hq, hkv, dim = torch.bfloat16, 12, 2, 128
    seq_lens = [40] * 200
    cpu, r2t, kv = random_decode_problem(gpu, dtype, hq, hkv, dim, seq_lens, seed=1)
    k_slab, v_slab = kv.get_kv_buffer(0)
    rows, seq = cpu["req_pool_indices"].to(gpu), cpu["seq_lens"].to(gpu)
    assert _hip.decode_workspace_bytes(200, hq, dim, 64) > (16 << 20)
    out = torch.full((200, hq * dim), 3.0, dtype=dtype, device=gpu)
    with pytest.raises(RuntimeError, match="bytes of scratch"):
        _hip.paged_decode(cpu["q"].to(gpu), k_slab, v_slab, r2t.req_to_token, rows, seq, n_splits=64, variant=2, out=out,
                          static_scratch=True)
    torch.cuda.synchronize()
    assert bool((out == 3.0).all())  # nothing was launched
    # the same batch with fewer splits fits and runs
    got = _hip.paged_decode(cpu["q"].to(gpu), k_slab, v_slab, r2t.req_to_token, rows, seq, n_splits=4, out=out, static_scratch=True)
    want = orc.decode_exact(cpu["q"], cpu["k_slab"], cpu["v_slab"], cpu["req_to_token"], cpu["req_pool_indices"], cpu["seq_lens"])
    assert_close(got, want, dtype, "200 requests, 4 splits")


def test_static_scratch_inside_a_hip_graph(gpu):
    """What the entry point is for: a decode step captured into a HIP graph (nothing is allocated
    while capturing) and replayed on NEW queries in the same buffers -- with unchanged queries a
    merge from the last replay's partial rows would go unnoticed -- against the workspace entry
    point called eagerly."""
    from baby_sglang_amd import _hip

    dtype, hq, hkv, dim = torch.bfloat16, 12, 2, 128
    seq_lens = [700, 33, 1200, 64, 2000, 1]
    cpu, r2t, kv = random_decode_problem(gpu, dtype, hq, hkv, dim, seq_lens, seed=11)
    k_slab, v_slab = kv.get_kv_buffer(0)
    k_slab.copy_(cpu["k_slab"])
    v_slab.copy_(cpu["v_slab"])
    rows, seq = cpu["req_pool_indices"].to(gpu), cpu["seq_lens"].to(gpu)
    q = cpu["q"].to(gpu).clone()
    out = torch.empty((len(seq_lens), hq * dim), dtype=dtype, device=gpu)
    n_splits = _hip.decode_num_splits(len(seq_lens), hkv, 4096)

    def call(static, out_):
        return _hip.paged_decode(q, k_slab, v_slab, r2t.req_to_token, rows, seq, n_splits=n_splits, out=out_, static_scratch=static)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        call(True, out)  # (first use of the array on this device: its address is looked up outside the capture)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        call(True, out)
    gen = torch.Generator().manual_seed(12)
    for _ in range(3):
        q.copy_(torch.randn(q.shape, generator=gen).to(dtype))
        graph.replay()
        torch.cuda.synchronize()
        want = torch.empty_like(out)
        call(False, want)
        torch.cuda.synchronize()
        assert torch.equal(out, want)
        exact = orc.decode_exact(q.cpu(), cpu["k_slab"], cpu["v_slab"], cpu["req_to_token"], cpu["req_pool_indices"], cpu["seq_lens"])
        assert_close(out, exact, dtype, "graph replay of the static-scratch call")
