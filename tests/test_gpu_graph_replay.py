"""The backend under HIP graph replay with inputs that CHANGE between replays.

The drop-in boundary promises that every entry point only validates and launches, so a caller may
capture a decode step (or a chunk of extend calls) into a HIP graph and replay it on new data in
the same buffers.  A replay whose inputs never change cannot tell a correct step from one that
merges the previous replay's partial rows, so every replay here runs on new queries and new K/V
rows and is compared with the float64 oracle and, bit for bit, with the same call made eagerly.
(Reference call site: SimpleAttentionBackend.forward, attn_backend.py:9-32.)"""
import types

import pytest
import torch

from oracle import attn_oracle as orc
from test_gpu_parity import assert_close, random_decode_problem

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype,hq,hkv,dim,seq_lens", [
    (torch.bfloat16, 12, 2, 128, [700, 33, 1200, 64, 2000, 1]),          # several waves per piece
    (torch.bfloat16, 12, 2, 128, [512] * 40 + [37, 1500, 9]),            # one wave per piece, ragged
    (torch.float16, 28, 4, 128, [300, 1024, 5, 77]),
    (torch.bfloat16, 14, 2, 64, [900] * 33),                             # two kv heads of 64 folded into one
])
def test_decode_step_replayed_on_new_inputs(gpu, dtype, hq, hkv, dim, seq_lens):
    import baby_sglang_amd as pkg

    n = len(seq_lens)
    cpu, r2t, kv = random_decode_problem(gpu, dtype, hq, hkv, dim, seq_lens, seed=n + hq)
    rows, seq = cpu["req_pool_indices"].to(gpu), cpu["seq_lens"].to(gpu)
    loc = r2t.req_to_token[rows, seq - 1].contiguous()  # the step's new token sits in each request's last slot
    layer = types.SimpleNamespace(layer_id=0, scaling=dim**-0.5)
    fb = pkg.SimplifiedForwardBatch.create_decode_batch(torch.zeros(n, dtype=torch.int64, device=gpu), rows, seq, loc, r2t, kv)
    fb.private_decode_slots = True  # (each request's last slot is its own: the decode kernel stores the new rows)
    q = torch.empty((n, hq, dim), dtype=dtype, device=gpu)
    k = torch.empty((n, hkv, dim), dtype=dtype, device=gpu)
    v = torch.empty((n, hkv, dim), dtype=dtype, device=gpu)
    gen = torch.Generator().manual_seed(1)

    def refill():
        for t in (q, k, v):
            t.copy_(torch.randn(t.shape, generator=gen).to(dtype))

    refill()
    out = torch.empty((n, hq * dim), dtype=dtype, device=gpu)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        out.copy_(pkg.SimpleAttentionBackend.forward(q, k, v, layer, fb))
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out.copy_(pkg.SimpleAttentionBackend.forward(q, k, v, layer, fb))
    k_ref, v_ref = cpu["k_slab"].clone(), cpu["v_slab"].clone()
    for _ in range(3):
        refill()
        graph.replay()
        torch.cuda.synchronize()
        got = out.clone()
        orc.kv_scatter_ref(k_ref, v_ref, loc.cpu(), k.cpu(), v.cpu())
        assert torch.equal(kv.k_buffer[0].cpu(), k_ref) and torch.equal(kv.v_buffer[0].cpu(), v_ref)
        want = orc.decode_exact(q.cpu(), k_ref, v_ref, cpu["req_to_token"], cpu["req_pool_indices"], cpu["seq_lens"])
        assert_close(got, want, dtype, "graph replay on new inputs")
        eager = pkg.SimpleAttentionBackend.forward(q, k, v, layer, fb)
        torch.cuda.synchronize()
        assert torch.equal(got, eager)


@pytest.mark.parametrize("dtype,hq,hkv,dim", [(torch.bfloat16, 12, 2, 128), (torch.float16, 28, 4, 128)])
def test_extend_call_replayed_on_new_inputs(gpu, dtype, hq, hkv, dim):
    """Extend attention with its KV write (one chunk of a chunked prefill: new tokens over a cached
    prefix) captured once and replayed on new rows."""
    import baby_sglang_amd as pkg

    seq_lens, prefix = [330, 70, 200, 129], [200, 0, 64, 128]
    cpu, r2t, kv = random_decode_problem(gpu, dtype, hq, hkv, dim, seq_lens, seed=hq)
    n = len(seq_lens)
    rows, seq = cpu["req_pool_indices"].to(gpu), cpu["seq_lens"].to(gpu)
    pre = torch.tensor(prefix, dtype=torch.int64, device=gpu)
    loc = torch.cat([r2t.req_to_token[rows[i], prefix[i] : seq_lens[i]] for i in range(n)]).contiguous()
    n_new = int(loc.numel())
    layer = types.SimpleNamespace(layer_id=0, scaling=dim**-0.5)
    fb = pkg.SimplifiedForwardBatch.create_prefill_batch(torch.zeros(n_new, dtype=torch.int64, device=gpu), rows, seq, loc, r2t, kv, pre)
    q = torch.empty((n_new, hq, dim), dtype=dtype, device=gpu)
    k = torch.empty((n_new, hkv, dim), dtype=dtype, device=gpu)
    v = torch.empty((n_new, hkv, dim), dtype=dtype, device=gpu)
    gen = torch.Generator().manual_seed(2)

    def refill():
        for t in (q, k, v):
            t.copy_(torch.randn(t.shape, generator=gen).to(dtype))

    refill()
    out = torch.empty((n_new, hq * dim), dtype=dtype, device=gpu)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        out.copy_(pkg.SimpleAttentionBackend.forward(q, k, v, layer, fb))
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out.copy_(pkg.SimpleAttentionBackend.forward(q, k, v, layer, fb))
    k_ref, v_ref = cpu["k_slab"].clone(), cpu["v_slab"].clone()
    for _ in range(2):
        refill()
        graph.replay()
        torch.cuda.synchronize()
        got = out.clone()
        orc.kv_scatter_ref(k_ref, v_ref, loc.cpu(), k.cpu(), v.cpu())
        assert torch.equal(kv.k_buffer[0].cpu(), k_ref) and torch.equal(kv.v_buffer[0].cpu(), v_ref)
        eager = pkg.SimpleAttentionBackend.forward(q, k, v, layer, fb)
        torch.cuda.synchronize()
        assert torch.equal(got, eager)
        want = orc.prefill_exact(q.cpu(), k_ref, v_ref, cpu["req_to_token"], cpu["req_pool_indices"], cpu["seq_lens"],
                                 pre.cpu(), (seq - pre).cpu())
        err = (got.double().cpu() - want).abs().max().item()
        assert err < (3e-2 if dtype == torch.bfloat16 else 4e-3), err  # (the element-wise bars are in test_gpu_parity.py)

