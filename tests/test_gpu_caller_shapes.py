"""What callers hand the backend besides dense tensors: q, k and v as column slices of one projection
output (the reference's qkv split, qwen2.py:285-290, leaves v -- and here also q and k -- with the row
stride of the merged row), through SimpleAttentionBackend.forward for a decode step and for an extend
call, against the float64 oracle with the cache rows compared bit for bit."""
import types

import pytest
import torch

from oracle import attn_oracle as orc
from test_gpu_parity import assert_close, random_decode_problem

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype,hq,hkv,dim", [(torch.bfloat16, 12, 2, 128), (torch.float16, 14, 2, 64), (torch.float32, 4, 4, 64)])
def test_projection_slices_as_q_k_v(gpu, dtype, hq, hkv, dim):
    import baby_sglang_amd as pkg

    g = torch.Generator().manual_seed(hq)
    seq_lens = [5, 700, 64, 1300, 2]
    n = len(seq_lens)
    cpu, r2t, kv = random_decode_problem(gpu, dtype, hq, hkv, dim, seq_lens, seed=3)
    rows, seq = cpu["req_pool_indices"].to(gpu), cpu["seq_lens"].to(gpu)
    layer = types.SimpleNamespace(layer_id=0, scaling=dim**-0.5)
    k_ref, v_ref = cpu["k_slab"].clone(), cpu["v_slab"].clone()

    def split(t_rows):
        qkv = torch.randn((t_rows, (hq + 2 * hkv) * dim), generator=g).to(dtype).to(gpu)
        q = qkv[:, : hq * dim].view(t_rows, hq, dim)
        k = qkv[:, hq * dim : (hq + hkv) * dim].view(t_rows, hkv, dim)
        v = qkv[:, (hq + hkv) * dim :].view(t_rows, hkv, dim)
        assert not q.is_contiguous() and not v.is_contiguous()
        return q, k, v

    # a decode step
    q, k, v = split(n)
    loc = r2t.req_to_token[rows, seq - 1].contiguous()
    fb = pkg.SimplifiedForwardBatch.create_decode_batch(torch.zeros(n, dtype=torch.int64, device=gpu), rows, seq, loc, r2t, kv)
    out = pkg.SimpleAttentionBackend.forward(q, k, v, layer, fb)
    orc.kv_scatter_ref(k_ref, v_ref, loc.cpu(), k.cpu(), v.cpu())
    want = orc.decode_exact(q.cpu(), k_ref, v_ref, cpu["req_to_token"], cpu["req_pool_indices"], cpu["seq_lens"])
    assert_close(out, want, dtype, "decode with strided q / k / v")
    assert torch.equal(kv.k_buffer[0].cpu(), k_ref) and torch.equal(kv.v_buffer[0].cpu(), v_ref)

    # an extend call over cached prefixes
    pre = torch.tensor([0, 500, 10, 1000, 1], dtype=torch.int64)
    ext = cpu["seq_lens"] - pre
    loc = torch.cat([r2t.req_to_token[rows[i], int(pre[i]) : seq_lens[i]] for i in range(n)]).contiguous()
    total = int(ext.sum())
    q, k, v = split(total)
    fb = pkg.SimplifiedForwardBatch.create_prefill_batch(torch.zeros(total, dtype=torch.int64, device=gpu), rows, seq, loc, r2t, kv, pre.to(gpu))
    out = pkg.SimpleAttentionBackend.forward(q, k, v, layer, fb)
    orc.kv_scatter_ref(k_ref, v_ref, loc.cpu(), k.cpu(), v.cpu())
    assert torch.equal(kv.k_buffer[0].cpu(), k_ref) and torch.equal(kv.v_buffer[0].cpu(), v_ref)
    want = orc.prefill_exact(q.cpu(), k_ref, v_ref, cpu["req_to_token"], cpu["req_pool_indices"], cpu["seq_lens"], pre, ext)
    err = (out.double().cpu() - want).abs()
    bound = {torch.bfloat16: 3e-2, torch.float16: 4e-3, torch.float32: 1e-4}[dtype]
    assert float(err.max()) < bound, float(err.max())  # (the element-wise bars are in test_gpu_parity.py)


def test_two_threads_on_two_streams(gpu):
    """The reference calls its backend from a worker thread on a side stream while the main thread
    may use another (overlap_worker.py:104-127): two threads, each on a stream of its own with a
    batch of its own, decode at the same time through ONE backend class; every result must be its
    own batch's (split-merge workspaces and call descriptors are kept per stream / per thread)."""
    import threading

    import baby_sglang_amd as pkg

    dtype, hq, hkv, dim = torch.bfloat16, 12, 2, 128
    layer = types.SimpleNamespace(layer_id=0, scaling=dim**-0.5)
    jobs = []
    for t, seq_lens in enumerate(([900, 40, 1300, 7, 512, 2000], [64, 1800, 3, 700])):
        cpu, r2t, kv = random_decode_problem(gpu, dtype, hq, hkv, dim, seq_lens, seed=20 + t)
        rows, seq = cpu["req_pool_indices"].to(gpu), cpu["seq_lens"].to(gpu)
        fb = pkg.SimplifiedForwardBatch.create_decode_batch(torch.zeros(len(seq_lens), dtype=torch.int64, device=gpu), rows, seq,
                                                            torch.zeros(len(seq_lens), dtype=torch.int32, device=gpu), r2t, kv)
        want = orc.decode_exact(cpu["q"], cpu["k_slab"], cpu["v_slab"], cpu["req_to_token"], cpu["req_pool_indices"], cpu["seq_lens"])
        jobs.append((cpu["q"].to(gpu), fb, want, torch.cuda.Stream()))
    torch.cuda.synchronize()
    errors = []

    def body(q, fb, want, stream):
        try:
            with torch.cuda.stream(stream):
                for _ in range(40):
                    out = pkg.SimpleAttentionBackend.forward(q, None, None, layer, fb, save_kv_cache=False)
                stream.synchronize()
                assert_close(out, want, dtype, "two threads, two streams")
        except BaseException as exc:  # noqa: BLE001  (re-raised on the main thread)
            errors.append(exc)

    threads = [threading.Thread(target=body, args=j) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]


def test_a_batch_that_names_its_longest_request(gpu):
    """A scheduler that knows its longest request on the host says so on the batch
    (``SimplifiedForwardBatch.max_seq_len``, not a field of the reference's descriptor); the backend
    hands it to the library as a hint.  Same batch object, with the hint, without it, and with one
    that is wrong: all three against the oracle, the new rows written by the kernel each time."""
    import baby_sglang_amd as pkg

    dtype, hq, hkv, dim = torch.bfloat16, 12, 2, 128
    layer = types.SimpleNamespace(layer_id=0, scaling=dim**-0.5)
    gen = torch.Generator().manual_seed(3)
    seq_lens = torch.randint(1, 201, (64,), generator=gen).tolist()
    cpu, r2t, kv = random_decode_problem(gpu, dtype, hq, hkv, dim, seq_lens, seed=91)
    k_new = torch.randn((64, hkv, dim), generator=gen).to(dtype)
    v_new = torch.randn((64, hkv, dim), generator=gen).to(dtype)
    loc = cpu["req_to_token"][cpu["req_pool_indices"], cpu["seq_lens"] - 1].to(torch.int32)
    orc.kv_scatter_ref(cpu["k_slab"], cpu["v_slab"], loc, k_new, v_new)
    want = orc.decode_exact(cpu["q"], cpu["k_slab"], cpu["v_slab"], cpu["req_to_token"], cpu["req_pool_indices"], cpu["seq_lens"])
    q, kn, vn = cpu["q"].to(gpu), k_new.to(gpu), v_new.to(gpu)
    for hint in (max(seq_lens), 0, 5):
        fb = pkg.SimplifiedForwardBatch.create_decode_batch(torch.zeros(64, dtype=torch.int64, device=gpu), cpu["req_pool_indices"].to(gpu),
                                                            cpu["seq_lens"].to(gpu), loc.to(gpu), r2t, kv)
        assert fb.max_seq_len == 0
        fb.max_seq_len = hint
        out = pkg.SimpleAttentionBackend.forward(q, kn, vn, layer, fb)
        assert_close(out, want, dtype, f"hint {hint}")
        k_slab, v_slab = kv.get_kv_buffer(0)
        assert torch.equal(k_slab.cpu(), cpu["k_slab"]) and torch.equal(v_slab.cpu(), cpu["v_slab"])


def test_one_batch_object_rebound_to_new_index_tensors(gpu):
    """A caller may keep ONE batch object and bind new index tensors to it from step to step (the
    dataclass is mutable; a static batch is a natural way to drive graph replays).  The backend keeps
    the pointers it looked up on the batch object: they must be looked up again when the tensors are
    other objects -- here the second step runs fewer tokens per request, from freshly made tensors,
    after the first step's tensors are gone."""
    import gc

    import baby_sglang_amd as pkg

    dtype, hq, hkv, dim = torch.bfloat16, 12, 2, 128
    seq_lens = [900, 40, 1500, 7]
    n = len(seq_lens)
    cpu, r2t, kv = random_decode_problem(gpu, dtype, hq, hkv, dim, seq_lens, seed=11)
    layer = types.SimpleNamespace(layer_id=0, scaling=dim**-0.5)
    g = torch.Generator().manual_seed(5)
    q = torch.randn((n, hq, dim), generator=g).to(dtype)
    rows = cpu["req_pool_indices"].to(gpu)
    seq = cpu["seq_lens"].to(gpu)
    loc = r2t.req_to_token[rows, seq - 1].contiguous()
    fb = pkg.SimplifiedForwardBatch.create_decode_batch(torch.zeros(n, dtype=torch.int64, device=gpu), rows, seq, loc, r2t, kv)
    out = pkg.SimpleAttentionBackend.forward(q.to(gpu), None, None, layer, fb)
    want = orc.decode_exact(q, cpu["k_slab"], cpu["v_slab"], cpu["req_to_token"], cpu["req_pool_indices"], cpu["seq_lens"])
    assert_close(out, want, dtype, "first step")
    # second step on the SAME batch object: shorter requests, other rows first, all in new tensors
    order = torch.tensor([2, 0, 3, 1])
    seq2_cpu = (cpu["seq_lens"][order] // 2).clamp_min(1)
    rows2_cpu = cpu["req_pool_indices"][order]
    del rows, seq, loc
    gc.collect()
    torch.cuda.empty_cache()
    junk = [torch.full((n,), 10**9, dtype=torch.int64, device=gpu) for _ in range(64)]  # (what freed index tensors may turn into)
    fb.req_pool_indices = rows2_cpu.to(gpu)
    fb.seq_lens = seq2_cpu.to(gpu)
    fb.out_cache_loc = r2t.req_to_token[fb.req_pool_indices, fb.seq_lens - 1].contiguous()
    fb.positions = fb.seq_lens - 1
    out2 = pkg.SimpleAttentionBackend.forward(q.to(gpu), None, None, layer, fb)
    want2 = orc.decode_exact(q, cpu["k_slab"], cpu["v_slab"], cpu["req_to_token"], rows2_cpu, seq2_cpu)
    assert_close(out2, want2, dtype, "second step, rebound tensors")
    del junk
