"""Parity at every size bench.py quotes a number for.

The workloads are built by bench.py's own constructors (``bench.Workload`` with one layer instead
of 28, ``bench.extend_problem``), so the page tables are exactly the ones timed: random
permutation, contiguous, radix-shared prefixes, ragged lengths, pair-shared rows.  At these sizes
the float64 oracle is run on a SAMPLE of the requests (it finishes in seconds); the new tokens'
cache rows are compared bit for bit over the whole batch.  Decode tolerance as everywhere
(bf16: rtol 8e-3, atol 2e-3); the bf16 matrix-core extend is held to the rounding of its terms
(probabilities rounded to bf16 once, as torch's SDPA does: see test_prefill_kernels_agree).
"""
import types

import pytest
import torch

from oracle import attn_oracle as orc
from test_gpu_parity import assert_close, random_decode_problem

pytestmark = pytest.mark.gpu

PICK = [0, 1, 8, 9, 23, 38, 47, 63]  # (both members of a pair, of a radix group, and the ends)


def _bench_args(**over):
    a = dict(batch=64, seq=2048, layers=1, max_context=4096, table="perm", kv_layout="interleaved", variant=0,
             splits=0, no_fuse=False)
    a.update(over)
    return types.SimpleNamespace(**a)


@pytest.fixture(scope="module")
def workload(gpu):
    import baby_sglang_amd as pkg
    import bench

    return bench.Workload(_bench_args(), gpu, pkg.TPLayout(0, 1, 12, 2))


@pytest.mark.parametrize("table", ["perm", "contig", "radix", "ragged", "pairs"])
def test_headline_batch_on_every_page_table_of_the_bench(gpu, workload, table):
    """64 x 2048 (ragged: lengths in [1024, 3072)), Qwen2.5-1.5B shapes, bf16, through the call
    bench.py times (``Workload.attention_layer``: new token written by the decode kernel)."""
    wl = workload
    wl.set_table(table)
    fused = table != "pairs"  # (requests 2i and 2i+1 name the SAME last slot there: no defined writer)
    wl.writes_kv = fused
    k_slab, v_slab = wl.kv.get_kv_buffer(0)
    k_before, v_before = k_slab.cpu(), v_slab.cpu()
    wl.attention_layer(0)
    torch.cuda.synchronize()
    got = wl.out.clone()
    k_ref, v_ref = k_before.clone(), v_before.clone()
    if fused:
        orc.kv_scatter_ref(k_ref, v_ref, wl.loc.cpu(), wl.k[0].cpu(), wl.v_rows(0).cpu())
    assert torch.equal(k_slab.cpu(), k_ref) and torch.equal(v_slab.cpu(), v_ref), "cache rows after the step"
    seq = wl.local_seq.cpu()
    want = orc.decode_exact(wl.q[0].cpu()[PICK], k_ref, v_ref, wl.r2t.req_to_token.cpu(), torch.tensor(PICK), seq[PICK])
    assert_close(got[PICK], want, torch.bfloat16, f"table {table}: {len(PICK)} requests vs the float64 oracle")
    wl.writes_kv = True
    # put the slabs back for the next table
    k_slab.copy_(k_before.to(gpu))
    v_slab.copy_(v_before.to(gpu))


def test_every_request_of_the_headline_batch_is_finite_and_split_invariant(gpu, workload):
    """All 64 rows, not a sample: finite, and equal (to the decode tolerance) between the
    length-balanced flat grid the bench runs and exactly-one-piece-per-request."""
    from baby_sglang_amd import _hip

    wl = workload
    wl.set_table("perm")
    k_slab, v_slab = wl.kv.get_kv_buffer(0)
    args = (wl.q[0], k_slab, v_slab, wl.r2t.req_to_token, wl.req, wl.local_seq)
    a = _hip.paged_decode(*args, n_splits=wl.n_splits).float()
    b = _hip.paged_decode(*args, n_splits=1).float()
    assert bool(torch.isfinite(a).all())
    assert torch.allclose(a, b, rtol=8e-3, atol=2e-3)


@pytest.mark.parametrize("seq", [1, 128, 2048, 4095])
def test_batch_one(gpu, seq):
    """BASELINE config 2: batch 1 decode at the context lengths the latency table quotes."""
    from baby_sglang_amd import _hip

    dtype, hq, hkv, dim = torch.bfloat16, 12, 2, 128
    cpu, r2t, kv = random_decode_problem(gpu, dtype, hq, hkv, dim, [seq], seed=seq)
    gen = torch.Generator().manual_seed(1)
    k_new = torch.randn((1, hkv, dim), generator=gen).to(dtype)
    v_new = torch.randn((1, hkv, dim), generator=gen).to(dtype)
    rows, seq_t = cpu["req_pool_indices"].to(gpu), cpu["seq_lens"].to(gpu)
    loc = r2t.req_to_token[rows, seq_t - 1].contiguous()
    k_ref, v_ref = cpu["k_slab"].clone(), cpu["v_slab"].clone()
    orc.kv_scatter_ref(k_ref, v_ref, loc.cpu(), k_new, v_new)
    want = orc.decode_exact(cpu["q"], k_ref, v_ref, cpu["req_to_token"], cpu["req_pool_indices"], cpu["seq_lens"])
    k_slab, v_slab = kv.get_kv_buffer(0)
    out = _hip.paged_decode(cpu["q"].to(gpu), k_slab, v_slab, r2t.req_to_token, rows, seq_t,
                            k_new=k_new.to(gpu), v_new=v_new.to(gpu), out_cache_loc=loc, fuse_kv_write=True)
    assert_close(out, want, dtype, f"batch 1 x {seq}")
    assert torch.equal(k_slab.cpu(), k_ref) and torch.equal(v_slab.cpu(), v_ref)


def test_extend_shape_of_the_bench(gpu):
    """BASELINE config 5 as bench.py times it: 7B shapes (28 q / 4 kv heads), 128 requests x 512 new
    tokens over 1536 cached, bf16, one layer call.  All rows finite; every row of six requests
    against the float64 oracle."""
    import bench
    from baby_sglang_amd import _hip

    p = bench.extend_problem(gpu)
    out = _hip.paged_prefill(p["q"], p["k_slab"], p["v_slab"], p["r2t"].req_to_token, p["rows"], p["seq"], p["pre"],
                             p["seq"] - p["pre"], out=p["out"])
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out.float()).all())
    pick = [0, 5, 64, 77, 126, 127]
    ext, hq, dim = p["ext"], p["hq"], p["dim"]
    k_cpu, v_cpu, table = p["k_slab"].cpu(), p["v_slab"].cpu(), p["r2t"].req_to_token.cpu()
    q_cpu = p["q"].view(p["batch"], ext, hq, dim)[pick].reshape(len(pick) * ext, hq, dim).cpu()
    rows, seq, pre = torch.tensor(pick), p["seq"].cpu()[pick], p["pre"].cpu()[pick]
    want = orc.prefill_exact(q_cpu, k_cpu, v_cpu, table, rows, seq, pre, seq - pre)
    spread = orc.prefill_exact(q_cpu, k_cpu, v_cpu.abs(), table, rows, seq, pre, seq - pre)
    got = out.view(p["batch"], ext, hq * dim)[pick].reshape(len(pick) * ext, hq * dim).double().cpu()
    err = (got - want).abs()
    bound = 2.0**-8 * want.abs() + 2.0**-8 * spread + 1e-4
    assert bool((err <= bound).all()), f"worst excess {(err - bound).max().item():.3e}"
    assert err.mean().item() < 1.5e-3


def test_long_contexts(gpu):
    """Contexts far beyond the bench's: a 40 000-token request decodes (many page-table blocks per
    piece) and a 700-token chunk extends over a 30 000-token prefix (the extend kernel streams 60
    page-table blocks); both against the float64 oracle."""
    from baby_sglang_amd import _hip

    dtype, hq, hkv, dim = torch.bfloat16, 12, 2, 128
    seq_lens = [40000, 1, 30700]
    cpu, r2t, kv = random_decode_problem(gpu, dtype, hq, hkv, dim, seq_lens, seed=77)
    k_slab, v_slab = kv.get_kv_buffer(0)
    rows, seq = cpu["req_pool_indices"].to(gpu), cpu["seq_lens"].to(gpu)
    want = orc.decode_exact(cpu["q"], cpu["k_slab"], cpu["v_slab"], cpu["req_to_token"], cpu["req_pool_indices"], cpu["seq_lens"])
    for n_splits in (None, 64, 3):
        got = _hip.paged_decode(cpu["q"].to(gpu), k_slab, v_slab, r2t.req_to_token, rows, seq, n_splits=n_splits)
        assert_close(got, want, dtype, f"long-context decode, n_splits {n_splits}")
    # extend: request 2 runs its last 700 tokens over the 30 000 before them
    pre = torch.tensor([30000], dtype=torch.int64)
    ext = torch.tensor([700], dtype=torch.int64)
    gen = torch.Generator().manual_seed(78)
    q = torch.randn((700, hq, dim), generator=gen).to(dtype)
    got = _hip.paged_prefill(q.to(gpu), k_slab, v_slab, r2t.req_to_token, rows[2:3], seq[2:3], pre.to(gpu), ext.to(gpu))
    want = orc.prefill_exact(q, cpu["k_slab"], cpu["v_slab"], cpu["req_to_token"], cpu["req_pool_indices"][2:3],
                             cpu["seq_lens"][2:3], pre, ext)
    err = (got.double().cpu() - want).abs()
    assert float(err.max()) < 2e-2 and float(err.mean()) < 1.5e-3, (float(err.max()), float(err.mean()))


@pytest.mark.parametrize("dtype,hq,hkv,dim,n,hi", [(torch.bfloat16, 12, 2, 128, 1100, 200), (torch.bfloat16, 12, 2, 128, 2100, 90),
                                                   (torch.float16, 16, 1, 128, 300, 700), (torch.bfloat16, 14, 2, 64, 1500, 120)])
def test_batches_beyond_the_piece_search(gpu, dtype, hq, hkv, dim, n, hi):
    """More than 1024 requests (the length-balanced piece list is built for up to 1024: above that
    the grid falls back to equal splits per request), the widest GQA group (16 q heads on one kv
    head) and a large folded-head batch, ragged lengths, each with the default, one and four splits."""
    from baby_sglang_amd import _hip

    g = torch.Generator().manual_seed(n)
    seq_lens = torch.randint(1, hi, (n,), generator=g).tolist()
    cpu, r2t, kv = random_decode_problem(gpu, dtype, hq, hkv, dim, seq_lens, seed=n)
    k_slab, v_slab = kv.get_kv_buffer(0)
    rows, seq = cpu["req_pool_indices"].to(gpu), cpu["seq_lens"].to(gpu)
    want = orc.decode_exact(cpu["q"], cpu["k_slab"], cpu["v_slab"], cpu["req_to_token"], cpu["req_pool_indices"], cpu["seq_lens"])
    for ns in (None, 1, 4):
        got = _hip.paged_decode(cpu["q"].to(gpu), k_slab, v_slab, r2t.req_to_token, rows, seq, n_splits=ns)
        assert_close(got, want, dtype, f"batch {n}, n_splits {ns}")


def test_hundreds_of_short_extends(gpu):
    """700 requests extending by 0 - 40 tokens each over ragged prefixes (the 16-position blocks with
    the q heads on the columns, empty requests in between)."""
    from baby_sglang_amd import _hip

    dtype, hq, hkv, dim, n = torch.bfloat16, 12, 2, 128, 700
    g = torch.Generator().manual_seed(5)
    seq_lens = torch.randint(2, 150, (n,), generator=g).tolist()
    cpu, r2t, kv = random_decode_problem(gpu, dtype, hq, hkv, dim, seq_lens, seed=n + 1)
    k_slab, v_slab = kv.get_kv_buffer(0)
    rows, seq = cpu["req_pool_indices"].to(gpu), cpu["seq_lens"].to(gpu)
    ext = torch.tensor([int(torch.randint(0, min(s, 40) + 1, (1,), generator=g)) for s in seq_lens], dtype=torch.int64)
    pre = cpu["seq_lens"] - ext
    q = torch.randn((int(ext.sum()), hq, dim), generator=g).to(dtype)
    got = _hip.paged_prefill(q.to(gpu), k_slab, v_slab, r2t.req_to_token, rows, seq, pre.to(gpu), ext.to(gpu))
    want = orc.prefill_exact(q, cpu["k_slab"], cpu["v_slab"], cpu["req_to_token"], cpu["req_pool_indices"], cpu["seq_lens"], pre, ext)
    err = (got.double().cpu() - want).abs()
    assert float(err.max()) < 3e-2 and float(err.mean()) < 1.5e-3, (float(err.max()), float(err.mean()))
