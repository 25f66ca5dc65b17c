"""BASELINE config 4 (tensor parallel over 2 / 4 / 8 ranks) through the HIP kernels on ONE GPU.

Every rank of a ``TPLayout`` (kv heads first, then round-robin sequence slices inside a kv
head) runs as a thread of this process (``LocalGroup``) with what it would hold on its own GPU:
its kv heads of the cache in its own pool, a local page table over the positions it owns, its q
heads of the query.  It calls the product's decode entry point (``bsgl_paged_decode`` through
``_hip.paged_decode``: fp32 output + log-sum-exp when the sequence is sliced), with the step's new
token written by the kernel on the rank that owns its position, and then the product's
``tp_attention_exchange`` (``bsgl_merge_attn_states`` across the slices of a kv head).  Only the
transport differs from a multi-GPU run: tensors move between threads instead of over RCCL.
The result on every rank is compared with the UNSHARDED float64 oracle; the cache rows of the new
tokens are compared bit for bit.  (The reference has the collectives only, tp_runtime.py:10-51;
the sharding is this build's.)
"""
import pytest
import torch

from oracle import attn_oracle as orc
from test_gpu_parity import TOL, assert_close

pytestmark = pytest.mark.gpu


def _global_problem(dtype, hq, hkv, dim, seq_lens, seed):
    """Per request: K / V rows of every position (the last one is the step's new token), q."""
    gen = torch.Generator().manual_seed(seed)
    B, S = len(seq_lens), max(seq_lens)
    k_all = torch.randn((B, S, hkv, dim), generator=gen).to(dtype)
    v_all = torch.randn((B, S, hkv, dim), generator=gen).to(dtype)
    q = torch.randn((B, hq, dim), generator=gen).to(dtype)
    return k_all, v_all, q


def _unsharded_oracle(k_all, v_all, q, seq_lens, pick=None):
    """decode_exact over one slab that holds every (request, position) row."""
    B, S = k_all.shape[0], k_all.shape[1]
    pick = list(range(B)) if pick is None else pick
    k_slab = torch.cat([k_all.new_zeros((1,) + k_all.shape[2:]), k_all.reshape((B * S,) + k_all.shape[2:])])
    v_slab = torch.cat([v_all.new_zeros((1,) + v_all.shape[2:]), v_all.reshape((B * S,) + v_all.shape[2:])])
    table = (torch.arange(B * S, dtype=torch.int32) + 1).view(B, S)
    rows = torch.tensor(pick)
    return orc.decode_exact(q[pick], k_slab, v_slab, table, rows, torch.tensor([seq_lens[i] for i in pick]))


def _rank_step(pkg, layout, group, dev, dtype, k_all, v_all, q, seq_lens, seed):
    """What one rank does in a decode step; returns (full output rows, its new-token cache rows ok)."""
    hip, tp = pkg._hip, pkg.tp_runtime
    B, S = k_all.shape[0], k_all.shape[1]
    hkv_l, dim = layout.local_kv_heads, k_all.shape[3]
    gen = torch.Generator().manual_seed(seed * 131 + layout.rank)
    seq = torch.tensor(seq_lens)
    local_len = layout.local_len(seq)
    n_local = int(local_len.sum())
    pool = pkg.MHATokenToKVPool(n_local + 8, dtype, hkv_l, dim, 1, str(dev))
    r2t = pkg.ReqToTokenPool(B, max(int(local_len.max()), 1) + 3, str(dev))
    slots = (torch.randperm(n_local + 8, generator=gen)[:n_local] + 1).to(torch.int32)
    table = torch.zeros(tuple(r2t.req_to_token.shape), dtype=torch.int32)
    k_loc = torch.zeros((n_local + 9, hkv_l, dim), dtype=dtype)
    v_loc = torch.zeros_like(k_loc)
    loc = torch.full((B,), -1, dtype=torch.int32)  # -1: the new token's position is another rank's
    used = 0
    k_mine, v_mine = layout.shard_kv(k_all.view(B * S, -1, dim)).view(B, S, hkv_l, dim), layout.shard_kv(v_all.view(B * S, -1, dim)).view(B, S, hkv_l, dim)
    for i, n in enumerate(seq_lens):
        pos = torch.arange(n)[layout.seq_rank :: layout.tp_seq]
        assert len(pos) == int(local_len[i])
        mine = slots[used : used + len(pos)]
        used += len(pos)
        table[i, : len(pos)] = mine
        k_loc[mine.long()] = k_mine[i, pos]
        v_loc[mine.long()] = v_mine[i, pos]
        if n > 0 and layout.owns_position(n - 1):
            loc[i] = mine[-1]
            k_loc[int(mine[-1])] = 0  # not in the cache yet: the kernel writes it
            v_loc[int(mine[-1])] = 0
    pool.k_buffer[0].copy_(k_loc)
    pool.v_buffer[0].copy_(v_loc)
    r2t.req_to_token.copy_(table)
    last = (seq - 1).clamp_min(0)
    k_new = k_mine[torch.arange(B), last].contiguous().to(dev)
    v_new = v_mine[torch.arange(B), last].contiguous().to(dev)
    k_slab, v_slab = pool.get_kv_buffer(0)
    sliced = layout.tp_seq > 1
    lse = torch.empty((B, layout.local_q_heads), dtype=torch.float32, device=dev) if sliced else None
    out = hip.paged_decode(
        layout.shard_q(q).contiguous().to(dev), k_slab, v_slab, r2t.req_to_token, torch.arange(B, device=dev), local_len.to(dev),
        out_dtype=torch.float32 if sliced else dtype, lse=lse,
        k_new=k_new, v_new=v_new, out_cache_loc=loc.to(dev), fuse_kv_write=True,
    )
    full = tp.tp_attention_exchange(layout, out, lse, group, dtype)
    own = loc >= 0
    rows_ok = torch.equal(k_slab[loc[own].long().to(dev)].cpu(), k_new.cpu()[own]) and torch.equal(v_slab[loc[own].long().to(dev)].cpu(), v_new.cpu()[own])
    # nothing but the new tokens' rows changed
    k_loc[loc[own].long()] = k_new.cpu()[own]
    v_loc[loc[own].long()] = v_new.cpu()[own]
    rest_ok = torch.equal(k_slab.cpu(), k_loc) and torch.equal(v_slab.cpu(), v_loc)
    return full, rows_ok and rest_ok


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("hq,hkv,dim", [(12, 2, 128), (28, 4, 128)])
@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_decode_step_equals_the_unsharded_oracle(gpu, world, hq, hkv, dim, dtype):
    import baby_sglang_amd as pkg

    # ragged lengths: shorter than the number of sequence slices (ranks left without a token),
    # tile and block borders per slice, lengths whose last position falls on every slice
    seq_lens = [1, 2, 3, 5, 8, 17, 64, 300, 1025, 7, 129, 2050, 33]
    k_all, v_all, q = _global_problem(dtype, hq, hkv, dim, seq_lens, seed=world * 100 + hq)
    want = _unsharded_oracle(k_all, v_all, q, seq_lens)
    group = pkg.LocalGroup(world)

    def body(rank):
        layout = pkg.TPLayout(rank, world, hq, hkv)
        return _rank_step(pkg, layout, group, gpu, dtype, k_all, v_all, q, seq_lens, seed=7)

    results = group.run(body)
    torch.cuda.synchronize()
    for rank, (full, rows_ok) in enumerate(results):
        assert rows_ok, f"rank {rank}: cache rows of the new tokens"
        assert tuple(full.shape) == (len(seq_lens), hq * dim) and full.dtype == dtype
        assert_close(full, want, dtype, f"world {world} rank {rank} vs the unsharded oracle")
    # every rank holds the same bits
    for full, _ in results[1:]:
        assert torch.equal(full, results[0][0])


@pytest.mark.parametrize("world", [8, 2])
def test_headline_batch_sharded_over_the_ranks(gpu, world):
    """Qwen2.5-1.5B shapes, 64 requests x 2048 tokens, bf16, over 8 ranks (2 kv-head shards x 4
    sequence slices) and over 2 (kv-head shards only): all ranks' rows equal, ten requests
    against the unsharded float64 oracle."""
    import baby_sglang_amd as pkg

    hq, hkv, dim, dtype = 12, 2, 128, torch.bfloat16
    seq_lens = [2048] * 64
    k_all, v_all, q = _global_problem(dtype, hq, hkv, dim, seq_lens, seed=world)
    pick = [0, 7, 13, 21, 30, 31, 42, 50, 57, 63]
    want = _unsharded_oracle(k_all, v_all, q, seq_lens, pick)
    group = pkg.LocalGroup(world)

    def body(rank):
        layout = pkg.TPLayout(rank, world, hq, hkv)
        return _rank_step(pkg, layout, group, gpu, dtype, k_all, v_all, q, seq_lens, seed=3)

    results = group.run(body)
    torch.cuda.synchronize()
    for rank, (full, rows_ok) in enumerate(results):
        assert rows_ok, f"rank {rank}: cache rows of the new tokens"
        assert torch.equal(full, results[0][0])
    assert_close(results[0][0][pick], want, dtype, f"64 x 2048 over {world} ranks vs the unsharded oracle")
    assert TOL[dtype]["rtol"] == 8e-3  # (the decode tolerance, not a looser one)
