"""Tensor-parallel runtime on CPU: two gloo ranks (no GPU).  Covers the four
reference wrappers (tp_runtime.py:10-51) and the attention sharding / exchange:
head sharding (world == kv heads) and sequence slicing (world > kv heads), each
checked against the unsharded float64 oracle."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _merge_fn(parts_o, parts_lse, dtype):
    from oracle import attn_oracle as orc

    out, _ = orc.merge_states_exact(parts_o, parts_lse)
    return out.to(dtype)


def _worker(rank, world, port, n_kv_heads):
    import sys

    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from baby_sglang_amd import tp_runtime as tp
    from oracle import attn_oracle as orc

    r, w, group = tp.init_tp(backend="gloo")
    assert (r, w) == (rank, world)

    # --- the four wrappers
    x = torch.full((3,), float(rank + 1))
    assert tp.tp_all_reduce_sum(x, group) is x and x.tolist() == [world * (world + 1) / 2] * 3
    g = tp.tp_all_gather(torch.full((2, 2), float(rank)), group)
    assert tuple(g.shape) == (2 * world, 2) and all(bool(g[2 * r : 2 * r + 2].eq(r).all()) for r in range(world))
    b = torch.tensor([7.0 if rank == 1 else 0.0])
    assert tp.tp_broadcast(b, 1, group) is b and b.item() == 7.0

    # --- sharded decode attention == unsharded
    hq, dim = 4 * n_kv_heads, 64
    layout = tp.TPLayout(rank, world, hq, n_kv_heads)
    gen = torch.Generator().manual_seed(0)  # same data on both ranks
    seq_lens = [5, 12, 1, 9, 30] if world > 2 else [5, 12, 1, 9]  # (5 rows over 2 slices: a padded share)
    n_slots = sum(seq_lens) + 1
    k_slab = torch.randn((n_slots, n_kv_heads, dim), generator=gen)
    v_slab = torch.randn((n_slots, n_kv_heads, dim), generator=gen)
    q = torch.randn((len(seq_lens), hq, dim), generator=gen)
    table = torch.zeros((len(seq_lens), 32), dtype=torch.int32)
    perm = torch.randperm(n_slots - 1, generator=gen) + 1
    used = 0
    for i, n in enumerate(seq_lens):
        table[i, :n] = perm[used : used + n].to(torch.int32)
        used += n
    rows = torch.arange(len(seq_lens))
    seq = torch.tensor(seq_lens)
    full = orc.decode_exact(q, k_slab, v_slab, table, rows, seq)

    # this rank's shard: its kv heads, and every tp_seq-th position starting at seq_rank
    local_table = torch.zeros_like(table)
    local_len = layout.local_len(seq)
    for i, n in enumerate(seq_lens):
        mine = table[i, :n][layout.seq_rank :: layout.tp_seq]
        assert len(mine) == int(local_len[i])
        local_table[i, : len(mine)] = mine
    o, lse = orc.decode_exact(layout.shard_q(q), layout.shard_kv(k_slab), layout.shard_kv(v_slab),
                              local_table, rows, local_len, return_lse=True)
    assert o.shape[1] == layout.local_q_heads * dim
    if layout.tp_seq == 1:
        got = tp.tp_attention_exchange(layout, o, None, group, torch.float64)
    else:
        got = tp.tp_attention_exchange(layout, o.float(), lse.float(), group, torch.float64, merge_fn=_merge_fn)
    assert tuple(got.shape) == tuple(full.shape)
    tol = 1e-12 if layout.tp_seq == 1 else 2e-6  # the sequence-sliced state travels as fp32
    assert (got - full).abs().max().item() < tol
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_kv_heads", [2, 1])
def test_two_ranks(n_kv_heads):
    mp.spawn(_worker, args=(2, _free_port(), n_kv_heads), nprocs=2, join=True)


def test_four_ranks_two_kv_shards_of_two_sequence_slices():
    """world 4 over 2 kv heads: the partial states are traded inside each kv head's pair of
    ranks (a sub-group), the merged rows gathered over all four."""
    mp.spawn(_worker, args=(4, _free_port(), 2), nprocs=4, join=True)


def _replica_worker(rank, world, port):
    import sys

    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from baby_sglang_amd import tp_runtime as tp

    r, w, world_group = tp.init_tp(backend="gloo")
    group, tp_rank, replica, n_replicas = tp.split_tp_groups(2, world_group)
    assert (tp_rank, replica, n_replicas) == (rank % 2, rank // 2, world // 2)
    assert dist.get_world_size(group=group) == 2 and dist.get_rank(group=group) == tp_rank
    # head-sharded exchange inside each replica, on the replica's own rows
    layout = tp.TPLayout(tp_rank, 2, 4, 2)
    rows, dim = 3, 8
    full = torch.arange(rows * 4 * dim, dtype=torch.float64).view(rows, 4, dim) + 1000.0 * replica
    got = tp.tp_attention_exchange(layout, layout.shard_q(full).reshape(rows, -1).contiguous(), None, group, torch.float64)
    assert torch.equal(got, full.view(rows, -1))
    # sums stay inside a replica
    x = torch.tensor([float(rank)])
    tp.tp_all_reduce_sum(x, group)
    assert x.item() == float(4 * replica + 1)
    # one group of all ranks: the world group itself
    g_all, r_all, rep, n_rep = tp.split_tp_groups(world, world_group)
    assert (r_all, rep, n_rep) == (rank, 0, 1) and dist.get_world_size(group=g_all) == world
    with pytest.raises(ValueError):
        tp.split_tp_groups(3, world_group)
    dist.barrier()
    dist.destroy_process_group()


def test_four_ranks_as_two_replicas_of_a_two_rank_group():
    """world 4 = 2 replicas x tp 2: each pair shards the kv heads of its own requests and
    gathers its own rows; nothing crosses between replicas."""
    mp.spawn(_replica_worker, args=(4, _free_port()), nprocs=4, join=True)


def _no_gpu_worker(rank, world, port):
    import sys

    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from baby_sglang_amd import tp_runtime as tp

    _, _, group = tp.init_tp(backend="gloo")
    # no GPU: no rank can make its window; every rank still goes through the same exchanges, all
    # agree not to switch over, and the sums keep going through the group's own all-reduce
    assert tp.use_p2p_all_reduce(group) is False
    assert tp._P2P == [None, None] and tp._P2P_WHY[0]
    x = torch.full((8,), float(rank + 1))
    assert tp.tp_all_reduce_sum(x, group) is x and x.tolist() == [world * (world + 1) / 2] * 8
    tp.release_p2p_all_reduce()  # (nothing live: no exchange)
    dist.barrier()
    dist.destroy_process_group()


def test_one_shot_all_reduce_is_declined_by_all_ranks_without_a_gpu():
    if torch.cuda.is_available():
        pytest.skip("the failure path: only where no window can be made")
    mp.spawn(_no_gpu_worker, args=(2, _free_port()), nprocs=2, join=True)


def test_layouts():
    import sys

    sys.path.insert(0, ROOT)
    from baby_sglang_amd.tp_runtime import TPLayout

    # Qwen2.5-1.5B: 12 q heads, 2 kv heads
    for world, (tp_kv, tp_seq) in {1: (1, 1), 2: (2, 1), 4: (2, 2), 8: (2, 4)}.items():
        seen = set()
        for rank in range(world):
            lay = TPLayout(rank, world, 12, 2)
            assert (lay.tp_kv, lay.tp_seq) == (tp_kv, tp_seq)
            assert lay.local_q_heads == 12 // tp_kv and lay.local_kv_heads == 2 // tp_kv
            seen.add((lay.kv_head_range, lay.seq_rank))
            lens = torch.tensor([0, 1, 7, 2048])
            total = sum(TPLayout(r, world, 12, 2).local_len(lens) for r in range(world) if r // tp_seq == lay.kv_rank)
            assert total.tolist() == lens.tolist()
        assert len(seen) == world
    with pytest.raises(ValueError):
        TPLayout(0, 3, 12, 2)  # 3 neither divides nor is a multiple of 2 kv heads
    with pytest.raises(ValueError):
        TPLayout(5, 4, 12, 2)


def test_world_of_one_needs_no_process_group():
    import sys

    sys.path.insert(0, ROOT)
    from baby_sglang_amd import tp_runtime as tp

    x = torch.ones(2)
    assert tp.tp_all_reduce_sum(x, None) is x
    assert tp.tp_all_gather(x, None) is x
    assert tp.tp_broadcast(x, 0, None) is x


def _sharded_decode_check(tp, orc, layout, group, n_kv_heads, seq_lens):
    """One rank's part of a sharded decode on the CPU oracle + the exchange; returns the error
    against the unsharded result."""
    hq, dim = 4 * n_kv_heads, 64
    gen = torch.Generator().manual_seed(0)  # same data on every rank
    n_slots = sum(seq_lens) + 1
    k_slab = torch.randn((n_slots, n_kv_heads, dim), generator=gen)
    v_slab = torch.randn((n_slots, n_kv_heads, dim), generator=gen)
    q = torch.randn((len(seq_lens), hq, dim), generator=gen)
    table = torch.zeros((len(seq_lens), 32), dtype=torch.int32)
    perm = torch.randperm(n_slots - 1, generator=gen) + 1
    used = 0
    for i, n in enumerate(seq_lens):
        table[i, :n] = perm[used : used + n].to(torch.int32)
        used += n
    rows, seq = torch.arange(len(seq_lens)), torch.tensor(seq_lens)
    full = orc.decode_exact(q, k_slab, v_slab, table, rows, seq)
    local_table, local_len = torch.zeros_like(table), layout.local_len(seq)
    for i, n in enumerate(seq_lens):
        mine = table[i, :n][layout.seq_rank :: layout.tp_seq]
        local_table[i, : len(mine)] = mine
    o, lse = orc.decode_exact(layout.shard_q(q), layout.shard_kv(k_slab), layout.shard_kv(v_slab),
                              local_table, rows, local_len, return_lse=True)
    if layout.tp_seq == 1:
        got = tp.tp_attention_exchange(layout, o, None, group, torch.float64)
    else:
        got = tp.tp_attention_exchange(layout, o.float(), lse.float(), group, torch.float64, merge_fn=_merge_fn)
    return (got - full).abs().max().item()


@pytest.mark.parametrize("world,n_kv_heads", [(2, 2), (2, 1), (4, 2), (8, 2), (8, 4), (3, 1)])
def test_ranks_as_threads_of_one_process(world, n_kv_heads):
    """``LocalGroup``: the ranks of a tensor-parallel group as threads (the one-GPU rehearsal of
    the multi-GPU path).  The four wrappers, sub-groups, the attention exchange, and a failure on
    one rank reaching the caller instead of hanging the others."""
    import sys

    sys.path.insert(0, ROOT)
    from baby_sglang_amd import tp_runtime as tp
    from oracle import attn_oracle as orc

    group = tp.LocalGroup(world)

    def rank_body(rank):
        x = torch.full((3,), float(rank + 1))
        assert tp.tp_all_reduce_sum(x, group) is x and x.tolist() == [world * (world + 1) / 2] * 3
        g = tp.tp_all_gather(torch.full((2, 2), float(rank)), group)
        assert tuple(g.shape) == (2 * world, 2) and all(bool(g[2 * r : 2 * r + 2].eq(r).all()) for r in range(world))
        b = torch.tensor([7.0 if rank == 1 else 0.0])
        assert tp.tp_broadcast(b, 1, group) is b and b.item() == 7.0
        if world % 2 == 0:
            sub, sub_rank, replica, n_rep = tp.split_tp_groups(2, group)
            assert (sub_rank, replica, n_rep) == (rank % 2, rank // 2, world // 2)
            y = torch.tensor([float(rank)])
            tp.tp_all_reduce_sum(y, sub)
            assert y.item() == float(4 * replica + 1)
        layout = tp.TPLayout(rank, world, 4 * n_kv_heads, n_kv_heads)
        return _sharded_decode_check(tp, orc, layout, group, n_kv_heads, [5, 12, 1, 9, 30])

    errs = group.run(rank_body)
    assert max(errs) < 2e-6

    def failing(rank):
        if rank == world - 1:
            raise KeyError("boom")
        tp.tp_all_gather(torch.ones(1), group)

    with pytest.raises(KeyError):
        group.run(failing)
