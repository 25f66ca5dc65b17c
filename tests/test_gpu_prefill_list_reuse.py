"""The extend launch's work list is made once per batch, not once per layer
(``BSGL_PREFILL_LIST_READY``; reference call pattern: one batch descriptor for all layers of a
forward, qwen2.py:292-315).  Through the product backend: the layers after the first skip the list's
launch and give what a fresh call gives, bit for bit; a new forward (the first layer again), another
batch, or the same batch object with rewritten lengths all get a new list."""
import types

import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(dev, ext_lens, prefix_lens, seed=0, layers=3, hq=12, hkv=2, dim=128, dtype=torch.bfloat16):
    import baby_sglang_amd as pkg

    gen = torch.Generator().manual_seed(seed)
    batch = len(ext_lens)
    seq = [p + e for p, e in zip(prefix_lens, ext_lens)]
    max_ctx = max(seq) + 8
    size = sum(seq) + 16
    r2t = pkg.ReqToTokenPool(batch, max_ctx, str(dev))
    kv = pkg.MHATokenToKVPool(size, dtype, hkv, dim, layer_num=layers, device=str(dev))
    kv.arena.normal_()
    perm = (torch.randperm(size, generator=gen) + 1).to(torch.int32)
    used = 0
    for i, n in enumerate(seq):
        r2t.req_to_token[i, :n] = perm[used : used + n].to(dev)
        used += n
    return pkg, r2t, kv, seq, gen


def _batch(pkg, dev, r2t, kv, seq, prefix_lens):
    batch = len(seq)
    loc = torch.cat([r2t.req_to_token[i, prefix_lens[i] : seq[i]] for i in range(batch)]).contiguous()
    return pkg.SimplifiedForwardBatch.create_prefill_batch(
        input_ids=torch.zeros(int(loc.numel()), dtype=torch.int64, device=dev),
        req_pool_indices=torch.arange(batch, device=dev), seq_lens=torch.tensor(seq, device=dev),
        out_cache_loc=loc, req_to_token_pool=r2t, token_to_kv_pool=kv,
        prefix_lens=torch.tensor(prefix_lens, device=dev))


def test_layers_after_the_first_reuse_the_list(gpu, monkeypatch):
    ext, pre = [8, 64, 1, 130], [200, 0, 513, 64]
    pkg, r2t, kv, seq, gen = _setup(gpu, ext, pre)
    from baby_sglang_amd import _hip

    seen = []
    real = _hip.paged_prefill

    def spy(*a, **kw):
        seen.append(bool(kw.get("list_ready", False)))
        return real(*a, **kw)

    monkeypatch.setattr(_hip, "paged_prefill", spy)
    n_tok = sum(ext)
    q = [torch.randn((n_tok, 12, 128), generator=gen).to(torch.bfloat16).to(gpu) for _ in range(3)]
    fb = _batch(pkg, gpu, r2t, kv, seq, pre)
    outs = []
    for layer in range(3):
        lay = types.SimpleNamespace(layer_id=layer, scaling=128**-0.5)
        outs.append(pkg.SimpleAttentionBackend.forward(q[layer], None, None, lay, fb, save_kv_cache=False))
    assert seen == [False, True, True]
    # a fresh batch object per layer never reuses: the same numbers, bit for bit
    for layer in range(3):
        lay = types.SimpleNamespace(layer_id=layer, scaling=128**-0.5)
        fresh = pkg.SimpleAttentionBackend.forward(q[layer], None, None, lay, _batch(pkg, gpu, r2t, kv, seq, pre), save_kv_cache=False)
        assert torch.equal(fresh, outs[layer])
    assert seen[3:] == [False, False, False]
    # the next forward of the SAME batch object starts at its first layer again: a new list
    seen.clear()
    lay0 = types.SimpleNamespace(layer_id=0, scaling=128**-0.5)
    again = pkg.SimpleAttentionBackend.forward(q[0], None, None, lay0, fb, save_kv_cache=False)
    assert seen == [False] and torch.equal(again, outs[0])


def test_a_batch_object_whose_lengths_are_rewritten_in_place(gpu):
    """One batch object kept from step to step, its tensors rewritten in place (a static batch): the
    first layer of the next forward must not take over the last forward's list."""
    ext1, pre1 = [40, 16, 8], [100, 300, 0]
    ext2, pre2 = [8, 16, 40], [132, 300, 0]  # same totals, same shapes, other lengths
    pkg, r2t, kv, seq1, gen = _setup(gpu, [48, 16, 40], [132, 300, 0], seed=1, layers=2)
    q = torch.randn((64, 12, 128), generator=gen).to(torch.bfloat16).to(gpu)
    fb = _batch(pkg, gpu, r2t, kv, [p + e for p, e in zip(pre1, ext1)], pre1)
    lays = [types.SimpleNamespace(layer_id=i, scaling=128**-0.5) for i in range(2)]
    for lay in lays:
        pkg.SimpleAttentionBackend.forward(q, None, None, lay, fb, save_kv_cache=False)
    # step 2: the same object and tensors, new contents
    fb.seq_lens.copy_(torch.tensor([p + e for p, e in zip(pre2, ext2)]))
    fb.prefix_lens.copy_(torch.tensor(pre2))
    fb.extended_lens.copy_(torch.tensor(ext2))
    got = [pkg.SimpleAttentionBackend.forward(q, None, None, lay, fb, save_kv_cache=False) for lay in lays]
    want_fb = _batch(pkg, gpu, r2t, kv, [p + e for p, e in zip(pre2, ext2)], pre2)
    for lay, g in zip(lays, got):
        want = pkg.SimpleAttentionBackend.forward(q, None, None, lay, _batch(pkg, gpu, r2t, kv, [p + e for p, e in zip(pre2, ext2)], pre2),
                                                  save_kv_cache=False)
        assert torch.equal(g, want)
    del want_fb
