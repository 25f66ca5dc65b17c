"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Run in the build container only (``python tests/golden/gen_golden.py``): it puts
``/root/reference`` on ``sys.path``, imports the reference's own
``memory_pool`` / ``forward_batch`` / ``attn_backend`` modules and records
inputs and outputs of ``SimpleAttentionBackend.forward`` on the CPU.  The
reference never travels: only the tensors written here are committed, and the
tests load them with ``torch.load(..., weights_only=True)``.

Files
  decode_*.pt     one decode step (attn_backend.py:112-149) incl. the KV write
  prefill_*.pt    one extend step over cached prefixes (attn_backend.py:34-110)
  pools.pt        alloc / free / clear traces of both pools (memory_pool.py:7-84)
  sched_trace.pt  ForwardBatch sequence produced by the reference Scheduler on a
                  tiny random-weight model: shared prefixes, two admission
                  waves, FIFO slot recycling (scheduler.py:242-507)
"""

from __future__ import annotations

import os
import sys
import types

import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)

import attn_backend as ref_backend  # noqa: E402
import forward_batch as ref_fb  # noqa: E402
import memory_pool as ref_pool  # noqa: E402

CONFIGS = {  # name: (n_q_heads, n_kv_heads, head_dim)
    "q15": (12, 2, 128),  # Qwen2.5-1.5B
    "q05": (14, 2, 64),  # Qwen2.5-0.5B
    "q7": (28, 4, 128),  # Qwen2.5-7B
}
DTYPES = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}


def build_tables(gen, seq_lens, share, n_extra_rows=2, slack=8):
    """Random-permutation slot assignment; ``share`` = (a, b, n): request b reuses
    the first n slots of request a (a radix-cache prefix hit)."""
    batch = len(seq_lens)
    size = sum(seq_lens) + slack
    max_ctx = max(seq_lens) + 3
    perm = (torch.randperm(size, generator=gen) + 1).to(torch.int32)
    table = torch.zeros((batch + n_extra_rows, max_ctx), dtype=torch.int32)
    rows = torch.randperm(batch + n_extra_rows, generator=gen)[:batch]
    used = 0
    for i, n in enumerate(seq_lens):
        table[rows[i], :n] = perm[used : used + n]
        used += n
    if share is not None:
        a, b, n = share
        table[rows[b], :n] = table[rows[a], :n]
    return size, max_ctx, table, rows.to(torch.int64)


def make_pools(size, max_req, max_ctx, n_kv, dim, dtype, gen, n_layers=2):
    r2t = ref_pool.ReqToTokenPool(size=max_req, max_context_len=max_ctx, device="cpu")
    kv = ref_pool.MHATokenToKVPool(
        size=size, dtype=dtype, head_num=n_kv, head_dim=dim, layer_num=n_layers, device="cpu"
    )
    for layer in range(n_layers):  # torch.empty slabs: give them reproducible contents
        kv.k_buffer[layer].copy_(torch.randn(kv.k_buffer[layer].shape, generator=gen).to(dtype))
        kv.v_buffer[layer].copy_(torch.randn(kv.v_buffer[layer].shape, generator=gen).to(dtype))
    return r2t, kv


def new_rows(gen, n_tok, hq, hkv, dim, dtype):
    """q, k contiguous; v a column slice of a merged qkv row (qwen2.py:288, :189)."""
    q = torch.randn((n_tok, hq, dim), generator=gen).to(dtype)
    k = torch.randn((n_tok, hkv, dim), generator=gen).to(dtype)
    qkv = torch.randn((n_tok, (hq + 2 * hkv) * dim), generator=gen).to(dtype)
    v = qkv[:, (hq + hkv) * dim :].view(n_tok, hkv, dim)
    return q, k, qkv, v


def record(name, case):
    path = os.path.join(OUT, name + ".pt")
    torch.save(case, path)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def slab_delta(before, after, loc):
    keep = torch.ones(before.shape[0], dtype=torch.bool)
    keep[loc.long()] = False
    return after[loc.long()].clone(), bool(torch.equal(before[keep], after[keep]))


def gen_decode(name, cfg, dt, seq_lens, share=None, save_kv=True, seed=0):
    hq, hkv, dim = CONFIGS[cfg]
    dtype = DTYPES[dt]
    gen = torch.Generator().manual_seed(seed)
    size, max_ctx, table, rows = build_tables(gen, seq_lens, share)
    r2t, kv = make_pools(size, table.shape[0], max_ctx, hkv, dim, dtype, gen)
    r2t.req_to_token.copy_(table)
    layer_id = 1
    seq = torch.tensor(seq_lens, dtype=torch.int64)
    loc = table[rows, seq - 1].clone()  # slot of the token written this step
    q, k, qkv, v = new_rows(gen, len(seq_lens), hq, hkv, dim, dtype)
    k_before, v_before = kv.k_buffer[layer_id].clone(), kv.v_buffer[layer_id].clone()
    fb = ref_fb.SimplifiedForwardBatch.create_decode_batch(
        input_ids=torch.zeros(len(seq_lens), dtype=torch.int64),
        req_pool_indices=rows,
        seq_lens=seq,
        out_cache_loc=loc,
        req_to_token_pool=r2t,
        token_to_kv_pool=kv,
    )
    layer = types.SimpleNamespace(layer_id=layer_id, scaling=dim**-0.5)
    out = ref_backend.SimpleAttentionBackend.forward(q, k, v, layer, fb, save_kv_cache=save_kv)
    k_rows, k_rest_same = slab_delta(k_before, kv.k_buffer[layer_id], loc)
    v_rows, v_rest_same = slab_delta(v_before, kv.v_buffer[layer_id], loc)
    record(name, {
        "kind": "decode", "n_q_heads": hq, "n_kv_heads": hkv, "head_dim": dim, "save_kv_cache": save_kv,
        "q": q, "k": k, "qkv": qkv, "v_col0": (hq + hkv) * dim,
        "k_slab": k_before, "v_slab": v_before, "req_to_token": table,
        "req_pool_indices": rows, "seq_lens": seq, "out_cache_loc": loc, "positions": fb.positions,
        "out": out, "k_rows_after": k_rows, "v_rows_after": v_rows,
        "rest_unchanged": k_rest_same and v_rest_same,
    })  # fmt: skip


def gen_prefill(name, cfg, dt, seq_lens, prefix_lens, share=None, seed=0):
    hq, hkv, dim = CONFIGS[cfg]
    dtype = DTYPES[dt]
    gen = torch.Generator().manual_seed(seed)
    size, max_ctx, table, rows = build_tables(gen, seq_lens, share)
    r2t, kv = make_pools(size, table.shape[0], max_ctx, hkv, dim, dtype, gen)
    r2t.req_to_token.copy_(table)
    layer_id = 0
    seq = torch.tensor(seq_lens, dtype=torch.int64)
    prefix = torch.tensor(prefix_lens, dtype=torch.int64)
    loc = torch.cat(
        [table[rows[i], prefix_lens[i] : seq_lens[i]] for i in range(len(seq_lens))]
    ).clone()
    n_tok = int(loc.numel())
    q, k, qkv, v = new_rows(gen, n_tok, hq, hkv, dim, dtype)
    k_before, v_before = kv.k_buffer[layer_id].clone(), kv.v_buffer[layer_id].clone()
    fb = ref_fb.SimplifiedForwardBatch.create_prefill_batch(
        input_ids=torch.zeros(n_tok, dtype=torch.int64),
        req_pool_indices=rows,
        seq_lens=seq,
        out_cache_loc=loc,
        req_to_token_pool=r2t,
        token_to_kv_pool=kv,
        prefix_lens=prefix,
    )
    layer = types.SimpleNamespace(layer_id=layer_id, scaling=dim**-0.5)
    out = ref_backend.SimpleAttentionBackend.forward(q, k, v, layer, fb)
    k_rows, k_rest_same = slab_delta(k_before, kv.k_buffer[layer_id], loc)
    v_rows, v_rest_same = slab_delta(v_before, kv.v_buffer[layer_id], loc)
    record(name, {
        "kind": "prefill", "n_q_heads": hq, "n_kv_heads": hkv, "head_dim": dim, "save_kv_cache": True,
        "q": q, "k": k, "qkv": qkv, "v_col0": (hq + hkv) * dim,
        "k_slab": k_before, "v_slab": v_before, "req_to_token": table,
        "req_pool_indices": rows, "seq_lens": seq, "prefix_lens": prefix,
        "extended_lens": fb.extended_lens, "out_cache_loc": loc, "positions": fb.positions,
        "out": out, "k_rows_after": k_rows, "v_rows_after": v_rows,
        "rest_unchanged": k_rest_same and v_rest_same,
    })  # fmt: skip


def gen_pools():
    """Op-by-op traces of the two allocators, with every return value."""
    trace = {}
    kv = ref_pool.MHATokenToKVPool(size=12, dtype=torch.float16, head_num=1, head_dim=8, layer_num=1, device="cpu")
    ops = []
    ops.append(("available", -1, torch.tensor([kv.available_size()])))
    a = kv.alloc(5); ops.append(("alloc", 5, a.clone()))  # noqa: E702
    b = kv.alloc(4); ops.append(("alloc", 4, b.clone()))  # noqa: E702
    ops.append(("alloc_none", 4, torch.tensor([int(kv.alloc(4) is None)])))
    kv.free(a[1:4]); ops.append(("free", -1, a[1:4].clone()))  # noqa: E702
    ops.append(("available", -1, torch.tensor([kv.available_size()])))
    c = kv.alloc(6); ops.append(("alloc", 6, c.clone()))  # noqa: E702
    kv.free(torch.tensor([], dtype=torch.int32)); ops.append(("free", -1, torch.tensor([], dtype=torch.int32)))  # noqa: E702
    kv.free(b); ops.append(("free", -1, b.clone()))  # noqa: E702
    d = kv.alloc(3); ops.append(("alloc", 3, d.clone()))  # noqa: E702
    kv.clear(); ops.append(("clear", -1, kv.free_slots.clone()))  # noqa: E702
    trace["kv_ops"] = [(n, k, t) for n, k, t in ops]
    trace["kv_shapes"] = (tuple(kv.k_buffer[0].shape), str(kv.k_buffer[0].dtype), str(kv.free_slots.dtype))

    r2t = ref_pool.ReqToTokenPool(size=5, max_context_len=7, device="cpu")
    rops = []
    x = r2t.alloc(3); rops.append(("alloc", 3, list(x)))  # noqa: E702
    rops.append(("alloc_none", 3, r2t.alloc(3) is None))
    r2t.free(x[1]); rops.append(("free", -1, [x[1]]))  # noqa: E702
    y = r2t.alloc(2); rops.append(("alloc", 2, list(y)))  # noqa: E702
    r2t.free([x[0], y[0]]); rops.append(("free", -1, [x[0], y[0]]))  # noqa: E702
    rops.append(("available", -1, r2t.available_size()))
    z = r2t.alloc(3); rops.append(("alloc", 3, list(z)))  # noqa: E702
    r2t.write(torch.tensor([1, 3]), torch.arange(14, dtype=torch.int32).view(2, 7))
    r2t.write((2, slice(1, 4)), torch.tensor([9, 8, 7], dtype=torch.int32))
    rops.append(("table", -1, r2t.req_to_token.clone()))
    r2t.clear(); rops.append(("clear", -1, list(r2t.free_slots)))  # noqa: E702
    trace["req_ops"] = rops
    record("pools", trace)


def gen_sched_trace():
    """Drive the reference Scheduler itself.  scheduler.py imports two modules this
    image cannot provide (constraints.py -> lmformatenforcer, not installed; a
    tokenizer base class from transformers<5 used only in annotations); both are
    off the attention path, so empty placeholder modules satisfy the imports."""
    for name in ("lmformatenforcer", "lmformatenforcer.integrations",
                 "lmformatenforcer.integrations.transformers", "lmformatenforcer.tokenenforcer"):  # fmt: skip
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["lmformatenforcer"].JsonSchemaParser = object
    sys.modules["lmformatenforcer.integrations.transformers"].build_token_enforcer_tokenizer_data = None
    sys.modules["lmformatenforcer.tokenenforcer"].TokenEnforcer = object
    tu = types.ModuleType("transformers.tokenization_utils")
    tu.PreTrainedTokenizerBase = object
    import transformers  # noqa: F401  (real package; only the submodule symbol is missing)

    sys.modules["transformers.tokenization_utils"] = tu

    os.environ["BABYSGL_MAX_REQUESTS"] = "4"
    os.environ["BABYSGL_MAX_TOTAL_TOKENS"] = "256"
    import scheduler as ref_sched
    from model_loader import BabyQwenConfig
    from qwen2 import BabyQwen2ForCausalLM
    from sample import SamplingParams

    torch.manual_seed(0)
    cfg = BabyQwenConfig(
        vocab_size=97, hidden_size=256, intermediate_size=128, num_hidden_layers=2,
        num_attention_heads=4, num_key_value_heads=2, max_position_embeddings=128,
    )  # fmt: skip
    model = BabyQwen2ForCausalLM(cfg).eval()
    sched = ref_sched.Scheduler(model, tokenizer=None)

    batches = []
    fwd = ref_backend.SimpleAttentionBackend.forward

    def spy(q, k, v, layer, fb, save_kv_cache=True):
        out = fwd(q, k, v, layer, fb, save_kv_cache)
        if layer.layer_id == 0:
            k_slab, v_slab = fb.token_to_kv_pool.get_kv_buffer(0)
            batches.append({
                "is_prefill": bool(fb.is_prefill),
                "req_pool_indices": fb.req_pool_indices.clone(),
                "seq_lens": fb.seq_lens.clone(),
                "out_cache_loc": fb.out_cache_loc.clone(),
                "positions": fb.positions.clone(),
                "prefix_lens": None if fb.prefix_lens is None else fb.prefix_lens.clone(),
                "extended_lens": None if fb.extended_lens is None else fb.extended_lens.clone(),
                "req_to_token": fb.req_to_token_pool.req_to_token.clone(),
                "q": q.clone(), "k": k.clone(), "v": v.clone(),
                "k_slab_after": k_slab.clone(), "v_slab_after": v_slab.clone(),
                "out": out.clone(),
            })  # fmt: skip
        return out

    ref_backend.SimpleAttentionBackend.forward = staticmethod(spy)
    gen = torch.Generator().manual_seed(1)
    shared = torch.randint(0, 97, (12,), generator=gen)
    prompts = [
        torch.cat([shared, torch.randint(0, 97, (n,), generator=gen)]) for n in (3, 5, 2, 7, 4, 6)
    ]
    sp = SamplingParams(max_new_tokens=5, temperature=0.0, top_k=0, top_p=1.0, do_sample=False, eos_id=-1)
    outs = sched.run_batch(prompts, sp)
    ref_backend.SimpleAttentionBackend.forward = staticmethod(fwd)
    record("sched_trace", {
        "n_q_heads": 4, "n_kv_heads": 2, "head_dim": 64,
        "prompts": prompts, "output_ids": [torch.tensor(o) for o in outs], "batches": batches,
    })  # fmt: skip
    print("scheduler batches:", [("P" if b["is_prefill"] else "D", b["seq_lens"].tolist()) for b in batches][:6], "...")


def main():
    torch.set_num_threads(4)
    torch.manual_seed(0)
    seed = 100
    for cfg in CONFIGS:
        for dt in DTYPES:
            seed += 1
            gen_decode(f"decode_{cfg}_{dt}", cfg, dt, [1, 17, 70, 33], share=(2, 3, 20), seed=seed)
    gen_decode("decode_q15_bf16_b1", "q15", "bf16", [1], seed=201)
    gen_decode("decode_q15_bf16_long", "q15", "bf16", [513, 70], seed=202)
    gen_decode("decode_q15_f16_nosave", "q15", "f16", [9, 4, 31], save_kv=False, seed=203)
    for dt in DTYPES:
        seed += 1
        gen_prefill(f"prefill_q15_{dt}", "q15", dt, [10, 7, 5], [4, 0, 4], share=(0, 2, 4), seed=seed)
    gen_prefill("prefill_q05_bf16", "q05", "bf16", [10, 7, 5, 9], [4, 0, 4, 8], seed=301)
    gen_prefill("prefill_q7_f16", "q7", "f16", [6, 12], [0, 3], seed=302)
    gen_prefill("prefill_q15_bf16_long", "q15", "bf16", [194, 33], [64, 0], seed=303)
    gen_prefill("prefill_q15_f16_ext0", "q15", "f16", [8, 5, 6], [3, 5, 0], seed=304)  # middle request: nothing new
    gen_pools()
    gen_sched_trace()


if __name__ == "__main__":
    main()
