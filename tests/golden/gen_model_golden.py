"""Generate the goldens of the steps either side of attention by RUNNING THE REFERENCE.

Run in the build container only (``python tests/golden/gen_model_golden.py``): imports the
reference's ``qwen2`` module from ``/root/reference`` and records inputs and outputs of its
RMSNorm and RotaryEmbedding modules and of a whole tiny random-weight
``BabyQwen2ForCausalLM`` over a prefill, two decode steps and an extend over a cached
prefix (fp32 and bf16 weights).  Only tensors are committed; tests load them with
``torch.load(..., weights_only=True)``.

Files
  model_ops.pt    RMSNorm (qwen2.py:125-141) and RotaryEmbedding (qwen2.py:224-268) cases
  tiny_model.pt   config, fp32 state_dict, four ForwardBatches and the logits of each
"""

from __future__ import annotations

import os
import sys

import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)

import forward_batch as ref_fb  # noqa: E402
import memory_pool as ref_pool  # noqa: E402
import qwen2 as ref_model  # noqa: E402
from model_loader import BabyQwenConfig  # noqa: E402

DTYPES = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}


def record(name, case):
    path = os.path.join(OUT, name + ".pt")
    torch.save(case, path)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def gen_ops():
    gen = torch.Generator().manual_seed(11)
    cases = []
    for dt_name, dt in DTYPES.items():
        for hidden in (256, 1536):
            norm = ref_model.RMSNorm(hidden, eps=1e-6)
            with torch.no_grad():
                norm.weight.copy_(1.0 + 0.1 * torch.randn(hidden, generator=gen))
            norm = norm.to(dt)
            x = (3.0 * torch.randn((5, hidden), generator=gen)).to(dt)
            res = torch.randn((5, hidden), generator=gen).to(dt)
            with torch.no_grad():
                out_plain, _ = norm(x)
                out_res, res_out = norm(x, res)
            cases.append({"op": "rmsnorm", "dtype": dt_name, "eps": 1e-6, "weight": norm.weight.detach().clone(),
                          "x": x, "residual": res, "out": out_plain, "out_with_residual": out_res, "residual_out": res_out})
        for hq, hkv, dim in ((12, 2, 128), (14, 2, 64)):
            rope = ref_model.RotaryEmbedding(dim, 32768, 1000000.0, None)
            pos = torch.tensor([0, 1, 17, 4095, 30000, 2], dtype=torch.int64)
            q = torch.randn((len(pos), hq, dim), generator=gen).to(dt)
            k = torch.randn((len(pos), hkv, dim), generator=gen).to(dt)
            with torch.no_grad():
                q_rot, k_rot = rope(pos, q, k)
            cases.append({"op": "rope", "dtype": dt_name, "inv_freq": rope.inv_freq.clone(), "positions": pos,
                          "q": q, "k": k, "q_rot": q_rot, "k_rot": k_rot})
    record("model_ops", {"cases": cases})


def gen_tiny_model():
    torch.manual_seed(3)
    cfg = BabyQwenConfig(
        vocab_size=64, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
        num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=256,
    )  # fmt: skip
    model = ref_model.BabyQwen2ForCausalLM(cfg).eval()
    with torch.no_grad():  # norm weights away from 1 so that they matter
        for name, p in model.named_parameters():
            if name.endswith("layernorm.weight") or name.endswith("norm.weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape))
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    gen = torch.Generator().manual_seed(4)
    head_dim = cfg.hidden_size // cfg.num_attention_heads
    size, max_req, max_ctx = 96, 5, 40

    # one schedule of batches, replayed for each dtype on fresh pools
    prompt_lens = [5, 9, 3]
    prompts = [torch.randint(0, cfg.vocab_size, (n,), generator=gen) for n in prompt_lens]
    perm = (torch.randperm(size, generator=gen) + 1).to(torch.int32)  # slots in a scrambled order
    out = {"config": {
        "vocab_size": cfg.vocab_size, "hidden_size": cfg.hidden_size, "intermediate_size": cfg.intermediate_size,
        "num_hidden_layers": cfg.num_hidden_layers, "num_attention_heads": cfg.num_attention_heads,
        "num_key_value_heads": cfg.num_key_value_heads, "head_dim": head_dim, "rms_norm_eps": cfg.rms_norm_eps,
        "rope_theta": cfg.rope_theta, "max_position_embeddings": cfg.max_position_embeddings,
    }, "state_dict": state, "pool": {"size": size, "max_req": max_req, "max_ctx": max_ctx}, "runs": {}}  # fmt: skip

    for dt_name in ("f32", "bf16"):
        dt = DTYPES[dt_name]
        m = ref_model.BabyQwen2ForCausalLM(cfg).eval()
        m.load_state_dict(state)
        m = m.to(dt)
        r2t = ref_pool.ReqToTokenPool(size=max_req, max_context_len=max_ctx, device="cpu")
        kv = ref_pool.MHATokenToKVPool(size=size, dtype=dt, head_num=cfg.num_key_value_heads, head_dim=head_dim,
                                       layer_num=cfg.num_hidden_layers, device="cpu")
        for layer in range(cfg.num_hidden_layers):
            kv.k_buffer[layer].zero_()
            kv.v_buffer[layer].zero_()
        used = 0
        rows = torch.tensor([3, 0, 2], dtype=torch.int64)
        seq = torch.tensor(prompt_lens, dtype=torch.int64)
        steps = []

        def take(n):
            nonlocal used
            s = perm[used : used + n].clone()
            used += n
            return s

        def snapshot(kind, fb, ids, logits):
            steps.append({
                "kind": kind, "input_ids": ids.clone(), "positions": fb.positions.clone(),
                "req_pool_indices": fb.req_pool_indices.clone(), "seq_lens": fb.seq_lens.clone(),
                "out_cache_loc": fb.out_cache_loc.clone(),
                "prefix_lens": None if fb.prefix_lens is None else fb.prefix_lens.clone(),
                "req_to_token": r2t.req_to_token.clone(), "logits": logits.clone(),
            })  # fmt: skip

        # 1. prefill of three prompts, no prefix
        loc = take(int(seq.sum()))
        o = 0
        for i, n in enumerate(prompt_lens):
            r2t.req_to_token[rows[i], :n] = loc[o : o + n]
            o += n
        ids = torch.cat(prompts)
        fb = ref_fb.SimplifiedForwardBatch.create_prefill_batch(ids, rows, seq, loc, r2t, kv, torch.zeros(3, dtype=torch.int64))
        logits = m(ids, fb.positions, fb)
        snapshot("prefill", fb, ids, logits)
        last = torch.cumsum(seq, 0) - 1
        nxt = logits[last].float().argmax(-1)
        # 2.-3. two greedy decode steps
        for _ in range(2):
            seq = seq + 1
            loc = take(3)
            for i in range(3):
                r2t.req_to_token[rows[i], seq[i] - 1] = loc[i]
            fb = ref_fb.SimplifiedForwardBatch.create_decode_batch(nxt, rows, seq, loc, r2t, kv)
            logits = m(nxt, fb.positions, fb)
            snapshot("decode", fb, nxt, logits)
            nxt = logits.float().argmax(-1)
        # 4. a new request that reuses the first 4 cached tokens of request 0 (radix-cache hit)
        new_row = torch.tensor([4], dtype=torch.int64)
        tail = torch.randint(0, cfg.vocab_size, (6,), generator=gen)
        loc = take(6)
        r2t.req_to_token[4, :4] = r2t.req_to_token[rows[0], :4]
        r2t.req_to_token[4, 4:10] = loc
        fb = ref_fb.SimplifiedForwardBatch.create_prefill_batch(
            tail, new_row, torch.tensor([10], dtype=torch.int64), loc, r2t, kv, torch.tensor([4], dtype=torch.int64))
        logits = m(tail, fb.positions, fb)
        snapshot("extend", fb, tail, logits)
        out["runs"][dt_name] = steps
    record("tiny_model", out)


if __name__ == "__main__":
    torch.set_num_threads(4)
    gen_ops()
    gen_tiny_model()
