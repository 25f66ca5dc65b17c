#!/usr/bin/env python3
"""Whole-model decode step on the HIP path: a random-weight Qwen2.5-1.5B-shaped
``BabyQwen2ForCausalLM`` (this package's mirror of the reference model), batch 64, every
request at 2048 cached tokens, bf16, replayed from a HIP graph.  Prints ms per step and
tokens/s, and the same with the attention call removed (GEMMs + norms + RoPE only) so the
attention share is visible.  Synthetic weights and KV: there is no checkpoint here."""
import argparse
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import baby_sglang_amd as pkg  # noqa: E402

SHAPES = {
    "0.5b": dict(hidden_size=896, intermediate_size=4864, num_hidden_layers=24, num_attention_heads=14,
                 num_key_value_heads=2, vocab_size=151936),
    "1.5b": dict(hidden_size=1536, intermediate_size=8960, num_hidden_layers=28, num_attention_heads=12,
                 num_key_value_heads=2, vocab_size=151936),
    "7b": dict(hidden_size=3584, intermediate_size=18944, num_hidden_layers=28, num_attention_heads=28,
               num_key_value_heads=4, vocab_size=152064),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", choices=list(SHAPES), default="1.5b")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--seq", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    dtype = torch.bfloat16
    cfg = types.SimpleNamespace(**SHAPES[args.model], rms_norm_eps=1e-6, rope_theta=1000000.0, rope_scaling=None,
                                max_position_embeddings=32768, tie_word_embeddings=args.model in ("0.5b", "1.5b"), hidden_act="silu")
    torch.manual_seed(0)
    with torch.device(dev):
        model = pkg.BabyQwen2ForCausalLM(cfg).to(dtype).eval()
    for p in model.parameters():
        p.data.normal_(0.0, 0.02)
    for name, p in model.named_parameters():
        if name.endswith("norm.weight"):
            p.data.fill_(1.0)
    B, S = args.batch, args.seq
    head_dim = cfg.hidden_size // cfg.num_attention_heads
    max_ctx = 4096
    gen = torch.Generator(device=dev).manual_seed(1)
    r2t = pkg.ReqToTokenPool(B, max_ctx, str(dev))
    kv = pkg.MHATokenToKVPool(B * max_ctx, dtype, cfg.num_key_value_heads, head_dim, cfg.num_hidden_layers, str(dev))
    kv.arena.normal_(generator=gen)
    r2t.req_to_token[:, :S] = (torch.randperm(B * S, generator=gen, device=dev) + 1).to(torch.int32).view(B, S)
    req = torch.arange(B, device=dev)
    seq = torch.full((B,), S, dtype=torch.int64, device=dev)
    loc = r2t.req_to_token[req, S - 1].contiguous()
    ids = torch.randint(0, cfg.vocab_size, (B,), generator=gen, device=dev)
    fb = pkg.SimplifiedForwardBatch.create_decode_batch(ids, req, seq, loc, r2t, kv)

    def timed(fn):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            fn()
        graph.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps):
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.steps

    full = timed(lambda: model(ids, fb.positions, fb))
    print(f"{args.model} batch={B} seq={S}: whole decode step {full:.3f} ms = {B / full * 1e3:,.0f} tokens/s", flush=True)

    # the same step with the attention launch skipped (its output replaced by q): what the rest costs
    real = pkg.SimpleAttentionBackend.forward
    pkg.SimpleAttentionBackend.forward = staticmethod(lambda q, k, v, layer, fb, save_kv_cache=True: q.view(q.shape[0], -1))
    try:
        rest = timed(lambda: model(ids, fb.positions, fb))
    finally:
        pkg.SimpleAttentionBackend.forward = real
    print(f"  without the attention launches: {rest:.3f} ms -> attention share {full - rest:.3f} ms "
          f"({(full - rest) / cfg.num_hidden_layers * 1e3:.1f} us/layer)", flush=True)


if __name__ == "__main__":
    main()
