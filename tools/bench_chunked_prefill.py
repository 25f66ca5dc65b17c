#!/usr/bin/env python3
"""Chunked prefill of a whole batch through the attention path on one GPU.

Qwen2.5-7B shapes by default (28 layers, 28 q heads, 4 kv heads, head_dim 128, bf16):
`--batch` prompts of `--prompt` tokens are admitted at once (slots and page-table rows
written up front, random-permutation slots) and run as extend calls of at most `--chunk`
tokens, cut exactly as the serving loop cuts them (`serving_loop.chunk_members`); per
call and layer: KV write of the call's tokens + paged extend attention over everything
the request has cached so far.  Reports attention-path time for the whole batch, extend
tokens/s, TFLOP/s (flops = 4 Hq D (ext prefix + ext (ext + 1) / 2) per request, layer and
call) and what the KV pool of this run and of a full 288 GB card holds.
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--prompt", type=int, default=2048)
    ap.add_argument("--chunk", type=int, default=8192)
    ap.add_argument("--layers", type=int, default=28)
    ap.add_argument("--hq", type=int, default=28)
    ap.add_argument("--hkv", type=int, default=4)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--weights-gb", type=float, default=15.2, help="bf16 weights kept beside the KV pool (7B: 15.2)")
    a = ap.parse_args()
    import baby_sglang_amd as pkg
    from baby_sglang_amd.serving_loop import chunk_members

    dev = torch.device("cuda:0")
    dim, dtype = 128, torch.bfloat16
    size = a.batch * a.prompt
    gen = torch.Generator(device=dev).manual_seed(0)
    kv = pkg.MHATokenToKVPool(size, dtype, a.hkv, dim, a.layers, "cuda:0")
    r2t = pkg.ReqToTokenPool(a.batch, a.prompt, "cuda:0")
    slots = torch.randperm(size, generator=gen, device=dev).to(torch.int32) + 1
    r2t.req_to_token.copy_(slots.view(a.batch, a.prompt))
    # one layer's worth of new q / k / v rows, reused by every layer (values do not matter for timing)
    q = torch.randn((size, a.hq, dim), generator=gen, device=dev).to(dtype)
    kn = torch.randn((size, a.hkv, dim), generator=gen, device=dev).to(dtype)
    vn = torch.randn((size, a.hkv, dim), generator=gen, device=dev).to(dtype)
    out = torch.empty((size, a.hq * dim), dtype=dtype, device=dev)

    calls, flops = [], 0.0
    done = [0] * a.batch
    first = [i * a.prompt for i in range(a.batch)]
    for members in chunk_members([a.prompt] * a.batch, a.chunk):
        idx = torch.tensor([m for m, _ in members], device=dev)
        prefix = torch.tensor([done[m] for m, _ in members], dtype=torch.int64, device=dev)
        takes = torch.tensor([t for _, t in members], dtype=torch.int64, device=dev)
        t0, t1 = first[members[0][0]] + done[members[0][0]], first[members[-1][0]] + done[members[-1][0]] + members[-1][1]
        # (requests are consecutive and whole prompts are new, so a call's tokens are one slice)
        loc = torch.cat([r2t.req_to_token[m, done[m] : done[m] + t] for m, t in members])
        ws = torch.empty(pkg._hip.prefill_workspace_bytes(len(members), t1 - t0, a.hq, dim), dtype=torch.uint8, device=dev)
        calls.append((idx, prefix, takes, prefix + takes, t0, t1, loc, ws))
        for m, t in members:
            flops += 4.0 * a.hq * dim * (t * done[m] + t * (t + 1) / 2)
            done[m] += t
    flops *= a.layers

    def run():
        for idx, prefix, takes, seq, t0, t1, loc, ws in calls:
            for layer in range(a.layers):
                k_slab, v_slab = kv.get_kv_buffer(layer)
                pkg._hip.kv_scatter(k_slab, v_slab, loc, kn[t0:t1], vn[t0:t1])
                pkg._hip.paged_prefill(q[t0:t1], k_slab, v_slab, r2t.req_to_token, idx, seq, prefix, takes,
                                       out=out[t0:t1], workspace=ws)

    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    per_tok = 2 * a.layers * a.hkv * dim * 2
    print(f"chunked prefill: {a.batch} prompts x {a.prompt} tokens in {len(calls)} calls of <= {a.chunk} tokens, "
          f"{a.layers} layers, Hq={a.hq} Hkv={a.hkv} bf16: attention path {ms:.1f} ms for the batch = "
          f"{size / ms * 1e3:,.0f} prompt tokens/s, {flops / ms / 1e9:.0f} TFLOP/s", flush=True)
    free_b, total_b = torch.cuda.mem_get_info()
    print(f"KV: {per_tok} B/token over {a.layers} layers; this pool {size:,} tokens = {size * per_tok / 2**30:.1f} GiB; "
          f"card {total_b / 2**30:.0f} GiB -> {(total_b - a.weights_gb * 2**30) * 0.95 / per_tok / 1e6:.2f} M tokens "
          f"beside {a.weights_gb} GiB of weights (5 % left for activations)", flush=True)


if __name__ == "__main__":
    main()
