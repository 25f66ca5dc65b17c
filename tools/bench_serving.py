#!/usr/bin/env python3
"""Continuous batching + prefix reuse, end to end, on one GPU.

A random-weight Qwen2.5-1.5B-shaped model on the HIP path is driven by the serving loop:
`--requests` prompts of `--prompt` tokens, in groups of `--group` that share their first
`--shared` tokens (held once by the prefix tree), `--new` generated tokens each, greedy, eos
off; pools sized for `--rows` page-table rows of `--ctx` tokens.  Timed: the whole job --
prefix matching, admission, prefill (chunked with `--chunk`), every decode step,
finished-request caching -- with `run_batch` (ids looked at every step) and with
`run_batch_overlap` (ids looked at one step late).  Synthetic weights and token ids: there is
no checkpoint or tokenizer here.
"""
import argparse
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import baby_sglang_amd as pkg  # noqa: E402
from baby_sglang_amd.serving_loop import SamplingParams, ServingLoop  # noqa: E402

SHAPES = {
    "1.5b": dict(hidden_size=1536, intermediate_size=8960, num_hidden_layers=28, num_attention_heads=12,
                 num_key_value_heads=2, vocab_size=151936),
    "0.5b": dict(hidden_size=896, intermediate_size=4864, num_hidden_layers=24, num_attention_heads=14,
                 num_key_value_heads=2, vocab_size=151936),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", choices=list(SHAPES), default="1.5b")
    ap.add_argument("--requests", type=int, default=128)
    ap.add_argument("--rows", type=int, default=64)
    ap.add_argument("--ctx", type=int, default=4096)
    ap.add_argument("--prompt", type=int, default=1984)
    ap.add_argument("--shared", type=int, default=1024)
    ap.add_argument("--group", type=int, default=8)
    ap.add_argument("--new", type=int, default=64)
    ap.add_argument("--chunk", type=int, default=16384)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = types.SimpleNamespace(**SHAPES[a.model], rms_norm_eps=1e-6, rope_theta=1000000.0, rope_scaling=None,
                                max_position_embeddings=32768, tie_word_embeddings=True, hidden_act="silu")
    torch.manual_seed(0)
    with torch.device(dev):
        model = pkg.BabyQwen2ForCausalLM(cfg).to(torch.bfloat16).eval()
    for name, p in model.named_parameters():
        p.data.fill_(1.0) if name.endswith("norm.weight") else p.data.normal_(0.0, 0.02)
    gen = torch.Generator().manual_seed(3)
    # request i belongs to group i mod n_groups: the members of a group arrive spread over the
    # admission waves, so later waves find their group's head in the prefix tree
    n_groups = max(1, a.requests // a.group)
    heads = [torch.randint(0, cfg.vocab_size, (a.shared,), generator=gen) for _ in range(n_groups)]
    prompts = [torch.cat([heads[i % n_groups], torch.randint(0, cfg.vocab_size, (a.prompt - a.shared,), generator=gen)])
               for i in range(a.requests)]
    sp = SamplingParams(max_new_tokens=a.new)
    total_tokens = a.rows * a.ctx

    def job(overlap):
        # overlap: False = step by step; "one stream" = ids looked at a step late, one thread and stream;
        # "worker" = the same with the model on a worker thread and a second stream (the reference's form)
        loop = ServingLoop(model, max_total_tokens=total_tokens, max_requests=a.rows,
                           chunked_prefill_size=None if overlap else a.chunk, overlap_worker=overlap == "worker")
        calls = {"prefill": 0, "decode": 0, "prefill_tokens": 0, "decode_tokens": 0}

        def count(mode, fb):
            calls[mode] += 1
            calls[mode + "_tokens"] += len(fb.input_ids)

        loop.on_batch = count
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = (loop.run_batch_overlap if overlap else loop.run_batch)(prompts, sp)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        loop.close()
        return dt, outs, calls

    first = None
    for overlap in (False, "one stream", "worker"):
        job(overlap)  # warm-up: allocator, rocBLAS kernels
        dt, outs, calls = job(overlap)
        if overlap:  # (the step-by-step job chunks its prefill: other roundings in bf16, so other ids are possible there)
            first = first if first is not None else outs
            assert outs == first, "the two overlap forms disagree on the ids"
        n_out = sum(len(o) for o in outs)
        print(f"{('overlap, ' + overlap) if overlap else 'step-by-step'}: {len(prompts)} requests x {a.prompt} prompt tokens "
              f"({a.shared} shared by groups of {a.group}) + {a.new} new: {dt * 1e3:.0f} ms, "
              f"{n_out / dt:,.0f} output tokens/s; {calls['prefill']} prefill calls over {calls['prefill_tokens']:,} tokens "
              f"(of {len(prompts) * a.prompt:,} prompt tokens), {calls['decode']} decode calls over {calls['decode_tokens']:,} tokens",
              flush=True)


if __name__ == "__main__":
    main()
