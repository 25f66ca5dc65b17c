#!/usr/bin/env python3
"""Time of one next-token selection for a decode batch on the GPU (batch x vocab logits, bf16 or
fp32): greedy, plain sampling, top-k, top-p, top-k + top-p.  Usage: python tools/bench_sampling.py [batch] [vocab]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from baby_sglang_amd import sampling  # noqa: E402


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    vocab = int(sys.argv[2]) if len(sys.argv) > 2 else 151936
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(0)
    for dtype in (torch.float32, torch.bfloat16):
        logits = (torch.randn((batch, vocab), generator=gen, device=dev) * 3).to(dtype)
        cases = {
            "greedy": dict(do_sample=False),
            "sample T=0.6": dict(do_sample=True, temperature=0.6),
            "top_k=50": dict(do_sample=True, temperature=0.6, top_k=50),
            "top_p=0.9": dict(do_sample=True, temperature=0.6, top_p=0.9),
            "top_k=50 top_p=0.9": dict(do_sample=True, temperature=0.6, top_k=50, top_p=0.9),
        }
        for name, kw in cases.items():
            for _ in range(3):
                sampling.sample_next_ids(logits, generator=gen, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                sampling.sample_next_ids(logits, generator=gen, **kw)
            e1.record()
            torch.cuda.synchronize()
            print(f"{str(dtype):16s} batch {batch} vocab {vocab} {name:22s}: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us", flush=True)


if __name__ == "__main__":
    main()
