#!/usr/bin/env python3
"""Decode latency of one layer call at small batch (BASELINE configs[1]: Qwen2.5-1.5B
bf16, batch 1): microseconds per paged_decode launch for several context lengths."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import baby_sglang_amd as pkg  # noqa: E402

dev = torch.device("cuda:0")
hq, hkv, dim = (int(x) for x in os.getenv("SHAPE", "12,2,128").split(","))  # q heads, kv heads, head dim
dtype = torch.bfloat16
gen = torch.Generator(device=dev).manual_seed(0)
for batch in [int(b) for b in os.getenv("BATCHES", "1,2,4,8").split(",")]:
    for seq in [int(x) for x in os.getenv("SEQS", "1,128,2048,4095").split(",")]:
        size = batch * 4096
        kv = pkg.MHATokenToKVPool(size, dtype, hkv, dim, 1, "cuda:0")
        kv.arena.normal_(generator=gen)
        table = (torch.randperm(size, generator=gen, device=dev).to(torch.int32) + 1).view(batch, 4096).contiguous()
        rows = torch.arange(batch, device=dev)
        sl = torch.full((batch,), seq, dtype=torch.int64, device=dev)
        q = torch.randn((batch, hq, dim), generator=gen, device=dev).to(dtype)
        k_new = torch.randn((batch, hkv, dim), generator=gen, device=dev).to(dtype)
        loc = table[rows, sl - 1].contiguous()
        k_slab, v_slab = kv.get_kv_buffer(0)
        n_splits = pkg._hip.decode_num_splits(batch, hkv, 4096)
        ws = pkg._hip.new_decode_workspace(pkg._hip.decode_workspace_bytes(batch, hq, dim, n_splits), dev)
        out = torch.empty((batch, hq * dim), dtype=dtype, device=dev)

        def call():
            pkg._hip.paged_decode(q, k_slab, v_slab, table, rows, sl, n_splits=n_splits, out=out, workspace=ws, variant=int(os.getenv("VARIANT", "0")),
                                  k_new=k_new, v_new=k_new, out_cache_loc=loc, fuse_kv_write=True)

        for _ in range(3):
            call()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20):
                call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g.replay()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 200
        print(f"batch={batch} seq={seq:5d} splits={n_splits:2d}: {us:7.2f} us per layer call (incl. launch boundary)", flush=True)
