#!/usr/bin/env python3
"""Decode latency of one layer call at small and medium batch (BASELINE configs[1]: Qwen2.5-1.5B
bf16, batch 1): microseconds per paged_decode launch, replayed from a HIP graph, for several
context lengths.  Environment: SHAPE="q heads,kv heads,head dim" (default 12,2,128), BATCHES and
SEQS (comma lists), VARIANT (0 automatic, 1 generic kernel, 2 matrix-core kernel), HINT=1 (the calls say how
long their longest request is); the library's
own knobs (BSGL_*) apply as always."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import baby_sglang_amd as pkg  # noqa: E402


def measure(hq, hkv, dim, batch, seq, variant=0, dev=None, max_ctx=4096, dtype=torch.bfloat16, hint=0):
    """(microseconds per layer call incl. the launch boundary, n_splits): KV write + paged decode of
    `batch` requests of `seq` tokens on a random-permutation page table of `max_ctx` columns."""
    dev = dev or torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(0)
    size = batch * max_ctx
    kv = pkg.MHATokenToKVPool(size, dtype, hkv, dim, 1, str(dev))
    kv.arena.normal_(generator=gen)
    table = (torch.randperm(size, generator=gen, device=dev).to(torch.int32) + 1).view(batch, max_ctx).contiguous()
    rows = torch.arange(batch, device=dev)
    sl = torch.full((batch,), seq, dtype=torch.int64, device=dev)
    q = torch.randn((batch, hq, dim), generator=gen, device=dev).to(dtype)
    k_new = torch.randn((batch, hkv, dim), generator=gen, device=dev).to(dtype)
    loc = table[rows, sl - 1].contiguous()
    k_slab, v_slab = kv.get_kv_buffer(0)
    n_splits = pkg._hip.decode_num_splits(batch, hkv, max_ctx)
    ws = pkg._hip.new_decode_workspace(pkg._hip.decode_workspace_bytes(batch, hq, dim, n_splits), dev)
    out = torch.empty((batch, hq * dim), dtype=dtype, device=dev)

    def call():
        pkg._hip.paged_decode(q, k_slab, v_slab, table, rows, sl, n_splits=n_splits, out=out, workspace=ws, variant=variant, max_seq_len=hint,
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
    return e0.elapsed_time(e1) * 1e3 / 200, n_splits


def main():
    hq, hkv, dim = (int(x) for x in os.getenv("SHAPE", "12,2,128").split(","))
    variant = int(os.getenv("VARIANT", "0"))
    for batch in [int(b) for b in os.getenv("BATCHES", "1,2,4,8").split(",")]:
        for seq in [int(x) for x in os.getenv("SEQS", "1,128,2048,4095").split(",")]:
            us, n_splits = measure(hq, hkv, dim, batch, seq, variant, hint=seq if os.getenv("HINT", "0") != "0" else 0)
            print(f"batch={batch} seq={seq:5d} splits={n_splits:2d}: {us:7.2f} us per layer call (incl. launch boundary)", flush=True)


if __name__ == "__main__":
    main()
