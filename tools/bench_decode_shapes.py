#!/usr/bin/env python3
"""Decode attention across model shapes, batch sizes and context lengths on one GPU:
per layer call (KV write + paged decode + split merge, replayed from a HIP graph of `--layers`
calls on distinct slabs), microseconds and the fraction of the 8 TB/s HBM peak that the
algorithmic bytes of the call (KV rows, page table, q, out, new rows) amount to."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import baby_sglang_amd as pkg  # noqa: E402
from baby_sglang_amd import _hip  # noqa: E402

# "1.5b-tp2": one rank's share of a two-rank tensor-parallel group (one kv head and its 6 q heads)
MODELS = {"0.5b": (14, 2, 64), "1.5b": (12, 2, 128), "7b": (28, 4, 128), "1.5b-tp2": (6, 1, 128)}


def run(model, batch, seq, layers, reps, dev):
    hq, hkv, dim = MODELS[model]
    dtype = torch.bfloat16
    gen = torch.Generator(device=dev).manual_seed(0)
    max_ctx = max(4096, seq)
    size = batch * seq + 8
    kv = pkg.MHATokenToKVPool(size, dtype, hkv, dim, layers, str(dev))
    kv.arena.normal_(generator=gen)
    r2t = pkg.ReqToTokenPool(batch, max_ctx, str(dev))
    r2t.req_to_token[:, :seq] = (torch.randperm(batch * seq, generator=gen, device=dev) + 1).to(torch.int32).view(batch, seq)
    req = torch.arange(batch, device=dev)
    lens = torch.full((batch,), seq, dtype=torch.int64, device=dev)
    loc = r2t.req_to_token[req, seq - 1].contiguous()
    q = torch.randn((layers, batch, hq, dim), generator=gen, device=dev).to(dtype)
    k = torch.randn((layers, batch, hkv, dim), generator=gen, device=dev).to(dtype)
    v = torch.randn((layers, batch, hkv, dim), generator=gen, device=dev).to(dtype)
    n_splits = _hip.decode_num_splits(batch, hkv, max_ctx)
    ws = _hip.new_decode_workspace(max(_hip.decode_workspace_bytes(batch, hq, dim, n_splits), 1 << 20), dev)
    out = torch.empty((batch, hq * dim), dtype=dtype, device=dev)

    def step():
        for layer in range(layers):
            ks, vs = kv.get_kv_buffer(layer)
            _hip.paged_decode(q[layer], ks, vs, r2t.req_to_token, req, lens, n_splits=n_splits, out=out, workspace=ws,
                              k_new=k[layer], v_new=v[layer], out_cache_loc=loc, fuse_kv_write=True)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps / layers * 1e3
    nbytes = 2 * batch * seq * hkv * dim * 2 + batch * seq * 4 + 2 * batch * hq * dim * 2 + 2 * batch * hkv * dim * 2
    print(f"{model:8s} batch {batch:4d} seq {seq:6d} splits<= {n_splits:2d}: {us:8.2f} us/layer  "
          f"{nbytes / us / 1e6:6.2f} TB/s = {nbytes / us / 1e6 / 8:.3f} of peak  ({batch / (us * 28) * 1e6:,.0f} tokens/s at 28 layers)", flush=True)
    del kv, graph
    torch.cuda.empty_cache()
    return {"us_per_layer": us, "algorithmic_bytes_per_launch": nbytes, "achieved_gbs": nbytes / us / 1e3, "frac_of_hbm_peak": nbytes / us / 1e6 / 8}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--cases", default="1.5b:1:2048,1.5b:8:2048,1.5b:32:2048,1.5b:64:512,1.5b:64:2048,1.5b:64:8192,"
                                       "1.5b:128:2048,1.5b:256:2048,1.5b:512:1024,7b:16:2048,7b:64:2048,7b:128:2048,0.5b:64:2048")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    for case in a.cases.split(","):
        model, batch, seq = case.split(":")
        run(model, int(batch), int(seq), a.layers, a.reps, dev)


if __name__ == "__main__":
    main()
