#!/usr/bin/env python3
"""bsgl_paged_decode_static in a FRESH process: a first-ever call on a device made while the stream is
being captured is refused (the look-up of the library's scratch array loads the code object's symbol
table, which a capture must not run into); after one eager call the same capture goes through, and
its replays on new inputs equal the workspace entry point bit for bit.  Prints one verdict line."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from baby_sglang_amd import _hip
    from test_gpu_parity import random_decode_problem

    gpu = torch.device("cuda:0")
    dtype, hq, hkv, dim = torch.bfloat16, 12, 2, 128
    seq_lens = [700, 33, 1200, 64, 2000, 1]
    n = len(seq_lens)
    cpu, r2t, kv = random_decode_problem(gpu, dtype, hq, hkv, dim, seq_lens, seed=4)
    rows, seq = cpu["req_pool_indices"].to(gpu), cpu["seq_lens"].to(gpu)
    k_slab, v_slab = kv.get_kv_buffer(0)
    q = cpu["q"].to(gpu)
    out = torch.empty((n, hq * dim), dtype=dtype, device=gpu)

    def call(static):
        return _hip.paged_decode(q, k_slab, v_slab, r2t.req_to_token, rows, seq, n_splits=8, out=out, static_scratch=static)

    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    refused = False
    graph = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(graph, stream=side):
            call(True)
    except NotImplementedError as exc:
        refused = "first call" in str(exc)
    torch.cuda.synchronize()
    assert refused, "the first-ever call under capture was not refused"
    call(True)  # the eager call a capture needs anyway
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        call(True)
    gen = torch.Generator().manual_seed(9)
    for _ in range(3):
        q.copy_(torch.randn(q.shape, generator=gen).to(dtype))
        graph.replay()
        torch.cuda.synchronize()
        got = out.clone()
        call(False)
        torch.cuda.synchronize()
        assert torch.equal(got, out), "replay differs from the workspace entry point"
    print("static decode under capture: first call refused, later capture replays right", flush=True)


if __name__ == "__main__":
    main()
