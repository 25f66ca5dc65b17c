#!/usr/bin/env python3
"""Host-side cost of one decode layer call through SimpleAttentionBackend.forward: cProfile of
bench.py's backend_step (28 calls per step) with the GPU kept busy, top functions by own time."""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    sys.argv = [sys.argv[0], "--cpu-seconds", "0", "--no-extras"]
    args = bench.parse_args()
    dev = torch.device("cuda", 0)
    import baby_sglang_amd as pkg

    layout = pkg.TPLayout(0, 1, 12, 2)
    wl = bench.Workload(args, dev, layout)
    for _ in range(3):
        wl.backend_step()
    torch.cuda.synchronize()
    # host time alone: issue without waiting for the GPU in between (the queue absorbs 50 steps)
    t0 = time.perf_counter()
    for _ in range(20):
        wl.backend_step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"backend.forward: {(t1 - t0) / 20 / args.layers * 1e6:.1f} us of host time per layer call; with the GPU: {(t2 - t0) / 20 / args.layers * 1e6:.1f} us")
    t0 = time.perf_counter()
    for _ in range(20):
        wl.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"_hip.paged_decode directly: {(t1 - t0) / 20 / args.layers * 1e6:.1f} us of host time per layer call")
    # the wrapper alone: the library call replaced by a Python no-op
    real, real_call = pkg._hip.lib.bsgl_paged_decode, pkg._hip.lib.bsgl_paged_decode_call
    pkg._hip.lib.bsgl_paged_decode = lambda *a: 0
    pkg._hip.lib.bsgl_paged_decode_call = lambda *a: 0
    t0 = time.perf_counter()
    for _ in range(20):
        wl.step()
    t1 = time.perf_counter()
    for _ in range(20):
        wl.backend_step()
    t2 = time.perf_counter()
    pkg._hip.lib.bsgl_paged_decode, pkg._hip.lib.bsgl_paged_decode_call = real, real_call
    print(f"without the library call: paged_decode wrapper {(t1 - t0) / 20 / args.layers * 1e6:.1f} us, backend.forward {(t2 - t1) / 20 / args.layers * 1e6:.1f} us per layer call")
    # backend.forward alone: the caller's own slicing of the projection output (v_rows) taken out of the loop
    fwd = pkg.SimpleAttentionBackend.forward
    calls = [(wl.q[l], wl.k[l], wl.v_rows(l), wl.layers[l]) for l in range(args.layers)]
    for _ in range(3):
        for a in calls:
            fwd(*a, wl.fb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        for a in calls:
            fwd(*a, wl.fb)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"backend.forward with ready-made arguments: {(t1 - t0) / 20 / args.layers * 1e6:.1f} us of host time per layer call")

    def each(label, fn, n=2000):
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        print(f"  {label}: {(t1 - t0) / n * 1e6:.2f} us")

    q, k, v, layer = calls[0]
    B = SimpleBackend = pkg.SimpleAttentionBackend
    pool, table = wl.fb.token_to_kv_pool, wl.fb.req_to_token_pool.req_to_token
    rec = B._decode_call(q, pool, table, 0)
    each("the library call alone (descriptor as it stands)", lambda: pkg._hip.lib.bsgl_paged_decode_call(rec[3]), 560)
    each("torch.empty of the output", lambda: torch.empty((q.shape[0], 1536), dtype=q.dtype, device=q.device))
    each("descriptor look-up (_decode_call)", lambda: B._decode_call(q, pool, table, 0))
    each("q checks", lambda: (q.is_cuda, q.is_contiguous(), q.dim(), q.dtype in pkg._hip._DTYPE_CODE))
    each("current stream", lambda: pkg._hip.current_stream_handle(q.device.index))
    each("k/v strides and shape", lambda: (k.stride(), v.stride(), k.shape, v.shape, k.dtype, v.dtype))
    each("five data_ptr()", lambda: (q.data_ptr(), k.data_ptr(), v.data_ptr(), q.data_ptr(), q.data_ptr()))
    each("an empty lambda", lambda: None)
    prof = cProfile.Profile()
    prof.enable()
    for _ in range(20):
        wl.backend_step()
    prof.disable()
    torch.cuda.synchronize()
    pstats.Stats(prof).sort_stats("tottime").print_stats(18)


if __name__ == "__main__":
    main()
