#!/usr/bin/env python3
"""Extend (prefill) attention timing on one GPU: Qwen2.5-7B shapes by default
(28 q heads, 4 kv heads, head_dim 128, bf16), batch of requests each extending
`--ext` tokens over a cached prefix of `--prefix` tokens (a chunk of a chunked
prefill is exactly that).  Reports ms per layer call and TFLOP/s with
flops = 4 * Hq * D * (ext * prefix + ext * (ext + 1) / 2) per request."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--ext", type=int, default=512)
    ap.add_argument("--prefix", type=int, default=1536)
    ap.add_argument("--hq", type=int, default=28)
    ap.add_argument("--hkv", type=int, default=4)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--table", default="perm")
    ap.add_argument("--dtype", choices=["bf16", "f16"], default="bf16")
    ap.add_argument("--profile", action="store_true", help="per-phase cycle counts of the matrix-core kernel")
    ap.add_argument("--warmup", type=int, default=1, help="untimed launches first (30 or more: the sustained rate; see --curve)")
    ap.add_argument("--curve", type=int, default=0, help="N: time N successive groups of 10 launches (how the rate settles under sustained load)")
    ap.add_argument("--knob", action="append", default=[], help="NAME=VALUE: set one of the library's tuning knobs, e.g. BSGL_PREFILL_BLK=64")
    a = ap.parse_args()
    import baby_sglang_amd as pkg

    for kv_ in a.knob:
        name, value = kv_.split("=")
        pkg._hip.set_knob(name, int(value))
    dev = torch.device("cuda:0")
    dim, dtype = a.dim, torch.bfloat16 if a.dtype == "bf16" else torch.float16
    seq = a.prefix + a.ext
    size = a.batch * seq
    gen = torch.Generator(device=dev).manual_seed(0)
    kv = pkg.MHATokenToKVPool(size, dtype, a.hkv, dim, 1, "cuda:0")
    kv.arena.normal_(generator=gen)
    r2t = pkg.ReqToTokenPool(a.batch, seq, "cuda:0")
    if a.table == "perm":
        slots = torch.randperm(size, generator=gen, device=dev).to(torch.int32) + 1
    else:
        slots = torch.arange(size, device=dev, dtype=torch.int32) + 1
    r2t.req_to_token.copy_(slots.view(a.batch, seq))
    q = torch.randn((a.batch * a.ext, a.hq, dim), generator=gen, device=dev).to(dtype)
    rows = torch.arange(a.batch, device=dev)
    seq_t = torch.full((a.batch,), seq, dtype=torch.int64, device=dev)
    pre_t = torch.full((a.batch,), a.prefix, dtype=torch.int64, device=dev)
    ext_t = seq_t - pre_t
    k_slab, v_slab = kv.get_kv_buffer(0)
    ws = torch.empty(pkg._hip.prefill_workspace_bytes(a.batch, q.shape[0], a.hq, dim), dtype=torch.uint8, device=dev)
    out = torch.empty((q.shape[0], a.hq * dim), dtype=dtype, device=dev)

    def call():
        return pkg._hip.paged_prefill(q, k_slab, v_slab, r2t.req_to_token, rows, seq_t, pre_t, ext_t,
                                      variant=a.variant, out=out, workspace=ws)

    for _ in range(max(1, a.warmup)):
        call()
    torch.cuda.synchronize()
    if a.curve:
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(a.curve + 1)]
        evs[0].record()
        for g in range(a.curve):
            for _ in range(10):
                call()
            evs[g + 1].record()
        torch.cuda.synchronize()
        fl = 4.0 * a.hq * dim * (a.ext * a.prefix + a.ext * (a.ext + 1) / 2) * a.batch
        print("TFLOP/s of successive groups of 10 launches: " + " ".join(f"{fl / (evs[g].elapsed_time(evs[g + 1]) / 10) / 1e9:.0f}" for g in range(a.curve)), flush=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    flops = 4.0 * a.hq * dim * (a.ext * a.prefix + a.ext * (a.ext + 1) / 2) * a.batch
    kv_bytes = a.batch * seq * a.hkv * dim * 2 * 2
    print(f"{a.dtype} batch={a.batch} ext={a.ext} prefix={a.prefix} Hq={a.hq} Hkv={a.hkv} variant={a.variant} table={a.table}: "
          f"{ms:.3f} ms/layer  {flops / ms / 1e9:.1f} TFLOP/s  (KV of the batch {kv_bytes / 1e6:.0f} MB, "
          f"{a.batch * a.ext / ms * 1e3:.0f} extend tokens/s/layer)", flush=True)


    if a.profile:
        n_wave = 1 << 20
        buf = torch.zeros((n_wave, 8), dtype=torch.int64, device=dev)
        pkg._hip.lib.bsgl_debug_prefill_profile(buf.data_ptr())
        call()
        torch.cuda.synchronize()
        pkg._hip.lib.bsgl_debug_prefill_profile(None)
        ti = buf.cpu()
        ti = ti[ti[:, 7] > 0]
        ticks = (ti[:, 7] >> 32).double()  # 100 MHz ticks of the wave's life (0 from kernels that do not stamp them)
        ti[:, 7] &= 0xFFFFFFFF
        t = ti.double()
        staging = (ti[:, 5] >> 32).double()
        t[:, 5] = (ti[:, 5] & 0xFFFFFFFF).double()
        steps = t[:, 5].sum()
        names = ("own-DMA wait", "barrier", "DMA issue", "score+softmax (32x32x16 kernel: whole step body)", "PV (16x16x32 kernel only)")
        ncol = 4 if ticks.sum() > 0 else 5  # (the 32x32x16 kernel's column 4 is a part of its column 3)
        tot = t[:, :ncol].sum()
        print(f"  {len(t)} waves, {int(steps)} wave-steps; cycles per wave-step by phase:")
        for i, n in enumerate(names[:ncol]):
            print(f"    {n:50s} {t[:, i].sum() / steps:8.0f}  ({t[:, i].sum() / tot:6.1%})")
        print(f"    {'sum':50s} {tot / steps:8.0f}")
        if ticks.sum() > 0:
            print(f"    clock while the waves ran: {t[:, 7].sum() / ticks.sum() / 10:.3f} GHz; score half of the body {t[:, 4].sum() / steps:.0f} cycles")
        pro = ti[:, 6] & 0xFFFFFFFF
        epi = ti[:, 6] >> 32
        life = t[:, 7]
        print(f"  per wave: whole life {life.mean():.0f} cycles = before the first page-table read {pro.double().mean():.0f} "
              f"+ staging the entries {staging.mean():.0f} + steps {t[:, :ncol].sum(1).mean():.0f} + after the last step {epi.double().mean():.0f} "
              f"+ rest {life.mean() - pro.double().mean() - staging.mean() - t[:, :ncol].sum(1).mean() - epi.double().mean():.0f}")


if __name__ == "__main__":
    main()
