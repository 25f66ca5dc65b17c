"""Where does a ragged decode batch lose time?  Times the decode call on 8 rotating layers
for: uniform / ragged lengths x compact / spread slot ranges x balanced / equal pieces."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import baby_sglang_amd as pkg  # noqa: E402
from baby_sglang_amd import _hip  # noqa: E402


def run(lens, spread, n_splits, uniform, layers=8, reps=6):
    dev = torch.device("cuda:0")
    B, hq, hkv, dim = len(lens), 12, 2, 128
    gen = torch.Generator(device=dev).manual_seed(1)
    total = int(lens.sum().item())
    size = int(total * spread) + 16
    max_ctx = 4096
    kv = pkg.MHATokenToKVPool(size, torch.bfloat16, hkv, dim, layers, str(dev))
    kv.arena.normal_(generator=gen)
    slots = (torch.randperm(size - 1, generator=gen, device=dev)[:total] + 1).to(torch.int32)
    table = torch.zeros((B, max_ctx), dtype=torch.int32, device=dev)
    o = 0
    for i, n in enumerate(lens.tolist()):
        table[i, :n] = slots[o : o + n]
        o += n
    q = torch.randn((layers, B, hq, dim), generator=gen, device=dev).to(torch.bfloat16)
    req = torch.arange(B, device=dev)
    seq = lens.to(dev)
    ws = _hip.new_decode_workspace(_hip.decode_workspace_bytes(B, hq, dim, n_splits), dev)
    out = torch.empty((B, hq * dim), dtype=torch.bfloat16, device=dev)

    def step():
        for l in range(layers):
            k, v = kv.get_kv_buffer(l)
            _hip.paged_decode(q[l], k, v, table, req, seq, n_splits=n_splits, variant=2, out=out, workspace=ws,
                              uniform_splits=uniform)

    step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        step()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * layers)


def main():
    B, S = 64, 2048
    g = torch.Generator().manual_seed(0)
    uni = torch.full((B,), S, dtype=torch.int64)
    rag = torch.randint(S // 2, 3 * S // 2, (B,), generator=g, dtype=torch.int64)
    two = torch.tensor([S // 2, 3 * S // 2] * (B // 2), dtype=torch.int64)
    blocked = torch.tensor([3 * S // 2] * (B // 2) + [S // 2] * (B // 2), dtype=torch.int64)
    by8 = torch.tensor(([3 * S // 2] * 8 + [S // 2] * 8) * (B // 16), dtype=torch.int64)
    rag_sorted = rag.sort(descending=True).values
    cases = (("uniform", uni), ("uni-2041", uni - 7), ("ragged", rag), ("rag-r256", (rag + 255) // 256 * 256),
             ("rag-r16", (rag + 15) // 16 * 16), ("two-level", two), ("two+7", two + 7),
             ("three", torch.tensor([S // 2, S, 3 * S // 2, S] * (B // 4), dtype=torch.int64)))
    for name, lens in cases:
        for spread in (1.0,):
            for n_splits, uniform in ((8, True), (16, False)):
                us = run(lens, spread, n_splits, uniform)
                print(f"{name:9s} mean {lens.float().mean():7.1f} spread {spread} splits {n_splits:2d} "
                      f"{'equal   ' if uniform else 'balanced'}: {us:6.1f} us/layer", flush=True)


if __name__ == "__main__":
    main()
