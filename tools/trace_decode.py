"""Per-workgroup timeline of one matrix-core decode launch (bsgl_debug_decode_trace)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import baby_sglang_amd as pkg  # noqa: E402
from baby_sglang_amd import _hip  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lens", default="ragged")
    ap.add_argument("--splits", type=int, default=16)
    ap.add_argument("--uniform", action="store_true")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--seq", type=int, default=2048)
    ap.add_argument("--kv-layout", default="interleaved")
    ap.add_argument("--shape", default="12,2,128", help="q heads, kv heads, head_dim")
    ap.add_argument("--launches", type=int, default=1, help="> 1: also print what part of the spread of streaming ends repeats from launch to launch")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    hq, hkv, dim = (int(x) for x in args.shape.split(","))
    B, S, layers = args.batch, args.seq, 4
    g = torch.Generator().manual_seed(0)
    if args.lens == "ragged":
        lens = torch.randint(S // 2, 3 * S // 2, (B,), generator=g, dtype=torch.int64)
    elif args.lens == "two":
        lens = torch.tensor([S // 2, 3 * S // 2] * (B // 2), dtype=torch.int64)
    else:
        lens = torch.full((B,), S, dtype=torch.int64)
    gen = torch.Generator(device=dev).manual_seed(1)
    total = int(lens.sum())
    size = total + 16
    kv = pkg.MHATokenToKVPool(size, torch.bfloat16, hkv, dim, layers, str(dev), kv_layout=args.kv_layout)
    kv.arena.normal_(generator=gen)
    slots = (torch.randperm(size - 1, generator=gen, device=dev)[:total] + 1).to(torch.int32)
    table = torch.zeros((B, 4096), dtype=torch.int32, device=dev)
    o = 0
    for i, n in enumerate(lens.tolist()):
        table[i, :n] = slots[o : o + n]
        o += n
    q = torch.randn((layers, B, hq, dim), generator=gen, device=dev).to(torch.bfloat16)
    req, seq = torch.arange(B, device=dev), lens.to(dev)
    ws = _hip.new_decode_workspace(_hip.decode_workspace_bytes(B, hq, dim, args.splits), dev)
    out = torch.empty((B, hq * dim), dtype=torch.bfloat16, device=dev)
    n_wg = 8192
    trace = torch.zeros((n_wg, 8), dtype=torch.int64, device=dev)

    def call(layer):
        k, v = kv.get_kv_buffer(layer)
        _hip.paged_decode(q[layer], k, v, table, req, seq, n_splits=args.splits, variant=0 if dim != 128 else 2, out=out, workspace=ws,
                          uniform_splits=args.uniform)

    for layer in range(layers):
        call(layer)
    torch.cuda.synchronize()
    _hip.lib.bsgl_debug_decode_trace(trace.data_ptr())
    call(0)
    torch.cuda.synchronize()
    _hip.lib.bsgl_debug_decode_trace(None)
    t = trace.cpu()
    live = t[:, 0] != 0
    t = t[live]
    t0 = int(t[:, 0].min())
    start = (t[:, 0] - t0).double() / 100.0  # us
    ran = t[:, 1] != 0
    pro = ((t[:, 1] - t[:, 0]).double() / 100.0)[ran]
    loop = ((t[:, 2] - t[:, 1]).double() / 100.0)[ran]
    end_raw = t[:, 3] & 0x7FFFFFFFFFFFFFFF
    merged = (t[:, 3] < 0)
    fin = ((end_raw - t0).double() / 100.0)
    tail = ((end_raw - t[:, 2]).double() / 100.0)[ran]
    print(f"workgroups launched {int(live.sum())}, ran {int(ran.sum())}, merged {int(merged.sum())}")
    print(f"start   us: min {start.min():.2f} median {start.median():.2f} p90 {start.quantile(0.9):.2f} max {start.max():.2f}")
    print(f"prolog  us: median {pro.median():.2f} p90 {pro.quantile(0.9):.2f} max {pro.max():.2f}")
    print(f"stream  us: min {loop.min():.2f} median {loop.median():.2f} p90 {loop.quantile(0.9):.2f} max {loop.max():.2f}")
    print(f"tail    us: median {tail.median():.2f} p90 {tail.quantile(0.9):.2f} max {tail.max():.2f}")
    print(f"finish  us: median {fin[ran].median():.2f} p90 {fin[ran].quantile(0.9):.2f} max {fin.max():.2f}")
    if False:  # (used while timing a wait-based merge protocol)
        m = merged
        ent = ((t[:, 5] - t[:, 2]).double() / 100.0)[m]
        spin = (((t[:, 4] & 0xFFFFFFFFFFFF) - t[:, 5]).double() / 100.0)[m]
        polls = (t[:, 4] >> 48)[m] & 0xFFFF
        rest = ((end_raw - (t[:, 4] & 0xFFFFFFFFFFFF)).double() / 100.0)[m]
        print(f"mergers: loop end -> merge entry median {ent.median():.2f} max {ent.max():.2f}; wait median {spin.median():.2f} max {spin.max():.2f} "
              f"(polls median {int(polls.median())} max {int(polls.max())}); merge itself median {rest.median():.2f} max {rest.max():.2f}")
    xcc = (t[:, 4] & 0xF)[ran]
    hw = t[:, 5][ran]
    cu = (hw >> 8) & 0xF
    se = (hw >> 13) & 0x7
    sh = (hw >> 12) & 0x1
    cu_key = xcc * 1024 + se * 64 + sh * 16 + cu
    uniq, counts = cu_key.unique(return_counts=True)
    print(f"distinct CUs used {len(uniq)}; workgroups per CU: min {int(counts.min())} median {int(counts.median())} max {int(counts.max())}")
    cyc = (t[:, 4] >> 8)[ran].double()
    mhz = cyc / loop  # core-clock cycles of the streaming phase per us of wall time
    for x in range(8):
        m = xcc == x
        if m.any():
            print(f"  XCC {x}: {int(m.sum())} workgroups, stream median {loop[m].median():.2f} us, last finish {fin[ran][m].max():.2f} us, "
                  f"core clock while streaming {mhz[m].median():.0f} MHz")
    # stream time against workgroups sharing the CU
    per_cu = dict(zip(uniq.tolist(), counts.tolist()))
    load = torch.tensor([per_cu[int(k)] for k in cu_key])
    for c in sorted(set(load.tolist())):
        m = load == c
        print(f"  CUs holding {c} workgroups: {int(m.sum())} workgroups, stream median {loop[m].median():.2f} us")
    rows_ = (t[:, 6] & 0xFFFFFFFF)[ran]
    team = rows_ * 64 + (t[:, 6] >> 32)[ran]
    end_stream = ((t[:, 2] - t0).double() / 100.0)[ran]
    spreads, firsts, lasts = [], [], []
    for k in team.unique().tolist():
        m = team == k
        if int(m.sum()) > 1:
            spreads.append(float(end_stream[m].max() - end_stream[m].min()))
            firsts.append(float(end_stream[m].min()))
            lasts.append(float(end_stream[m].max()))
    if spreads:
        sp = torch.tensor(spreads)
        print(f"  within a (row, head group) team: streaming ends spread over median {sp.median():.2f} us, p90 {sp.quantile(0.9):.2f}, max {sp.max():.2f}; "
              f"team's last end: median {torch.tensor(lasts).median():.2f} max {max(lasts):.2f}; mean of all ends {end_stream.mean():.2f}")
    hgs = (t[:, 6] >> 32)[ran]
    for h in sorted(set(hgs.tolist())):
        m = hgs == h
        print(f"  head group {h}: {int(m.sum())} workgroups, stream median {loop[m].median():.2f} us, finish median {fin[ran][m].median():.2f} max {fin[ran][m].max():.2f}")
    par = (torch.arange(len(live))[live][ran] % 2)
    for h in (0, 1):
        m = par == h
        print(f"  launch-index parity {h}: stream median {loop[m].median():.2f} us")
    pieces = (t[:, 7] >> 32)[ran]
    for c in sorted(set(pieces.tolist())):
        m = pieces == c
        print(f"  rows cut in {c}: {int(m.sum())} workgroups, stream median {loop[m].median():.2f} us, tail median {tail[m].median():.2f}")


    if args.launches > 1:
        # the same launch traced again and again: per-workgroup mean of the end of streaming
        ends, starts = [], []
        for _ in range(args.launches):
            trace.zero_()
            _hip.lib.bsgl_debug_decode_trace(trace.data_ptr())
            call(0)
            torch.cuda.synchronize()
            _hip.lib.bsgl_debug_decode_trace(None)
            tt = trace.cpu()
            ok = tt[:, 1] != 0
            base = int(tt[tt[:, 0] != 0][:, 0].min())
            ends.append(torch.where(ok, (tt[:, 2] - base).double() / 100.0, torch.nan))
            starts.append(torch.where(ok, (tt[:, 1] - base).double() / 100.0, torch.nan))
        e = torch.stack(ends)  # [launches, workgroups]
        okc = ~torch.isnan(e).any(0)
        e = e[:, okc]
        st = torch.stack(starts)[:, okc]
        idx = torch.arange(trace.shape[0])[okc]
        mean_wg = e.mean(0)
        print(f"{args.launches} launches, {int(okc.sum())} workgroups: end of streaming: overall mean {e.mean():.2f} us; "
              f"std of a workgroup's mean over launches {mean_wg.std():.2f} us (what repeats), "
              f"std within a workgroup across launches {e.std(0).mean():.2f} us (what does not); "
              f"mean of each launch's last end {e.max(1).values.mean():.2f} us")
        for mod, name in ((8, "launch index mod 8"), (2, "launch index mod 2"), (16, "launch index mod 16")):
            row = "  ".join(f"{k}: {mean_wg[idx % mod == k].mean():.2f}" for k in range(mod))
            print(f"  mean end by {name}:  {row}")
        piece_i = idx // 2
        for k in range(8):
            m = piece_i % 8 == k
            print(f"  split {k} of its row: mean end {mean_wg[m].mean():.2f}, mean start of streaming {st.mean(0)[m].mean():.2f}")
        q = torch.tensor([0.0, 0.1, 0.5, 0.9, 1.0], dtype=torch.double)
        print("  per-workgroup mean end, quantiles 0 / 10 / 50 / 90 / 100 %: " + " ".join(f"{v:.2f}" for v in mean_wg.quantile(q).tolist()))


if __name__ == "__main__":
    main()
