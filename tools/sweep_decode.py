#!/usr/bin/env python3
"""Kernel-level sweep of the decode path on one GPU: HIP-event time per layer call
(decode + merge launches) for kernel variant / pipeline depth / heads per
workgroup / split count / KV layout / page-table order.  Prints one line per
configuration; correctness of each configuration is spot-checked against the
generic kernel with one split.

  python tools/sweep_decode.py [--quick]
"""
import argparse
import itertools
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--layouts", default="interleaved,split")
    ap.add_argument("--tables", default="perm,contig")
    ap.add_argument("--splits", default="4,8,12,16,24,32")
    ap.add_argument("--stages", default="2,3")
    ap.add_argument("--hpt", default="2,1")
    ap.add_argument("--variants", default="2,1")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--seq", type=int, default=2048)
    ap.add_argument("--nt", default="0")
    opt = ap.parse_args()
    import baby_sglang_amd as pkg

    dev = torch.device("cuda:0")
    base = bench.parse_args.__wrapped__() if hasattr(bench.parse_args, "__wrapped__") else None
    sys.argv = [sys.argv[0]]
    args = bench.parse_args()
    args.seq = opt.seq
    layout = pkg.TPLayout(0, 1, 12, 2)
    total_b = None
    for kv_layout in opt.layouts.split(","):
        args.kv_layout = kv_layout
        wl = bench.Workload(args, dev, layout)
        total_b, kv_b = wl.algorithmic_bytes_per_layer()
        # reference result for the spot check: generic kernel, one split, layer 0
        k_slab, v_slab = wl.kv.get_kv_buffer(0)
        for table in opt.tables.split(","):
            wl.set_table(table)
            ref = pkg._hip.paged_decode(wl.q[0], k_slab, v_slab, wl.r2t.req_to_token, wl.req, wl.local_seq,
                                        n_splits=1, variant=1).float()
            for variant in (int(v) for v in opt.variants.split(",")):
                stage_list = opt.stages.split(",") if variant == 2 else ["-"]
                hpt_list = opt.hpt.split(",") if variant == 2 else ["-"]
                nt_list = opt.nt.split(",") if variant == 2 else ["-"]
                for stages, hpt, nt, splits in itertools.product(stage_list, hpt_list, nt_list, (int(s) for s in opt.splits.split(","))):
                    if variant == 2:
                        pkg._hip.set_knob("BSGL_MFMA_NT", int(nt))
                        pkg._hip.set_knob("BSGL_MFMA_STAGES", int(stages))
                        pkg._hip.set_knob("BSGL_MFMA_HPT", int(hpt))
                    wl.args.variant = variant
                    wl.n_splits = splits
                    ws = pkg._hip.decode_workspace_bytes(args.batch, wl.hq_l, wl.dim, splits, False)
                    if wl.workspace.numel() < ws:
                        wl.workspace = pkg._hip.new_decode_workspace(ws, dev)
                    got = pkg._hip.paged_decode(wl.q[0], k_slab, v_slab, wl.r2t.req_to_token, wl.req, wl.local_seq,
                                                n_splits=splits, variant=variant, workspace=wl.workspace).float()
                    err = (got - ref).abs().max().item()
                    us = bench.kernel_event_times(wl, opt.reps) * 1e3
                    print(f"layout={kv_layout:11s} table={table:6s} variant={variant} stages={stages} hpt={hpt} nt={nt} seq={opt.seq} "
                          f"splits={splits:2d}  {us:7.2f} us/layer  {total_b / us / 1e3:7.1f} GB/s  "
                          f"frac={total_b / us / 1e3 / 8000:.3f}  maxdiff={err:.2e}", flush=True)
        del wl
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
