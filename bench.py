#!/usr/bin/env python3
"""Decode-step benchmark of the paged-attention path on MI355X.

A step = one pass of the attention hot path over one synthetic decode batch:
for each of the 28 layers, ``SimpleAttentionBackend.forward(q, k, v, layer, fb)``
(KV write of the new token + paged decode attention over the whole context).
Workload (BASELINE.json configs[2], Qwen2.5-1.5B shapes): batch 64, every
request at sequence length 2048, 12 q heads / 2 kv heads / head_dim 128, bf16,
one KV slab pair per layer, random-permutation page table (worst-case scatter).

  python bench.py [--gpus N --steps K --warmup W]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Typed without a launcher (``WORLD_SIZE`` unset) and ``--gpus N > 1``, it starts the N ranks itself
(``self_launch``): N fresh child processes with the ``RANK`` / ``LOCAL_RANK`` / ``WORLD_SIZE`` /
``MASTER_ADDR`` / ``MASTER_PORT`` environment that ``init_tp`` reads (reference: tp_runtime.py:10-25),
started before this process has touched the GPU; rank 0's JSON line is forwarded, and the exit code
is non-zero when any rank fails or the job outlives its watchdog.

Prints ONE JSON line (rank 0).  ``value`` is decode tokens/s of the whole job
with every input resident in HBM.  ``roofline`` prices the dominant kernel (the
split-KV decode kernel) against HBM bandwidth with algorithmic bytes;
``cpu_baseline`` times the oracle's reference-style torch-CPU backend on a
bounded sample of the same workload on this box's host cores.
"""

from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--seq", type=int, default=2048)
    ap.add_argument("--layers", type=int, default=28)
    ap.add_argument("--max-context", type=int, default=4096)
    ap.add_argument("--table", choices=["perm", "contig", "radix", "ragged", "pairs"], default="perm")
    ap.add_argument("--kv-layout", choices=["interleaved", "split"], default="interleaved")
    ap.add_argument("--variant", type=int, default=0, help="0 auto, 1 generic, 2 matrix-core decode kernel")
    ap.add_argument("--splits", type=int, default=0, help="sequence splits (0 = library heuristic)")
    ap.add_argument("--mode", choices=["graph", "eager"], default="graph",
                    help="replay the 28-layer step from a HIP graph, or launch it from Python each step")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--no-fuse", action="store_true", help="KV write as its own launch instead of inside the decode kernel")
    ap.add_argument("--no-extras", action="store_true", help="skip the contiguous-table and eager/graph side measurements")
    ap.add_argument("--tp", type=int, default=0,
                    help="ranks per tensor-parallel group (0 = N: ONE group of all ranks, kv heads first, then sequence "
                         "slices inside a kv head).  A smaller value runs N / tp such groups side by side as replicas on "
                         "different requests (--tp 2: each rank owns a whole kv head, no sequence slices)")
    ap.add_argument("--collectives", choices=["auto", "torch", "capi"], default="torch",
                    help="N > 1: torch = torch.distributed (RCCL) issued from Python per layer; capi = the library's own "
                         "RCCL communicator (bsgl_all_gather enqueued on the launch stream, no per-call work objects; "
                         "kv-head shards only); auto = capi when its communicator comes up on every rank and its first "
                         "gather equals torch.distributed's, torch otherwise.  The default is torch: the library's "
                         "communicator has never come up between two GPUs (one GPU per box where this was built), "
                         "and a first run should not depend on it")
    ap.add_argument("--tp-graph", action="store_true",
                    help="N > 1 with capi collectives: replay the whole step, collectives included, from one HIP graph")
    ap.add_argument("--launch-timeout", type=float, default=1500.0,
                    help="N > 1 started without a launcher: seconds the ranks this process starts may run before they are ended")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="N > 1 GPUs: weak = N x batch requests over the tensor-parallel group (per-GPU KV bytes fixed), "
                         "strong = the same batch cut N ways")
    return ap.parse_args()


def self_launch(n_ranks, child_argv=None, timeout_s=1500.0, poll_s=0.2, out=None, local_ranks=None):
    """Start ``n_ranks`` fresh processes of this program with the environment a launcher would give
    them (``RANK``, ``LOCAL_RANK``, ``WORLD_SIZE``, ``MASTER_ADDR=127.0.0.1``, a free ``MASTER_PORT``),
    forward what rank 0 writes to its standard output to ``out`` and return the job's exit code:
    0 when every rank exited 0; the first failing rank's code (the other ranks are ended at once, so
    that nobody waits in a collective for a peer that is gone); 124 when the ranks outlive
    ``timeout_s`` (all are ended).  The calling process must not have initialised the GPU."""
    out = out if out is not None else sys.stdout
    if child_argv is None:
        child_argv = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    if local_ranks is None:
        local_ranks = list(range(n_ranks))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = dict(os.environ, WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                LOCAL_WORLD_SIZE=str(n_ranks), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = []
    try:
        for r in range(n_ranks):
            env = dict(base, RANK=str(r), LOCAL_RANK=str(local_ranks[r]))
            # rank 0's standard output carries the JSON line; the other ranks print nothing there
            procs.append(subprocess.Popen(child_argv, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
        os.set_blocking(procs[0].stdout.fileno(), False)
        deadline = time.monotonic() + timeout_s
        rc = None
        while rc is None:
            chunk = procs[0].stdout.read()
            if chunk:
                out.write(chunk.decode("utf-8", "replace"))
                out.flush()
            codes = [p.poll() for p in procs]
            bad = [c for c in codes if c not in (None, 0)]
            if bad:
                rc = bad[0] if bad[0] > 0 else 128 - bad[0]  # (a rank ended by signal n: 128 + n)
            elif all(c == 0 for c in codes):
                rc = 0
            elif time.monotonic() > deadline:
                print(f"bench.py: the {n_ranks} ranks did not finish within {timeout_s:.0f} s; ending them", file=sys.stderr)
                rc = 124
            else:
                time.sleep(poll_s)
        if rc == 0:
            rest = procs[0].stdout.read()
            if rest:
                out.write(rest.decode("utf-8", "replace"))
                out.flush()
        return rc
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.monotonic() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()


class Workload:
    """Per-rank synthetic decode state in HBM."""

    def __init__(self, args, dev, layout):
        import baby_sglang_amd as pkg

        self.pkg = pkg
        self.args = args
        self.dev = dev
        self.layout = layout
        B, S, L = args.batch, args.seq, args.layers
        self.hq, self.hkv, self.dim = 12, 2, 128
        self.hq_l, self.hkv_l = layout.local_q_heads, layout.local_kv_heads
        self.dtype = torch.bfloat16
        gen = torch.Generator(device=dev).manual_seed(1234 + getattr(args, "seed_rank", layout.rank))
        # this rank's share of every request's tokens (round-robin by position when tp_seq > 1)
        seq = torch.full((B,), S, dtype=torch.int64, device=dev)
        self.local_seq = layout.local_len(seq)
        s_loc = int(self.local_seq.max().item())
        self.s_loc = s_loc
        size = B * args.max_context // layout.tp_seq
        self.r2t = pkg.ReqToTokenPool(B, args.max_context, str(dev))
        self.kv = pkg.MHATokenToKVPool(size, self.dtype, self.hkv_l, self.dim, L, str(dev), kv_layout=args.kv_layout)
        self.kv.arena.normal_(generator=gen)
        self.tables = {}
        perm = torch.randperm(B * s_loc, generator=gen, device=dev).to(torch.int32) + 1
        t = torch.zeros((B, args.max_context), dtype=torch.int32, device=dev)
        t[:, :s_loc] = perm.view(B, s_loc)
        self.tables["perm"] = t
        t = torch.zeros((B, args.max_context), dtype=torch.int32, device=dev)
        t[:, :s_loc] = (torch.arange(B * s_loc, device=dev, dtype=torch.int32) + 1).view(B, s_loc)
        self.tables["contig"] = t
        # radix-cache reuse: groups of 8 requests share the slots of their first half
        # (a common prompt prefix held once by the radix tree, radix_cache.py:142-205)
        t = self.tables["perm"].clone()
        half = s_loc // 2
        for g0 in range(0, B, 8):
            t[g0 : g0 + 8, :half] = t[g0, :half]
        self.tables["radix"] = t
        # requests 2i and 2i+1 name the same slots throughout: every KV byte is wanted twice, at the same time
        t = self.tables["perm"].clone()
        t[1::2] = t[0::2]
        self.tables["pairs"] = t
        # ragged batch: lengths uniform in [S/2, 3S/2) with the same total work on average
        self.ragged_seq = torch.randint(max(1, S // 2), 3 * S // 2, (B,), generator=gen, device=dev, dtype=torch.int64)
        self.ragged_seq = layout.local_len(self.ragged_seq.clamp_max(args.max_context))
        t = torch.zeros((B, args.max_context), dtype=torch.int32, device=dev)
        n_r = int(self.ragged_seq.max().item())
        t[:, :n_r] = (torch.randperm(B * n_r, generator=gen, device=dev).to(torch.int32) % size + 1).view(B, n_r)
        self.tables["ragged"] = t
        self.uniform_seq = self.local_seq
        self.q = torch.randn((L, B, self.hq_l, self.dim), generator=gen, device=dev).to(self.dtype)
        self.k = torch.randn((L, B, self.hkv_l, self.dim), generator=gen, device=dev).to(self.dtype)
        # v as the model hands it over: a column slice of a wider projection row
        self.v_parent = torch.randn((L, B, (self.hq_l + 2 * self.hkv_l) * self.dim), generator=gen, device=dev).to(self.dtype)
        self.layers = [types.SimpleNamespace(layer_id=i, scaling=self.dim**-0.5) for i in range(L)]
        self.req = torch.arange(B, dtype=torch.int64, device=dev)
        self.writes_kv = layout.owns_position(S - 1)
        self.set_table(args.table)
        self.n_splits = args.splits or pkg._hip.decode_num_splits(B, self.hkv_l, args.max_context)
        self.out_f32 = layout.tp_seq > 1
        ws = pkg._hip.decode_workspace_bytes(B, self.hq_l, self.dim, self.n_splits, self.out_f32)
        self.workspace = pkg._hip.new_decode_workspace(ws, dev)
        self.lse = torch.empty((B, self.hq_l), dtype=torch.float32, device=dev) if self.out_f32 else None
        self.out = torch.empty((B, self.hq_l * self.dim), dtype=torch.float32 if self.out_f32 else self.dtype, device=dev)
        self.full_out = None

    def set_table(self, kind):
        self.r2t.req_to_token.copy_(self.tables[kind])
        self.local_seq = self.ragged_seq if kind == "ragged" else self.uniform_seq
        # slot of the token written this step = last local position of each request
        self.loc = self.r2t.req_to_token[self.req, (self.local_seq - 1).clamp_min(0)].contiguous()
        self.fb = self.pkg.SimplifiedForwardBatch.create_decode_batch(
            torch.zeros(self.args.batch, dtype=torch.int64, device=self.dev), self.req, self.local_seq,
            self.loc, self.r2t, self.kv,
        )
        # (every synthetic table but "pairs" gives each request a last slot of its own)
        self.fb.private_decode_slots = kind != "pairs"

    def v_rows(self, layer):
        c0 = (self.hq_l + self.hkv_l) * self.dim
        return self.v_parent[layer][:, c0:].view(self.args.batch, self.hkv_l, self.dim)

    def attention_layer(self, layer, tp_group=None):
        """KV write + paged decode (+ the TP exchange) of one layer."""
        hip = self.pkg._hip
        k_slab, v_slab = self.kv.get_kv_buffer(layer)
        new = {}
        if self.writes_kv:
            new = dict(k_new=self.k[layer], v_new=self.v_rows(layer), out_cache_loc=self.loc,
                       fuse_kv_write=not self.args.no_fuse)
        hip.paged_decode(
            self.q[layer], k_slab, v_slab, self.r2t.req_to_token, self.req, self.local_seq,
            n_splits=self.n_splits, variant=self.args.variant, out=self.out, lse=self.lse,
            workspace=self.workspace, **new,
        )
        if self.layout.world > 1:
            self.full_out = self.pkg.tp_attention_exchange(self.layout, self.out, self.lse, tp_group, self.dtype)

    def step(self, tp_group=None):
        for layer in range(self.args.layers):
            self.attention_layer(layer, tp_group)

    def backend_step(self):
        """The same step through the reference-shaped entry point (1 GPU)."""
        fwd = self.pkg.SimpleAttentionBackend.forward
        for layer in range(self.args.layers):
            fwd(self.q[layer], self.k[layer], self.v_rows(layer), self.layers[layer], self.fb)

    def algorithmic_bytes_per_layer(self, shared_rows_once=False):
        """Bytes one layer call has to move: the K and V rows of every visible token, the page-table
        entries, q, out and the new rows.  ``shared_rows_once``: a cache row that several requests'
        page tables name (a shared prefix) is counted once, not once per request."""
        B, D, eb = self.args.batch, self.dim, 2
        tokens = int(self.local_seq.sum().item())
        kv_rows = tokens
        if shared_rows_once:
            table = self.r2t.req_to_token[self.req]
            visible = torch.arange(table.shape[1], device=table.device)[None, :] < self.local_seq[:, None]
            kv_rows = int(torch.unique(table[visible]).numel())
        kv = 2 * kv_rows * self.hkv_l * D * eb
        table = tokens * 4
        q = B * self.hq_l * D * eb
        out = B * self.hq_l * D * (4 if self.out_f32 else eb)
        new_kv = 2 * B * self.hkv_l * D * eb if self.writes_kv else 0
        return kv + table + q + out + new_kv, kv


def barrier(world):
    if world > 1:
        torch.distributed.barrier()


LAST_EVENT_MS = [0.0]


def timed_steps(fn, steps, warmup, world):
    """Barrier + device sync on both sides; wall time of exactly `steps` calls; max over
    ranks.  A HIP event pair on the launch stream brackets the same region
    (LAST_EVENT_MS): device time of the steps without the host-side sync cost."""
    for _ in range(warmup):
        fn()
    stream = torch.cuda.current_stream()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    barrier(world)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(steps):
        fn()
    ev1.record(stream)
    torch.cuda.synchronize()
    barrier(world)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    LAST_EVENT_MS[0] = ev0.elapsed_time(ev1)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def make_runner(wl, mode, tp_group):
    if mode == "eager":
        return lambda: wl.step(tp_group)
    # capture the whole 28-layer step once, replay it per step
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            wl.step(tp_group)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        wl.step(tp_group)
    return graph.replay


def kernel_event_times(wl, steps):
    """HIP-event time of the decode launches alone (split kernel + merge), one
    event pair per layer call on the launch stream, averaged over steps * layers."""
    hip = wl.pkg._hip
    L = wl.args.layers
    stream = torch.cuda.current_stream()
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(L)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(L)]
    total_ms, n = 0.0, 0
    for it in range(steps + 1):
        for layer in range(L):
            k_slab, v_slab = wl.kv.get_kv_buffer(layer)
            starts[layer].record(stream)
            hip.paged_decode(wl.q[layer], k_slab, v_slab, wl.r2t.req_to_token, wl.req, wl.local_seq,
                             n_splits=wl.n_splits, variant=wl.args.variant, out=wl.out, lse=wl.lse,
                             workspace=wl.workspace)
            ends[layer].record(stream)
        torch.cuda.synchronize()
        if it == 0:
            continue  # warm-up round
        total_ms += sum(s.elapsed_time(e) for s, e in zip(starts, ends))
        n += L
    return total_ms / n


def cpu_baseline(args, budget_s):
    """Reference-style torch-CPU backend (oracle/attn_oracle.py: per-request
    gather -> GQA repeat -> SDPA, bf16) on ONE layer of the same workload, repeated
    for about `budget_s` seconds; tokens/s = batch / (layers * seconds per layer)."""
    from oracle import attn_oracle as orc

    B, S = args.batch, args.seq
    hq, hkv, dim, dtype = 12, 2, 128, torch.bfloat16
    gen = torch.Generator().manual_seed(0)
    n_slots = B * S + 1
    k_slab = torch.randn((n_slots, hkv, dim), generator=gen).to(dtype)
    v_slab = torch.randn((n_slots, hkv, dim), generator=gen).to(dtype)
    table = torch.zeros((B, args.max_context), dtype=torch.int32)
    table[:, :S] = (torch.randperm(B * S, generator=gen) + 1).to(torch.int32).view(B, S)
    req = torch.arange(B)
    seq = torch.full((B,), S, dtype=torch.int64)
    q = torch.randn((B, hq, dim), generator=gen).to(dtype)
    k = torch.randn((B, hkv, dim), generator=gen).to(dtype)
    v = torch.randn((B, hkv, dim), generator=gen).to(dtype)
    loc = table[req, seq - 1]

    def one_layer():
        return orc.forward_ref(q, k, v, k_slab, v_slab, table, req, seq, loc, is_prefill=False)

    one_layer()
    reps, t0 = 0, time.perf_counter()
    while True:
        one_layer()
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or reps >= 2000:
            break
    per_layer = dt / reps
    return {
        "value": B / (per_layer * args.layers),
        "unit": "tokens/s",
        "cores": torch.get_num_threads(),
        "kind": "port",
        "sample": f"1 of {args.layers} layers of the same batch ({B} x seq {S}, bf16, random-permutation table), "
                  f"{reps} reps in {dt:.1f} s, scaled x{args.layers}; host has {os.cpu_count()} logical CPUs",
        "ms_per_layer": per_layer * 1e3,
    }


def extend_problem(dev, batch=128, ext=512, prefix=1536, hq=28, hkv=4, dim=128):
    """The extend side measurement's inputs (7B shapes by default): `batch` requests, `ext` new
    tokens each over `prefix` cached ones, bf16, random-permutation page table."""
    import baby_sglang_amd as pkg

    seq = prefix + ext
    gen = torch.Generator(device=dev).manual_seed(0)
    kv = pkg.MHATokenToKVPool(batch * seq, torch.bfloat16, hkv, dim, 1, str(dev))
    kv.arena.normal_(generator=gen)
    r2t = pkg.ReqToTokenPool(batch, seq, str(dev))
    r2t.req_to_token.copy_((torch.randperm(batch * seq, generator=gen, device=dev).to(torch.int32) + 1).view(batch, seq))
    q = torch.randn((batch * ext, hq, dim), generator=gen, device=dev).to(torch.bfloat16)
    k_slab, v_slab = kv.get_kv_buffer(0)
    return dict(batch=batch, ext=ext, prefix=prefix, hq=hq, hkv=hkv, dim=dim, kv=kv, r2t=r2t, q=q,
                k_slab=k_slab, v_slab=v_slab, rows=torch.arange(batch, device=dev),
                seq=torch.full((batch,), seq, dtype=torch.int64, device=dev),
                pre=torch.full((batch,), prefix, dtype=torch.int64, device=dev),
                out=torch.empty((q.shape[0], hq * dim), dtype=torch.bfloat16, device=dev))


def extend_extras(dev, reps=100, warmup=150):
    """Extend (prefill) attention beside the headline: one layer call of a 7B-shaped chunked
    prefill (128 requests, 512 new tokens over 1536 cached, Hq=28 / Hkv=4, bf16, random-permutation
    page table) -- ms per layer call, TFLOP/s with flops = 4 Hq D (ext * prefix + ext (ext + 1) / 2)
    per request, and the fraction of the dense bf16 MFMA peak (2.5 PFLOP/s)."""
    import baby_sglang_amd as pkg

    p = extend_problem(dev)
    batch, ext, prefix, hq, dim = p["batch"], p["ext"], p["prefix"], p["hq"], p["dim"]
    q, out = p["q"], p["out"]
    ws = torch.empty(pkg._hip.prefill_workspace_bytes(batch, q.shape[0], hq, dim), dtype=torch.uint8, device=dev)

    def call():
        pkg._hip.paged_prefill(q, p["k_slab"], p["v_slab"], p["r2t"].req_to_token, p["rows"], p["seq"], p["pre"],
                               p["seq"] - p["pre"], out=out, workspace=ws)

    def timed(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            call()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    # A chunked prefill makes this call 28 times per chunk, chunk after chunk, for seconds: the
    # figure quoted is the sustained one.  This matrix-core-bound kernel runs 10 - 20 % slower in
    # its first tens of launches after a memory-bound phase (the clock the chip holds comes up over
    # some 0.3 s of such load); the first five launches are reported beside it.
    call()
    torch.cuda.synchronize()
    ms_first = timed(5)
    for _ in range(warmup):
        call()
    torch.cuda.synchronize()
    ms = timed(reps)
    flops = 4.0 * hq * dim * (ext * prefix + ext * (ext + 1) / 2) * batch
    return {"extend_workload": "7B shapes: 128 requests x 512 new tokens over 1536 cached, Hq=28 Hkv=4 D=128 bf16, one layer call",
            "extend_reps": reps, "extend_launches_before": warmup + 6, "extend_first5_ms_per_layer": ms_first, "extend_first5_tflops": flops / ms_first / 1e9,
            "extend_ms_per_layer": ms, "extend_tflops": flops / ms / 1e9, "extend_frac_of_bf16_mfma_peak": flops / ms / 1e9 / 2500.0}


CAPI_NOTE = [None]


def _all_ok(ok, world_group, dev):
    """MIN over every rank of the job: did the step just taken succeed everywhere?"""
    dist = torch.distributed
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev if dist.get_backend(world_group) != "gloo" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=world_group)
    return int(flag.item()) == 1


def try_capi_collectives(pkg, tp_group, world_group, tp_rank, dev):
    """Bring up the library's own RCCL communicator for this rank's tensor-parallel group and check
    its first all-gather against torch.distributed's.  Every rank makes the SAME sequence of
    collective calls whatever happens locally -- a failure is carried as a marker and agreed on by
    a MIN all-reduce over the whole job before the next step -- so a rank that fails can never
    leave its peers inside a collective it skipped: (1) every rank loads RCCL (making an id does
    that), rank 0 of the group broadcasts its id or a failure marker; agree.  (2) every rank enters
    the blocking communicator set-up; agree.  (3) every rank runs both gathers and compares; agree."""
    dist = torch.distributed
    hip, tpr = pkg._hip, pkg.tp_runtime
    src = dist.get_global_rank(tp_group, 0) if tp_group is not None else 0
    # (1) id hand-over
    ok, ident = True, None
    try:
        ident = hip.comm_unique_id()
    except Exception as exc:  # RCCL not loadable here
        ok, CAPI_NOTE[0] = False, repr(exc)[:300]
    box = [ident if (tp_rank == 0 and ok) else None]
    dist.broadcast_object_list(box, src=src, group=tp_group)
    ok = ok and box[0] is not None
    if not _all_ok(ok, world_group, dev):
        CAPI_NOTE[0] = CAPI_NOTE[0] or "another rank could not load RCCL through the library"
        return False
    # (2) communicator (blocks until every rank of the group has entered: all of them do)
    world = dist.get_world_size(group=tp_group)
    try:
        hip.comm_init(tp_rank, world, box[0])
    except Exception as exc:
        ok, CAPI_NOTE[0] = False, repr(exc)[:300]
    if not _all_ok(ok, world_group, dev):
        CAPI_NOTE[0] = CAPI_NOTE[0] or "another rank could not bring up the library's communicator"
        if ok:
            hip.comm_destroy()
        return False
    # (3) probe gather, both ways, on every rank
    probe = torch.full((4, 8), float(tp_rank + 1), device=dev, dtype=torch.bfloat16)
    want = torch.empty((world * 4, 8), device=dev, dtype=torch.bfloat16)
    dist.all_gather_into_tensor(want, probe, group=tp_group)
    try:
        got = hip.all_gather(probe)
        torch.cuda.synchronize()
        if not torch.equal(got, want):
            ok, CAPI_NOTE[0] = False, "the library's all-gather disagreed with torch.distributed's"
    except Exception as exc:
        ok, CAPI_NOTE[0] = False, repr(exc)[:300]
    if not _all_ok(ok, world_group, dev):
        CAPI_NOTE[0] = CAPI_NOTE[0] or "the library's all-gather failed its check on another rank"
        hip.comm_destroy()
        return False
    tpr._CAPI_GROUP[0], tpr._CAPI_GROUP[1] = tp_group, True
    return True


def read_traffic():
    """HBM bytes per decode launch from the committed rocprofv3 --pmc summary (PMC counters cannot
    be read from inside a timed run), and where that number comes from."""
    path = os.path.join(ROOT, "profiles", "traffic_latest.json")
    try:
        with open(path) as f:
            t = json.load(f)
        src = f"profiles/traffic_latest.json (rocprofv3 --pmc passes: {t.get('source', 'tools/pmc_traffic.sh')}; library at commit {t.get('commit', 'unrecorded')})"
        return t.get("hbm_bytes_per_launch"), src
    except (OSError, ValueError):
        return None, None


def measure(args, wl, world, world_group, tp_group, tp, layout, n_replicas, dev, pkg, tp_rank):
    """Time args.steps steps of `wl`; returns the fields of one measurement."""
    note = None
    collectives = None
    mode = args.mode
    if tp > 1:
        want_capi = args.collectives in ("auto", "capi") and layout.tp_seq == 1
        got_capi = want_capi and try_capi_collectives(pkg, tp_group, world_group, tp_rank, dev)
        collectives = "capi" if got_capi else "torch"
        if want_capi and not got_capi:
            note = CAPI_NOTE[0] or "another rank could not bring up the library's communicator"
        # torch.distributed calls are issued from Python each layer; the library's own enqueue-only
        # collectives may also be captured with the step
        mode = "graph" if (got_capi and args.tp_graph) else "eager"
    runner = make_runner(wl, mode, tp_group)
    dt = timed_steps(runner, args.steps, args.warmup, world)
    rccl_ranks = None
    if tp > 1:
        backend = torch.distributed.get_backend(tp_group)
        rccl_ranks = pkg._hip.comm_world() if collectives == "capi" else (torch.distributed.get_world_size(group=tp_group) if backend == "nccl" else 0)
    return dict(dt=dt, event_ms=LAST_EVENT_MS[0], mode=mode, collectives=collectives, note=note, rccl_ranks=rccl_ranks)


HBM_ACHIEVABLE_GBS = 6290.0  # what a streaming copy reaches on MI355X (measured figure of the hardware guide)


def parallelism_label(world, n_replicas, tp, layout):
    if world == 1:
        return "single GPU"
    how = ("no exchange" if tp == 1 else
           "one all-gather of the output rows per layer" if layout.tp_seq == 1 else
           "per layer an all-to-all of partial (o, lse) states among the ranks of a kv head, a merge, and an all-gather of the merged rows")
    groups = "ONE tensor-parallel group of all ranks" if n_replicas == 1 else f"{n_replicas} replica(s) x tp{tp}"
    return f"{groups}: tp{tp} = {layout.tp_kv} kv-head shards x {layout.tp_seq} sequence slices; {how}"


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # typed without a launcher: this process only starts the ranks (it never touches the GPU;
        # counting the devices does not initialise it)
        n_dev = torch.cuda.device_count()
        gloo = os.getenv("BABYSGL_TP_BACKEND") == "gloo"
        if n_dev < args.gpus and not (gloo and n_dev >= 1):
            raise SystemExit(f"--gpus {args.gpus}: {n_dev} GPU(s) visible and RCCL wants one per rank "
                             "(BABYSGL_TP_BACKEND=gloo rehearses several ranks on fewer GPUs)")
        local = [r if n_dev >= args.gpus else r % n_dev for r in range(args.gpus)]
        import signal

        def ended(signum, frame):  # SIGTERM to this process ends its ranks too (self_launch's clean-up runs)
            raise SystemExit(128 + signum)

        signal.signal(signal.SIGTERM, ended)
        sys.exit(self_launch(args.gpus, timeout_s=args.launch_timeout, local_ranks=local))
    # stdout carries the ONE JSON line and nothing else: RCCL prints a five-line version banner on
    # stdout when a communicator comes up (every rank), so file descriptor 1 is pointed at stderr
    # for the run and the line goes out through a copy of the original descriptor.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    world = int(os.getenv("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world
    import baby_sglang_amd as pkg

    rank, world, world_group = pkg.init_tp()
    dev = torch.device("cuda", torch.cuda.current_device())
    per_gpu_batch = args.batch
    # N > 1: ONE tensor-parallel group of all N ranks (BASELINE configs[3]: TP = 2 / 4 / 8): kv heads
    # are split first (2 for these shapes), the ranks of a kv head then split every request's tokens
    # round-robin by position.  --tp T < N runs N / T such groups as replicas on different requests.
    tp = args.tp if args.tp > 0 else world
    tp_group, tp_rank, replica, n_replicas = pkg.split_tp_groups(tp, world_group)
    base_batch = args.batch

    def sized(tp_, n_replicas_):
        a = argparse.Namespace(**vars(args))
        if args.scaling == "weak":
            # every rank holds its kv-head / sequence share of tp x batch requests, i.e. the KV
            # bytes of `batch` whole requests, whatever N
            a.batch = base_batch * tp_
        elif n_replicas_ > 1:
            if base_batch % n_replicas_:
                raise SystemExit("--scaling strong: the batch must divide over the replicas")
            a.batch = base_batch // n_replicas_
        a.seed_rank = rank  # replicas hold different requests
        return a

    wargs = sized(tp, n_replicas)
    layout = pkg.TPLayout(tp_rank, tp, 12, 2)
    job_batch = wargs.batch * n_replicas
    wl = Workload(wargs, dev, layout)
    m = measure(wargs, wl, world, world_group, tp_group, tp, layout, n_replicas, dev, pkg, tp_rank)
    dt = m["dt"]
    event_ms_per_step = m["event_ms"] / args.steps
    ms_per_step = dt / args.steps * 1e3
    tokens_per_s = job_batch * args.steps / dt

    result = {
        "metric": "decode tokens/s, paged-attention path, Qwen2.5-1.5B shapes bs=64 seq=2048",
        "value": tokens_per_s,
        "unit": "tokens/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": args.scaling if world > 1 else "weak",
        "vs_baseline": None,
        "dtype": "bf16",
        "data": "synthetic (N(0,1) q/k/v and KV slabs, random-permutation page table, torch.manual_seed)",
        "config": {
            "workload": f"BASELINE configs[{2 if world == 1 else 3}]: Qwen2.5-1.5B bf16 decode, batch={job_batch}, seq={args.seq}, "
                        f"{args.layers} layers x (KV write + paged decode attention"
                        + ("" if world == 1 else " + tensor-parallel exchange of the output rows") +
                        f"), Hq=12 Hkv=2 D=128, page table={args.table}, kv_layout={args.kv_layout}",
            "global_batch": job_batch,
            "kv_bytes_per_gpu_equal_to_batch": per_gpu_batch if args.scaling == "weak" or world == 1 else job_batch / world,
            "seq_len": args.seq,
            "parallelism": parallelism_label(world, n_replicas, tp, layout),
            "launch": m["mode"],
            "collectives": m["collectives"],
            "collectives_note": m["note"],
            "rccl_ranks": m["rccl_ranks"],
            "n_splits": wl.n_splits,
            "decode_variant": args.variant,
        },
    }

    if rank == 0 or world > 1:
        # Dominant kernel = the decode kernel, launched once per layer (it also stores the new
        # token and merges its splits).  Its average launch duration is the HIP-event time of the
        # timed region divided by the launches in it; at one GPU the region holds nothing else, so
        # this is the kernel time plus the boundary between dependent launches (an upper bound on
        # what rocprofv3 reports for the kernel alone).  With several GPUs the region also holds
        # the collectives, so the launches are timed on their own there (this rank's GPU).
        if tp == 1:
            ev_ms = event_ms_per_step / args.layers
        else:
            ev_ms = kernel_event_times(wl, max(2, min(args.steps, 10)))
        total_b, kv_b = wl.algorithmic_bytes_per_layer()
        achieved = total_b / (ev_ms * 1e-3) / 1e9
        traffic, traffic_source = read_traffic() if world == 1 else (None, None)
        result["roofline"] = {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "frac_of_achievable": achieved / HBM_ACHIEVABLE_GBS,
            "achievable": HBM_ACHIEVABLE_GBS,
            "traffic": traffic,
            "traffic_source": traffic_source,
            "kernel": "decode_mfma_kernel (paged decode of one layer incl. new-token KV write and split merge); "
                      "HIP events over the timed region / launches" if world == 1 else
                      "decode launches of one layer on this rank's GPU (per-GPU figure), HIP events around each call",
            "algorithmic_bytes_per_launch": total_b,
            "kv_bytes_per_launch": kv_b,
            "avg_launch_us": ev_ms * 1e3,
        }
        result["step_hbm_gbs"] = total_b * args.layers / (ms_per_step * 1e-3) / 1e9
        if tp > 1:
            # the exchange step broken out: decode launches alone vs the whole step
            result["attention_only_ms_per_step"] = ev_ms * args.layers
            result["exchange_ms_per_step"] = max(0.0, ms_per_step - ev_ms * args.layers)

    if world > 2 and args.tp == 0 and not args.no_extras and args.scaling == "weak":
        # Beside the tensor-parallel line: the same GPUs as REPLICAS of a two-rank group (each rank a
        # whole kv head, no sequence slices; replicas serve different requests and exchange nothing).
        # Not config 4 -- its curve is linear by construction -- but what a serving stack would run.
        extras = result.setdefault("extras", {})
        if rank == 0:
            # this side measurement makes new process groups and collectives of its own: should it
            # stall, the line measured so far still goes out (and the job is ended)
            import signal

            def give_up(signum, frame):
                extras["replicas_x_tp2_error"] = "timed out"
                json_out.write(json.dumps(result) + "\n")
                json_out.flush()
                os._exit(3)

            signal.signal(signal.SIGALRM, give_up)
            signal.alarm(240)
        try:
            pkg.tp_runtime.release_capi_collectives()
            g2, r2, rep2, nrep2 = pkg.split_tp_groups(2, world_group)
            a2 = sized(2, nrep2)
            lay2 = pkg.TPLayout(r2, 2, 12, 2)
            del wl
            torch.cuda.empty_cache()
            wl2 = Workload(a2, dev, lay2)
            m2 = measure(a2, wl2, world, world_group, g2, 2, lay2, nrep2, dev, pkg, r2)
            extras["replicas_x_tp2"] = {
                "label": parallelism_label(world, nrep2, 2, lay2) + " -- replicas, not tensor parallelism over all ranks",
                "global_batch": a2.batch * nrep2,
                "ms_per_step": m2["dt"] / args.steps * 1e3,
                "tokens_per_s": a2.batch * nrep2 * args.steps / m2["dt"],
                "collectives": m2["collectives"], "launch": m2["mode"], "rccl_ranks": m2["rccl_ranks"],
            }
        except Exception as exc:  # a side measurement: it must not take the headline line with it
            extras["replicas_x_tp2_error"] = repr(exc)[:300]
        if rank == 0:
            signal.alarm(0)

    if world == 1 and not args.no_extras:
        extras = {}
        other = "eager" if args.mode == "graph" else "graph"
        r2 = make_runner(wl, other, tp_group)
        extras[f"{other}_ms_per_step"] = timed_steps(r2, args.steps, args.warmup, 1) / args.steps * 1e3
        extras["backend_forward_eager_ms_per_step"] = timed_steps(wl.backend_step, args.steps, args.warmup, 1) / args.steps * 1e3
        for alt in ("contig", "radix", "ragged", "perm"):
            if alt == args.table:
                continue
            wl.set_table(alt)
            r3 = make_runner(wl, args.mode, tp_group)
            ms = timed_steps(r3, args.steps, args.warmup, 1) / args.steps * 1e3
            extras[f"{alt}_table_ms_per_step"] = ms
            extras[f"{alt}_table_tokens_per_s"] = args.batch / ms * 1e3
            # the same roofline arithmetic as the headline's, on this table's own bytes (whole step / 28
            # launches, launch boundaries included); the radix table counts a shared cache row once
            alt_b, _ = wl.algorithmic_bytes_per_layer(shared_rows_once=alt == "radix")
            extras[f"{alt}_table_algorithmic_bytes_per_launch"] = alt_b
            extras[f"{alt}_table_frac_of_hbm_peak"] = alt_b * args.layers / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            if alt == "radix":
                extras["radix_table_note"] = ("groups of 8 requests share the cache rows of their first half: shared rows counted "
                                              "once in the bytes (they are read from HBM once and served from cache after that)")
            if alt == "ragged":
                extras["ragged_mean_seq"] = float(wl.local_seq.float().mean().item())
        wl.set_table(args.table)
        try:
            extras.update(extend_extras(dev))
        except Exception as exc:  # a side measurement: it must not take the headline line with it
            extras["extend_error"] = repr(exc)
        try:
            # larger decode launches of the same kernel (the launch's fixed costs weigh less): per layer call,
            # KV write and split merge inside, 4 layers on distinct slabs replayed from a HIP graph
            from tools import bench_decode_shapes as shapes

            del wl
            torch.cuda.empty_cache()
            extras["larger_decode_launches"] = {
                f"{model} shapes, batch {b} x {s_} tokens": shapes.run(model, b, s_, 4, 10, dev)
                for model, b, s_ in (("1.5b", 64, 8192), ("1.5b", 256, 2048), ("7b", 128, 4096))
            }
        except Exception as exc:
            extras["larger_decode_launches_error"] = repr(exc)
        try:
            # BASELINE configs[0] and [1] as single layer calls: one request (and eight) on the 1.5B and the
            # 0.5B shapes, microseconds per KV write + decode launch incl. the launch boundary, graph replay
            from tools import bench_latency as latency

            torch.cuda.empty_cache()
            extras["small_batch_decode_us_per_layer"] = {
                f"{name} shapes, batch {b} x {s_} tokens": round(latency.measure(hq, hkv, d, b, s_, 0, dev)[0], 2)
                for name, (hq, hkv, d) in (("1.5b", (12, 2, 128)), ("0.5b", (14, 2, 64)))
                for b in (1, 8) for s_ in (128, 2048, 4095)
            }
        except Exception as exc:
            extras["small_batch_decode_error"] = repr(exc)
        result["extras"] = extras

    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        result["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)

    if rank == 0:
        json_out.write(json.dumps(result) + "\n")
        json_out.flush()
    if world > 1:
        torch.distributed.barrier()
        pkg.tp_runtime.release_capi_collectives()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
