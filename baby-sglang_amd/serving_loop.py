"""Continuous batching over the paged pools with radix-cache prefix reuse.

Behavioural counterpart of the reference's ``Scheduler.run_batch`` loop (reference:
scheduler.py:107-168 loop, :242-301 prefill admission, :303-369 decode selection and
retraction, :381-455 prefill batch, :457-507 decode batch, :553-587 result handling;
radix_cache.py:141-207 caching of finished / running requests): the same admission
policy, the same FIFO slot and row allocation, the same prefix-cache bookkeeping, so
that an identical request stream produces identical ``SimplifiedForwardBatch``es --
request rows, sequence lengths, ``out_cache_loc`` and page-table rows -- which is what
the golden traces pin.

What is arranged differently, for a GPU that should not wait on the host:
  * every request keeps a host copy of its page-table row, so caching a request in the
    prefix tree needs no device read;
  * a decode step appends its new slots to the page table with one indexed device
    write instead of one scalar write per request;
  * greedy sampling is one ``argmax`` over the batch and one device-to-host copy per
    step (the ids are needed on the host for finish checks and cache keys).

Overlap mode (``run_batch_overlap``; reference: scheduler.py:599-761 with
overlap_worker.py:27-160): the next batch is selected, built and launched while the
previous one is still running, and a batch's sampled ids are looked at one step late.
Two forms.  The reference's: a worker thread runs the model and the sampling on a second
stream while the loop's own thread selects and builds the next batch on its stream
(``overlap_worker=True``, the default on a GPU; ``_ForwardWorker`` below).  The batch's
metadata copies are ordered in front of the worker's forward by an event the worker's
stream waits for (the reference makes the host wait for the scheduler's stream instead,
overlap_worker.py:83), the sampled ids stay on the device, and the loop looks at them one
step late behind the event of their device-to-host copy (overlap_worker.py:155-160).
Where the reference feeds the next step negative placeholder ids that its worker
resolves from a "future token" map (overlap_worker.py:39-44, :114-121), the ids here
stay in a per-page-table-row buffer that the worker gathers a decode batch's inputs
from: the same hand-over, keyed by the row instead of by a ring position.  The other
form (``overlap_worker=False``, and always on the CPU, where the reference's worker
refuses to run, overlap_worker.py:50-53) does the same on ONE thread and ONE stream:
every host-to-device copy of batch metadata goes through
pinned staging memory without blocking, and the only wait of a step is on the event
behind the PREVIOUS step's device-to-host copy of its ids.  As in the reference, a
request that stops on its eos token is found out a step late, so it runs one decode step
too many whose slot is handed back (scheduler.py:740-746); a request that stops on its
token budget is not scheduled again, the count of tokens in flight being known.

Chunked prefill (not in the reference, whose scheduler admits whole prompts,
scheduler.py:286-299, although its backend's extend semantics allow it,
attn_backend.py:43-59): with ``chunked_prefill_size = N`` an admitted prefill batch of
more than N new tokens goes through the model as several extend calls of at most N
tokens each.  Admission, slots and page-table rows are exactly those of the unchunked
batch -- every row is written up front -- and a chunk is an extend over
``prefix_len + tokens already run``; a request cut by a chunk border continues in the
next call, and a request is sampled from the call that holds its last prompt token.
"""

from __future__ import annotations

import os
import queue
import threading
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import torch

from .forward_batch import SimplifiedForwardBatch
from .memory_pool import MHATokenToKVPool, ReqToTokenPool
from .prefix_cache import PrefixCache, PrefixNode
from .sampling import SamplingParams, sample_next_ids


@dataclass(eq=False)
class Sequence_:
    """One request in flight (counterpart of request.py:9-46)."""

    input_ids: List[int]
    max_new_tokens: int
    eos_id: int
    output_ids: List[int] = field(default_factory=list)
    finished: bool = False
    row: Optional[int] = None  # page-table row
    seq_len: int = 0
    table: List[int] = field(default_factory=list)  # host copy of the page-table row, seq_len entries
    prefix_slots: torch.Tensor = field(default_factory=lambda: torch.empty(0, dtype=torch.int32))
    last_node: Optional[PrefixNode] = None
    num_cached: int = 0
    in_flight: int = 0  # steps launched whose ids have not been looked at yet (overlap mode)

    def reset(self) -> None:
        self.output_ids, self.finished, self.row, self.seq_len, self.table = [], False, None, 0, []
        self.prefix_slots, self.last_node, self.num_cached = torch.empty(0, dtype=torch.int32), None, 0


def chunk_members(n_new: Sequence[int], limit: int):
    """Cut a batch whose request i brings ``n_new[i]`` tokens into calls of at most
    ``limit`` tokens: yields, per call, ``[(request index, tokens taken), ...]`` in batch
    order; a request cut by a call's border continues at the head of the next call."""
    done = [0] * len(n_new)
    i = 0
    while i < len(n_new):
        room, members = limit, []
        j = i
        while j < len(n_new) and room > 0:
            take = min(room, n_new[j] - done[j])
            members.append((j, take))
            room -= take
            done[j] += take
            if done[j] < n_new[j]:
                break
            j += 1
        while i < len(n_new) and done[i] == n_new[i]:
            i += 1
        yield members


class _ForwardWorker:
    """The model's forward pass and the sampling on a thread and a HIP stream of their own
    (reference: overlap_worker.py:27-160).  ``submit`` hands over a built batch and returns at
    once; ``resolve`` gives the oldest submitted batch's ids on the host, waiting only for the
    event behind their device-to-host copy.  A failure inside the worker is re-raised by
    ``resolve``."""

    def __init__(self, loop: "ServingLoop"):
        self.loop = loop
        self.stream = torch.cuda.Stream(device=loop.device)
        self.inbox: "queue.Queue" = queue.Queue()
        self.outbox: "queue.Queue" = queue.Queue()
        self.thread = threading.Thread(target=self._main, name="bsgl-forward-worker", daemon=True)
        self.thread.start()

    def submit(self, mode: str, fb: SimplifiedForwardBatch, batch, logits_rows, ids_host) -> None:
        # everything the batch needs was enqueued on the caller's stream: the forward stream waits for
        # it on the device (the reference waits on the host instead, overlap_worker.py:83)
        # (the batch's tensors were allocated on the caller's stream: the allocator must not hand their
        # memory out again before the forward stream is done with them, whoever drops them first)
        for t in list(vars(fb).values()) + [logits_rows]:
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(self.stream)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.loop.device))
        self.inbox.put((mode, fb, batch, logits_rows, ids_host, ready))

    def resolve(self):
        item = self.outbox.get()
        if isinstance(item, BaseException):
            raise item
        done, ids_host = item
        done.synchronize()
        return ids_host

    def close(self) -> None:
        self.inbox.put(None)
        self.thread.join(timeout=30)
        self.stream.synchronize()

    def _main(self) -> None:
        loop = self.loop
        torch.cuda.set_device(loop.device)
        with torch.cuda.stream(self.stream), torch.no_grad():
            while True:
                work = self.inbox.get()
                if work is None:
                    return
                try:
                    mode, fb, batch, logits_rows, ids_host, ready = work
                    self.stream.wait_event(ready)
                    if mode == "decode":
                        # the ids the requests' last steps left on the device (overlap_worker.py:114-121)
                        fb.input_ids = loop._last_ids[fb.req_pool_indices]
                    next_ids = loop._pick(loop._model_logits(fb, logits_rows), batch)
                    loop._last_ids[fb.req_pool_indices] = next_ids
                    ids_host.copy_(next_ids, non_blocking=True)
                    done = torch.cuda.Event()
                    done.record(self.stream)
                    self.outbox.put((done, ids_host))
                except BaseException as exc:  # noqa: BLE001  (handed to the loop's thread by resolve)
                    self.outbox.put(exc)


class ServingLoop:
    def __init__(self, model, max_total_tokens: Optional[int] = None, max_requests: Optional[int] = None,
                 chunked_prefill_size: Optional[int] = None, overlap_worker: Optional[bool] = None):
        self.model = model
        if chunked_prefill_size is not None and chunked_prefill_size < 1:
            raise ValueError("chunked_prefill_size must be positive")
        self.chunked_prefill_size = chunked_prefill_size
        self.max_total_tokens = int(max_total_tokens or os.getenv("BABYSGL_MAX_TOTAL_TOKENS", "131072"))
        self.max_requests = int(max_requests or os.getenv("BABYSGL_MAX_REQUESTS", "32"))
        p = next(model.parameters())
        self.device = p.device
        cfg = model.config
        self.req_to_token_pool = ReqToTokenPool(self.max_requests, self.max_total_tokens // self.max_requests, str(self.device))
        self.token_to_kv_pool = MHATokenToKVPool(
            self.max_total_tokens, p.dtype, cfg.num_key_value_heads, cfg.hidden_size // cfg.num_attention_heads,
            cfg.num_hidden_layers, str(self.device),
        )  # fmt: skip
        self.tree_cache = PrefixCache()
        # admission estimates (scheduler.py:63-69)
        self.r_init, self.min_r = 0.5, 0.1
        self.decay_step = (self.r_init - self.min_r) / 50
        self.retract_decode_steps = 20
        self.clip_max_new_tokens_estimation = 512
        self.waiting: List[Sequence_] = []
        self.decoding: List[Sequence_] = []
        self.finished: List[Sequence_] = []
        self.est_new_token_ratio = self.r_init
        self.on_batch = None  # optional hook: on_batch(mode, forward_batch) before every model call
        # pinned staging for host-to-device copies that must not block (three arenas in rotation:
        # one per step being built, in flight, and just finished)
        self._staging = None
        self._staging_turn = 0
        self._staging_used = 0
        self._last_ids = None  # device: the id each page-table row's request produced last
        self._sampling = SamplingParams()
        self.generator: Optional[torch.Generator] = None  # for do_sample; None = the device's default generator
        self._ids_host = None  # pinned landing buffers of the per-step id copies, in the same rotation
        # overlap mode: the model on a worker thread and a second stream (None: wherever there is a GPU)
        self.overlap_worker = (self.device.type == "cuda") if overlap_worker is None else bool(overlap_worker)
        if self.overlap_worker and self.device.type != "cuda":
            raise NotImplementedError("the forward worker needs a GPU stream of its own (as the reference's, overlap_worker.py:50-53)")
        self._worker: Optional[_ForwardWorker] = None

    def reset(self) -> None:
        self.req_to_token_pool.clear()
        self.token_to_kv_pool.clear()
        self.tree_cache.reset()
        self.waiting, self.decoding, self.finished = [], [], []
        self.est_new_token_ratio = self.r_init

    # ---------------------------------------------------------------------------- copies
    def _next_staging(self) -> None:
        if self.device.type != "cuda":
            return
        if self._staging is None:
            n = self.max_requests * self.req_to_token_pool.max_context_len // 2 + (1 << 16)
            self._staging = [torch.empty(n, dtype=torch.int64).pin_memory() for _ in range(3)]
        self._staging_turn = (self._staging_turn + 1) % 3
        self._staging_used = 0

    def _h2d(self, data, dtype: torch.dtype) -> torch.Tensor:
        """``data`` (list or CPU tensor) as a device tensor of ``dtype``, without making the
        host wait for the stream: through this step's pinned arena when it fits."""
        t = data.to(dtype) if isinstance(data, torch.Tensor) else torch.tensor(data, dtype=dtype)
        if self.device.type != "cuda":
            return t.to(self.device)
        n64 = (t.numel() * t.element_size() + 7) // 8
        if self._staging is None or self._staging_used + n64 > self._staging[0].numel():
            return t.to(self.device)  # (blocking copy: correct, only slower)
        arena = self._staging[self._staging_turn]
        stage = arena[self._staging_used : self._staging_used + n64].view(torch.uint8)[: t.numel() * t.element_size()]
        stage = stage.view(dtype).view(t.shape)
        stage.copy_(t)
        self._staging_used += n64
        return stage.to(self.device, non_blocking=True)

    # ---------------------------------------------------------------------------- the loop
    def run_batch(self, requests: Sequence[torch.Tensor], sampling: Optional[SamplingParams] = None) -> List[List[int]]:
        sampling = sampling or SamplingParams()
        self._sampling = sampling
        self.waiting, self.decoding, self.finished = [], [], []
        self.est_new_token_ratio = self.r_init
        order = []
        for ids in requests:
            seq = Sequence_(input_ids=[int(t) for t in ids.tolist()],
                            max_new_tokens=min(sampling.max_new_tokens, self.max_total_tokens), eos_id=sampling.eos_id)
            self.waiting.append(seq)
            order.append(seq)
        while len(self.finished) < len(order):
            batch, mode = self._select()
            if mode == "error":
                raise RuntimeError("Insufficient memory to process any requests")
            if mode == "prefill":
                next_ids = self._run_prefill(batch)
            else:
                fb = self._decode_batch(batch)
                next_ids = self._pick(self._forward(mode, fb), batch).tolist()
            self._finish_step(next_ids, batch, mode)
        return [s.output_ids for s in order]

    def run_batch_overlap(self, requests: Sequence[torch.Tensor], sampling: Optional[SamplingParams] = None) -> List[List[int]]:
        """``run_batch`` with a batch's ids looked at one step late, so that selecting,
        building and launching the next batch overlaps the current one on the GPU.  Same
        output ids as ``run_batch``."""
        sampling = sampling or SamplingParams()
        # (the repetition penalty is left out in this mode, as in the reference, scheduler.py:632:
        # it needs the ids of the step in flight)
        self._sampling = SamplingParams(**{**sampling.__dict__, "repetition_penalty": 1.0})
        if self.chunked_prefill_size is not None:
            raise NotImplementedError("overlap mode runs whole-prompt prefill batches")
        self.waiting, self.decoding, self.finished = [], [], []
        self.est_new_token_ratio = self.r_init
        if self._last_ids is None:
            self._last_ids = torch.zeros(self.max_requests, dtype=torch.long, device=self.device)
        order = []
        for ids in requests:
            seq = Sequence_(input_ids=[int(t) for t in ids.tolist()],
                            max_new_tokens=min(sampling.max_new_tokens, self.max_total_tokens), eos_id=sampling.eos_id)
            self.waiting.append(seq)
            order.append(seq)
        if self.overlap_worker:
            if self._worker is None:
                self._worker = _ForwardWorker(self)
            return self._run_overlap_worker(order)
        pending = None  # (batch, mode, ids on the host, copy-done event, new slots on the host)
        while len(self.finished) < len(order) or pending is not None:
            batch, mode = self._select_overlap(pending is not None)
            if mode == "error" and pending is None:
                raise RuntimeError("Insufficient memory to process any requests")
            submitted = None
            if batch:
                self._next_staging()
                if mode == "prefill":
                    fb = self._prefill_batch(batch)
                    n_new = [len(s.input_ids) - s.num_cached for s in batch]
                    ends = self._h2d(torch.cumsum(torch.tensor(n_new), dim=0) - 1, torch.long)
                    next_ids = self._pick(self._forward(mode, fb, ends), batch)
                    host_loc = None
                    for s in batch:  # decodable from the next step on (scheduler.py:676-681)
                        self.waiting.remove(s)
                        self.decoding.append(s)
                else:
                    fb = self._decode_batch(batch, ids_on_device=True)
                    next_ids = self._pick(self._forward(mode, fb), batch)
                    host_loc = [s.table[s.seq_len - 1] for s in batch]
                self._last_ids[fb.req_pool_indices] = next_ids
                for s in batch:
                    s.in_flight += 1
                if self.device.type == "cuda":
                    if self._ids_host is None:
                        self._ids_host = [torch.empty(self.max_requests, dtype=torch.long).pin_memory() for _ in range(3)]
                    ids_host = self._ids_host[self._staging_turn][: len(batch)]
                    ids_host.copy_(next_ids, non_blocking=True)
                    done = torch.cuda.Event()
                    done.record()
                else:
                    ids_host, done = next_ids.clone(), None
                submitted = (batch, mode, ids_host, done, host_loc)
            if pending is not None:
                self._commit_overlap(*pending)
            pending = submitted
        return [s.output_ids for s in order]

    def _run_overlap_worker(self, order: List[Sequence_]) -> List[List[int]]:
        """The overlap loop with the forward pass on the worker's thread and stream
        (scheduler.py:647-698): build and submit batch t + 1, then take in batch t's ids."""
        worker = self._worker
        if self._ids_host is None:
            self._ids_host = [torch.empty(self.max_requests, dtype=torch.long).pin_memory() for _ in range(3)]
        pending = None  # (batch, mode, new slots on the host)
        try:
            while len(self.finished) < len(order) or pending is not None:
                batch, mode = self._select_overlap(pending is not None)
                if mode == "error" and pending is None:
                    raise RuntimeError("Insufficient memory to process any requests")
                submitted = None
                if batch:
                    self._next_staging()
                    rows_for_logits = None
                    if mode == "prefill":
                        fb = self._prefill_batch(batch)
                        n_new = [len(s.input_ids) - s.num_cached for s in batch]
                        rows_for_logits = self._h2d(torch.cumsum(torch.tensor(n_new), dim=0) - 1, torch.long)
                        host_loc = None
                        for s in batch:  # decodable from the next step on (scheduler.py:676-681)
                            self.waiting.remove(s)
                            self.decoding.append(s)
                    else:
                        fb = self._decode_batch(batch, ids_on_device=True, gather_ids=False)
                        host_loc = [s.table[s.seq_len - 1] for s in batch]
                    if self.on_batch is not None:
                        self.on_batch(mode, fb)
                    for s in batch:
                        s.in_flight += 1
                    worker.submit(mode, fb, batch, rows_for_logits, self._ids_host[self._staging_turn][: len(batch)])
                    submitted = (batch, mode, host_loc)
                if pending is not None:
                    p_batch, p_mode, p_loc = pending
                    self._commit_overlap(p_batch, p_mode, worker.resolve(), None, p_loc)
                pending = submitted
        except BaseException:
            # leave no batch behind on the worker: the next run starts from an empty queue
            if pending is not None:
                try:
                    worker.resolve()
                except BaseException:  # noqa: BLE001
                    pass
            raise
        return [s.output_ids for s in order]

    def close(self) -> None:
        """End the forward worker's thread (a loop that never ran in overlap mode has none)."""
        if self._worker is not None:
            self._worker.close()
            self._worker = None

    def _select_overlap(self, step_in_flight: bool) -> Tuple[List[Sequence_], str]:
        """``_select`` for a loop that has not seen the last step's ids yet: requests that have
        stopped, or whose tokens in flight already fill their budget, are left out of decode
        batches; a decode batch that could only be made by sending requests back to the queue is
        put off ("wait") until the step in flight has been taken in."""
        if self.waiting:
            admitted = self._admit_prefill()
            if admitted:
                return admitted, "prefill"
        live = [s for s in self.decoding if not s.finished and len(s.output_ids) + s.in_flight < s.max_new_tokens]
        if not live:
            return [], "error"
        if step_in_flight and len(live) > self.token_to_kv_pool.available_size() + self.tree_cache.evictable_size():
            return [], "wait"
        held, self.decoding = self.decoding, live
        try:
            batch = self._select_decode()
        finally:
            kept = set(map(id, self.decoding))
            live_ids = set(map(id, live))
            self.decoding = [s for s in held if id(s) in kept or id(s) not in live_ids]
        return (batch, "decode") if batch else ([], "error")

    def _commit_overlap(self, batch, mode, ids_host, done, host_loc) -> None:
        """Take in a finished step's ids (scheduler.py:700-761)."""
        if done is not None:
            done.synchronize()
        for i, (s, tok) in enumerate(zip(batch, ids_host.tolist())):
            s.in_flight -= 1
            if s.finished:
                # stopped on eos in the step before, found out too late: hand the extra slot back
                s.table.pop()
                s.seq_len -= 1
                self.token_to_kv_pool.free(torch.tensor([host_loc[i]], dtype=torch.int32))
                continue
            s.output_ids.append(int(tok))
            s.finished = len(s.output_ids) >= s.max_new_tokens or (s.eos_id != -1 and tok == s.eos_id)
            if s.finished and s.in_flight == 0:
                self._retire(s)
            elif not s.finished and mode == "prefill":
                self._cache_running(s)
        for s in batch:
            if s.finished and s.in_flight == 0 and s in self.decoding:
                self._retire(s)

    def _retire(self, s: Sequence_) -> None:
        if s in self.decoding:
            self.decoding.remove(s)
        if s not in self.finished:
            self._cache_finished(s)
            self.finished.append(s)

    def _pick(self, logits: torch.Tensor, batch: List[Sequence_]) -> torch.Tensor:
        """One id per request from its row of ``logits``: the whole batch at once, on the device
        (scheduler.py:523-546 does it request by request)."""
        sp = self._sampling
        prev = [s.output_ids for s in batch] if sp.repetition_penalty > 1.0 else None
        return sample_next_ids(logits, sp.do_sample, sp.temperature, sp.top_k, sp.top_p, prev, sp.repetition_penalty,
                               generator=self.generator)

    def _forward(self, mode: str, fb: SimplifiedForwardBatch, rows: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Logits of the batch's tokens, or of token ``rows`` only (a model that takes
        ``logits_rows`` then skips the vocabulary projection of every other token)."""
        if self.on_batch is not None:
            self.on_batch(mode, fb)
        return self._model_logits(fb, rows)

    def _model_logits(self, fb: SimplifiedForwardBatch, rows: Optional[torch.Tensor] = None) -> torch.Tensor:
        with torch.no_grad():
            if rows is not None and getattr(self.model, "supports_logits_rows", False):
                return self.model(fb.input_ids, fb.positions, fb, logits_rows=rows)
            logits = self.model(fb.input_ids, fb.positions, fb)
        return logits if rows is None else logits[rows]

    def _run_prefill(self, batch: List[Sequence_]) -> List[int]:
        """The admitted batch through the model, whole or in chunks; the token each
        request's last prompt position predicts."""
        fb = self._prefill_batch(batch)
        n_new = [len(s.input_ids) - s.num_cached for s in batch]
        limit = self.chunked_prefill_size
        if limit is None or sum(n_new) <= limit:
            ends = self._h2d(torch.cumsum(torch.tensor(n_new), dim=0) - 1, torch.long)
            return self._pick(self._forward("prefill", fb, ends), batch).tolist()
        next_ids: List[Optional[int]] = [None] * len(batch)
        first = [0]
        for n in n_new:
            first.append(first[-1] + n)  # offset of each request's new tokens in the batch's token list
        done = [0] * len(batch)  # new tokens of each request already run
        for members in chunk_members(n_new, limit):
            tok = torch.cat([torch.arange(first[m] + done[m], first[m] + done[m] + t) for m, t in members]).to(self.device)
            idx = torch.tensor([m for m, _ in members], device=self.device)
            prefix = torch.tensor([batch[m].num_cached + done[m] for m, _ in members], dtype=torch.long, device=self.device)
            takes = torch.tensor([t for _, t in members], dtype=torch.long, device=self.device)
            chunk = SimplifiedForwardBatch.create_prefill_batch(
                input_ids=fb.input_ids[tok], req_pool_indices=fb.req_pool_indices[idx], seq_lens=prefix + takes,
                out_cache_loc=fb.out_cache_loc[tok], req_to_token_pool=fb.req_to_token_pool,
                token_to_kv_pool=fb.token_to_kv_pool, prefix_lens=prefix,
            )  # fmt: skip
            ends, run = [], 0
            for _, t in members:
                run += t
                ends.append(run - 1)
            finished = [(m, e) for (m, t), e in zip(members, ends) if done[m] + t == n_new[m]]
            # (a call that completes no prompt still runs: it fills the cache for the next one)
            rows = self._h2d([e for _, e in finished] or [0], torch.long)
            logits = self._forward("prefill", chunk, rows)
            if finished:
                for (m, _), tok_id in zip(finished, self._pick(logits, [batch[m] for m, _ in finished]).tolist()):
                    next_ids[m] = tok_id
            for m, t in members:
                done[m] += t
        return next_ids

    # ---------------------------------------------------------------------------- selection
    def _select(self) -> Tuple[List[Sequence_], str]:
        if self.waiting:
            admitted = self._admit_prefill()
            if admitted:
                return admitted, "prefill"
        if self.decoding:
            batch = self._select_decode()
            if batch:
                return batch, "decode"
        return [], "error"

    def _admit_prefill(self) -> List[Sequence_]:
        tree = self.tree_cache
        for s in self.waiting:
            # never the whole prompt: at least one token must run through the model
            s.prefix_slots, s.last_node = tree.match_prefix(s.input_ids[:-1])
            s.num_cached = len(s.prefix_slots)
        self.waiting.sort(key=lambda s: len(s.prefix_slots), reverse=True)
        rows_free = self.req_to_token_pool.available_size()
        reserved = sum(
            min(int(self.est_new_token_ratio * (s.max_new_tokens - len(s.output_ids))), self.clip_max_new_tokens_estimation)
            for s in self.decoding
        )
        budget = self.token_to_kv_pool.available_size() - reserved
        admitted: List[Sequence_] = []
        for s in self.waiting:
            tree.inc_lock_ref(s.last_node)  # its own prefix does not count as evictable
            need = len(s.input_ids) - len(s.prefix_slots) + s.max_new_tokens
            fits = len(admitted) + 1 <= rows_free and need <= budget + tree.evictable_size()
            if fits:
                budget -= need
                tree.inc_lock_ref(s.last_node)  # held while the request runs
                admitted.append(s)
            tree.dec_lock_ref(s.last_node)
            if not fits:
                break
        return admitted

    def _select_decode(self) -> List[Sequence_]:
        pool, tree = self.token_to_kv_pool, self.tree_cache
        bs = len(self.decoding)
        if bs > pool.available_size():
            tree.evict(bs, pool.free)
        if bs <= pool.available_size():
            self.est_new_token_ratio = max(self.min_r, self.est_new_token_ratio - self.decay_step)
            return list(self.decoding)
        # still short: send the newest requests back to the queue (scheduler.py:322-369)
        while len(self.decoding) * self.retract_decode_steps > pool.available_size():
            if not self.decoding:
                break
            s = self.decoding.pop()
            pool.free(torch.tensor(s.table[s.num_cached : s.seq_len], dtype=torch.int32))
            self.req_to_token_pool.free(s.row)
            tree.dec_lock_ref(s.last_node)
            s.reset()
            self.waiting.append(s)
            short = max(0, len(self.decoding) * self.retract_decode_steps - pool.available_size())
            if short > 0:
                tree.evict(short, pool.free)
        assert self.decoding
        done = sum(len(s.output_ids) for s in self.decoding)
        cap = sum(s.max_new_tokens for s in self.decoding)
        self.est_new_token_ratio = min(1.0, (done + len(self.decoding) * self.retract_decode_steps) / cap)
        return list(self.decoding)

    # ---------------------------------------------------------------------------- batches
    def _alloc_slots(self, n: int) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
        """(host copy, device tensor) of the next n free slots, or (None, None): the pool's
        ``alloc`` (first n of the free list, memory_pool.py:68-74) with the copy to the device
        made through pinned memory."""
        pool = self.token_to_kv_pool
        if n > pool.available_size():
            return None, None
        host = pool.free_slots[:n].clone()
        pool.free_slots = pool.free_slots[n:]
        return host, self._h2d(host, torch.int32)

    def _prefill_batch(self, batch: List[Sequence_]) -> SimplifiedForwardBatch:
        r2t = self.req_to_token_pool
        rows = r2t.alloc(len(batch))
        if rows is None:
            raise RuntimeError("Failed to allocate request pool indices")
        new_ids = [t for s in batch for t in s.input_ids[s.num_cached :]]
        host_loc, out_cache_loc = self._alloc_slots(len(new_ids))
        if host_loc is None:
            self.tree_cache.evict(len(new_ids), self.token_to_kv_pool.free)
            host_loc, out_cache_loc = self._alloc_slots(len(new_ids))
            if host_loc is None:
                avail = self.token_to_kv_pool.available_size() + self.tree_cache.evictable_size()
                raise RuntimeError(f"Prefill OOM: need {len(new_ids)}, available {avail}")
        mapping = torch.zeros((len(batch), r2t.max_context_len), dtype=torch.int32)
        used = 0
        for i, s in enumerate(batch):
            n_new = len(s.input_ids) - s.num_cached
            s.row, s.seq_len = rows[i], len(s.input_ids)
            s.table = s.prefix_slots.tolist() + host_loc[used : used + n_new].tolist()
            mapping[i, : s.seq_len] = torch.tensor(s.table, dtype=torch.int32)
            used += n_new
        rows_t = self._h2d(rows, torch.long)
        r2t.write(rows_t, self._h2d(mapping, torch.int32))
        return SimplifiedForwardBatch.create_prefill_batch(
            input_ids=self._h2d(new_ids, torch.long),
            req_pool_indices=rows_t,
            seq_lens=self._h2d([s.seq_len for s in batch], torch.long),
            out_cache_loc=out_cache_loc,
            req_to_token_pool=r2t,
            token_to_kv_pool=self.token_to_kv_pool,
            prefix_lens=self._h2d([s.num_cached for s in batch], torch.long),
        )

    def _decode_batch(self, batch: List[Sequence_], ids_on_device: bool = False, gather_ids: bool = True) -> SimplifiedForwardBatch:
        """One new token per request.  ``ids_on_device``: the tokens to feed are the ones the
        requests' last steps left in the per-row id buffer (overlap mode), not ``output_ids[-1]``;
        with ``gather_ids=False`` they are not gathered here (the forward worker does it on its own
        stream, behind the step that produces them) and ``input_ids`` is a placeholder."""
        host_loc, new_slots = self._alloc_slots(len(batch))
        if host_loc is None:
            raise RuntimeError("Failed to allocate KV slots for decode")
        for s, slot in zip(batch, host_loc.tolist()):
            if s.row is None or (not ids_on_device and not s.output_ids):
                raise RuntimeError("request scheduled for decode without a page-table row or a token to feed")
            s.table.append(slot)  # position seq_len
            s.seq_len += 1
        rows_t = self._h2d([s.row for s in batch], torch.long)
        seq_t = self._h2d([s.seq_len for s in batch], torch.long)
        # one indexed write appends every request's new slot to its page-table row
        self.req_to_token_pool.req_to_token[rows_t, seq_t - 1] = new_slots
        if ids_on_device and not gather_ids:
            input_ids = torch.empty(len(batch), dtype=torch.long, device=self.device)
        elif ids_on_device:
            input_ids = self._last_ids[rows_t]
        else:
            input_ids = self._h2d([s.output_ids[-1] for s in batch], torch.long)
        fb = SimplifiedForwardBatch.create_decode_batch(
            input_ids=input_ids,
            req_pool_indices=rows_t,
            seq_lens=seq_t,
            out_cache_loc=new_slots,
            req_to_token_pool=self.req_to_token_pool,
            token_to_kv_pool=self.token_to_kv_pool,
        )
        fb.max_seq_len = max(s.seq_len for s in batch)  # (known here without asking the device)
        # every new slot comes straight from the free list and is written into its own request's row
        # only; a slot the prefix tree shares is never a decode step's new slot
        fb.private_decode_slots = True
        return fb

    # ---------------------------------------------------------------------------- results
    def _finish_step(self, next_ids: List[int], batch: List[Sequence_], mode: str) -> None:
        for s, tok in zip(batch, next_ids):
            s.output_ids.append(int(tok))
            s.finished = len(s.output_ids) >= s.max_new_tokens or (s.eos_id != -1 and tok == s.eos_id)
        for s in batch:
            if s.finished:
                self._cache_finished(s)
                self.finished.append(s)
            elif mode == "prefill":
                self._cache_running(s)
                self.decoding.append(s)
            if mode == "prefill" and s in self.waiting:
                self.waiting.remove(s)
        if mode == "decode":
            self.decoding = [s for s in self.decoding if not s.finished]

    def _cache_finished(self, s: Sequence_) -> None:
        """radix_cache.py:141-173: the tokens whose K/V are written (all but the last
        generated one) go into the tree; duplicates of already cached slots are freed."""
        tokens = (s.input_ids + s.output_ids)[:-1] if s.output_ids else list(s.input_ids)
        if tokens:
            slots = torch.tensor(s.table[: len(tokens)], dtype=torch.int32)
            had = self.tree_cache.insert(tokens, slots)
            if had > len(s.prefix_slots):
                self.token_to_kv_pool.free(slots[len(s.prefix_slots) : had])
        self.req_to_token_pool.free(s.row)
        self.tree_cache.dec_lock_ref(s.last_node)

    def _cache_running(self, s: Sequence_) -> None:
        """radix_cache.py:175-207: cache the prompt of a request that goes on decoding
        and point its page-table row at the tree's own slots for it."""
        tokens = list(s.input_ids)
        if not tokens:
            return
        tree = self.tree_cache
        slots = torch.tensor(s.table[: len(tokens)], dtype=torch.int32)
        had = tree.insert(tokens, slots)
        n_old = len(s.prefix_slots)
        if had > n_old:
            self.token_to_kv_pool.free(slots[n_old:had])
        canon, node = tree.match_prefix(tokens)
        assert len(canon) == len(tokens)
        if len(canon) > n_old:
            s.table[n_old : len(canon)] = canon[n_old:].tolist()
            self.req_to_token_pool.write((s.row, slice(n_old, len(canon))), self._h2d(canon[n_old:], torch.int32))
        tree.dec_lock_ref(s.last_node)
        tree.inc_lock_ref(node)
        s.prefix_slots, s.last_node = canon, node
