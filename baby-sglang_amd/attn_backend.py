"""MI355X paged-attention backend behind the reference's backend interface.

Drop-in for the reference's ``SimpleAttentionBackend`` (reference:
attn_backend.py:6-149): same class name, same static ``forward(q, k, v,
attention_layer, forward_batch, save_kv_cache=True)``, same result
``[T, n_q_heads * head_dim]`` in q's dtype.  What runs underneath is the HIP
library (KV scatter, split-KV paged decode, paged extend) instead of the
per-request gather / repeat_interleave / SDPA loop, and there is no host read
of a device scalar anywhere on the path.

Behaviour kept from the reference, deliberately:
  * new K/V rows are written before attention reads the cache
    (attn_backend.py:20-23), and skipped when ``save_kv_cache`` is false or
    ``k``/``v`` is ``None``;
  * the softmax scale is ``head_dim ** -0.5``: the reference reads
    ``attention_layer.scaling`` (attn_backend.py:18) but never passes it to
    SDPA (attn_backend.py:94-101, :141-143);
  * kv head of q head h is ``h // (n_q_heads // n_kv_heads)``
    (repeat_interleave, attn_backend.py:63-66, :132-135);
  * extend rows see page-table positions ``0 .. prefix_len + row``
    (attn_backend.py:74-92); requests with no extend tokens own no rows.
"""

from __future__ import annotations

import collections
import os
import threading
import weakref

import torch

from . import _hip
from .forward_batch import SimplifiedForwardBatch


class _PerStream:
    """One buffer per (device, stream), at most ``keep`` of them: the least recently used entry goes
    when a new stream shows up (a serving process has a main stream, a worker's side stream and
    perhaps a capture stream; a process that keeps making streams must not keep a buffer for each).
    A dropped buffer's memory goes back to torch's allocator, which hands it out again in the order
    of the stream it was allocated on -- the stream whose calls used it."""

    def __init__(self, keep: int = 8):
        self.keep = keep
        self._items = collections.OrderedDict()
        self._lock = threading.Lock()

    def get(self, key, n_bytes: int, make):
        with self._lock:
            buf = self._items.get(key)
            if buf is not None and buf.numel() >= n_bytes:
                self._items.move_to_end(key)
                return buf
        buf = make(n_bytes)
        with self._lock:
            self._items[key] = buf
            self._items.move_to_end(key)
            while len(self._items) > self.keep:
                self._items.popitem(last=False)
        return buf

    def __len__(self):
        return len(self._items)

    def clear(self):
        with self._lock:
            self._items.clear()


class SimpleAttentionBackend:
    # decode kernel selection: _hip.DECODE_AUTO picks the matrix-core kernel
    # when it covers the shape, the generic kernel otherwise
    decode_variant = _hip.DECODE_AUTO
    # A decode step's new K/V rows can be stored by the decode kernel itself instead of a scatter
    # launch ahead of it (2.8 us per layer).  That is only right when every slot in out_cache_loc is
    # fresh: named by its own request's page-table row alone.  Otherwise a request whose row names
    # ANOTHER request's new slot would read that row while it is being written; the reference
    # writes all new rows before any request reads (attn_backend.py:20-23).  So the kernel only
    # stores the rows for batches whose producer says so (SimplifiedForwardBatch.private_decode_slots;
    # this package's ServingLoop does), or when the integrator vouches for every batch with
    # ``assume_private_decode_slots`` (the reference's scheduler gives every decode token a fresh
    # slot, scheduler.py:459-490, but its batch descriptor cannot say so).  ``fuse_decode_kv_write``
    # = False keeps the write a launch of its own whatever the batch says.
    fuse_decode_kv_write = True
    assume_private_decode_slots = False
    # Debug aid for a producer's promise (BABYSGL_CHECK_FUSED_KV=1, or set the attribute): before
    # every decode call that fuses the write, check on the device -- with a host read, so not for
    # production -- that no slot in out_cache_loc is named by ANOTHER request's visible page-table
    # entries (or twice in out_cache_loc); a producer that breaks its word gets a ValueError
    # instead of silently wrong numbers.
    check_fused_kv_invariant = os.getenv("BABYSGL_CHECK_FUSED_KV", "0") not in ("", "0")
    # one zero-initialised decode workspace per (device, stream): calls on
    # different streams may overlap (overlap_worker.py:104-127) and must not
    # share split-merge tickets; likewise one extend workspace per stream
    _workspaces = _PerStream()
    _prefill_workspaces = _PerStream()

    @staticmethod
    def forward(
        q: torch.Tensor,
        k: torch.Tensor,
        v: torch.Tensor,
        attention_layer,
        forward_batch: SimplifiedForwardBatch,
        save_kv_cache: bool = True,
    ) -> torch.Tensor:
        layer_id = attention_layer.layer_id
        pool = forward_batch.token_to_kv_pool
        write_kv = save_kv_cache and k is not None and v is not None

        if forward_batch.is_prefill:
            if write_kv:
                pool.set_kv_buffer(layer_id, forward_batch.out_cache_loc, k, v)
            return SimpleAttentionBackend._prefill_attention(q, forward_batch, layer_id)
        if write_kv and k.dtype != pool.dtype:
            # the pool casts on write (memory_pool.py:163-165); keep that path separate
            pool.set_kv_buffer(layer_id, forward_batch.out_cache_loc, k, v)
            write_kv = False
        return SimpleAttentionBackend._decode_attention(
            q, k if write_kv else None, v if write_kv else None, forward_batch, layer_id
        )

    # (batch, heads, dim, columns) -> (n_splits, workspace bytes): two library calls per shape, not per layer
    _plans = {}

    @staticmethod
    def _decode_workspace(q, n_q_heads: int, n_kv_heads: int, max_ctx: int):
        batch, dim = q.shape[0], q.shape[-1]
        shape = (batch, n_q_heads, n_kv_heads, dim, max_ctx)
        plan = SimpleAttentionBackend._plans.get(shape)
        if plan is None:
            n_splits = _hip.decode_num_splits(batch, n_kv_heads, max_ctx)
            plan = (n_splits, _hip.decode_workspace_bytes(batch, n_q_heads, dim, n_splits))
            if len(SimpleAttentionBackend._plans) > 4096:
                SimpleAttentionBackend._plans.clear()
            SimpleAttentionBackend._plans[shape] = plan
        n_splits, need = plan
        stream = torch.cuda.current_stream(q.device).cuda_stream
        ws = SimpleAttentionBackend._workspaces.get(
            (q.device, stream), need, lambda n: _hip.new_decode_workspace(max(n, 1 << 20), q.device))
        return n_splits, ws, (need, stream)

    @staticmethod
    def _check_fused_kv(forward_batch: SimplifiedForwardBatch) -> None:
        table = forward_batch.req_to_token_pool.req_to_token
        rows = forward_batch.req_pool_indices.long()
        seq = forward_batch.seq_lens.long()
        loc = forward_batch.out_cache_loc.long()
        width = int(seq.max().item()) if seq.numel() else 0
        visible = table[rows, :width].long()  # [B, width]
        live = torch.arange(width, device=table.device)[None, :] < seq[:, None]
        # hits[i, j]: request j's visible entries name request i's new slot
        hits = ((visible[None, :, :] == loc[:, None, None]) & live[None, :, :]).any(dim=-1)
        hits.fill_diagonal_(False)
        dup = loc.numel() != loc.unique().numel()
        if dup or bool(hits.any().item()):
            bad = hits.nonzero()[:4].tolist()
            raise ValueError(
                "fuse_decode_kv_write: a slot in out_cache_loc is named by another request's page-table row "
                f"(new slot of request i, named by request j: {bad}; duplicate slots: {dup}); such a batch "
                "must not set private_decode_slots"
            )

    # (pool, layer, batch, q heads, dtype, page table) -> a filled-in _hip.DecodeCall: everything
    # about a layer's decode call that does not change from step to step
    _calls = {}

    @staticmethod
    def _decode_call(q, pool, table, layer_id: int):
        """The call descriptor of this (layer, batch shape): slabs, page table, head counts,
        splits and variant filled in once."""
        batch, hq, dim = q.shape
        # (one descriptor per calling thread: a worker thread and the main thread never fill the same struct)
        key = (id(pool), layer_id, batch, hq, q.dtype, table.data_ptr(), threading.get_ident())
        hit = SimpleAttentionBackend._calls.get(key)
        if hit is not None and hit[1] is pool and hit[2] is table:
            return hit
        if pool.store_dtype != pool.dtype:
            return None
        k_slab, v_slab = pool.get_kv_buffer(layer_id)
        if k_slab.dtype != q.dtype or k_slab.stride(0) != v_slab.stride(0) or k_slab.shape[2] != dim:
            return None
        if table.dtype != torch.int32 or table.stride(1) != 1:
            return None
        hkv = k_slab.shape[1]
        n_splits = _hip.decode_num_splits(batch, hkv, table.shape[1])
        c = _hip.DecodeCall()
        c.out_row_stride = hq * dim
        c.q_row_stride = hq * dim
        c.k_slab, c.v_slab = k_slab.data_ptr(), v_slab.data_ptr()
        c.slab_row_stride, c.n_slots = k_slab.stride(0), k_slab.shape[0]
        c.req_to_token = table.data_ptr()
        c.table_row_stride, c.table_rows, c.table_cols = table.stride(0), table.shape[0], table.shape[1]
        c.batch, c.n_q_heads, c.n_kv_heads, c.head_dim = batch, hq, hkv, dim
        c.scale = dim**-0.5
        c.dtype = c.out_dtype = _hip._DTYPE_CODE[q.dtype]
        c.n_splits = n_splits
        c.workspace_bytes = _hip.decode_workspace_bytes(batch, hq, dim, n_splits)
        if len(SimpleAttentionBackend._calls) > 4096:
            SimpleAttentionBackend._calls.clear()
        # [descriptor, pool, table, pointer to the descriptor, the step whose pointers it holds]
        rec = [c, pool, table, _hip.ctypes.byref(c), None]
        SimpleAttentionBackend._calls[key] = rec
        return rec

    @staticmethod
    def _decode_fast(q, k, v, forward_batch: SimplifiedForwardBatch, layer_id: int):
        """One foreign call per layer: the descriptor of `_decode_call` with the step's pointers
        put in.  Returns None when the call is not of the plain shape (the general wrapper then
        takes it): q dense, int64 index arrays, int32 slots, new rows in q's dtype."""
        pool = forward_batch.token_to_kv_pool
        table = forward_batch.req_to_token_pool.req_to_token
        if not q.is_cuda or not q.is_contiguous() or q.dim() != 3 or q.dtype not in _hip._DTYPE_CODE:
            return None
        rec = SimpleAttentionBackend._decode_call(q, pool, table, layer_id)
        if rec is None:
            return None
        c = rec[0]
        # what the batch brings, looked up once per step (the same object comes back for every layer)
        # (the pointers are only good for the tensors they were taken from: a caller that reuses one
        # batch object and binds new index tensors to it gets a new look-up, three identity compares)
        step = getattr(forward_batch, "_bsgl_step", None)
        stream = _hip.current_stream_handle(q.device.index)
        rows, seq, loc = forward_batch.req_pool_indices, forward_batch.seq_lens, forward_batch.out_cache_loc
        if step is None or step[0] != stream or step[6] is not rows or step[7] is not seq or step[8] is not loc:
            if rows.dtype != torch.int64 or seq.dtype != torch.int64 or not rows.is_contiguous() or not seq.is_contiguous():
                return None
            if rows.shape[0] != c.batch or seq.shape[0] != c.batch:
                return None
            loc_ptr = 0
            if loc is not None and loc.dtype == torch.int32 and loc.is_contiguous() and loc.shape[0] == c.batch:
                loc_ptr = loc.data_ptr()
            ws = SimpleAttentionBackend._workspaces.get(
                (q.device, stream), c.workspace_bytes, lambda n: _hip.new_decode_workspace(max(n, 1 << 20), q.device))
            step = (stream, rows.data_ptr(), seq.data_ptr(), loc_ptr, ws.data_ptr(), ws, rows, seq, loc)
            try:
                forward_batch._bsgl_step = step
            except AttributeError:  # (a batch object that takes no new attributes)
                pass
        variant = SimpleAttentionBackend.decode_variant
        row = c.out_row_stride
        if k is not None:
            if step[3] == 0 or k.dtype != q.dtype or v.dtype != q.dtype:
                return None
            ks, vs, shape = k.stride(), v.stride(), k.shape
            # ([batch, kv heads, dim] with dense rows; v is a column slice of the projection's output, qwen2.py:288)
            if len(shape) != 3 or shape != v.shape or ks[1:] != (shape[2], 1) or vs[1:] != (shape[2], 1) \
                    or shape[0] != c.batch or shape[1] != c.n_kv_heads or shape[2] != c.head_dim:
                return None
            if SimpleAttentionBackend._fuses(forward_batch):
                variant |= _hip.DECODE_FUSE_KV_WRITE
            out = torch.empty((shape[0], row), dtype=q.dtype, device=q.device)
            # the ten leading fields of the descriptor in one write: out, its row stride, lse, q, its
            # row stride, the new rows with their strides, their slots
            _hip.DECODE_CALL_HEAD.pack_into(c, 0, out.data_ptr(), row, 0, q.data_ptr(), row,
                                            k.data_ptr(), v.data_ptr(), ks[0], vs[0], step[3])
        else:
            out = torch.empty((q.shape[0], row), dtype=q.dtype, device=q.device)
            _hip.DECODE_CALL_HEAD.pack_into(c, 0, out.data_ptr(), row, 0, q.data_ptr(), row, 0, 0, 0, 0, 0)
        if rec[4] is not step:
            c.req_pool_indices, c.seq_lens = step[1], step[2]
            c.workspace = step[4] if c.workspace_bytes else None
            c.stream = stream
            # (a scheduler that knows its longest request says so on the batch; the reference's does not)
            c.max_seq_len = getattr(forward_batch, "max_seq_len", 0) or 0
            rec[4] = step
        c.variant = variant
        rc = _hip.lib.bsgl_paged_decode_call(rec[3])
        if rc:
            _hip._check(rc, "bsgl_paged_decode_call")
        return out

    @staticmethod
    def _fuses(forward_batch) -> bool:
        """Whether the decode kernel may store this batch's new rows itself."""
        cls = SimpleAttentionBackend
        return cls.fuse_decode_kv_write and (cls.assume_private_decode_slots or getattr(forward_batch, "private_decode_slots", False))

    @staticmethod
    def _decode_attention(q, k, v, forward_batch: SimplifiedForwardBatch, layer_id: int):
        fuses = k is not None and SimpleAttentionBackend._fuses(forward_batch)
        if fuses and SimpleAttentionBackend.check_fused_kv_invariant:
            SimpleAttentionBackend._check_fused_kv(forward_batch)
        out = SimpleAttentionBackend._decode_fast(q, k, v, forward_batch, layer_id)
        if out is not None:
            return out
        pool = forward_batch.token_to_kv_pool
        if pool.store_dtype != pool.dtype:
            raise NotImplementedError("float8 KV cache write is not supported in baby-sglang MVP")
        k_slab, v_slab = pool.get_kv_buffer(layer_id)
        table = forward_batch.req_to_token_pool.req_to_token
        n_splits, ws, plan = SimpleAttentionBackend._decode_workspace(q, q.shape[1], k_slab.shape[1], table.shape[1])
        return _hip.paged_decode(
            q,
            k_slab,
            v_slab,
            table,
            forward_batch.req_pool_indices,
            forward_batch.seq_lens,
            n_splits=n_splits,
            variant=SimpleAttentionBackend.decode_variant,
            workspace=ws,
            k_new=k,
            v_new=v,
            out_cache_loc=forward_batch.out_cache_loc if k is not None else None,
            fuse_kv_write=fuses,
            _plan=plan,
        )

    @staticmethod
    def _prefill_attention(q, forward_batch: SimplifiedForwardBatch, layer_id: int):
        k_slab, v_slab = forward_batch.token_to_kv_pool.get_kv_buffer(layer_id)
        workspace = None
        list_ready = False
        if q.is_cuda and q.dim() == 3:
            # the extend launch's work list lives in a per-stream buffer instead of a fresh allocation per layer call
            need = _hip.prefill_workspace_bytes(int(forward_batch.req_pool_indices.shape[0]), q.shape[0], q.shape[1], q.shape[2])
            stream = _hip.current_stream_handle(q.device.index)
            workspace = SimpleAttentionBackend._prefill_workspaces.get(
                (q.device, stream), need, lambda n: torch.empty(max(n, 1 << 16), dtype=torch.uint8, device=q.device))
            # ... and is made once per batch, not once per layer: the layers of a model call the backend with
            # ONE batch descriptor (qwen2.py:292-315), so from the second layer on the list of the first is
            # still in the buffer.  Whose list it holds is kept ON the buffer, by identity (weak references:
            # the note must not keep a batch alive): the batch object, its extended_lens tensor, then the
            # shape of q, the dtype, the kv heads and the knobs' epoch -- and the layer that used it last: only
            # a LATER layer takes the list over, so a caller that keeps one batch object and rewrites its
            # tensors in place from step to step gets a new list with the first layer of every forward.
            ext = forward_batch.extended_lens
            facts = (tuple(q.shape), q.dtype, k_slab.shape[1], _hip.knob_epoch)
            held = getattr(workspace, "_bsgl_list_of", None)
            list_ready = (held is not None and held[0]() is forward_batch and held[1]() is ext and held[2] == facts
                          and layer_id > held[3])
            workspace._bsgl_list_of = (weakref.ref(forward_batch), weakref.ref(ext), facts, layer_id)
        return _hip.paged_prefill(
            q,
            k_slab,
            v_slab,
            forward_batch.req_to_token_pool.req_to_token,
            forward_batch.req_pool_indices,
            forward_batch.seq_lens,
            forward_batch.prefix_lens,
            forward_batch.extended_lens,
            workspace=workspace,
            list_ready=list_ready,
        )
