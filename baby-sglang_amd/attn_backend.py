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

import torch

from . import _hip
from .forward_batch import SimplifiedForwardBatch


class SimpleAttentionBackend:
    # decode kernel selection: _hip.DECODE_AUTO picks the matrix-core kernel
    # when it covers the shape, the generic kernel otherwise
    decode_variant = _hip.DECODE_AUTO

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

        if save_kv_cache and k is not None and v is not None:
            pool.set_kv_buffer(layer_id, forward_batch.out_cache_loc, k, v)

        if forward_batch.is_prefill:
            return SimpleAttentionBackend._prefill_attention(q, forward_batch, layer_id)
        return SimpleAttentionBackend._decode_attention(q, forward_batch, layer_id)

    @staticmethod
    def _decode_attention(q, forward_batch: SimplifiedForwardBatch, layer_id: int):
        k_slab, v_slab = forward_batch.token_to_kv_pool.get_kv_buffer(layer_id)
        return _hip.paged_decode(
            q,
            k_slab,
            v_slab,
            forward_batch.req_to_token_pool.req_to_token,
            forward_batch.req_pool_indices,
            forward_batch.seq_lens,
            variant=SimpleAttentionBackend.decode_variant,
        )

    @staticmethod
    def _prefill_attention(q, forward_batch: SimplifiedForwardBatch, layer_id: int):
        k_slab, v_slab = forward_batch.token_to_kv_pool.get_kv_buffer(layer_id)
        return _hip.paged_prefill(
            q,
            k_slab,
            v_slab,
            forward_batch.req_to_token_pool.req_to_token,
            forward_batch.req_pool_indices,
            forward_batch.seq_lens,
            forward_batch.prefix_lens,
            forward_batch.extended_lens,
        )
