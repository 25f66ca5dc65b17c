"""Qwen2 decoder over the HIP path: the callers of the attention backend.

Same module tree, class names and parameter names as the reference model (reference:
qwen2.py:9-315), so a reference ``state_dict`` loads unchanged and
``model(input_ids, positions, forward_batch)`` returns the logits of every input
token.  What differs is what runs between the GEMMs:

  * RMSNorm (+ residual add) is one HIP launch (``bsgl_rmsnorm``; qwen2.py:125-141);
  * the QKV split, the rotary embedding of q and k and the KV-cache write are one
    HIP launch (``bsgl_rope_kv_write``; qwen2.py:246-268, :285-290,
    memory_pool.py:152-171), so attention is then called without new K/V rows;
  * attention is ``SimpleAttentionBackend`` of this package.

GEMMs, embedding lookup and SiLU stay torch ops (rocBLAS / hipBLASLt): plumbing here.
``inv_freq`` is a module buffer as in the reference, so ``model.to(torch.bfloat16)``
rounds it to bf16 exactly as the reference's does.

Tensor parallel (the reference has the collectives, tp_runtime.py:10-51, and no sharded model):
``BabyQwen2ForCausalLM(config, tp_layout=TPLayout(rank, world, Hq, Hkv, replicate_kv=True),
tp_group=group)`` builds one rank's shard -- column-parallel q / k / v and gate / up
projections, row-parallel ``o_proj`` and ``down_proj`` each followed by ``tp_all_reduce_sum``
(tp_runtime.py:28-33) -- over a KV pool of ``tp_layout.local_kv_heads`` heads; embeddings, norms and
the vocabulary projection are replicated.  ``shard_state_dict`` cuts a whole (reference)
``state_dict`` down to a rank's.
"""

from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _hip
from .attn_backend import SimpleAttentionBackend
from .forward_batch import SimplifiedForwardBatch
from .tp_runtime import TPLayout, tp_all_reduce_sum


def _model_layout(tp_layout: Optional[TPLayout]) -> Optional[TPLayout]:
    if tp_layout is None or tp_layout.world == 1:
        return None
    if tp_layout.tp_seq != 1:
        raise NotImplementedError(
            "a tensor-parallel model shards heads (TPLayout(..., replicate_kv=True)); sequence slices "
            "inside a kv head are the attention path's own layout and have no extend (prefill) form"
        )
    return tp_layout


class RMSNorm(nn.Module):
    def __init__(self, hidden_size: int, eps: float = 1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps

    def forward(self, hidden_states: torch.Tensor, residual: Optional[torch.Tensor] = None):
        return _hip.rmsnorm(hidden_states, self.weight, self.variance_epsilon, residual)


class RotaryEmbedding(nn.Module):
    """Holds ``inv_freq`` (qwen2.py:243: from the constructor's ``base``; a "linear"
    rope_scaling entry changes nothing there either).  The rotation itself happens in
    ``Qwen2Attention`` together with the QKV split and the cache write."""

    def __init__(self, head_dim: int, max_position_embeddings: int = 32768, base: float = 10000.0, rope_scaling=None):
        super().__init__()
        self.head_dim = head_dim
        self.max_position_embeddings = max_position_embeddings
        self.base = base
        self.register_buffer("inv_freq", 1.0 / (base ** (torch.arange(0, head_dim, 2).float() / head_dim)), persistent=False)

    def forward(self, positions: torch.Tensor, q: torch.Tensor, k: torch.Tensor):
        """Rotated ``(q, k)`` for ``[T, heads, D]`` inputs (no cache write)."""
        n_tok, hq, dim = q.shape
        hkv = k.shape[1]
        qkv = torch.cat((q.reshape(n_tok, -1), k.reshape(n_tok, -1), k.reshape(n_tok, -1)), dim=-1)
        q_rot, k_rot, _ = _hip.rope_kv_write(qkv, positions, self.inv_freq, hq, hkv, dim, return_kv=True)
        return q_rot, k_rot


class QKVLinear(nn.Module):
    def __init__(self, input_size: int, num_heads: int, num_kv_heads: int, head_dim: int, bias: bool = True):
        super().__init__()
        self.q_size = num_heads * head_dim
        self.kv_size = num_kv_heads * head_dim
        self.qkv_proj = nn.Linear(input_size, self.q_size + 2 * self.kv_size, bias=bias)

    def forward(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        return self.qkv_proj(x).split([self.q_size, self.kv_size, self.kv_size], dim=-1)


class GateUpLinear(nn.Module):
    def __init__(self, input_size: int, intermediate_size: int, bias: bool = False):
        super().__init__()
        self.gate_up_proj = nn.Linear(input_size, 2 * intermediate_size, bias=bias)

    def forward(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        return self.gate_up_proj(x).chunk(2, dim=-1)


class BabyRadixAttention(nn.Module):
    """The shim the backend reads ``layer_id`` / ``scaling`` from (qwen2.py:292-315)."""

    def __init__(self, num_heads: int, head_dim: int, scaling: float, num_kv_heads: int, layer_id: int):
        super().__init__()
        self.tp_q_head_num = num_heads
        self.tp_k_head_num = num_kv_heads
        self.tp_v_head_num = num_kv_heads
        self.head_dim = head_dim
        self.scaling = scaling
        self.layer_id = layer_id

    def forward(self, q, k, v, forward_batch: SimplifiedForwardBatch, save_kv_cache: bool = True):
        return SimpleAttentionBackend.forward(q, k, v, self, forward_batch, save_kv_cache)


class Qwen2Attention(nn.Module):
    def __init__(self, hidden_size: int, num_heads: int, num_kv_heads: int, layer_id: int = 0,
                 rope_theta: float = 1000000, rope_scaling=None, max_position_embeddings: int = 32768,
                 tp_layout: Optional[TPLayout] = None, tp_group=None):  # fmt: skip
        super().__init__()
        self.head_dim = hidden_size // num_heads
        self.tp_layout, self.tp_group = _model_layout(tp_layout), tp_group
        if self.tp_layout is not None:
            # this rank's heads: the projections are cut by output column (q / k / v) and input
            # column (o_proj) accordingly
            num_heads, num_kv_heads = self.tp_layout.local_q_heads, self.tp_layout.local_kv_heads
        self.num_heads = num_heads
        self.num_kv_heads = num_kv_heads
        self.scaling = self.head_dim**-0.5
        self.qkv_proj = QKVLinear(hidden_size, num_heads, num_kv_heads, self.head_dim, bias=True)
        self.o_proj = nn.Linear(num_heads * self.head_dim, hidden_size, bias=False)
        self.rotary_emb = RotaryEmbedding(self.head_dim, max_position_embeddings, rope_theta, rope_scaling)
        self.attn = BabyRadixAttention(num_heads, self.head_dim, self.scaling, num_kv_heads, layer_id)

    def forward(self, positions: torch.Tensor, hidden_states: torch.Tensor, forward_batch: SimplifiedForwardBatch):
        qkv = self.qkv_proj.qkv_proj(hidden_states)
        pool = forward_batch.token_to_kv_pool
        if pool.store_dtype == qkv.dtype:
            # one launch: split, rotate, and put k / v into their cache rows
            k_slab, v_slab = pool.get_kv_buffer(self.attn.layer_id)
            q, _, _ = _hip.rope_kv_write(
                qkv, positions, self.rotary_emb.inv_freq, self.num_heads, self.num_kv_heads, self.head_dim,
                k_slab, v_slab, forward_batch.out_cache_loc,
            )
            attn = self.attn(q, None, None, forward_batch, save_kv_cache=False)
        else:
            # the pool casts on write (memory_pool.py:163-165): rotate only, let the pool store
            q, k, v = _hip.rope_kv_write(qkv, positions, self.rotary_emb.inv_freq, self.num_heads,
                                         self.num_kv_heads, self.head_dim, return_kv=True)  # fmt: skip
            attn = self.attn(q, k, v, forward_batch)
        out = self.o_proj(attn)
        if self.tp_layout is not None:
            # row-parallel projection: every rank holds the partial sum over its heads
            out = tp_all_reduce_sum(out, self.tp_group)
        return out


class Qwen2MLP(nn.Module):
    def __init__(self, hidden_size: int, intermediate_size: int, hidden_act: str = "silu",
                 tp_layout: Optional[TPLayout] = None, tp_group=None):
        super().__init__()
        if hidden_act != "silu":
            raise ValueError(f"Unsupported activation: {hidden_act}. Only silu supported.")
        self.tp_layout, self.tp_group = _model_layout(tp_layout), tp_group
        if self.tp_layout is not None:
            if intermediate_size % self.tp_layout.world:
                raise ValueError(f"intermediate size {intermediate_size} does not divide over {self.tp_layout.world} ranks")
            intermediate_size //= self.tp_layout.world
        self.gate_up_proj = GateUpLinear(hidden_size, intermediate_size, bias=False)
        self.down_proj = nn.Linear(intermediate_size, hidden_size, bias=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        gate, up = self.gate_up_proj(x)
        out = self.down_proj(F.silu(gate) * up)
        if self.tp_layout is not None:
            out = tp_all_reduce_sum(out, self.tp_group)
        return out


class Qwen2DecoderLayer(nn.Module):
    def __init__(self, config, layer_id: int = 0, tp_layout: Optional[TPLayout] = None, tp_group=None):
        super().__init__()
        self.self_attn = Qwen2Attention(
            hidden_size=config.hidden_size,
            num_heads=config.num_attention_heads,
            num_kv_heads=config.num_key_value_heads,
            layer_id=layer_id,
            rope_theta=getattr(config, "rope_theta", 1000000),
            rope_scaling=getattr(config, "rope_scaling", None),
            max_position_embeddings=getattr(config, "max_position_embeddings", 32768),
            tp_layout=tp_layout,
            tp_group=tp_group,
        )
        self.mlp = Qwen2MLP(config.hidden_size, config.intermediate_size, getattr(config, "hidden_act", "silu"),
                            tp_layout=tp_layout, tp_group=tp_group)
        self.input_layernorm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.post_attention_layernorm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)

    def forward(self, positions, hidden_states, forward_batch, residual=None):
        if residual is None:
            residual = hidden_states
            hidden_states, _ = self.input_layernorm(hidden_states)
        else:
            hidden_states, residual = self.input_layernorm(hidden_states, residual)
        hidden_states = self.self_attn(positions, hidden_states, forward_batch)
        hidden_states, residual = self.post_attention_layernorm(hidden_states, residual)
        return self.mlp(hidden_states), residual


class BabyQwen2Model(nn.Module):
    def __init__(self, config, tp_layout: Optional[TPLayout] = None, tp_group=None):
        super().__init__()
        self.config = config
        self.vocab_size = config.vocab_size
        self.embed_tokens = nn.Embedding(config.vocab_size, config.hidden_size)
        self.layers = nn.ModuleList(
            Qwen2DecoderLayer(config, layer_id=i, tp_layout=tp_layout, tp_group=tp_group) for i in range(config.num_hidden_layers)
        )
        self.norm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)

    def forward(self, input_ids, positions, forward_batch, input_embeds=None):
        hidden_states = self.embed_tokens(input_ids) if input_embeds is None else input_embeds
        residual = None
        for layer in self.layers:
            hidden_states, residual = layer(positions, hidden_states, forward_batch, residual)
        hidden_states, _ = self.norm(hidden_states, residual)
        return hidden_states


def shard_state_dict(state_dict, config, tp_layout: TPLayout):
    """One rank's part of a whole model's ``state_dict`` (reference parameter names): the rows of
    ``qkv_proj`` (and its bias) that make this rank's q, k and v heads, the columns of ``o_proj``
    that take its q heads, its slice of the MLP's intermediate columns; everything else whole."""
    lay = _model_layout(tp_layout)
    if lay is None:
        return dict(state_dict)
    hq, hkv = config.num_attention_heads, config.num_key_value_heads
    dim = getattr(config, "head_dim", None) or config.hidden_size // hq
    q_lo, q_hi = (x * dim for x in lay.q_head_range)
    kv_lo, kv_hi = (x * dim for x in lay.kv_head_range)
    inter = config.intermediate_size
    i_lo, i_hi = lay.rank * inter // lay.world, (lay.rank + 1) * inter // lay.world
    out = {}
    for name, w in state_dict.items():
        if name.endswith("qkv_proj.qkv_proj.weight") or name.endswith("qkv_proj.qkv_proj.bias"):
            q, k, v = w.split([hq * dim, hkv * dim, hkv * dim], dim=0)
            w = torch.cat((q[q_lo:q_hi], k[kv_lo:kv_hi], v[kv_lo:kv_hi]), dim=0)
        elif name.endswith("self_attn.o_proj.weight"):
            w = w[:, q_lo:q_hi]
        elif name.endswith("gate_up_proj.gate_up_proj.weight"):
            gate, up = w.chunk(2, dim=0)
            w = torch.cat((gate[i_lo:i_hi], up[i_lo:i_hi]), dim=0)
        elif name.endswith("mlp.down_proj.weight"):
            w = w[:, i_lo:i_hi]
        out[name] = w.contiguous()
    return out


class BabyQwen2ForCausalLM(nn.Module):
    def __init__(self, config, tp_layout: Optional[TPLayout] = None, tp_group=None):
        super().__init__()
        self.config = config
        self.model = BabyQwen2Model(config, tp_layout=tp_layout, tp_group=tp_group)
        if getattr(config, "tie_word_embeddings", False):
            self.lm_head = self.model.embed_tokens
        else:
            self.lm_head = nn.Linear(config.hidden_size, config.vocab_size, bias=False)

    supports_logits_rows = True

    @torch.no_grad()
    def forward(self, input_ids, positions, forward_batch, input_embeds=None, logits_rows=None):
        """Logits ``[T, vocab]`` of every input token (qwen2.py:21-37), or, with
        ``logits_rows`` (token indices), of those tokens only: a prefill batch is sampled
        from the last token of each request, and the other rows of the vocabulary projection
        are most of the work and memory of a long prompt."""
        hidden_states = self.model(input_ids, positions, forward_batch, input_embeds)
        if logits_rows is not None:
            hidden_states = hidden_states[logits_rows]
        return F.linear(hidden_states, self.lm_head.weight)
