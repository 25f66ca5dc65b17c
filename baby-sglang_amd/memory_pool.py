"""Request page table and per-layer KV slabs for the paged-attention path.

Host-side mirror of the reference pools (reference: memory_pool.py:7-44
``ReqToTokenPool``, memory_pool.py:46-102 ``BaseTokenToKVPool``,
memory_pool.py:104-171 ``MHATokenToKVPool``).  Names, argument meaning,
return types and error behaviour are the reference's, so the reference
scheduler and radix cache drive these objects unchanged.

What differs is the storage behind ``k_buffer`` / ``v_buffer``: every layer's
K and V rows live in ONE device arena laid out
``[layer][slot][K|V][kv_head][head_dim]`` (``kv_layout="interleaved"``), so a
token's K row and V row are adjacent in HBM (1 KiB per token for 2 kv heads
of 128 bf16) and one DRAM page serves both reads of the decode gather.
``kv_layout="split"`` gives the reference's separate contiguous K and V
tensors.  Either way ``k_buffer[l]`` / ``v_buffer[l]`` are
``[size + 1, kv_head, head_dim]`` tensors whose row stride the kernels take
as an argument; slot 0 is the never-allocated dummy row
(memory_pool.py:82-84).
"""

from __future__ import annotations

import logging
from typing import List, Optional, Tuple, Union

import torch

logger = logging.getLogger(__name__)

_FP8_DTYPES = tuple(
    getattr(torch, n) for n in ("float8_e5m2", "float8_e4m3fn") if hasattr(torch, n)
)


class ReqToTokenPool:
    """Page table ``req_to_token[max_req, max_context_len]`` (int32, device).

    Row r, column p holds the KV slot of token p of the request that owns row
    r; page size is one token.  Free rows are a Python list handed out from
    the front (reference: memory_pool.py:7-44).
    """

    def __init__(self, size: int, max_context_len: int, device: str):
        self.size = size
        self.max_context_len = max_context_len
        self.device = device
        self.req_to_token = torch.zeros(
            (size, max_context_len), dtype=torch.int32, device=device
        )
        self.free_slots: List[int] = list(range(size))

    def write(self, indices, values) -> None:
        # indices is a tensor of rows or a (row, slice) tuple (radix_cache.py:196-199)
        self.req_to_token[indices] = values

    def available_size(self) -> int:
        return len(self.free_slots)

    def alloc(self, need_size: int) -> Optional[List[int]]:
        if need_size > len(self.free_slots):
            return None
        taken, self.free_slots = (
            self.free_slots[:need_size],
            self.free_slots[need_size:],
        )
        return taken

    def free(self, free_index: Union[int, List[int]]) -> None:
        if isinstance(free_index, int):
            self.free_slots.append(free_index)
        else:
            self.free_slots.extend(free_index)

    def clear(self) -> None:
        self.free_slots = list(range(self.size))


class BaseTokenToKVPool:
    """Slot allocator: FIFO over a CPU int32 tensor of free slot ids.

    ``alloc(n)`` hands out the first n free ids as a device tensor, ``free``
    appends, ``None`` signals exhaustion (reference: memory_pool.py:46-84).
    """

    def __init__(self, size: int, dtype: torch.dtype, device: str):
        self.size = size
        self.dtype = dtype
        self.store_dtype = torch.uint8 if dtype in _FP8_DTYPES else dtype
        self.device = device
        self.free_slots: torch.Tensor = None
        self.clear()

    def available_size(self) -> int:
        return len(self.free_slots)

    def alloc(self, need_size: int) -> Optional[torch.Tensor]:
        if need_size > len(self.free_slots):
            return None
        taken = self.free_slots[:need_size]
        self.free_slots = self.free_slots[need_size:]
        return taken.to(self.device, non_blocking=True)

    def free(self, free_index: torch.Tensor) -> None:
        if free_index.numel() == 0:
            return
        self.free_slots = torch.concat((self.free_slots, free_index.cpu()))

    def clear(self) -> None:
        # slot 0 is the pad row; valid slots are 1..size
        self.free_slots = torch.arange(1, self.size + 1, dtype=torch.int32)

    def get_key_buffer(self, layer_id: int) -> torch.Tensor:
        raise NotImplementedError()

    def get_value_buffer(self, layer_id: int) -> torch.Tensor:
        raise NotImplementedError()

    def get_kv_buffer(self, layer_id: int) -> Tuple[torch.Tensor, torch.Tensor]:
        raise NotImplementedError()

    def set_kv_buffer(self, layer_id, loc, cache_k, cache_v) -> None:
        raise NotImplementedError()


class MHATokenToKVPool(BaseTokenToKVPool):
    """Per-layer K and V slabs ``[size + 1, head_num, head_dim]``.

    Reference: memory_pool.py:104-171.  ``set_kv_buffer`` always goes through
    the HIP scatter kernel: slabs that are not on a GPU are an error, there is
    no torch fallback.
    """

    def __init__(
        self,
        size: int,
        dtype: torch.dtype,
        head_num: int,
        head_dim: int,
        layer_num: int,
        device: str,
        kv_layout: str = "interleaved",
    ):
        super().__init__(size, dtype, device)
        if kv_layout not in ("interleaved", "split"):
            raise ValueError(f"unknown kv_layout {kv_layout!r}")
        self.layer_num = layer_num
        self.head_num = head_num
        self.head_dim = head_dim
        self.kv_layout = kv_layout
        self._create_buffers()

    def _create_buffers(self) -> None:
        rows, h, d = self.size + 1, self.head_num, self.head_dim
        if self.kv_layout == "interleaved":
            self.arena = torch.empty(
                (self.layer_num, rows, 2, h, d),
                dtype=self.store_dtype,
                device=self.device,
            )
            self.k_buffer = [self.arena[l, :, 0] for l in range(self.layer_num)]
            self.v_buffer = [self.arena[l, :, 1] for l in range(self.layer_num)]
        else:
            self.arena = torch.empty(
                (self.layer_num, 2, rows, h, d),
                dtype=self.store_dtype,
                device=self.device,
            )
            self.k_buffer = [self.arena[l, 0] for l in range(self.layer_num)]
            self.v_buffer = [self.arena[l, 1] for l in range(self.layer_num)]
        # elements between consecutive slots of one slab
        self.row_stride = self.k_buffer[0].stride(0)

    def _check_readable(self) -> None:
        if self.store_dtype != self.dtype:
            raise NotImplementedError(
                "float8 KV cache read is not supported in baby-sglang MVP"
            )

    def get_key_buffer(self, layer_id: int) -> torch.Tensor:
        self._check_readable()
        return self.k_buffer[layer_id]

    def get_value_buffer(self, layer_id: int) -> torch.Tensor:
        self._check_readable()
        return self.v_buffer[layer_id]

    def get_kv_buffer(self, layer_id: int) -> Tuple[torch.Tensor, torch.Tensor]:
        return self.get_key_buffer(layer_id), self.get_value_buffer(layer_id)

    def set_kv_buffer(
        self,
        layer_id: int,
        loc: torch.Tensor,
        cache_k: torch.Tensor,
        cache_v: torch.Tensor,
    ) -> None:
        if cache_k.dtype != self.dtype:
            cache_k = cache_k.to(self.dtype)
            cache_v = cache_v.to(self.dtype)
        if self.store_dtype != self.dtype:
            raise NotImplementedError(
                "float8 KV cache write is not supported in baby-sglang MVP"
            )
        from . import _hip  # raises if the HIP library is not built

        _hip.kv_scatter(
            self.k_buffer[layer_id], self.v_buffer[layer_id], loc, cache_k, cache_v
        )
