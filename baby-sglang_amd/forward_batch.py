"""Batch descriptor handed from the scheduler down to the attention backend.

Mirror of the reference's ``SimplifiedForwardBatch`` (reference:
forward_batch.py:8-84): same field names and dtypes, same two constructors.

Field dtypes as the reference scheduler produces them (scheduler.py:389-406,
:459-498): ``req_pool_indices``, ``seq_lens``, ``prefix_lens``,
``extended_lens``, ``positions``, ``input_ids`` int64; ``out_cache_loc`` int32.
Decode ``seq_lens`` already count the token written this step
(scheduler.py:485-490).
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

from .memory_pool import MHATokenToKVPool, ReqToTokenPool


@dataclass
class SimplifiedForwardBatch:
    batch_size: int
    input_ids: torch.Tensor
    req_pool_indices: torch.Tensor
    seq_lens: torch.Tensor
    out_cache_loc: torch.Tensor
    positions: torch.Tensor
    req_to_token_pool: ReqToTokenPool
    token_to_kv_pool: MHATokenToKVPool
    is_prefill: bool = True

    # prefix-cache metadata (prefill / extend only)
    prefix_lens: Optional[torch.Tensor] = None
    extended_lens: Optional[torch.Tensor] = None

    # Not in the reference: the longest request of a decode batch, when the scheduler knows it on
    # the host (0 = unknown).  The backend hands it to the library as a hint for the shape of the
    # launch; results never depend on it.
    max_seq_len: int = 0

    # Not in the reference: the producer's word, for a decode batch, that every slot in out_cache_loc
    # is fresh -- named by its own request's page-table row (at position seq_len - 1) and by nothing
    # else, and by no other entry of out_cache_loc.  The backend then lets the decode kernel store the
    # new rows itself; without it the rows are written by a launch of their own before attention reads
    # the cache, which is right for any page table (reference order: attn_backend.py:20-23).
    private_decode_slots: bool = False

    @classmethod
    def create_prefill_batch(
        cls,
        input_ids: torch.Tensor,
        req_pool_indices: torch.Tensor,
        seq_lens: torch.Tensor,
        out_cache_loc: torch.Tensor,
        req_to_token_pool: ReqToTokenPool,
        token_to_kv_pool: MHATokenToKVPool,
        prefix_lens: torch.Tensor,
    ) -> "SimplifiedForwardBatch":
        """Extend batch: request i contributes tokens ``prefix_lens[i]..seq_lens[i]-1``.

        Reference: forward_batch.py:24-59.  Positions are built without a
        per-request host loop: position of extend token j of request i is
        ``prefix_lens[i] + j``.
        """
        device = input_ids.device
        extended_lens = (seq_lens - prefix_lens).to(dtype=torch.long, device=device)
        starts = prefix_lens.to(dtype=torch.long, device=device)
        # (the batch carries exactly its requests' extend tokens: no device read needed for the total)
        total = int(input_ids.shape[0])
        owner = torch.repeat_interleave(
            torch.arange(len(req_pool_indices), device=device),
            extended_lens,
            output_size=total,
        )
        first = torch.cumsum(extended_lens, 0) - extended_lens
        positions = (
            torch.arange(total, device=device, dtype=torch.long)
            - first[owner]
            + starts[owner]
        )
        return cls(
            batch_size=len(req_pool_indices),
            input_ids=input_ids,
            req_pool_indices=req_pool_indices,
            seq_lens=seq_lens,
            out_cache_loc=out_cache_loc,
            positions=positions,
            req_to_token_pool=req_to_token_pool,
            token_to_kv_pool=token_to_kv_pool,
            is_prefill=True,
            prefix_lens=prefix_lens,
            extended_lens=extended_lens,
        )

    @classmethod
    def create_decode_batch(
        cls,
        input_ids: torch.Tensor,
        req_pool_indices: torch.Tensor,
        seq_lens: torch.Tensor,
        out_cache_loc: torch.Tensor,
        req_to_token_pool: ReqToTokenPool,
        token_to_kv_pool: MHATokenToKVPool,
    ) -> "SimplifiedForwardBatch":
        """One new token per request; its position is ``seq_lens - 1``.

        Reference: forward_batch.py:61-84.
        """
        return cls(
            batch_size=len(req_pool_indices),
            input_ids=input_ids,
            req_pool_indices=req_pool_indices,
            seq_lens=seq_lens,
            out_cache_loc=out_cache_loc,
            positions=seq_lens - 1,
            req_to_token_pool=req_to_token_pool,
            token_to_kv_pool=token_to_kv_pool,
            is_prefill=False,
        )
