"""Next-token selection on the device, for whole batches at once.

Counterpart of the reference's ``sample.py`` (reference: sample.py:8-17 ``SamplingParams``,
:19-37 repetition penalty, :40-90 ``sample_next_ids``).  The reference is called once per
request with a ``[1, vocab]`` row; here every function takes the ``[batch, vocab]`` logits of a
step and stays on the device (no per-request loop, no host read).  The filtered distribution
is the reference's bit for bit -- temperature, then top-k (ties with the k-th value kept),
then nucleus filtering over the survivors (the first token always kept), then softmax -- but
top-k and top-p share ONE descending sort: masking the values below the k-th only turns the
tail of the sorted row into -inf, which is exactly the sorted row of the masked logits.
The random draw itself (``torch.multinomial``) depends on the generator state and is not
comparable across implementations.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Sequence

import torch


@dataclass
class SamplingParams:
    """Fields of the reference's ``SamplingParams`` (sample.py:8-17; ``json_schema`` belongs to
    constrained decoding, which is not part of this build)."""

    max_new_tokens: int = 64
    temperature: float = 1.0
    top_k: int = 0
    top_p: float = 1.0
    eos_id: int = -1
    do_sample: bool = False
    repetition_penalty: float = 1.0


def apply_repetition_penalty_(logits: torch.Tensor, prev_ids: Sequence[Sequence[int]], penalty: float) -> torch.Tensor:
    """In place: logits of tokens a request has already produced or been given are divided by
    ``penalty`` when positive and multiplied by it otherwise (sample.py:19-37), for all rows
    with one indexed update.  ``prev_ids[i]`` are row i's tokens (duplicates count once)."""
    if penalty is None or penalty <= 1.0:
        return logits
    rows, cols = [], []
    for i, ids in enumerate(prev_ids):
        uniq = set(int(t) for t in ids)
        rows.extend([i] * len(uniq))
        cols.extend(uniq)
    if not rows:
        return logits
    r = torch.tensor(rows, dtype=torch.long, device=logits.device)
    c = torch.tensor(cols, dtype=torch.long, device=logits.device)
    val = logits[r, c]
    logits[r, c] = torch.where(val > 0, val / penalty, val * penalty)
    return logits


def next_token_probs(logits: torch.Tensor, temperature: float = 1.0, top_k: int = 0, top_p: float = 1.0) -> torch.Tensor:
    """The distribution the reference draws from (sample.py:59-89), ``[batch, vocab]``."""
    if temperature is None or temperature <= 0:
        temperature = 1.0
    scaled = logits / temperature
    vocab = scaled.shape[-1]
    use_k = bool(top_k) and top_k > 0
    use_p = bool(top_p) and top_p < 1.0
    if not use_k and not use_p:
        return torch.softmax(scaled, dim=-1)
    minus_inf = torch.full((), float("-inf"), dtype=scaled.dtype, device=scaled.device)
    if not use_p:
        kth = torch.topk(scaled, k=min(top_k, vocab), dim=-1).values[..., -1:]
        return torch.softmax(torch.where(scaled < kth, minus_inf, scaled), dim=-1)
    ordered, order = torch.sort(scaled, descending=True, dim=-1)
    if use_k:
        kth = ordered[..., min(top_k, vocab) - 1 : min(top_k, vocab)]
        ordered = torch.where(ordered < kth, minus_inf, ordered)
    mass = torch.cumsum(torch.softmax(ordered, dim=-1), dim=-1)
    drop = mass > top_p
    drop[..., 0] = False
    ordered = torch.where(drop, minus_inf, ordered)
    kept = torch.full_like(scaled, float("-inf")).scatter_(-1, order, ordered)
    return torch.softmax(kept, dim=-1)


def sample_next_ids(
    logits: torch.Tensor,
    do_sample: bool = False,
    temperature: float = 1.0,
    top_k: int = 0,
    top_p: float = 1.0,
    prev_ids: Optional[Sequence[Sequence[int]]] = None,
    repetition_penalty: float = 1.0,
    generator: Optional[torch.Generator] = None,
) -> torch.Tensor:
    """One id per row of ``logits`` (a copy is penalised, never the caller's tensor)."""
    if prev_ids is not None and repetition_penalty is not None and repetition_penalty > 1.0:
        logits = apply_repetition_penalty_(logits.clone(), prev_ids, repetition_penalty)
    if not do_sample:
        return torch.argmax(logits, dim=-1)
    probs = next_token_probs(logits, temperature, top_k, top_p)
    return torch.multinomial(probs, num_samples=1, generator=generator).squeeze(-1)
