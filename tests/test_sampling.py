"""Next-token selection against distributions captured from the reference's sample.py
(tests/golden/sampling.pt, made by gen_sampling_golden.py): the filtered distribution is the
reference's bit for bit for every (temperature, top-k, top-p, repetition penalty) case,
computed for the whole batch at once; greedy ids equal; draws come from that distribution."""
import torch

from conftest import load_golden
from baby_sglang_amd import sampling as smp

CASES = load_golden("sampling")["cases"]


def test_distribution_is_the_reference_s_bit_for_bit():
    for c in CASES:
        logits = smp.apply_repetition_penalty_(c["logits"].clone(), c["prev_ids"], c["repetition_penalty"])
        probs = smp.next_token_probs(logits, c["temperature"], c["top_k"], c["top_p"])
        assert torch.equal(probs, c["probs"]), (c["temperature"], c["top_k"], c["top_p"], c["repetition_penalty"])


def test_greedy_ids_and_caller_tensor_untouched():
    for c in CASES:
        before = c["logits"].clone()
        ids = smp.sample_next_ids(c["logits"], False, c["temperature"], c["top_k"], c["top_p"], c["prev_ids"], c["repetition_penalty"])
        assert torch.equal(ids, c["greedy_ids"])
        assert torch.equal(c["logits"], before)


def test_draws_stay_inside_the_kept_set_and_follow_the_generator():
    c = next(c for c in CASES if c["top_k"] == 10)
    g = torch.Generator().manual_seed(1)
    a = torch.stack([smp.sample_next_ids(c["logits"], True, c["temperature"], c["top_k"], c["top_p"], generator=g) for _ in range(50)])
    kept = c["probs"] > 0
    assert bool(kept.gather(1, a.t()).all())
    g2 = torch.Generator().manual_seed(1)
    b = torch.stack([smp.sample_next_ids(c["logits"], True, c["temperature"], c["top_k"], c["top_p"], generator=g2) for _ in range(50)])
    assert torch.equal(a, b)
    assert (kept.sum(-1) >= 10).all() and (kept.sum(-1) <= 11).all()  # the top tie may keep one more
