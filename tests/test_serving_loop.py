"""The serving loop against traces of the reference scheduler (CPU only).

``tests/golden/driver_traces.pt`` records every model call the reference
``Scheduler.run_batch`` made for four request streams driven by a stand-in model whose
logits depend only on (token id, position).  The same streams through
``ServingLoop`` must give the same calls -- mode, input ids, positions, request rows,
sequence lengths, new slots, prefix lengths and page-table rows, bit for bit -- and the
same output ids.  The prefix tree is also exercised directly."""
import types

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from conftest import load_golden
from baby_sglang_amd.prefix_cache import PrefixCache
from baby_sglang_amd.serving_loop import SamplingParams, ServingLoop

TRACES = load_golden("driver_traces")


class StandInModel(nn.Module):
    """Same function as the one the goldens were generated with."""

    def __init__(self, vocab):
        super().__init__()
        self.config = types.SimpleNamespace(num_key_value_heads=1, hidden_size=64, num_attention_heads=1,
                                            num_hidden_layers=1, vocab_size=vocab)
        self.dummy = nn.Parameter(torch.zeros(1))

    def forward(self, input_ids, positions, forward_batch, input_embeds=None):
        return F.one_hot((7 * input_ids + 3 * positions + 1) % self.config.vocab_size, self.config.vocab_size).float()


@pytest.mark.parametrize("name", sorted(TRACES["scenarios"]))
def test_same_batches_as_the_reference_scheduler(name):
    sc = TRACES["scenarios"][name]
    loop = ServingLoop(StandInModel(TRACES["vocab"]), max_total_tokens=sc["max_total_tokens"], max_requests=sc["max_requests"])
    seen = []

    def record(mode, fb):
        if not fb.is_prefill:
            # the loop vouches for fresh slots (the decode kernel may then store the new rows itself):
            # each new slot is named once among the batch's visible page-table entries, by its own row
            assert fb.private_decode_slots
            table = fb.req_to_token_pool.req_to_token[fb.req_pool_indices]
            visible = torch.arange(table.shape[1])[None, :] < fb.seq_lens[:, None]
            for i, slot in enumerate(fb.out_cache_loc.tolist()):
                hits = ((table == slot) & visible).nonzero().tolist()
                assert hits == [[i, int(fb.seq_lens[i]) - 1]], (slot, hits)
        seen.append({
            "mode": mode, "input_ids": fb.input_ids.clone(), "positions": fb.positions.clone(),
            "req_pool_indices": fb.req_pool_indices.clone(), "seq_lens": fb.seq_lens.clone(),
            "out_cache_loc": fb.out_cache_loc.clone(),
            "prefix_lens": None if fb.prefix_lens is None else fb.prefix_lens.clone(),
            "table_rows": fb.req_to_token_pool.req_to_token[fb.req_pool_indices].clone(),
        })  # fmt: skip

    loop.on_batch = record
    outs = loop.run_batch(sc["prompts"], SamplingParams(max_new_tokens=sc["max_new_tokens"], eos_id=sc["eos_id"]))
    assert [list(o) for o in outs] == [o.tolist() for o in sc["output_ids"]]
    assert len(seen) == len(sc["steps"])
    for i, (got, want) in enumerate(zip(seen, sc["steps"])):
        assert got["mode"] == want["mode"], i
        for key in ("input_ids", "positions", "req_pool_indices", "seq_lens", "out_cache_loc", "prefix_lens", "table_rows"):
            if want[key] is None:
                assert got[key] is None, (i, key)
            else:
                assert torch.equal(got[key].to(want[key].dtype), want[key]), (i, key)
    if name == "eviction":
        assert sc["counts"]["evict"] > 0  # the trace really went through eviction
    # everything went back: rows free, slots free or owned by the tree
    assert loop.req_to_token_pool.available_size() == sc["max_requests"]
    assert loop.token_to_kv_pool.available_size() + loop.tree_cache.evictable_size() == sc["max_total_tokens"]
    # a second run on the same loop reuses what the tree kept
    outs2 = loop.run_batch(sc["prompts"], SamplingParams(max_new_tokens=sc["max_new_tokens"], eos_id=sc["eos_id"]))
    assert [list(o) for o in outs2] == [o.tolist() for o in sc["output_ids"]]


@pytest.mark.parametrize("name", sorted(TRACES["scenarios"]))
@pytest.mark.parametrize("limit", [1, 7, 16])
def test_chunked_prefill_makes_the_same_decisions(name, limit):
    """Prefill batches cut into extend calls of at most ``limit`` tokens: same output ids,
    same decode calls, and the chunks of a prefill batch put together are that batch."""
    sc = TRACES["scenarios"][name]
    loop = ServingLoop(StandInModel(TRACES["vocab"]), max_total_tokens=sc["max_total_tokens"],
                       max_requests=sc["max_requests"], chunked_prefill_size=limit)
    calls = []
    loop.on_batch = lambda mode, fb: calls.append((mode, fb))
    outs = loop.run_batch(sc["prompts"], SamplingParams(max_new_tokens=sc["max_new_tokens"], eos_id=sc["eos_id"]))
    assert [list(o) for o in outs] == [o.tolist() for o in sc["output_ids"]]
    i = 0
    for want in sc["steps"]:
        if want["mode"] == "decode":
            mode, fb = calls[i]
            i += 1
            assert mode == "decode" and torch.equal(fb.out_cache_loc, want["out_cache_loc"]) and torch.equal(fb.seq_lens, want["seq_lens"])
            continue
        parts, n_tok = [], 0
        while n_tok < len(want["input_ids"]):
            mode, fb = calls[i]
            i += 1
            assert mode == "prefill" and 0 < len(fb.input_ids) <= limit
            assert int(fb.extended_lens.sum()) == len(fb.input_ids) and bool((fb.seq_lens == fb.prefix_lens + fb.extended_lens).all())
            parts.append(fb)
            n_tok += len(fb.input_ids)
        for key in ("input_ids", "positions", "out_cache_loc"):
            assert torch.equal(torch.cat([getattr(fb, key) for fb in parts]), want[key]), key
        # every request of the batch appears, in order; one cut by a border appears in consecutive calls
        rows = torch.cat([fb.req_pool_indices for fb in parts]).tolist()
        assert [r for k, r in enumerate(rows) if k == 0 or rows[k - 1] != r] == want["req_pool_indices"].tolist()
        # a request's last chunk ends at its full length
        last_len = {}
        for fb in parts:
            for r, n in zip(fb.req_pool_indices.tolist(), fb.seq_lens.tolist()):
                last_len[r] = n
        assert [last_len[r] for r in want["req_pool_indices"].tolist()] == want["seq_lens"].tolist()
    assert i == len(calls)


def test_prefix_tree_split_insert_lock_evict():
    tree = PrefixCache()
    freed = []
    a = torch.arange(10, 16, dtype=torch.int32)
    assert tree.insert([1, 2, 3, 4, 5, 6], a) == 0
    assert tree.evictable_size() == 6
    # a key that leaves the cached one after three tokens: the node splits, three slots match
    slots, node = tree.match_prefix([1, 2, 3, 9, 9])
    assert slots.tolist() == [10, 11, 12] and node.key == [1, 2, 3]
    assert [c.key for c in node.children.values()] == [[4, 5, 6]]
    # inserting the diverging key reports the three tokens already present
    b = torch.arange(20, 25, dtype=torch.int32)
    assert tree.insert([1, 2, 3, 9, 9], b) == 3
    assert tree.evictable_size() == 8  # 3 shared + 3 + 2
    # lock one branch: its nodes (and the shared head) leave the evictable count
    _, leaf = tree.match_prefix([1, 2, 3, 4, 5, 6])
    tree.inc_lock_ref(leaf)
    assert tree.evictable_size() == 2
    tree.evict(100, lambda s: freed.extend(s.tolist()))
    assert freed == [23, 24]  # only the unlocked leaf went; its parent still has a child
    tree.dec_lock_ref(leaf)
    tree.evict(100, lambda s: freed.extend(s.tolist()))
    assert sorted(freed) == [10, 11, 12, 13, 14, 15, 23, 24] and tree.evictable_size() == 0
    assert tree.match_prefix([1, 2, 3])[0].numel() == 0


def test_least_recently_used_leaf_goes_first():
    tree = PrefixCache()
    tree.insert([1, 1], torch.tensor([1, 2], dtype=torch.int32))
    tree.insert([2, 2], torch.tensor([3, 4], dtype=torch.int32))
    tree.insert([3, 3], torch.tensor([5, 6], dtype=torch.int32))
    tree.match_prefix([1, 1])  # touch the oldest
    freed = []
    tree.evict(2, lambda s: freed.extend(s.tolist()))
    assert freed == [3, 4]


def test_sampled_decoding_is_reproducible_and_differs_from_greedy():
    class Spread(StandInModel):
        def forward(self, input_ids, positions, forward_batch, input_embeds=None):
            v = self.config.vocab_size
            base = torch.arange(v, dtype=torch.float32)
            return torch.cos(base[None, :] * (1 + input_ids[:, None] % 7) + positions[:, None].float())

    prompts = [torch.tensor([1, 2, 3, 4]), torch.tensor([9, 8, 7])]
    sp = SamplingParams(max_new_tokens=12, do_sample=True, temperature=0.9, top_k=5, top_p=0.95, repetition_penalty=1.2)
    outs = []
    for _ in range(2):
        loop = ServingLoop(Spread(23), max_total_tokens=128, max_requests=2)
        loop.generator = torch.Generator().manual_seed(4)
        outs.append(loop.run_batch(prompts, sp))
    assert outs[0] == outs[1]
    greedy = ServingLoop(Spread(23), max_total_tokens=128, max_requests=2).run_batch(prompts, SamplingParams(max_new_tokens=12))
    assert greedy != outs[0]
    loop = ServingLoop(Spread(23), max_total_tokens=128, max_requests=2)
    loop.generator = torch.Generator().manual_seed(4)
    over = loop.run_batch_overlap(prompts, SamplingParams(max_new_tokens=12, do_sample=True, temperature=0.9, top_k=5))
    assert all(len(o) == 12 for o in over)


@pytest.mark.parametrize("name", sorted(TRACES["scenarios"]))
def test_overlap_mode_gives_the_same_ids(name):
    """Ids looked at one step late (eos stops found out a step late, budget stops known ahead):
    same outputs as the step-by-step loop, and everything handed back at the end."""
    sc = TRACES["scenarios"][name]
    loop = ServingLoop(StandInModel(TRACES["vocab"]), max_total_tokens=sc["max_total_tokens"], max_requests=sc["max_requests"])
    modes = []
    loop.on_batch = lambda mode, fb: modes.append((mode, len(fb.req_pool_indices)))
    sp = SamplingParams(max_new_tokens=sc["max_new_tokens"], eos_id=sc["eos_id"])
    outs = loop.run_batch_overlap(sc["prompts"], sp)
    assert [list(o) for o in outs] == [o.tolist() for o in sc["output_ids"]]
    assert loop.req_to_token_pool.available_size() == sc["max_requests"]
    assert loop.token_to_kv_pool.available_size() + loop.tree_cache.evictable_size() == sc["max_total_tokens"]
    assert all(s.in_flight == 0 for s in loop.finished)
    # decode tokens run: one per produced token after the first, plus one extra step per eos stop
    n_decode_tokens = sum(n for m, n in modes if m == "decode")
    produced = sum(len(o) - 1 for o in outs)
    assert produced <= n_decode_tokens <= produced + sum(1 for o in outs if sc["eos_id"] != -1 and o[-1] == sc["eos_id"])
    # and again on the same loop, now with the prompts in the prefix tree
    assert [list(o) for o in loop.run_batch_overlap(sc["prompts"], sp)] == [o.tolist() for o in sc["output_ids"]]
    assert [list(o) for o in loop.run_batch(sc["prompts"], sp)] == [o.tolist() for o in sc["output_ids"]]


def _expected(prompt, n_new, vocab, eos):
    """What the stand-in model makes of a prompt: each id is a function of the id before it and its position."""
    ids, last, pos = [], int(prompt[-1]), len(prompt) - 1
    for _ in range(n_new):
        last = (7 * last + 3 * pos + 1) % vocab
        pos += 1
        ids.append(last)
        if eos != -1 and last == eos:
            break
    return ids


@pytest.mark.parametrize("seed", range(8))
def test_random_streams_in_every_mode(seed):
    """Random request streams (shared prefixes of random depth, repeated prompts, tight pools that
    force eviction, eos on or off): step-by-step, chunked and overlapped loops all give what the
    stand-in model defines request by request, and hand every row and slot back."""
    gen = torch.Generator().manual_seed(100 + seed)
    vocab = 31
    rows = int(torch.randint(2, 6, (1,), generator=gen))
    ctx = int(torch.randint(40, 64, (1,), generator=gen))
    n_new = int(torch.randint(1, 9, (1,), generator=gen))
    eos = int(torch.randint(0, vocab, (1,), generator=gen)) if seed % 2 else -1
    stems = [torch.randint(0, vocab, (int(torch.randint(4, 20, (1,), generator=gen)),), generator=gen) for _ in range(3)]
    prompts = []
    for _ in range(int(torch.randint(6, 14, (1,), generator=gen))):
        stem = stems[int(torch.randint(0, 3, (1,), generator=gen))]
        keep = int(torch.randint(1, len(stem) + 1, (1,), generator=gen))
        tail = torch.randint(0, vocab, (int(torch.randint(0, 12, (1,), generator=gen)),), generator=gen)
        prompts.append(torch.cat([stem[:keep], tail]))
    prompts.append(prompts[0].clone())  # an exact repeat
    want = [_expected(p.tolist(), n_new, vocab, eos) for p in prompts]
    sp = SamplingParams(max_new_tokens=n_new, eos_id=eos)
    for mode in ("plain", "chunked", "overlap"):
        loop = ServingLoop(StandInModel(vocab), max_total_tokens=rows * ctx, max_requests=rows,
                           chunked_prefill_size=5 if mode == "chunked" else None)
        for _ in range(2):  # the second pass runs against a warm prefix tree
            outs = (loop.run_batch_overlap if mode == "overlap" else loop.run_batch)(prompts, sp)
            assert [list(o) for o in outs] == want, (mode, seed)
            assert loop.req_to_token_pool.available_size() == rows
            assert loop.token_to_kv_pool.available_size() + loop.tree_cache.evictable_size() == rows * ctx
