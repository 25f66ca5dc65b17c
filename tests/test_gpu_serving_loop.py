"""The serving loop on the GPU: continuous batching + prefix reuse over the HIP path.

``tests/golden/driver_traces.pt`` (``tiny_model_e2e``) holds what the reference
``Scheduler.run_batch`` did with the tiny random-weight reference model on five prompts
that share a prefix, three page-table rows (two admission waves), fp32 on the CPU:
every model call's batch and the greedy output ids.  ``ServingLoop`` over the HIP-path
model with the same weights must make the same calls and produce the same ids, and a
second run on the same loop, which now finds every prompt in the prefix tree, too.
"""
import types

import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _tiny_model(dev, dtype=torch.float32):
    import baby_sglang_amd as pkg

    g = load_golden("tiny_model")
    cfg = types.SimpleNamespace(**g["config"], tie_word_embeddings=False, hidden_act="silu", rope_scaling=None)
    model = pkg.BabyQwen2ForCausalLM(cfg).eval()
    model.load_state_dict(g["state_dict"], strict=True)
    return model.to(dtype).to(dev)


def test_tiny_model_end_to_end_matches_reference_scheduler(gpu):
    from baby_sglang_amd.serving_loop import SamplingParams, ServingLoop

    e2e = load_golden("driver_traces")["tiny_model_e2e"]
    loop = ServingLoop(_tiny_model(gpu), max_total_tokens=e2e["max_total_tokens"], max_requests=e2e["max_requests"])
    seen = []
    loop.on_batch = lambda mode, fb: seen.append((mode, fb))
    sp = SamplingParams(max_new_tokens=e2e["max_new_tokens"], eos_id=e2e["eos_id"])
    outs = loop.run_batch(e2e["prompts"], sp)
    assert [list(o) for o in outs] == [o.tolist() for o in e2e["output_ids"]]
    assert len(seen) == len(e2e["steps"])
    for i, ((mode, fb), want) in enumerate(zip(seen, e2e["steps"])):
        assert mode == want["mode"], i
        for key in ("input_ids", "positions", "req_pool_indices", "seq_lens", "out_cache_loc", "prefix_lens"):
            got = getattr(fb, key)
            if want[key] is None:
                assert got is None, (i, key)
            else:
                assert torch.equal(got.cpu().to(want[key].dtype), want[key]), (i, key)
    # second pass: the prompts are in the prefix tree now, so prefill runs over one token each
    seen.clear()
    outs2 = loop.run_batch(e2e["prompts"], sp)
    assert [list(o) for o in outs2] == [o.tolist() for o in e2e["output_ids"]]
    first_prefill = next(fb for mode, fb in seen if mode == "prefill")
    assert int(first_prefill.extended_lens.max()) == 1


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_batch_of_shared_prefixes(gpu, dtype):
    """More requests than page-table rows, a long shared prefix: two fresh loops give the
    same ids (both dtypes); in fp32 the ids do not depend on whether a request ran alone
    (whole prompt through prefill) or batched behind a cached prefix (extend + batched decode)."""
    from baby_sglang_amd.serving_loop import SamplingParams, ServingLoop

    model = _tiny_model(gpu, dtype)
    vocab = model.config.vocab_size
    gen = torch.Generator().manual_seed(5)
    shared = torch.randint(0, vocab, (40,), generator=gen)
    prompts = [torch.cat([shared, torch.randint(0, vocab, (n,), generator=gen)]) for n in (3, 9, 1, 17, 6, 2, 11)]
    sp = SamplingParams(max_new_tokens=8)
    a = ServingLoop(model, max_total_tokens=512, max_requests=4).run_batch(prompts, sp)
    b = ServingLoop(model, max_total_tokens=512, max_requests=4).run_batch(prompts, sp)
    assert a == b
    assert all(len(o) == 8 for o in a)
    if dtype == torch.float32:
        for i in (0, 3, 6):
            alone = ServingLoop(model, max_total_tokens=512, max_requests=4).run_batch([prompts[i]], sp)
            assert alone[0] == a[i], i


@pytest.mark.parametrize("limit", [5, 16])
def test_chunked_prefill_gives_the_reference_ids(gpu, limit):
    """Prompts run through the model as extend calls of at most ``limit`` tokens (each over
    what the earlier calls cached): the ids are those of the reference's whole-prompt prefill."""
    from baby_sglang_amd.serving_loop import SamplingParams, ServingLoop

    e2e = load_golden("driver_traces")["tiny_model_e2e"]
    loop = ServingLoop(_tiny_model(gpu), max_total_tokens=e2e["max_total_tokens"], max_requests=e2e["max_requests"],
                       chunked_prefill_size=limit)
    sizes = []
    loop.on_batch = lambda mode, fb: sizes.append(len(fb.input_ids)) if mode == "prefill" else None
    outs = loop.run_batch(e2e["prompts"], SamplingParams(max_new_tokens=e2e["max_new_tokens"], eos_id=e2e["eos_id"]))
    assert [list(o) for o in outs] == [o.tolist() for o in e2e["output_ids"]]
    assert max(sizes) <= limit and len(sizes) > sum(1 for s in e2e["steps"] if s["mode"] == "prefill")


@pytest.mark.parametrize("worker", [True, False])
def test_overlap_mode_on_the_gpu(gpu, worker):
    """Batches launched before the previous one's ids have been looked at -- with the model on a
    worker thread and a second stream (the reference's arrangement, overlap_worker.py:27-160) and
    on one thread and one stream (ids fed from the device-side buffer, metadata through pinned
    staging): the reference's ids, also on a second run that finds the prompts in the prefix tree,
    also with an eos that stops some requests early (their extra step's slot is handed back)."""
    from baby_sglang_amd.serving_loop import SamplingParams, ServingLoop

    e2e = load_golden("driver_traces")["tiny_model_e2e"]
    loop = ServingLoop(_tiny_model(gpu), max_total_tokens=e2e["max_total_tokens"], max_requests=e2e["max_requests"],
                       overlap_worker=worker)
    assert loop.overlap_worker is worker
    sp = SamplingParams(max_new_tokens=e2e["max_new_tokens"], eos_id=e2e["eos_id"])
    want = [o.tolist() for o in e2e["output_ids"]]
    assert [list(o) for o in loop.run_batch_overlap(e2e["prompts"], sp)] == want
    assert [list(o) for o in loop.run_batch_overlap(e2e["prompts"], sp)] == want
    # stop on the token that request 0 produces third
    eos = want[0][2]
    sp_eos = SamplingParams(max_new_tokens=e2e["max_new_tokens"], eos_id=eos)
    plain = ServingLoop(_tiny_model(gpu), max_total_tokens=e2e["max_total_tokens"], max_requests=e2e["max_requests"])
    a = plain.run_batch(e2e["prompts"], sp_eos)
    b = loop.run_batch_overlap(e2e["prompts"], sp_eos)
    assert a == b and len(a[0]) <= 3
    assert loop.req_to_token_pool.available_size() == e2e["max_requests"]
    assert loop.token_to_kv_pool.available_size() + loop.tree_cache.evictable_size() == e2e["max_total_tokens"]
    loop.close()


class _StandIn(torch.nn.Module):
    """The stand-in model the reference scheduler's traces were made with (tests/test_serving_loop.py):
    logits depend on (token id, position) only."""

    def __init__(self, vocab):
        super().__init__()
        self.config = types.SimpleNamespace(num_key_value_heads=1, hidden_size=64, num_attention_heads=1,
                                            num_hidden_layers=1, vocab_size=vocab)
        self.dummy = torch.nn.Parameter(torch.zeros(1))

    def forward(self, input_ids, positions, forward_batch, input_embeds=None):
        v = self.config.vocab_size
        return torch.nn.functional.one_hot((7 * input_ids + 3 * positions + 1) % v, v).float()


@pytest.mark.parametrize("name", sorted(load_golden("driver_traces")["scenarios"]))
def test_two_stream_overlap_gives_the_reference_schedulers_ids(gpu, name):
    """The request streams of the reference scheduler's traces (shared prefixes, two admission
    waves, eviction, eos stops) through the overlap loop with the forward pass on the worker's
    thread and stream: the ids the reference's ``run_batch`` produced, twice on one loop (the second
    run finds the prompts in the prefix tree), every row and slot handed back.  The ORDER of batches
    in this mode is not pinned: the reference's worker refuses to run without CUDA
    (overlap_worker.py:50-53), so no trace of it could be recorded."""
    from baby_sglang_amd.serving_loop import SamplingParams, ServingLoop

    traces = load_golden("driver_traces")
    sc = traces["scenarios"][name]
    model = _StandIn(traces["vocab"]).to(gpu)
    loop = ServingLoop(model, max_total_tokens=sc["max_total_tokens"], max_requests=sc["max_requests"], overlap_worker=True)
    sp = SamplingParams(max_new_tokens=sc["max_new_tokens"], eos_id=sc["eos_id"])
    want = [o.tolist() for o in sc["output_ids"]]
    threads = set()
    loop.on_batch = lambda mode, fb: threads.add(torch.cuda.current_stream().cuda_stream)
    for _ in range(2):
        assert [list(o) for o in loop.run_batch_overlap(sc["prompts"], sp)] == want
        assert loop.req_to_token_pool.available_size() == sc["max_requests"]
        assert loop.token_to_kv_pool.available_size() + loop.tree_cache.evictable_size() == sc["max_total_tokens"]
        assert all(s.in_flight == 0 for s in loop.finished)
    assert loop._worker is not None and loop._worker.stream.cuda_stream not in threads  # batches are BUILT on the loop's stream
    loop.close()


def test_a_failure_inside_the_forward_worker_reaches_the_caller(gpu):
    from baby_sglang_amd.serving_loop import SamplingParams, ServingLoop

    class Broken(_StandIn):
        def forward(self, input_ids, positions, forward_batch, input_embeds=None):
            if not forward_batch.is_prefill:
                raise ValueError("decode is broken in this model")
            return super().forward(input_ids, positions, forward_batch)

    loop = ServingLoop(Broken(31).to(gpu), max_total_tokens=256, max_requests=2, overlap_worker=True)
    with pytest.raises(ValueError, match="decode is broken"):
        loop.run_batch_overlap([torch.tensor([1, 2, 3]), torch.tensor([4, 5])], SamplingParams(max_new_tokens=4))
    loop.close()
