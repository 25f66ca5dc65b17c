"""Generate the goldens of the serving loop by RUNNING THE REFERENCE SCHEDULER.

Run in the build container only (``python tests/golden/gen_driver_golden.py``).  The
reference ``Scheduler.run_batch`` (scheduler.py:107-168) is driven on the CPU
  * with a stand-in model whose logits are a fixed function of (token id, position), so
    that scheduling decisions can be replayed anywhere without any attention kernel:
    scenarios with prefix sharing, with radix-cache eviction and with eos stops;
  * with the tiny random-weight ``BabyQwen2ForCausalLM`` of tiny_model.pt (fp32), whose
    greedy output ids pin the whole stack end to end.
Every model call is recorded: mode, input ids, positions, request rows, sequence
lengths, new slots, prefix lengths and the page-table rows of the batch.

scheduler.py imports two modules this image cannot provide (constraints.py ->
lmformatenforcer, not installed; a tokenizer base class of transformers<5 used only in
annotations); both are off this path, so empty placeholder modules satisfy the imports.
"""

from __future__ import annotations

import os
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)

for name in ("lmformatenforcer", "lmformatenforcer.integrations",
             "lmformatenforcer.integrations.transformers", "lmformatenforcer.tokenenforcer"):  # fmt: skip
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["lmformatenforcer"].JsonSchemaParser = object
sys.modules["lmformatenforcer.integrations.transformers"].build_token_enforcer_tokenizer_data = None
sys.modules["lmformatenforcer.tokenenforcer"].TokenEnforcer = object
_tu = types.ModuleType("transformers.tokenization_utils")
_tu.PreTrainedTokenizerBase = object
import transformers  # noqa: E402,F401

sys.modules["transformers.tokenization_utils"] = _tu


class StandInModel(nn.Module):
    """logits[t] = one-hot((7 * id + 3 * position + 1) mod vocab): no attention, no state."""

    def __init__(self, vocab):
        super().__init__()
        self.config = types.SimpleNamespace(num_key_value_heads=1, hidden_size=64, num_attention_heads=1,
                                            num_hidden_layers=1, vocab_size=vocab)
        self.dummy = nn.Parameter(torch.zeros(1))

    def forward(self, input_ids, positions, forward_batch, input_embeds=None):
        return F.one_hot((7 * input_ids + 3 * positions + 1) % self.config.vocab_size, self.config.vocab_size).float()


def run_reference(model, prompts, max_new_tokens, max_total, max_req, eos_id=-1):
    os.environ["BABYSGL_MAX_REQUESTS"] = str(max_req)
    os.environ["BABYSGL_MAX_TOTAL_TOKENS"] = str(max_total)
    import scheduler as ref_sched
    from sample import SamplingParams

    sched = ref_sched.Scheduler(model, tokenizer=None)
    steps = []
    counts = {"evict": 0, "retract": 0}
    real_forward = model.forward

    def spy(input_ids, positions, fb, input_embeds=None):
        steps.append({
            "mode": "prefill" if fb.is_prefill else "decode", "input_ids": input_ids.clone(),
            "positions": positions.clone(), "req_pool_indices": fb.req_pool_indices.clone(),
            "seq_lens": fb.seq_lens.clone(), "out_cache_loc": fb.out_cache_loc.clone(),
            "prefix_lens": None if fb.prefix_lens is None else fb.prefix_lens.clone(),
            "table_rows": fb.req_to_token_pool.req_to_token[fb.req_pool_indices].clone(),
        })  # fmt: skip
        return real_forward(input_ids, positions, fb)

    model.forward = spy
    real_evict = sched.tree_cache.evict

    def counting_evict(n, cb):
        counts["evict"] += 1
        return real_evict(n, cb)

    sched.tree_cache.evict = counting_evict
    from request import Request

    real_reset = Request.reset

    def counting_reset(self):
        counts["retract"] += 1
        return real_reset(self)

    Request.reset = counting_reset
    try:
        sp = SamplingParams(max_new_tokens=max_new_tokens, temperature=0.0, top_k=0, top_p=1.0, do_sample=False, eos_id=eos_id)
        outs = sched.run_batch(prompts, sp)
    finally:
        model.forward = real_forward
        Request.reset = real_reset
    return {"prompts": prompts, "max_new_tokens": max_new_tokens, "max_total_tokens": max_total, "max_requests": max_req,
            "eos_id": eos_id, "output_ids": [torch.tensor(o) for o in outs], "steps": steps, "counts": counts}


def main():
    torch.set_num_threads(2)
    gen = torch.Generator().manual_seed(21)
    vocab = 53
    scenarios = {}
    shared = torch.randint(0, vocab, (12,), generator=gen)
    prompts = [torch.cat([shared, torch.randint(0, vocab, (n,), generator=gen)]) for n in (3, 5, 2, 7, 4, 6)]
    scenarios["shared_prefix"] = run_reference(StandInModel(vocab), prompts, 5, 256, 4)
    # many distinct prompts through a small pool: finished requests fill the tree, later ones evict it
    prompts = [torch.randint(0, vocab, (int(n),), generator=gen) for n in torch.randint(14, 24, (10,), generator=gen)]
    scenarios["eviction"] = run_reference(StandInModel(vocab), prompts, 6, 96, 3)
    # two families of prompts sharing long prefixes + eos stops
    fam = [torch.randint(0, vocab, (16,), generator=gen) for _ in range(2)]
    prompts = [torch.cat([fam[i % 2], torch.randint(0, vocab, (3 + i,), generator=gen)]) for i in range(8)]
    scenarios["families_eos"] = run_reference(StandInModel(vocab), prompts, 9, 160, 4, eos_id=11)
    # (retraction, scheduler.py:322-369, cannot be reached through run_batch: a page-table row
    # holds max_total / max_requests tokens, so the running requests can never need more slots
    # than the pool has)
    for k, v in scenarios.items():
        print(k, "steps", len(v["steps"]), v["counts"], "modes", "".join(s["mode"][0] for s in v["steps"])[:60])

    # the real tiny model, end to end
    import qwen2 as ref_model
    from model_loader import BabyQwenConfig

    g = torch.load(os.path.join(OUT, "tiny_model.pt"), weights_only=True)
    c = g["config"]
    cfg = BabyQwenConfig(vocab_size=c["vocab_size"], hidden_size=c["hidden_size"], intermediate_size=c["intermediate_size"],
                         num_hidden_layers=c["num_hidden_layers"], num_attention_heads=c["num_attention_heads"],
                         num_key_value_heads=c["num_key_value_heads"], max_position_embeddings=c["max_position_embeddings"])
    model = ref_model.BabyQwen2ForCausalLM(cfg).eval()
    model.load_state_dict(g["state_dict"])
    shared = torch.randint(0, c["vocab_size"], (10,), generator=gen)
    prompts = [torch.cat([shared, torch.randint(0, c["vocab_size"], (n,), generator=gen)]) for n in (4, 2, 6, 3, 5)]
    e2e = run_reference(model, prompts, 6, 192, 3)
    for s in e2e["steps"]:
        s.pop("table_rows")  # pinned by the stand-in scenarios; keep this file small
    print("tiny model:", [o.tolist() for o in e2e["output_ids"]], e2e["counts"])
    path = os.path.join(OUT, "driver_traces.pt")
    torch.save({"vocab": vocab, "scenarios": scenarios, "tiny_model_e2e": e2e}, path)
    print(f"driver_traces: {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
