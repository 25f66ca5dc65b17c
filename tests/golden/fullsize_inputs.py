"""Inputs of the full-size golden cases, rebuilt from a seed.

The cases of ``gen_golden.py`` carry their inputs; at the sizes the bench quotes (requests of 2048 -
4095 tokens, the 7B extend of 512 new tokens over 1536 cached) the KV slabs alone would be tens of
megabytes per case.  So these cases keep only what cannot be rebuilt -- rows of the output the REFERENCE
computed (``gen_fullsize_golden.py`` imports it in the build container) -- and both the generator and
the tests build the inputs with the function below (torch's CPU generator: the same stream of numbers
wherever the same torch runs; checksums of the inputs are stored with the outputs and compared first).

A case is a dict in the layout of the other goldens (``conftest.load_golden``), so the same helpers run it.
"""
import torch

CONFIGS = {  # name: (n_q_heads, n_kv_heads, head_dim)
    "q15": (12, 2, 128),  # Qwen2.5-1.5B
    "q7": (28, 4, 128),  # Qwen2.5-7B
}
DTYPES = {"f16": torch.float16, "bf16": torch.bfloat16}

# name: (kind, config, dtype, seq_lens, prefix_lens or None, output rows kept, seed)
CASES = {
    "full_decode_q15_bf16": ("decode", "q15", "bf16", [4095, 2048, 3000, 1, 513, 4095, 1024, 2047], None, 8, 101),
    "full_decode_q7_f16": ("decode", "q7", "f16", [4095, 4095, 2048, 100], None, 4, 102),
    "full_prefill_q7_bf16": ("prefill", "q7", "bf16", [2048, 2048], [1536, 1536], 96, 103),
    "full_prefill_q15_f16": ("prefill", "q15", "f16", [2048, 1024], [1536, 0], 96, 104),
}


def build(name):
    kind, cfg, dt, seq_lens, prefix_lens, n_keep, seed = CASES[name]
    hq, hkv, dim = CONFIGS[cfg]
    dtype = DTYPES[dt]
    gen = torch.Generator().manual_seed(seed)
    batch = len(seq_lens)
    size = sum(seq_lens) + 8
    max_ctx = max(seq_lens) + 1
    perm = (torch.randperm(size, generator=gen) + 1).to(torch.int32)
    table = torch.zeros((batch + 2, max_ctx), dtype=torch.int32)
    rows = torch.randperm(batch + 2, generator=gen)[:batch].to(torch.int64)
    used = 0
    for i, n in enumerate(seq_lens):
        table[rows[i], :n] = perm[used : used + n]
        used += n
    k_slab = torch.randn((size + 1, hkv, dim), generator=gen).to(dtype)
    v_slab = torch.randn((size + 1, hkv, dim), generator=gen).to(dtype)
    seq = torch.tensor(seq_lens, dtype=torch.int64)
    if kind == "decode":
        loc = table[rows, seq - 1].clone()
        prefix = None
    else:
        prefix = torch.tensor(prefix_lens, dtype=torch.int64)
        loc = torch.cat([table[rows[i], prefix_lens[i] : seq_lens[i]] for i in range(batch)]).clone()
    n_tok = int(loc.numel())
    q = torch.randn((n_tok, hq, dim), generator=gen).to(dtype)
    k = torch.randn((n_tok, hkv, dim), generator=gen).to(dtype)
    qkv = torch.randn((n_tok, (hq + 2 * hkv) * dim), generator=gen).to(dtype)
    keep = torch.sort(torch.randperm(n_tok, generator=gen)[: min(n_keep, n_tok)]).values
    case = {
        "kind": kind, "n_q_heads": hq, "n_kv_heads": hkv, "head_dim": dim, "save_kv_cache": True,
        "q": q, "k": k, "qkv": qkv, "v_col0": (hq + hkv) * dim,
        "k_slab": k_slab, "v_slab": v_slab, "req_to_token": table,
        "req_pool_indices": rows, "seq_lens": seq, "out_cache_loc": loc, "rows_kept": keep,
    }  # fmt: skip
    if kind == "prefill":
        case["prefix_lens"] = prefix
        case["extended_lens"] = seq - prefix
    return case


def checksums(case):
    """A few float64 sums of the inputs: the stored ones must be reproduced before any output is compared."""
    return torch.stack([case[k].double().sum() for k in ("q", "k", "qkv", "k_slab", "v_slab")]
                       + [case["req_to_token"].double().sum(), case["out_cache_loc"].double().sum()])


def host_fingerprint():
    """What decides how torch's CPU kernels cut a long reduction: threads and vector width."""
    return f"{torch.get_num_threads()} threads, {torch.backends.cpu.get_cpu_capability()}, torch {torch.__version__}"
