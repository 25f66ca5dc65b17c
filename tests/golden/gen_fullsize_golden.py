"""Generate the full-size golden outputs (tests/golden/full_*.pt) by RUNNING THE REFERENCE.

Run in the build container only (``python tests/golden/gen_fullsize_golden.py``): it puts
``/root/reference`` on ``sys.path``, imports the reference's ``memory_pool`` / ``forward_batch`` /
``attn_backend`` and runs ``SimpleAttentionBackend.forward`` on the CPU over the inputs of
``fullsize_inputs.build`` -- decode batches with requests of up to 4095 tokens (attn_backend.py:112-149)
and extends of 512 / 1024 new tokens over 1536 / 0 cached ones (attn_backend.py:34-110) on the
1.5B and 7B head shapes.  Only ``rows_kept`` rows of the reference's output, the new K/V rows it wrote
and checksums of the inputs are stored; the reference never travels.
"""
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")

import attn_backend as ref_backend  # noqa: E402
import forward_batch as ref_fb  # noqa: E402
import memory_pool as ref_pool  # noqa: E402
from fullsize_inputs import CASES, build, checksums, host_fingerprint  # noqa: E402


def run_reference(case):
    hkv, dim = case["n_kv_heads"], case["head_dim"]
    dtype = case["q"].dtype
    table = case["req_to_token"]
    r2t = ref_pool.ReqToTokenPool(size=table.shape[0], max_context_len=table.shape[1], device="cpu")
    r2t.req_to_token.copy_(table)
    kv = ref_pool.MHATokenToKVPool(size=case["k_slab"].shape[0] - 1, dtype=dtype, head_num=hkv, head_dim=dim,
                                   layer_num=1, device="cpu")
    kv.k_buffer[0].copy_(case["k_slab"])
    kv.v_buffer[0].copy_(case["v_slab"])
    q, k = case["q"], case["k"]
    v = case["qkv"][:, case["v_col0"] :].view(q.shape[0], hkv, dim)
    common = dict(input_ids=torch.zeros(q.shape[0], dtype=torch.int64), req_pool_indices=case["req_pool_indices"],
                  seq_lens=case["seq_lens"], out_cache_loc=case["out_cache_loc"], req_to_token_pool=r2t, token_to_kv_pool=kv)
    if case["kind"] == "prefill":
        fb = ref_fb.SimplifiedForwardBatch.create_prefill_batch(prefix_lens=case["prefix_lens"], **common)
    else:
        fb = ref_fb.SimplifiedForwardBatch.create_decode_batch(**common)
    layer = types.SimpleNamespace(layer_id=0, scaling=dim**-0.5)
    out = ref_backend.SimpleAttentionBackend.forward(q, k, v, layer, fb, save_kv_cache=True)
    loc = case["out_cache_loc"].long()
    return out, kv.k_buffer[0][loc].clone(), kv.v_buffer[0][loc].clone()


def main():
    for name in CASES:
        case = build(name)
        out, k_rows, v_rows = run_reference(case)
        keep = case["rows_kept"]
        path = os.path.join(HERE, name + ".pt")
        torch.save({"out_rows": out[keep].clone(), "rows_kept": keep, "out_shape": torch.tensor(out.shape),
                    "k_rows_kept": k_rows[keep].clone(), "v_rows_kept": v_rows[keep].clone(),
                    "checksums": checksums(case), "host": host_fingerprint()}, path)
        print(f"{name}: {tuple(out.shape)} -> {len(keep)} rows kept, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
