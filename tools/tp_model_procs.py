"""One rank of the tensor-parallel model as a PROCESS of its own (tools/rehearse_tp_model.sh starts
two of them on the one GPU): init_tp() from the torchrun-style environment, gloo as the transport
(RCCL wants a GPU per rank), column-parallel qkv / gate_up, row-parallel o_proj / down_proj summed
with tp_all_reduce_sum (reference: tp_runtime.py:10-33, qwen2.py:179-197).  Every rank's logits over
the golden runs (prefill, two decode steps, an extend over a cached prefix) are compared with the
reference model's unsharded logits, to the bars of tests/test_gpu_model_ops.py.  With
BABYSGL_P2P_ALLREDUCE=1 the sums go through the library's one-shot all-reduce instead of gloo."""
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import baby_sglang_amd as pkg  # noqa: E402

BARS = {"f32": (torch.float32, 2e-4, 2e-5), "bf16": (torch.bfloat16, 6e-2, 6e-2)}


def main():
    rank, world, group = pkg.init_tp()
    assert world > 1, "start me through tools/rehearse_tp_model.sh"
    dev = torch.device("cuda", int(os.getenv("LOCAL_RANK", "0")))
    one_shot = os.getenv("BABYSGL_P2P_ALLREDUCE", "0") not in ("", "0")
    if one_shot:  # the sums on the device, through windows mapped by IPC handles, instead of gloo's host sums
        assert pkg.use_p2p_all_reduce(group), f"rank {rank}: no one-shot all-reduce: {pkg.tp_runtime._P2P_WHY[0]}"
    g = torch.load(os.path.join(ROOT, "tests", "golden", "tiny_model.pt"), weights_only=True)
    cfg = types.SimpleNamespace(**g["config"], tie_word_embeddings=False, hidden_act="silu", rope_scaling=None)
    for dt_name, (dtype, rtol, atol) in BARS.items():
        layout = pkg.TPLayout(rank, world, cfg.num_attention_heads, cfg.num_key_value_heads, replicate_kv=True)
        model = pkg.BabyQwen2ForCausalLM(cfg, tp_layout=layout, tp_group=group).eval()
        model.load_state_dict(pkg.shard_state_dict(g["state_dict"], cfg, layout), strict=True)
        model = model.to(dtype).to(dev)
        pool = g["pool"]
        r2t = pkg.ReqToTokenPool(pool["max_req"], pool["max_ctx"], str(dev))
        kv = pkg.MHATokenToKVPool(pool["size"], dtype, layout.local_kv_heads, cfg.head_dim, cfg.num_hidden_layers, str(dev))
        kv.arena.zero_()
        worst = 0.0
        for step in g["runs"][dt_name]:
            r2t.req_to_token.copy_(step["req_to_token"])
            common = dict(input_ids=step["input_ids"].to(dev), req_pool_indices=step["req_pool_indices"].to(dev),
                          seq_lens=step["seq_lens"].to(dev), out_cache_loc=step["out_cache_loc"].to(dev),
                          req_to_token_pool=r2t, token_to_kv_pool=kv)
            if step["prefix_lens"] is None:
                fb = pkg.SimplifiedForwardBatch.create_decode_batch(**common)
            else:
                fb = pkg.SimplifiedForwardBatch.create_prefill_batch(prefix_lens=step["prefix_lens"].to(dev), **common)
            logits = model(common["input_ids"], fb.positions, fb).float().cpu()
            want = step["logits"].float()
            ok = torch.isclose(logits, want, rtol=rtol, atol=atol)
            assert bool(ok.all()), f"rank {rank} {dt_name} {step['kind']}: {(~ok).sum().item()} logits off"
            if dt_name == "f32":
                assert torch.equal(logits.argmax(-1), want.argmax(-1)), step["kind"]
            worst = max(worst, float((logits - want).abs().max()))
        print(f"rank {rank}/{world} {dt_name}: {len(g['runs'][dt_name])} runs equal the reference model's logits "
              f"(max |diff| {worst:.2e}; rtol {rtol}, atol {atol})", flush=True)
    if one_shot:
        status, calls = pkg.tp_runtime._P2P[1].status()
        assert status == 0 and calls > 1, (status, calls)
        print(f"rank {rank}/{world}: {calls} of the sums were one-shot all-reduces through IPC windows", flush=True)
        pkg.release_p2p_all_reduce()
    torch.cuda.synchronize()
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
