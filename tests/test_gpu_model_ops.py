"""GPU parity of the steps either side of attention (RMSNorm, RoPE + QKV split + KV write)
and of the whole decoder built on them, against vectors captured from the reference and
against the CPU oracle on seeded inputs.  Everything goes through the C ABI.

The kernels round exactly where the reference rounds, so bf16 / f16 results are bit exact
except where a last-place difference of cos / sin / a sum order moves a rounding: the
tests demand that almost every element is identical and that no element is more than
one unit in the last place (of the dtype) away.  fp32 results carry those last-place
differences unrounded and are held to 4e-6 relative + 1e-6 absolute.
"""
import types

import pytest
import torch

from conftest import load_golden
from oracle import model_oracle as morc

pytestmark = pytest.mark.gpu

DT = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}
ULP = {torch.float32: 2.0**-23, torch.float16: 2.0**-10, torch.bfloat16: 2.0**-7}


def assert_ulp_close(got, want, what, min_equal=0.995, max_ulps=1.0):
    got, want = got.detach().cpu(), want.detach().cpu()
    assert got.shape == want.shape and got.dtype == want.dtype, what
    if got.dtype == torch.float32:
        # nothing rounds to a coarser grid here, so a last-place difference of cos / sin / rsqrt
        # shows in most elements (and, relative to a result that cancels, as many ulps): bound
        # the error against the operands' size, O(1), instead
        err = (got.double() - want.double()).abs()
        bound = 4e-6 * want.double().abs() + 1e-6
        assert bool((err <= bound).all()), f"{what}: max abs err {err.max().item():.3e}"
        return
    same = (got == want).double().mean().item()
    # one unit in the last place of the larger of the two magnitudes (power-of-two floor)
    mag = torch.maximum(got.double().abs(), want.double().abs()).clamp_min(1e-30)
    ulp = torch.exp2(torch.floor(torch.log2(mag))) * ULP[got.dtype]
    worst = ((got.double() - want.double()).abs() / ulp).max().item()
    assert same >= min_equal and worst <= max_ulps + 1e-9, f"{what}: {same:.4%} identical, worst {worst:.2f} ulp"


OPS = load_golden("model_ops")["cases"]


@pytest.mark.parametrize("i", [i for i, c in enumerate(OPS) if c["op"] == "rmsnorm"])
def test_rmsnorm_golden(gpu, i):
    from baby_sglang_amd import _hip

    c = OPS[i]
    x, w, r = c["x"].to(gpu), c["weight"].to(gpu), c["residual"].to(gpu)
    out, res = _hip.rmsnorm(x, w, c["eps"])
    assert res is None
    assert_ulp_close(out, c["out"], f"rmsnorm {c['dtype']}")
    out, res = _hip.rmsnorm(x, w, c["eps"], r)
    assert torch.equal(res.cpu(), c["residual_out"])  # the residual add itself is exact
    assert_ulp_close(out, c["out_with_residual"], f"rmsnorm+residual {c['dtype']}")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("rows,hidden", [(1, 1536), (64, 1536), (7, 3584), (3, 896), (2, 8192), (2, 16384)])
def test_rmsnorm_random(gpu, dtype, rows, hidden):
    from baby_sglang_amd import _hip

    gen = torch.Generator().manual_seed(rows * 10000 + hidden)
    parent = (2.0 * torch.randn((rows, hidden + 64), generator=gen)).to(dtype)
    x = parent[:, :hidden]  # strided rows
    r = torch.randn((rows, hidden), generator=gen).to(dtype)
    w = (1.0 + 0.2 * torch.randn(hidden, generator=gen)).to(dtype)
    want, want_res = morc.rmsnorm_ref(x, w, 1e-6, r)
    out, res = _hip.rmsnorm(parent.to(gpu)[:, :hidden], w.to(gpu), 1e-6, r.to(gpu))
    assert torch.equal(res.cpu(), want_res)
    assert_ulp_close(out, want, f"rmsnorm {dtype} {rows}x{hidden}")


@pytest.mark.parametrize("i", [i for i, c in enumerate(OPS) if c["op"] == "rope"])
def test_rope_golden(gpu, i):
    from baby_sglang_amd import _hip

    c = OPS[i]
    q, k = c["q"], c["k"]
    n_tok, hq, dim = q.shape
    hkv = k.shape[1]
    v = torch.arange(n_tok * hkv * dim, dtype=torch.float32).view(n_tok, hkv * dim).to(q.dtype) / 64
    qkv = torch.cat((q.reshape(n_tok, -1), k.reshape(n_tok, -1), v), dim=-1).to(gpu)
    # cache write into scrambled slots of a slab with a wider row (interleaved K|V arena)
    n_slots = 20
    arena = torch.zeros((n_slots, 2, hkv, dim), dtype=q.dtype, device=gpu)
    k_slab, v_slab = arena[:, 0], arena[:, 1]
    loc = torch.tensor([7, 3, 19, 1, 12, 5], dtype=torch.int32, device=gpu)
    q_rot, k_rot, v_out = _hip.rope_kv_write(qkv, c["positions"].to(gpu), c["inv_freq"].to(gpu), hq, hkv, dim,
                                             k_slab, v_slab, loc, return_kv=True)
    assert_ulp_close(q_rot, c["q_rot"], f"rope q {c['dtype']}")
    assert_ulp_close(k_rot, c["k_rot"], f"rope k {c['dtype']}")
    assert torch.equal(v_out.cpu().view(n_tok, -1), v)
    # cache rows hold exactly what was returned; untouched slots stay zero
    assert torch.equal(k_slab[loc.long()], k_rot) and torch.equal(v_slab[loc.long()], v_out)
    keep = torch.ones(n_slots, dtype=torch.bool)
    keep[loc.long().cpu()] = False
    assert float(arena[keep.to(gpu)].abs().max()) == 0.0


def test_rope_bad_slots_and_errors(gpu):
    from baby_sglang_amd import _hip

    hq, hkv, dim = 4, 2, 64
    qkv = torch.randn((3, (hq + 2 * hkv) * dim), device=gpu).to(torch.bfloat16)
    pos = torch.tensor([0, 5, 9], device=gpu)
    inv = morc.inv_freq_ref(dim, 1e6).to(gpu)
    k_slab = torch.zeros((8, hkv, dim), dtype=torch.bfloat16, device=gpu)
    v_slab = torch.zeros_like(k_slab)
    loc = torch.tensor([2, -1, 99], dtype=torch.int32, device=gpu)  # two slots outside the slab: skipped
    _hip.rope_kv_write(qkv, pos, inv, hq, hkv, dim, k_slab, v_slab, loc)
    torch.cuda.synchronize()
    assert float(k_slab[2].abs().max()) > 0 and float(k_slab[[0, 1, 3, 4, 5, 6, 7]].abs().max()) == 0.0
    with pytest.raises(ValueError):
        _hip.rope_kv_write(qkv[:, :-8], pos, inv, hq, hkv, dim, k_slab, v_slab, loc)
    with pytest.raises(RuntimeError):
        _hip.rmsnorm(torch.zeros((2, 64)), torch.ones(64), 1e-6)  # CPU tensors: no CPU path


def build_model(g, dtype, dev):
    import baby_sglang_amd as pkg

    cfg = types.SimpleNamespace(**g["config"], tie_word_embeddings=False, hidden_act="silu", rope_scaling=None)
    model = pkg.BabyQwen2ForCausalLM(cfg).eval()
    missing = model.load_state_dict(g["state_dict"], strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    model = model.to(dtype).to(dev)
    pool = g["pool"]
    r2t = pkg.ReqToTokenPool(pool["max_req"], pool["max_ctx"], str(dev))
    kv = pkg.MHATokenToKVPool(pool["size"], dtype, cfg.num_key_value_heads, cfg.head_dim,
                              cfg.num_hidden_layers, str(dev))
    kv.arena.zero_()
    return pkg, model, r2t, kv


@pytest.mark.parametrize("dt_name,rtol,atol", [("f32", 2e-4, 2e-5), ("bf16", 6e-2, 6e-2)])
def test_whole_model_against_reference_logits(gpu, dt_name, rtol, atol):
    """Prefill, two decode steps and an extend over a cached prefix of a tiny random-weight
    model: logits of the HIP-path model against the reference model's (fp32: the GEMMs'
    summation order is the only difference; bf16: every layer rounds, so the bound is loose
    and the greedy tokens are compared as well)."""
    g = load_golden("tiny_model")
    pkg, model, r2t, kv = build_model(g, DT[dt_name], gpu)
    for step in g["runs"][dt_name]:
        r2t.req_to_token.copy_(step["req_to_token"])
        common = dict(input_ids=step["input_ids"].to(gpu), req_pool_indices=step["req_pool_indices"].to(gpu),
                      seq_lens=step["seq_lens"].to(gpu), out_cache_loc=step["out_cache_loc"].to(gpu),
                      req_to_token_pool=r2t, token_to_kv_pool=kv)
        if step["prefix_lens"] is None:
            fb = pkg.SimplifiedForwardBatch.create_decode_batch(**common)
        else:
            fb = pkg.SimplifiedForwardBatch.create_prefill_batch(prefix_lens=step["prefix_lens"].to(gpu), **common)
        assert torch.equal(fb.positions.cpu(), step["positions"])
        logits = model(common["input_ids"], fb.positions, fb)
        want = step["logits"]
        assert logits.shape == want.shape and logits.dtype == want.dtype
        ok = torch.isclose(logits.float().cpu(), want.float(), rtol=rtol, atol=atol)
        assert bool(ok.all()), f"{step['kind']}: {(~ok).sum().item()} logits off, max " \
                               f"{(logits.float().cpu() - want.float()).abs().max().item():.3e}"
        if dt_name == "f32":
            assert torch.equal(logits.float().cpu().argmax(-1), want.float().argmax(-1)), step["kind"]


@pytest.mark.parametrize("dt_name,rtol,atol", [("f32", 2e-4, 2e-5), ("bf16", 6e-2, 6e-2)])
def test_tensor_parallel_model_against_reference_logits(gpu, dt_name, rtol, atol):
    """The same runs with the model cut over TWO ranks (threads of this process on the one GPU,
    ``LocalGroup``): each rank holds one of the two q heads, its own copy of the single kv head in
    its own pool, half of the MLP; ``o_proj`` and ``down_proj`` are row-parallel and summed with
    ``tp_all_reduce_sum`` (tp_runtime.py:28-33).  Every rank's logits against the reference
    model's (unsharded) goldens, to the tolerances of the unsharded test."""
    import baby_sglang_amd as pkg

    g = load_golden("tiny_model")
    cfg = types.SimpleNamespace(**g["config"], tie_word_embeddings=False, hidden_act="silu", rope_scaling=None)
    dtype, world = DT[dt_name], 2
    group = pkg.LocalGroup(world)

    def rank_body(rank):
        layout = pkg.TPLayout(rank, world, cfg.num_attention_heads, cfg.num_key_value_heads, replicate_kv=True)
        model = pkg.BabyQwen2ForCausalLM(cfg, tp_layout=layout, tp_group=group).eval()
        missing = model.load_state_dict(pkg.shard_state_dict(g["state_dict"], cfg, layout), strict=True)
        assert not missing.missing_keys and not missing.unexpected_keys
        model = model.to(dtype).to(gpu)
        pool = g["pool"]
        r2t = pkg.ReqToTokenPool(pool["max_req"], pool["max_ctx"], str(gpu))
        kv = pkg.MHATokenToKVPool(pool["size"], dtype, layout.local_kv_heads, cfg.head_dim, cfg.num_hidden_layers, str(gpu))
        kv.arena.zero_()
        outs = []
        for step in g["runs"][dt_name]:
            r2t.req_to_token.copy_(step["req_to_token"])
            common = dict(input_ids=step["input_ids"].to(gpu), req_pool_indices=step["req_pool_indices"].to(gpu),
                          seq_lens=step["seq_lens"].to(gpu), out_cache_loc=step["out_cache_loc"].to(gpu),
                          req_to_token_pool=r2t, token_to_kv_pool=kv)
            if step["prefix_lens"] is None:
                fb = pkg.SimplifiedForwardBatch.create_decode_batch(**common)
            else:
                fb = pkg.SimplifiedForwardBatch.create_prefill_batch(prefix_lens=step["prefix_lens"].to(gpu), **common)
            outs.append(model(common["input_ids"], fb.positions, fb).float().cpu())
        return outs

    per_rank = group.run(rank_body)
    for rank, outs in enumerate(per_rank):
        for step, logits in zip(g["runs"][dt_name], outs):
            want = step["logits"].float()
            ok = torch.isclose(logits, want, rtol=rtol, atol=atol)
            assert bool(ok.all()), f"rank {rank} {step['kind']}: {(~ok).sum().item()} logits off, max {(logits - want).abs().max().item():.3e}"
            if dt_name == "f32":
                assert torch.equal(logits.argmax(-1), want.argmax(-1)), step["kind"]
    for a, b in zip(per_rank[0], per_rank[1]):
        assert torch.equal(a, b)  # (the sums are taken in rank order on every rank)
