"""The oracle of the steps either side of attention against vectors captured from the
reference's own modules (CPU only): every restatement must be bit exact."""
import pytest
import torch

from conftest import load_golden
from oracle import model_oracle as morc

OPS = load_golden("model_ops")["cases"]


@pytest.mark.parametrize("i", [i for i, c in enumerate(OPS) if c["op"] == "rmsnorm"])
def test_rmsnorm_restatement_is_bit_exact(i):
    c = OPS[i]
    out, res = morc.rmsnorm_ref(c["x"], c["weight"], c["eps"])
    assert res is None and torch.equal(out, c["out"])
    out, res = morc.rmsnorm_ref(c["x"], c["weight"], c["eps"], c["residual"])
    assert torch.equal(out, c["out_with_residual"]) and torch.equal(res, c["residual_out"])


@pytest.mark.parametrize("i", [i for i, c in enumerate(OPS) if c["op"] == "rope"])
def test_rope_restatement_is_bit_exact(i):
    c = OPS[i]
    dim = c["q"].shape[-1]
    assert torch.equal(morc.inv_freq_ref(dim, 1000000.0), c["inv_freq"])
    assert torch.equal(morc.rope_ref(c["positions"], c["q"], c["inv_freq"]), c["q_rot"])
    assert torch.equal(morc.rope_ref(c["positions"], c["k"], c["inv_freq"]), c["k_rot"])


@pytest.mark.parametrize("dt_name,dtype", [("f32", torch.float32), ("bf16", torch.bfloat16)])
def test_whole_model_restatement_is_bit_exact(dt_name, dtype):
    g = load_golden("tiny_model")
    cfg, pool = g["config"], g["pool"]
    weights = {k: v.to(dtype) for k, v in g["state_dict"].items()}
    layers, hkv, dim = cfg["num_hidden_layers"], cfg["num_key_value_heads"], cfg["head_dim"]
    k_slabs = [torch.zeros((pool["size"] + 1, hkv, dim), dtype=dtype) for _ in range(layers)]
    v_slabs = [torch.zeros((pool["size"] + 1, hkv, dim), dtype=dtype) for _ in range(layers)]
    kinds = []
    for step in g["runs"][dt_name]:
        logits = morc.model_forward_ref(
            weights, cfg, step["input_ids"], step["positions"], k_slabs, v_slabs, step["req_to_token"],
            step["req_pool_indices"], step["seq_lens"], step["out_cache_loc"], step["prefix_lens"],
        )
        assert torch.equal(logits, step["logits"]), step["kind"]
        kinds.append(step["kind"])
    assert kinds == ["prefill", "decode", "decode", "extend"]
