"""Checkpoint tensors -> tensor-parallel shards, CPU side: the oracle's restatement of the reference's
shard rules (ColumnParallelLinear / RowParallelLinear::load_weight, linear.rs:154-171,249-267; packed qkv and gate_up,
:300-340,378-454; vocabulary shards, embed_head.rs:57-59) reassembles to the full tensors, through a .safetensors file.
The device side of the same rules is tests/test_engine_gpu.py::test_checkpoint_*."""
import numpy as np
import pytest

from oracle import model_oracle as mo


def _state(mcfg, rng):
    Hd, I, H, KVH, D, V = mcfg.hidden_size, mcfg.intermediate_size, mcfg.num_attention_heads, mcfg.num_key_value_heads, mcfg.hd(), mcfg.vocab_size
    w = lambda *shape: (rng.standard_normal(shape) * 0.05).astype(np.float16)
    sd = {"model.embed_tokens.weight": w(V, Hd), "model.norm.weight": w(Hd), "lm_head.weight": w(V, Hd)}
    for l in range(mcfg.num_hidden_layers):
        pre = f"model.layers.{l}."
        sd.update({pre + "input_layernorm.weight": w(Hd), pre + "post_attention_layernorm.weight": w(Hd),
                   pre + "self_attn.q_proj.weight": w(H * D, Hd), pre + "self_attn.k_proj.weight": w(KVH * D, Hd),
                   pre + "self_attn.v_proj.weight": w(KVH * D, Hd), pre + "self_attn.o_proj.weight": w(Hd, H * D),
                   pre + "mlp.gate_proj.weight": w(I, Hd), pre + "mlp.up_proj.weight": w(I, Hd), pre + "mlp.down_proj.weight": w(Hd, I)})
    return sd


@pytest.mark.parametrize("tp", [1, 2])
def test_oracle_shards_reassemble_through_safetensors(tmp_path, tp):
    from safetensors.numpy import load_file, save_file
    mcfg = mo.small(seed=1)
    sd = _state(mcfg, np.random.default_rng(5))
    path = str(tmp_path / "m.safetensors")
    save_file(sd, path)
    sd2 = load_file(path)
    ranks = [mo.OracleModel(mcfg, 2, 16, True, r, tp) for r in range(tp)]
    for rk in ranks:
        assert rk.load_state_dict(sd2) == []
    H, KVH, D, I = ranks[0].H, ranks[0].KVH, ranks[0].D, ranks[0].I
    f = lambda k: sd[k].astype(np.float32)
    for l in range(mcfg.num_hidden_layers):
        pre = f"model.layers.{l}."
        assert np.array_equal(np.concatenate([rk.layers[l]["qkv"][:H * D] for rk in ranks]), f(pre + "self_attn.q_proj.weight"))
        assert np.array_equal(np.concatenate([rk.layers[l]["qkv"][H * D:(H + KVH) * D] for rk in ranks]), f(pre + "self_attn.k_proj.weight"))
        assert np.array_equal(np.concatenate([rk.layers[l]["qkv"][(H + KVH) * D:] for rk in ranks]), f(pre + "self_attn.v_proj.weight"))
        assert np.array_equal(np.concatenate([rk.layers[l]["o"] for rk in ranks], 1), f(pre + "self_attn.o_proj.weight"))
        assert np.array_equal(np.concatenate([rk.layers[l]["gate_up"][:I] for rk in ranks]), f(pre + "mlp.gate_proj.weight"))
        assert np.array_equal(np.concatenate([rk.layers[l]["gate_up"][I:] for rk in ranks]), f(pre + "mlp.up_proj.weight"))
        assert np.array_equal(np.concatenate([rk.layers[l]["down"] for rk in ranks], 1), f(pre + "mlp.down_proj.weight"))
        assert all(np.array_equal(rk.layers[l]["ln1"], f(pre + "input_layernorm.weight")) for rk in ranks)
    assert np.array_equal(np.concatenate([rk.lm_head for rk in ranks]), f("lm_head.weight"))
    assert all(np.array_equal(rk.embed, f("model.embed_tokens.weight")) for rk in ranks)


def test_oracle_loader_errors_and_unknown_names():
    mcfg = mo.small(seed=1)
    om = mo.OracleModel(mcfg, 2, 16, True)
    with pytest.raises(ValueError, match="Partition weight shape mismatch"):
        om.load_state_dict({"layers.0.mlp.down_proj.weight": np.zeros((mcfg.hidden_size, mcfg.intermediate_size + 1), np.float16)})
    assert om.load_state_dict({"layers.0.self_attn.q_norm.weight": np.ones(mcfg.hd(), np.float16),
                               "layers.7.mlp.down_proj.weight": np.zeros((2, 2), np.float16)}) == \
        ["layers.0.self_attn.q_norm.weight", "layers.7.mlp.down_proj.weight"]
    tied = mo.tiny(tie_word_embeddings=True)
    ot = mo.OracleModel(tied, 2, 16, True)
    e = (np.random.default_rng(0).standard_normal((tied.vocab_size, tied.hidden_size)) * 0.1).astype(np.float16)
    ot.load_state_dict({"embed_tokens.weight": e})
    assert np.array_equal(ot.lm_head, e.astype(np.float32))
    with pytest.raises(ValueError, match="tie_word_embeddings"):
        ot.load_state_dict({"lm_head.weight": e})
