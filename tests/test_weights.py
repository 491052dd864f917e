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


def _write_safetensors(path, tensors):
    """Minimal writer (the format: u64 header length, JSON header, payload) so that a BF16 file exists without torch."""
    import json
    header, blobs, off = {}, [], 0
    for name, (tag, arr) in tensors.items():
        raw = np.ascontiguousarray(arr).tobytes()
        header[name] = dict(dtype=tag, shape=list(arr.shape), data_offsets=[off, off + len(raw)])
        blobs.append(raw); off += len(raw)
    hj = json.dumps(header).encode()
    hj += b" " * (-len(hj) % 8)
    with open(path, "wb") as f:
        f.write(len(hj).to_bytes(8, "little")); f.write(hj)
        for b in blobs:
            f.write(b)


def test_safetensors_reader_bf16_f16_f32(tmp_path):
    """safetensors' numpy front end raises on BF16 (the dtype of the published Qwen3 checkpoints); the
    package's own reader hands bf16 down as 16-bit patterns and agrees with the safetensors package on f16 / f32."""
    import nvr_import
    nvr = nvr_import.load()
    rng = np.random.default_rng(3)
    f32 = rng.standard_normal((5, 8)).astype(np.float32)
    bf = (f32.view(np.uint32) >> 16).astype(np.uint16)                      # truncated bf16 bit patterns
    f16 = rng.standard_normal((3, 4)).astype(np.float16)
    path = str(tmp_path / "m.safetensors")
    _write_safetensors(path, {"a.weight": ("BF16", bf), "b.weight": ("F16", f16), "c.weight": ("F32", f32), "d": ("F32", np.float32(2.5).reshape(()))})
    got = dict(nvr.iter_safetensors(path))
    assert got["a.weight"].dtype == np.uint16 and np.array_equal(got["a.weight"], bf)
    assert np.array_equal(got["b.weight"], f16) and np.array_equal(got["c.weight"], f32) and got["d"].shape == ()
    from safetensors.numpy import load_file, save_file
    p2 = str(tmp_path / "n.safetensors")
    save_file({"x": f16, "y": f32}, p2)
    ref, mine = load_file(p2), dict(nvr.iter_safetensors(p2))
    assert all(np.array_equal(ref[k], mine[k]) for k in ref)
    _write_safetensors(path, {"q": ("I64", np.arange(4))})
    with pytest.raises(nvr.NvrError):
        list(nvr.iter_safetensors(path))
