"""Config.dtype = "bfloat16" (reference src/config.rs:51,113-116) on the HIP path (-m gpu).

Every kernel of the library exists twice — fp16 (namespace nvr::k) and bfloat16 (nvr::kb, the same sources compiled with -DNVR_BF16:
bf16 MFMA / dot2 / transposed LDS reads, round-to-nearest-even at the oracle's rounding points) — and the runner picks one build at
creation.  These tests run the bf16 engine against the oracle in its bf16-faithful mode (oracle.round_bf16 wherever the fp16 mode
rounds to fp16; KV cache, activations, weights and residual stream all bf16, logits f32) through the SAME scenarios the fp16 build
is held to: greedy decode (hipGraph and eager), every kernel-variant shape, preemption + prefix cache, chunked prefill, the
four-launch decode chain, the shared-prefix attention pass, q/k-norm checkpoints, launch-ahead, and the weight-loading conversions.
Tolerance: BF16_TOL = 8 x the fp16 tolerance (8 mantissa bits against 11); the measured distances are in the docstring of BF16_TOL."""
import numpy as np
import pytest

import nvr_import
import oracle
from oracle import model_oracle as mo
from test_engine_gpu import BF16_TOL, _model_cfgs, _run_pair, _run_product

nvr = nvr_import.load()
pytestmark = pytest.mark.gpu
BF = dict(dtype="bfloat16")


def test_bfloat16_small_model_greedy_graph_and_eager():
    mcfg = mo.small()
    ecfg = dict(max_num_seqs=8, max_num_batched_tokens=512, max_model_len=512, kvcache_block_size=16, num_kvcache_blocks=64)
    prompts = [oracle.fill_tokens(n, 1, i, mcfg.vocab_size).tolist() for i, n in enumerate([5, 16, 17, 40, 1])]
    sps = [dict(temperature=0.0, max_tokens=24, ignore_eos=True)] * len(prompts)
    r = _run_pair(mcfg, ecfg, prompts, sps, **BF)
    assert r["decode_steps"] >= 23 and len(r["finished"]) == 5 and r["near_ties"] <= 2, r
    assert 1e-4 < r["max_err"] < BF16_TOL                      # not bit-equal to anything: a bf16 pipeline against a bf16 restatement
    e = _run_pair(mcfg, ecfg, prompts, sps, enforce_eager=True, **BF)
    assert e["finished"] == r["finished"]


@pytest.mark.parametrize("shape", [
    dict(h=128, i=256, H=2, KVH=2, D=64, V=512, bs=4, lens=[3, 9, 21, 70]),
    dict(h=256, i=384, H=8, KVH=1, D=64, V=1008, bs=16, lens=[40, 5, 17]),
    dict(h=512, i=768, H=4, KVH=4, D=128, V=2048, bs=48, lens=[100, 47, 140, 1]),
    dict(h=2048, i=1024, H=16, KVH=4, D=128, V=4096, bs=32, lens=[33, 64, 2]),
    dict(h=4096, i=512, H=8, KVH=8, D=64, V=256, bs=16, lens=[20, 8]),
])
def test_bfloat16_engine_parity_across_kernel_variants(shape):
    """The shapes of test_engine_gpu.test_engine_parity_across_kernel_variants (GQA groups 1/2/4/8, head dims 64/128, block sizes 4-48,
    hidden 128-4096) on the bf16 build."""
    mcfg = mo.ModelConfig(vocab_size=shape["V"], hidden_size=shape["h"], intermediate_size=shape["i"], num_hidden_layers=2,
                          num_attention_heads=shape["H"], num_key_value_heads=shape["KVH"], head_dim=shape["D"], rms_norm_eps=1e-6,
                          rope_theta=10000.0, tie_word_embeddings=False, max_position_embeddings=512, init_std=0.05, seed=21)
    ecfg = dict(max_num_seqs=8, max_num_batched_tokens=512, max_model_len=256, kvcache_block_size=shape["bs"], num_kvcache_blocks=400 // shape["bs"] * 8)
    prompts = [oracle.fill_tokens(n, 4, i, mcfg.vocab_size).tolist() for i, n in enumerate(shape["lens"])]
    sps = [dict(temperature=0.0, max_tokens=10, ignore_eos=True)] * len(prompts)
    r = _run_pair(mcfg, ecfg, prompts, sps, **BF)
    assert r["near_ties"] <= 3, r


def test_bfloat16_preemption_prefix_cache_and_chunked_prefill():
    mcfg = mo.small(seed=5)
    ecfg = dict(max_num_seqs=6, max_num_batched_tokens=256, max_model_len=256, kvcache_block_size=16, num_kvcache_blocks=11)
    shared = oracle.fill_tokens(32, 9, 99, mcfg.vocab_size).tolist()
    prompts = [shared + oracle.fill_tokens(6 + i, 9, i, mcfg.vocab_size).tolist() for i in range(4)]
    sps = [dict(temperature=0.0, max_tokens=30, ignore_eos=True)] * 4
    r = _run_pair(mcfg, ecfg, prompts, sps, **BF)
    assert r["oracle"].scheduler.stats.preemptions > 0 and r["near_ties"] <= 3, r
    # A-23: prompts cut into 48-token chunks; later chunks reach the earlier ones through the block table (paged flash kernel, bf16)
    mcfg = mo.small()
    ecfg = dict(max_num_seqs=6, max_num_batched_tokens=48, max_model_len=256, kvcache_block_size=16, num_kvcache_blocks=80, enable_chunked_prefill=True)
    prompts = [oracle.fill_tokens(n, 1, i, mcfg.vocab_size).tolist() for i, n in enumerate([130, 20, 75, 48, 3, 49])]
    sps = [dict(temperature=0.0, max_tokens=8, ignore_eos=True)] * len(prompts)
    c = _run_pair(mcfg, ecfg, prompts, sps, **BF)
    assert c["near_ties"] <= 2 and c["steps"] > 12, c


def test_bfloat16_shared_prefix_and_launch_ahead():
    """the shared-prefix MFMA attention pass and async_decode on the bf16 build: oracle parity for the first, token-stream identity for
    launch-ahead."""
    for shape in ("d64_g2", "d128_g2"):
        mcfg = mo.small(seed=9) if shape == "d64_g2" else \
            mo.small(seed=9, hidden_size=512, num_attention_heads=4, num_key_value_heads=2, head_dim=128, intermediate_size=768)
        V = mcfg.vocab_size
        ecfg = dict(max_num_seqs=12, max_num_batched_tokens=2048, max_model_len=512, kvcache_block_size=64, num_kvcache_blocks=60)
        system = oracle.fill_tokens(150, 4, 7, V).tolist()
        prompts = [system + oracle.fill_tokens(3 + 9 * i, 4, 100 + i, V).tolist() for i in range(9)]
        sps = [dict(temperature=0.0, max_tokens=8 + 3 * i, ignore_eos=True) for i in range(9)]
        on = dict(shared_prefix_min_seqs=4)
        r = _run_pair(mcfg, ecfg, prompts, sps, product_kw=on, **BF)
        assert r["near_ties"] <= 2 and r["shared_steps"] >= 8, r
        gen = {}
        for is_prefill, seq_ids, tokens, _, _ in _run_product(mcfg, ecfg, prompts, sps, async_decode=1, **on, **BF):
            for sid, tok in zip(seq_ids, tokens):
                if tok >= 0:
                    gen.setdefault(sid, []).append(tok)
        assert {sid: toks[-len(gen[sid]):] for sid, toks in r["finished"].items()} == gen


def test_bfloat16_qk_norm_checkpoint_and_weight_conversions(tmp_path):
    """load_tensor / load_safetensors into a bf16 runner: BF16 payloads land bit for bit (the published Qwen3 checkpoints need no
    conversion at all), F32 rounds to nearest even, F16 widens exactly and rounds once; q_norm / k_norm weights (A-27) run through the
    bf16 head-norm + RoPE + KV-store launch in parity with the oracle."""
    from safetensors.numpy import save_file
    mcfg = mo.small(seed=5, qk_norm=True)
    rng = np.random.default_rng(78)
    D, L, V = mcfg.hd(), mcfg.num_hidden_layers, mcfg.vocab_size
    sd = {}
    for l in range(L):
        sd[f"model.layers.{l}.self_attn.q_norm.weight"] = (1 + 0.3 * rng.standard_normal(D)).astype(np.float16)
        sd[f"model.layers.{l}.self_attn.k_norm.weight"] = (1 + 0.3 * rng.standard_normal(D)).astype(np.float32)
    path = str(tmp_path / "qk.safetensors")
    save_file(sd, path)
    ecfg = dict(max_num_seqs=4, max_num_batched_tokens=256, max_model_len=128, kvcache_block_size=16, num_kvcache_blocks=40)
    prompts = [oracle.fill_tokens(n, 8, i, V).tolist() for i, n in enumerate([23, 70, 5])]
    sps = [dict(temperature=0.0, max_tokens=14, ignore_eos=True)] * 3
    r = _run_pair(mcfg, ecfg, prompts, sps, checkpoint=(sd, path), **BF)
    assert r["near_ties"] <= 2 and r["decode_steps"] >= 13, r

    m2 = mo.small(seed=3)
    Hd, I = m2.hidden_size, m2.intermediate_size
    mr = nvr.ModelRunner(nvr.Config(skip_block_size_check=1, max_num_seqs=2, max_num_batched_tokens=64, max_model_len=64, kvcache_block_size=16,
                                    num_kvcache_blocks=4, **BF), _model_cfgs(m2))
    assert mr.bf16
    # the synthetic fill itself: generated unrounded, rounded to bf16 once (the oracle's fill_w in bf16 mode)
    om = mo.OracleModel(m2, 2, 16, fp16=False, bf16=True, max_pos=64)
    assert np.array_equal(mr.weight("layers.0.down"), om.layers[0]["down"]) and np.array_equal(mr.weight("norm"), om.norm)
    a32 = (rng.standard_normal((Hd, I)) * 0.05).astype(np.float32)
    a32[0, :4] = [1.0 + 2.0 ** -8, 1.0 + 3 * 2.0 ** -8, -(1.0 + 2.0 ** -8), 3.0e38]      # ties to even both ways; a value fp16 cannot hold
    mr.load_tensor("model.layers.0.mlp.down_proj.weight", a32)
    assert np.array_equal(mr.weight("layers.0.down"), oracle.round_bf16(a32))
    assert mr.weight("layers.0.down")[0, 0] == 1.0 and mr.weight("layers.0.down")[0, 1] == 1.0 + 2.0 ** -6 and mr.weight("layers.0.down")[0, 3] > 2.9e38
    bits = oracle.to_bf16_bits(a32 * 3)
    mr.load_tensor("model.layers.0.mlp.down_proj.weight", bits)                          # dtype 1: bf16 bit patterns, stored as they are
    assert np.array_equal(mr.weight("layers.0.down").view(np.uint32) >> 16, bits)
    a16 = (rng.standard_normal((Hd, I)) * 0.05).astype(np.float16)
    mr.load_tensor("model.layers.0.mlp.down_proj.weight", a16)
    assert np.array_equal(mr.weight("layers.0.down"), oracle.round_bf16(a16.astype(np.float32)))


def test_bfloat16_holds_magnitudes_fp16_cannot():
    """Why the dtype exists (config.rs:113-116 lists it beside float16): bf16 keeps f32's exponent range.  A final-norm weight of 3e5
    overflows fp16 (max 65 504) — the fp16 build's logits are non-finite — while the bf16 build stays finite and in parity with the
    bf16 oracle (relative tolerance: the logits are O(1e5))."""
    mcfg = mo.small(seed=4)
    ecfg = dict(max_num_seqs=2, max_num_batched_tokens=64, max_model_len=64, kvcache_block_size=16, num_kvcache_blocks=8)
    big = np.full(mcfg.hidden_size, 3.0e5, np.float32)
    prompt = oracle.fill_tokens(12, 1, 0, mcfg.vocab_size).tolist()
    out = {}
    for dt in ("float16", "bfloat16"):
        nvr.lib().nvr_seq_reset_id_counter()
        p = nvr.LLMEngine(nvr.Config(skip_block_size_check=1, dtype=dt, **ecfg), _model_cfgs(mcfg))
        p.model_runner.load_tensor("model.norm.weight", big)
        p.add_request(prompt, nvr.SamplingParams(temperature=0.0, max_tokens=2, ignore_eos=True))
        p.step()
        out[dt] = p.model_runner.logits(1).copy()
    assert not np.isfinite(out["float16"]).all()
    assert np.isfinite(out["bfloat16"]).all() and np.abs(out["bfloat16"]).max() > 1e4
    from oracle import engine_oracle as eo
    eo.reset_sequence_counter()
    o = mo.OracleEngine(mcfg, eo.Config(**ecfg), fp16=False, bf16=True, max_pos=64)
    assert o.ranks[0].load_state_dict({"model.norm.weight": big}) == []
    o.add_request(prompt, eo.SamplingParams(temperature=0.0, max_tokens=2, ignore_eos=True))
    ol = o.step()["logits"]
    assert np.abs(out["bfloat16"] - ol).max() <= BF16_TOL * np.abs(ol).max()


@pytest.mark.parametrize("seed", [111, 222, 333])
def test_bfloat16_random_workloads_end_to_end(seed):
    """The randomised soak of the fp16 build (test_engine_gpu.test_random_workloads_end_to_end) on the bf16 build: prompt lengths from one
    token to several blocks, shared prefixes of whole blocks (cached prefixes: K/V through the block tables), an EOS id some sequences
    honour, a KV pool small enough to preempt, more requests than max_num_seqs — batches, logits and tokens against the bf16 oracle."""
    rng = np.random.default_rng(seed)
    mcfg = mo.small(seed=seed % 7)
    bs = int(rng.choice([16, 32]))
    nreq = int(rng.integers(5, 11))
    ecfg = dict(max_num_seqs=int(rng.integers(3, 7)), max_num_batched_tokens=int(rng.choice([160, 256, 512])), max_model_len=256,
                kvcache_block_size=bs, num_kvcache_blocks=int(rng.integers(14, 30)) * (32 // bs), eos_token_id=int(rng.integers(0, mcfg.vocab_size)))
    shared = oracle.fill_tokens(2 * bs, seed, 9999, mcfg.vocab_size).tolist()
    prompts, sps = [], []
    for i in range(nreq):
        own = oracle.fill_tokens(int(rng.integers(1, 70)), seed, i, mcfg.vocab_size).tolist()
        pr = (shared[:bs * int(rng.integers(1, 3))] + own) if rng.random() < 0.5 else own
        prompts.append(pr[:ecfg["max_num_batched_tokens"] - 1])
        sps.append(dict(temperature=0.0, max_tokens=int(rng.integers(1, 40)), ignore_eos=bool(rng.random() < 0.6)))
    r = _run_pair(mcfg, ecfg, prompts, sps, max_steps=2000, **BF)
    assert len(r["finished"]) == nreq and r["near_ties"] <= 8, r
    for sid, toks in r["finished"].items():
        assert toks[:len(prompts[sid])] == prompts[sid] and 1 <= len(toks) - len(prompts[sid]) <= sps[sid]["max_tokens"]
