"""GPU end-to-end parity (-m gpu): the product engine (C++ scheduler + HIP model runner, through the
C ABI) against the oracle engine on the same synthetic weights and prompts.

What is asserted, per step: identical batch composition and block tables (integer state: bit-exact);
logits within LOGIT_TOL of the fp16-faithful oracle; greedy token ids identical wherever the
oracle's top-1/top-2 margin exceeds 2*LOGIT_TOL (a smaller margin is a numerical tie between two
fp16 pipelines that differ only in f32 summation order; it is counted and bounded, never silently
accepted).  The oracle is teacher-forced with the product's tokens so one near-tie cannot cascade."""
import os

import numpy as np
import pytest

import nvr_import
import oracle
from oracle import engine_oracle as eo
from oracle import model_oracle as mo

nvr = nvr_import.load()
pytestmark = pytest.mark.gpu

LOGIT_TOL = 2e-2        # |logit_gpu - logit_oracle|, logits are O(1) f32 built from fp16 activations
F32_TOL = 2e-4          # Config.dtype = "float32": the f32 path against the oracle's f32 arithmetic (other summation orders only; measured ~1e-5)
BF16_TOL = 1.6e-1       # Config.dtype = "bfloat16": 8 mantissa bits against fp16's 11, so 8 x LOGIT_TOL against the bf16-faithful oracle
                        # (measured: 2.4e-2 on the small model, 4.5e-2 on Qwen3-0.6B, i.e. 6-8 x the fp16 build's 3.8e-3 / 5.5e-3)


def _model_cfgs(mcfg: mo.ModelConfig):
    m = nvr.ModelConfig(vocab_size=mcfg.vocab_size, hidden_size=mcfg.hidden_size,
                        intermediate_size=mcfg.intermediate_size, num_hidden_layers=mcfg.num_hidden_layers,
                        num_attention_heads=mcfg.num_attention_heads, num_key_value_heads=mcfg.num_key_value_heads,
                        head_dim=mcfg.head_dim or 0, max_position_embeddings=mcfg.max_position_embeddings,
                        rms_norm_eps=mcfg.rms_norm_eps, rope_theta=mcfg.rope_theta,
                        tie_word_embeddings=mcfg.tie_word_embeddings, init_std=mcfg.init_std, seed=mcfg.seed,
                        qk_norm=mcfg.qk_norm, use_bias=mcfg.use_bias)
    return m


def _run_pair(mcfg, ecfg: dict, prompts, sps, max_steps=400, fp16=True, enforce_eager=False, checkpoint=None, product_kw=None, dtype="float16"):
    """dtype = "bfloat16": the product's bf16 kernels against the oracle with bf16 at every 16-bit rounding point, at BF16_TOL."""
    eo.reset_sequence_counter()
    nvr.lib().nvr_seq_reset_id_counter()
    bf16, f32 = dtype == "bfloat16", dtype == "float32"
    tol = F32_TOL if f32 else BF16_TOL if bf16 else LOGIT_TOL
    o = mo.OracleEngine(mcfg, eo.Config(**ecfg), fp16=fp16 and not bf16 and not f32, bf16=bf16, max_pos=ecfg["max_model_len"])
    p = nvr.LLMEngine(nvr.Config(skip_block_size_check=1, enforce_eager=enforce_eager, dtype=dtype, **ecfg, **(product_kw or {})), _model_cfgs(mcfg))
    if checkpoint is not None:                       # (state dict, .safetensors path): same tensors into both engines
        sd, path = checkpoint
        for rk in o.ranks:
            assert rk.load_state_dict(sd) == []
        assert p.model_runner.load_safetensors(path) == []
    for pr, sp in zip(prompts, sps):
        o.add_request(pr, eo.SamplingParams(**sp))
        p.add_request(pr, nvr.SamplingParams(**sp))
    near_ties, steps, max_err, decode_steps, shared_steps = 0, 0, 0.0, 0, 0
    while not p.is_finished():
        rec = p.step()
        shared_steps += int(not rec["is_prefill"] and p.model_runner.last_shared_prefix_len() > 0)
        logits = p.model_runner.logits(rec["num_seqs"])
        orec = o.step(forced_tokens=rec["tokens"])
        assert orec["is_prefill"] == rec["is_prefill"] and orec["seq_ids"] == rec["seq_ids"], f"step {steps}: batch differs"
        assert [t == -1 for t in rec["tokens"]] == [t == -1 for t in orec["tokens"]], f"step {steps}: unfinished prompts differ"
        live = [i for i, t in enumerate(rec["tokens"]) if t != -1]           # a partial prompt chunk (A-23) has no logits row to compare
        err = np.abs(logits[live] - orec["logits"][live]).max() if live else 0.0
        max_err = max(max_err, float(err))
        assert err < tol, f"step {steps}: logits differ by {err}"
        srt = np.sort(orec["logits"], axis=1)
        margin = srt[:, -1] - srt[:, -2]
        for i, (tg, to) in enumerate(zip(rec["tokens"], orec["tokens"])):
            if tg != to:
                assert margin[i] <= 2 * tol, f"step {steps} row {i}: token {tg} != {to} at margin {margin[i]}"
                near_ties += 1
        steps += 1
        decode_steps += int(not rec["is_prefill"])
        assert steps < max_steps
    assert o.scheduler.is_finished()
    ost, pst = o.scheduler.stats, p.scheduler.get_stats()
    assert (ost.finished_sequences, ost.preemptions, ost.prefill_batches, ost.decode_batches) == \
           (pst["finished_sequences"], pst["preemptions"], pst["prefill_batches"], pst["decode_batches"])
    fin = {s.seq_id: s.token_ids for s in p.take_finished()}
    return dict(steps=steps, decode_steps=decode_steps, near_ties=near_ties, max_err=max_err, finished=fin, oracle=o, shared_steps=shared_steps)


def test_small_model_greedy_end_to_end():
    mcfg = mo.small()
    ecfg = dict(max_num_seqs=8, max_num_batched_tokens=512, max_model_len=512, kvcache_block_size=16, num_kvcache_blocks=64)
    prompts = [oracle.fill_tokens(n, 1, i, mcfg.vocab_size).tolist() for i, n in enumerate([5, 16, 17, 40, 1])]
    sps = [dict(temperature=0.0, max_tokens=24, ignore_eos=True)] * len(prompts)
    r = _run_pair(mcfg, ecfg, prompts, sps)
    assert r["decode_steps"] >= 23 and len(r["finished"]) == 5
    assert r["near_ties"] <= 2, r
    # the oracle was teacher-forced with the product's tokens: its sequences must equal the product's
    for sid, toks in r["finished"].items():
        assert len(toks) == len(prompts[sid]) + 24


def test_small_model_eager_equals_graph():
    """hipGraph replay and eager execution of the decode step produce identical token streams."""
    mcfg = mo.small(seed=3)
    ecfg = dict(max_num_seqs=4, max_num_batched_tokens=256, max_model_len=256, kvcache_block_size=16, num_kvcache_blocks=32)
    prompts = [oracle.fill_tokens(n, 2, i, mcfg.vocab_size).tolist() for i, n in enumerate([9, 30, 3])]
    sps = [dict(temperature=0.0, max_tokens=20, ignore_eos=True)] * 3
    a = _run_pair(mcfg, ecfg, prompts, sps, enforce_eager=False)
    b = _run_pair(mcfg, ecfg, prompts, sps, enforce_eager=True)
    assert a["finished"] == b["finished"]


def test_preemption_and_prefix_cache_end_to_end():
    """Block pressure: sequences are preempted (recompute-style) and re-prefilled; shared prefixes are
    deduplicated by the BlockManager.  Batches, tables and statistics must match the oracle exactly."""
    mcfg = mo.small(seed=5)
    ecfg = dict(max_num_seqs=6, max_num_batched_tokens=256, max_model_len=256, kvcache_block_size=16, num_kvcache_blocks=11)
    shared = oracle.fill_tokens(32, 9, 99, mcfg.vocab_size).tolist()
    prompts = [shared + oracle.fill_tokens(6 + i, 9, i, mcfg.vocab_size).tolist() for i in range(4)]
    sps = [dict(temperature=0.0, max_tokens=30, ignore_eos=True)] * 4
    r = _run_pair(mcfg, ecfg, prompts, sps)
    st = r["oracle"].scheduler.stats
    assert st.preemptions > 0, "scenario must exercise preemption"
    assert r["near_ties"] <= 3, r


def _run_product(mcfg, ecfg, prompts, sps, **cfg_kw):
    nvr.lib().nvr_seq_reset_id_counter()
    p = nvr.LLMEngine(nvr.Config(skip_block_size_check=1, **ecfg, **cfg_kw), _model_cfgs(mcfg))
    for pr, sp in zip(prompts, sps):
        p.add_request(pr, nvr.SamplingParams(**sp))
    trace = []
    while not p.is_finished():
        rec = p.step()
        trace.append((rec["is_prefill"], rec["seq_ids"], rec["tokens"], rec["num_tokens"], p.model_runner.logits(rec["num_seqs"]).copy()))
        assert len(trace) < 400
    return trace


def test_cached_prefix_skipping_is_bit_identical():
    """a prefill step computes only the tokens after a sequence's cached prefix and reaches the prefix
    through the block table.  The K/V rows are the same bits either way (a GEMM row depends only on its own input row)
    and the flash kernel walks the keys in the same 64-key steps, so logits and tokens must be IDENTICAL to the
    recompute-everything path of the reference (model_runner.rs:176-182) while far fewer rows go through the model.
    Bit-identity needs both batches on the same GEMM kernels (the K-accumulation order differs between the weight-streaming
    and the tiled kernels; linear.hip routes by T and T*N): 441 and 696 rows are on the same side of every threshold of this
    model's shapes.  Across a routing boundary the two paths agree to fp16 rounding like any other pair of kernels."""
    mcfg = mo.small(seed=8)
    ecfg = dict(max_num_seqs=16, max_num_batched_tokens=1024, max_model_len=256, kvcache_block_size=16, num_kvcache_blocks=120)
    shared = oracle.fill_tokens(32, 3, 77, mcfg.vocab_size).tolist()                 # 2 full blocks
    prompts = [shared + oracle.fill_tokens(40 + i, 3, i, mcfg.vocab_size).tolist() for i in range(8)]
    prompts.append(list(shared))                                                       # fully cached prompt: last token still computed
    prompts.append(oracle.fill_tokens(60, 3, 500, mcfg.vocab_size).tolist())           # nothing shared
    sps = [dict(temperature=0.0, max_tokens=12, ignore_eos=True)] * len(prompts)
    skip = _run_product(mcfg, ecfg, prompts, sps)
    full = _run_product(mcfg, ecfg, prompts, sps, recompute_cached_prefix=1)
    assert len(skip) == len(full)
    for a, b in zip(skip, full):
        assert a[:3] == b[:3]
        assert np.array_equal(a[4], b[4]), "logits differ between skipping and recomputing the cached prefix"
    pre_skip = sum(t[3] for t in skip if t[0]); pre_full = sum(t[3] for t in full if t[0])
    assert pre_full == sum(len(p) for p in prompts)
    assert pre_full == 696 and pre_skip == pre_full - 7 * 32 - 31, (pre_skip, pre_full)   # the first prompt fills the blocks; 7 more skip 32 tokens, the bare prefix 31
    # and the skipping path against the oracle engine (which recomputes), teacher-forced
    r = _run_pair(mcfg, ecfg, prompts, sps)
    assert r["near_ties"] <= 2, r


@pytest.mark.parametrize("shape", [
    # hidden, inter, heads, kv heads, head_dim, vocab, block size, prompt lengths
    dict(h=128, i=256, H=2, KVH=2, D=64, V=512, bs=4, lens=[3, 9, 21, 70]),          # G=1, tiny blocks (row groups straddle blocks)
    dict(h=256, i=384, H=8, KVH=1, D=64, V=1008, bs=16, lens=[40, 5, 17]),           # G=8: outside the MFMA flash kernel (row-kernel prefill)
    dict(h=256, i=384, H=8, KVH=1, D=64, V=1008, bs=16, lens=[200, 200, 90]),        # the same, one prefill of 490 of the 512 budgeted tokens (no flash tiles are laid out on this path)
    dict(h=512, i=768, H=4, KVH=4, D=128, V=2048, bs=48, lens=[100, 47, 140, 1]),    # non-power-of-two block size, G=1, D=128
    dict(h=2048, i=1024, H=16, KVH=4, D=128, V=4096, bs=32, lens=[33, 64, 2]),       # hidden 2048 (16-wave GEMMs, 4 row chunks in the slab norm)
    dict(h=4096, i=512, H=8, KVH=8, D=64, V=256, bs=16, lens=[20, 8]),               # hidden 4096: no split-k slabs, plain norm path, LM head K > 2048
])
def test_engine_parity_across_kernel_variants(shape):
    """Model shapes that steer the runner through its other kernel variants (GQA group sizes, head dims, block sizes,
    hidden sizes beyond the fused paths' limits): the engine must stay in parity with the oracle on all of them."""
    mcfg = mo.ModelConfig(vocab_size=shape["V"], hidden_size=shape["h"], intermediate_size=shape["i"], num_hidden_layers=2,
                          num_attention_heads=shape["H"], num_key_value_heads=shape["KVH"], head_dim=shape["D"], rms_norm_eps=1e-6,
                          rope_theta=10000.0, tie_word_embeddings=False, max_position_embeddings=512, init_std=0.05, seed=21)
    ecfg = dict(max_num_seqs=8, max_num_batched_tokens=512, max_model_len=256, kvcache_block_size=shape["bs"], num_kvcache_blocks=400 // shape["bs"] * 8)
    prompts = [oracle.fill_tokens(n, 4, i, mcfg.vocab_size).tolist() for i, n in enumerate(shape["lens"])]
    sps = [dict(temperature=0.0, max_tokens=10, ignore_eos=True)] * len(prompts)
    r = _run_pair(mcfg, ecfg, prompts, sps)
    assert r["near_ties"] <= 3, r


def test_checkpoint_weights_end_to_end(tmp_path):
    """a checkpoint in the HF / reference naming (separate q/k/v and gate/up projections, RMSNorm weights
    that are NOT all ones, untied LM head; stored as f16, bf16 and f32 tensors) goes through a .safetensors file into the
    packed device parameters, and the engine stays in parity with the oracle that loaded the same tensors."""
    from safetensors.numpy import save_file
    mcfg = mo.small(seed=3)
    rng = np.random.default_rng(77)
    Hd, I, H, KVH, D, V, L = mcfg.hidden_size, mcfg.intermediate_size, mcfg.num_attention_heads, mcfg.num_key_value_heads, mcfg.hd(), mcfg.vocab_size, mcfg.num_hidden_layers
    w = lambda *shape: (rng.standard_normal(shape) * 0.05).astype(np.float16)
    sd = {"model.embed_tokens.weight": w(V, Hd), "model.norm.weight": (1 + 0.2 * rng.standard_normal(Hd)).astype(np.float32),
          "lm_head.weight": w(V, Hd).astype(np.float32)}
    for l in range(L):
        pre = f"model.layers.{l}."
        sd[pre + "input_layernorm.weight"] = (1 + 0.2 * rng.standard_normal(Hd)).astype(np.float16)
        sd[pre + "post_attention_layernorm.weight"] = (1 + 0.2 * rng.standard_normal(Hd)).astype(np.float16)
        sd[pre + "self_attn.q_proj.weight"] = w(H * D, Hd)
        sd[pre + "self_attn.k_proj.weight"] = w(KVH * D, Hd)
        sd[pre + "self_attn.v_proj.weight"] = w(KVH * D, Hd).astype(np.float32)
        sd[pre + "self_attn.o_proj.weight"] = w(Hd, H * D)
        sd[pre + "mlp.gate_proj.weight"] = w(I, Hd)
        sd[pre + "mlp.up_proj.weight"] = w(I, Hd)
        sd[pre + "mlp.down_proj.weight"] = w(Hd, I)
    path = str(tmp_path / "model.safetensors")
    save_file(sd, path)
    ecfg = dict(max_num_seqs=4, max_num_batched_tokens=256, max_model_len=128, kvcache_block_size=16, num_kvcache_blocks=40)
    prompts = [oracle.fill_tokens(n, 6, i, V).tolist() for i, n in enumerate([19, 40, 7])]
    sps = [dict(temperature=0.0, max_tokens=12, ignore_eos=True)] * 3
    r = _run_pair(mcfg, ecfg, prompts, sps, checkpoint=(sd, path))
    assert r["near_ties"] <= 2, r
    # the synthetic-weight run of the same prompts gives other tokens: the checkpoint really is what ran
    r0 = _run_pair(mcfg, ecfg, prompts, sps)
    assert r0["finished"] != r["finished"]


def test_checkpoint_shards_dtypes_and_errors():
    """Tensor-parallel slices of a full checkpoint (no communicator needed to look at them): each rank's packed tensors
    equal the oracle rank's, bit for bit; bf16 / f32 sources are converted to fp16 (RNE); wrong shapes and names outside
    the reference graph are reported with the reference's error text / UNSUPPORTED."""
    mcfg = mo.small(seed=3)
    rng = np.random.default_rng(78)
    Hd, I, H, KVH, D, V = mcfg.hidden_size, mcfg.intermediate_size, mcfg.num_attention_heads, mcfg.num_key_value_heads, mcfg.hd(), mcfg.vocab_size
    f32 = lambda *shape: (rng.standard_normal(shape) * 0.05).astype(np.float32)
    sd = {"layers.1.self_attn.qkv_proj.weight": f32((H + 2 * KVH) * D, Hd), "layers.1.self_attn.o_proj.weight": f32(Hd, H * D),
          "layers.0.mlp.gate_up_proj.weight": f32(2 * I, Hd), "layers.0.mlp.down_proj.weight": f32(Hd, I),
          "embed_tokens.weight": f32(V, Hd), "lm_head.weight": f32(V, Hd), "norm.weight": f32(Hd),
          "layers.0.input_layernorm.weight": f32(Hd)}
    for rank in range(2):
        om = mo.OracleModel(mcfg, 4, 16, True, rank, 2)
        assert om.load_state_dict(sd) == []
        mr = nvr.ModelRunner(nvr.Config(skip_block_size_check=1, max_num_seqs=2, max_num_batched_tokens=64, max_model_len=64, kvcache_block_size=16,
                                        num_kvcache_blocks=4, tensor_parallel_size=2, tensor_parallel_rank=rank), _model_cfgs(mcfg))
        for k, v in sd.items():
            mr.load_tensor(k, v)
        for local, ref in [("layers.1.qkv", om.layers[1]["qkv"]), ("layers.1.o", om.layers[1]["o"]), ("layers.0.gate_up", om.layers[0]["gate_up"]),
                           ("layers.0.down", om.layers[0]["down"]), ("embed", om.embed), ("lm_head", om.lm_head), ("norm", om.norm),
                           ("layers.0.ln1", om.layers[0]["ln1"])]:
            got = mr.weight(local).astype(np.float32)
            assert got.shape == ref.shape and np.array_equal(got, ref), (rank, local)
        # an untouched tensor keeps its synthetic values
        assert np.array_equal(mr.weight("layers.0.qkv").astype(np.float32), om.layers[0]["qkv"])
    # bf16 source: bits -> f32 -> fp16
    vals = f32(Hd)
    bf = (vals.view(np.uint32) >> 16).astype(np.uint16)
    mr.load_tensor("model.layers.0.post_attention_layernorm.weight", bf)
    back = (bf.astype(np.uint32) << 16).view(np.float32).astype(np.float16)
    assert np.array_equal(mr.weight("layers.0.ln2"), back)
    with pytest.raises(nvr.NvrError) as e:
        mr.load_tensor("layers.0.mlp.down_proj.weight", f32(Hd, I + 16))
    assert e.value.code == -4 and "Partition weight shape mismatch" in str(e.value)
    with pytest.raises(nvr.NvrError) as e:
        mr.load_tensor("layers.0.self_attn.q_norm.weight", f32(D))
    assert e.value.code == -10
    with pytest.raises(nvr.NvrError) as e:
        mr.load_tensor("layers.9.mlp.down_proj.weight", f32(Hd, I))
    assert e.value.code == -7


def test_checkpoint_bf16_safetensors_file(tmp_path):
    """a BF16 .safetensors file (the dtype of the published Qwen3 checkpoints) loads through load_safetensors
    (bits -> f32 -> fp16 RNE on the way in); a tensor outside the reference graph (q_norm) is reported AND warned about."""
    import json
    import warnings
    mcfg = mo.small(seed=3)
    rng = np.random.default_rng(79)
    Hd, I = mcfg.hidden_size, mcfg.intermediate_size
    tens = {"model.layers.0.mlp.down_proj.weight": (rng.standard_normal((Hd, I)) * 0.05).astype(np.float32),
            "model.norm.weight": (1 + 0.2 * rng.standard_normal(Hd)).astype(np.float32),
            "model.layers.0.self_attn.q_norm.weight": np.ones(mcfg.hd(), np.float32)}
    header, blobs, off = {}, [], 0
    for name, a in tens.items():
        raw = (a.view(np.uint32) >> 16).astype(np.uint16).tobytes()
        header[name] = dict(dtype="BF16", shape=list(a.shape), data_offsets=[off, off + len(raw)])
        blobs.append(raw); off += len(raw)
    hj = json.dumps(header).encode(); hj += b" " * (-len(hj) % 8)
    path = str(tmp_path / "bf16.safetensors")
    with open(path, "wb") as f:
        f.write(len(hj).to_bytes(8, "little")); f.write(hj); f.write(b"".join(blobs))
    mr = nvr.ModelRunner(nvr.Config(skip_block_size_check=1, max_num_seqs=2, max_num_batched_tokens=64, max_model_len=64, kvcache_block_size=16,
                                    num_kvcache_blocks=4), _model_cfgs(mcfg))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        skipped = mr.load_safetensors(path)
    assert skipped == ["model.layers.0.self_attn.q_norm.weight"] and any("q/k-norm" in str(x.message) for x in w)
    for local, name in [("layers.0.down", "model.layers.0.mlp.down_proj.weight"), ("norm", "model.norm.weight")]:
        a = tens[name]
        back = ((a.view(np.uint32) >> 16) << 16).view(np.float32).astype(np.float16)
        assert np.array_equal(mr.weight(local), back), local
    with pytest.raises(nvr.NvrError):
        mr.load_safetensors(path, strict=True)


def test_engine_stats_health_and_shutdown():
    """LLMEngine::get_stats / health_check / shutdown (llm_engine.rs:312-357) over the C ABI."""
    mcfg = mo.small(seed=2)
    nvr.lib().nvr_seq_reset_id_counter()
    eng = nvr.LLMEngine(nvr.Config(skip_block_size_check=1, max_num_seqs=4, max_num_batched_tokens=256, max_model_len=128, kvcache_block_size=16,
                                   num_kvcache_blocks=20), _model_cfgs(mcfg))
    for i, n in enumerate([33, 17, 50]):
        eng.add_request(oracle.fill_tokens(n, 7, i, mcfg.vocab_size).tolist(), nvr.SamplingParams(temperature=0.0, max_tokens=40, ignore_eos=True))
    st = eng.get_stats()
    assert st["is_running"] and st["scheduler"]["waiting_sequences"] == 3 and st["memory"] == dict(total_blocks=20, free_blocks=20, used_blocks=0, utilization=0.0)
    assert eng.host_times() == dict(schedule_us=0.0, postprocess_us=0.0, steps=0)
    eng.step(); eng.step()
    ht = eng.host_times()                                       # nvr_engine_host_times: the integer side of the two steps, always counted
    assert ht["steps"] == 2 and 0.0 < ht["schedule_us"] < 5e4 and 0.0 < ht["postprocess_us"] < 5e4, ht
    st, h = eng.get_stats(), eng.health_check()
    used = 3 + 2 + 4                                            # ceil(34/16) + ceil(18/16) + ceil(51/16) blocks after one decode step
    assert st["memory"]["used_blocks"] == used and abs(st["memory"]["utilization"] - used / 20 * 100) < 1e-9
    assert st["scheduler"]["running_sequences"] == 3 and st["scheduler"]["prefill_batches"] == 1 and st["scheduler"]["decode_batches"] == 1
    assert h == dict(is_healthy=True, memory_pressure=st["memory"]["utilization"], active_sequences=3, waiting_sequences=0)
    eng.shutdown()
    st = eng.get_stats()
    # preempt_all (scheduler.rs:314-319) does not refresh the queue-length snapshot in the stats; the queues themselves moved
    assert not st["is_running"] and eng.scheduler.get_queue_lengths() == (3, 0)
    assert st["memory"]["used_blocks"] == 0 and st["scheduler"]["preemptions"] == 3


def test_gqa4_head_dim_128_model():
    """Qwen3-8B-like head geometry (32:8 grouping scaled down, D=128) and block size 256."""
    mcfg = mo.ModelConfig(vocab_size=2048, hidden_size=512, intermediate_size=1024, num_hidden_layers=2,
                          num_attention_heads=8, num_key_value_heads=2, head_dim=128, rope_theta=1e6,
                          tie_word_embeddings=True, max_position_embeddings=1024, init_std=0.04, seed=11)
    ecfg = dict(max_num_seqs=4, max_num_batched_tokens=1024, max_model_len=1024, kvcache_block_size=256, num_kvcache_blocks=12)
    prompts = [oracle.fill_tokens(n, 4, i, mcfg.vocab_size).tolist() for i, n in enumerate([250, 300, 7])]
    sps = [dict(temperature=0.0, max_tokens=12, ignore_eos=True)] * 3
    r = _run_pair(mcfg, ecfg, prompts, sps)
    assert r["near_ties"] <= 2, r


def test_eos_and_stochastic_sampling_paths():
    mcfg = mo.small(seed=7)
    ecfg = dict(max_num_seqs=4, max_num_batched_tokens=256, max_model_len=256, kvcache_block_size=16, num_kvcache_blocks=32,
                eos_token_id=5)
    prompts = [oracle.fill_tokens(n, 6, i, mcfg.vocab_size).tolist() for i, n in enumerate([8, 12, 20])]
    sps = [dict(temperature=0.0, max_tokens=16), dict(temperature=0.8, top_k=20, max_tokens=16, ignore_eos=True),
           dict(temperature=1.0, top_p=0.9, max_tokens=16, ignore_eos=True)]
    eo.reset_sequence_counter(); nvr.lib().nvr_seq_reset_id_counter()
    o = mo.OracleEngine(mcfg, eo.Config(**ecfg), fp16=True, max_pos=256, sample_seed=42)
    p = nvr.LLMEngine(nvr.Config(skip_block_size_check=1, sample_seed=42, **ecfg), _model_cfgs(mcfg))
    for pr, sp in zip(prompts, sps):
        o.add_request(pr, eo.SamplingParams(**sp)); p.add_request(pr, nvr.SamplingParams(**sp))
    agree = total = 0
    while not p.is_finished():
        rec = p.step()
        orec = o.step(forced_tokens=rec["tokens"])
        assert orec["seq_ids"] == rec["seq_ids"]
        for tg, to in zip(rec["tokens"], orec["tokens"]):
            agree += int(tg == to); total += 1
    # same counter-RNG keys on both sides: the stochastic rows agree except at numerical near-ties
    assert agree >= total - 3, (agree, total)


_SOAK_SEEN = dict(preemptions=0, early_stops=0, runs=0)


@pytest.mark.parametrize("seed", [101, 202, 303, 404, 505, 606])
def test_random_workloads_end_to_end(seed):
    """Randomised soak: prompt lengths from 1 token to several blocks, shared prefixes of whole blocks, mixed max_tokens, an
    EOS id that some sequences honour, a KV pool small enough to force preemption on most seeds, more requests than
    max_num_seqs.  Every step: same batch, same block tables (through the logits), logits within tolerance, tokens equal
    outside numerical near-ties; at the end the same scheduler statistics and the same finished sequences."""
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
        pr = pr[:ecfg["max_num_batched_tokens"] - 1]
        prompts.append(pr)
        sps.append(dict(temperature=0.0, max_tokens=int(rng.integers(1, 40)), ignore_eos=bool(rng.random() < 0.6)))
    r = _run_pair(mcfg, ecfg, prompts, sps, max_steps=2000)
    assert len(r["finished"]) == nreq
    assert r["near_ties"] <= 4, r
    st = r["oracle"].scheduler.stats
    _SOAK_SEEN["preemptions"] += st.preemptions
    _SOAK_SEEN["early_stops"] += sum(len(t) - len(prompts[i]) < sps[i]["max_tokens"] for i, t in r["finished"].items())
    _SOAK_SEEN["runs"] += 1
    for sid, toks in r["finished"].items():
        assert toks[:len(prompts[sid])] == prompts[sid]
        assert 1 <= len(toks) - len(prompts[sid]) <= sps[sid]["max_tokens"]
        if len(toks) - len(prompts[sid]) < sps[sid]["max_tokens"]:                     # stopped early: only the EOS rule allows that
            assert not sps[sid]["ignore_eos"] and toks[-1] == ecfg["eos_token_id"]


def test_random_workloads_did_exercise_preemption():
    if _SOAK_SEEN["runs"] < 6:
        pytest.skip("runs after the six soak cases")
    assert _SOAK_SEEN["preemptions"] > 0, _SOAK_SEEN


# ------------------------------------------------------------------------------------------------------------------
# text in, SequenceOutput out (LLMEngine::generate / generate_stream, llm_engine.rs:70-128)
_GEN_ECFG = dict(max_num_seqs=4, max_num_batched_tokens=256, max_model_len=256, kvcache_block_size=16, num_kvcache_blocks=40)
_GEN_PROMPTS = ["hello world", "abc", "The quick brown fox jumps over the lazy dog", "x" * 140, "Zz"]


def _gen_engine(mcfg, **kw):
    nvr.lib().nvr_seq_reset_id_counter()
    return nvr.LLMEngine(nvr.Config(skip_block_size_check=1, **{**_GEN_ECFG, **kw}), _model_cfgs(mcfg))


def test_generate_returns_sequence_outputs_in_prompt_order():
    """generate() == the same requests stepped by hand, == the oracle engine (teacher-forced run of _run_pair)."""
    mcfg = mo.small(seed=5)
    sp = dict(temperature=0.0, max_tokens=12, ignore_eos=True)
    outs = _gen_engine(mcfg).generate(_GEN_PROMPTS, nvr.SamplingParams(**sp))
    assert [o.seq_id for o in outs] == list(range(len(_GEN_PROMPTS)))                  # prompt order (5 prompts > max_num_seqs 4)
    ids = [eo.tokenize(p) for p in _GEN_PROMPTS]
    assert [o.num_prompt_tokens for o in outs] == [11, 3, 43, 100, 2]                  # the 140-char prompt is cut at 100 chars
    r = _run_pair(mcfg, _GEN_ECFG, ids, [sp] * len(ids))                               # product stepped by hand vs oracle, per step
    assert r["near_ties"] <= 2
    for o, pid in zip(outs, ids):
        assert o.token_ids == r["finished"][o.seq_id] and o.token_ids[:o.num_prompt_tokens] == pid
        assert o.completion_token_ids == o.token_ids[o.num_prompt_tokens:] and o.num_completion_tokens == 12
        assert o.text == eo.detokenize(o.completion_token_ids) == nvr.detokenize(o.completion_token_ids)
        assert o.status == eo.FINISHED
    # the oracle's own SequenceOutput emission over the teacher-forced sequences
    exp = [eo.sequence_output(r["oracle"].finished[i]) for i in range(len(ids))]
    assert [(e.seq_id, e.text, e.token_ids, e.completion_token_ids, e.num_prompt_tokens, e.num_completion_tokens, e.status) for e in exp] == \
           [(o.seq_id, o.text, o.token_ids, o.completion_token_ids, o.num_prompt_tokens, o.num_completion_tokens, o.status) for o in outs]


def test_generate_token_prompts_empty_list_and_errors():
    mcfg = mo.small(seed=5)
    eng = _gen_engine(mcfg)
    assert eng.generate([]) == []                                                      # llm_engine.rs:76-78
    sp = nvr.SamplingParams(temperature=0.0, max_tokens=5, ignore_eos=True)
    a = eng.generate([[1, 2, 3], [1000, 7]], sp)
    assert [o.token_ids[:o.num_prompt_tokens] for o in a] == [[1, 2, 3], [1000, 7]] and all(o.num_completion_tokens == 5 for o in a)
    # a second call on the same engine starts from an empty scheduler and returns only its own prompts
    b = eng.generate(["abc"], sp)
    assert len(b) == 1 and b[0].num_prompt_tokens == 3 and eng.is_finished() and eng.take_finished() == []
    # requests the model cannot embed are refused before anything is queued (vocab 1024: '€' = 8364)
    for bad in (["ok", "caf€"], [[5, 1024]], [[5, -1]], ["ok", ""]):
        with pytest.raises(nvr.NvrError):
            eng.generate(bad, sp)
        assert eng.is_finished() and eng.scheduler.get_queue_lengths() == (0, 0)
    with pytest.raises(nvr.NvrError):
        eng.add_request([1, 2, 4096])
    with pytest.raises(nvr.NvrError):
        eng.generate(["abc"], nvr.SamplingParams(temperature=-1.0))
    assert eng.add_prompt("hi", sp) >= 0 and eng.scheduler.get_queue_lengths() == (1, 0)
    # a sequence queued through add_prompt is stepped by generate too but stays with take_finished
    c = eng.generate(["yo"], sp)
    assert len(c) == 1 and [len(s.token_ids) for s in eng.take_finished()] == [2 + 5]


def test_generate_stream_delivers_every_step_and_stops_when_the_receiver_drops():
    mcfg = mo.small(seed=5)
    sp = nvr.SamplingParams(temperature=0.0, max_tokens=6, ignore_eos=True)
    prompts = _GEN_PROMPTS[:3]
    final = {o.seq_id: o for o in _gen_engine(mcfg).generate(prompts, sp)}
    eng = _gen_engine(mcfg)
    got = eng.generate_stream(prompts, sp)
    assert eng.is_finished() and len(got) == 3 * 6                                     # prefill step + 5 decode steps, 3 sequences each
    per = {}
    for o in got:
        per.setdefault(o.seq_id, []).append(o)
    for sid, lst in per.items():
        assert [o.num_completion_tokens for o in lst] == [1, 2, 3, 4, 5, 6]            # cumulative, one token per step
        assert [o.status for o in lst] == [eo.RUNNING] * 5 + [eo.FINISHED]
        for o in lst:
            assert o.token_ids == final[sid].token_ids[:len(o.token_ids)] and o.text == final[sid].text[:o.num_completion_tokens]
        assert lst[-1] == final[sid]
    # the oracle's stream reports the same sequence of (seq_id, completion length, status)
    eo.reset_sequence_counter()
    oe = mo.OracleEngine(mcfg, eo.Config(**_GEN_ECFG), fp16=True, max_pos=_GEN_ECFG["max_model_len"])
    ostream = []
    oe.generate(prompts, eo.SamplingParams(temperature=0.0, max_tokens=6, ignore_eos=True), on_output=lambda o: ostream.append(o) and False)
    assert [(o.seq_id, o.num_completion_tokens, o.status) for o in ostream] == [(o.seq_id, o.num_completion_tokens, o.status) for o in got]
    # dropped receiver (llm_engine.rs:250-253): the loop stops after the delivery that returned non-zero
    eng = _gen_engine(mcfg)
    seen = []
    got = eng.generate_stream(prompts, sp, on_output=lambda o: seen.append(o) or len(seen) == 4)
    assert len(got) == 4 and not eng.is_finished() and eng.scheduler.get_queue_lengths() == (0, 3)
    while not eng.is_finished():                                                       # the queued sequences are still good
        eng.step()
    assert sorted(tuple(s.token_ids) for s in eng.take_finished()) == sorted(tuple(o.token_ids) for o in final.values())
    # an exception inside the callback stops the stream and is re-raised on the Python side
    eng = _gen_engine(mcfg)
    with pytest.raises(ZeroDivisionError):
        eng.generate_stream(prompts, sp, on_output=lambda o: 1 / 0)
    eng.shutdown()


def test_admission_limits_and_length_clamp():
    """ADVICE r01 (engine.cpp): a prompt longer than max_model_len or than one prefill batch is refused when it is added (it
    could never execute and would wedge the scheduler); max_tokens is clamped so that a sequence stops when its next decode
    step would pass max_model_len; requests around it keep running."""
    mcfg = mo.small()
    ecfg = dict(max_num_seqs=4, max_num_batched_tokens=64, max_model_len=48, kvcache_block_size=16, num_kvcache_blocks=16)
    p = nvr.LLMEngine(nvr.Config(skip_block_size_check=1, **ecfg), _model_cfgs(mcfg))
    with pytest.raises(nvr.NvrError, match="max_model_len"):
        p.add_request(oracle.fill_tokens(49, 1, 0, mcfg.vocab_size).tolist(), nvr.SamplingParams(temperature=0.0, max_tokens=4))
    p2 = nvr.LLMEngine(nvr.Config(skip_block_size_check=1, **dict(ecfg, max_model_len=128, num_kvcache_blocks=40)), _model_cfgs(mcfg))
    with pytest.raises(nvr.NvrError, match="max_num_batched_tokens"):
        p2.add_request(oracle.fill_tokens(65, 1, 0, mcfg.vocab_size).tolist(), nvr.SamplingParams(temperature=0.0, max_tokens=4))
    # 40-token prompt, max_tokens 64 -> clamped to 48 - 40 + 1 = 9 tokens; a short neighbour is unaffected
    a = p.add_request(oracle.fill_tokens(40, 1, 1, mcfg.vocab_size).tolist(), nvr.SamplingParams(temperature=0.0, max_tokens=64, ignore_eos=True))
    b = p.add_request(oracle.fill_tokens(5, 1, 2, mcfg.vocab_size).tolist(), nvr.SamplingParams(temperature=0.0, max_tokens=12, ignore_eos=True))
    steps = 0
    while not p.is_finished():
        p.step(); steps += 1
        assert steps < 40
    fin = {s.seq_id: s for s in p.take_finished()}
    assert len(fin[a].token_ids) == 49 and len(fin[b].token_ids) == 17
    assert p.scheduler.get_block_stats()["used_blocks"] == 0


@pytest.mark.parametrize("mml", [100, 120, 128])
def test_graph_decode_small_max_model_len_full_batch(mml):
    """ADVICE r01 (model_runner.cpp:113): the graph path launches attention with the 256-token context bucket, which exceeds
    a small max_model_len; the split-KV workspace must cover that bucket (it was sized from max_model_len and the partials of
    a full batch ran past it).  Graph == eager token streams, and the logits stay within tolerance of the oracle."""
    mcfg = mo.small()
    ecfg = dict(max_num_seqs=4, max_num_batched_tokens=512, max_model_len=mml, kvcache_block_size=16, num_kvcache_blocks=40)
    prompts = [oracle.fill_tokens(n, 1, i, mcfg.vocab_size).tolist() for i, n in enumerate([70, 66, 80, 75])]
    sps = [dict(temperature=0.0, max_tokens=16, ignore_eos=True)] * 4
    g = _run_pair(mcfg, ecfg, prompts, sps)
    e = _run_pair(mcfg, ecfg, prompts, sps, enforce_eager=True)
    assert g["finished"] == e["finished"] and g["near_ties"] <= 2


def test_config_device_and_dtype_gate_the_runner():
    """Config.device / Config.dtype (config.rs:48-51): the names validate like the reference's; a runner exists only for the
    HIP device and the three dtypes (fp16, bf16, float32 — tensor-parallel ranks of any of them) — anything else fails loudly instead of falling back."""
    mcfg = mo.small()
    base = dict(skip_block_size_check=1, max_num_seqs=2, max_num_batched_tokens=64, max_model_len=64, kvcache_block_size=16, num_kvcache_blocks=4)
    for ok in (dict(), dict(device="cuda"), dict(device="hip", dtype="float16"), dict(dtype="bfloat16"), dict(dtype="float32"),
               dict(dtype="float32", tensor_parallel_size=2, tensor_parallel_rank=1)):
        nvr.ModelRunner(nvr.Config(**base, **ok), _model_cfgs(mcfg))
    for bad in (dict(device="cpu"), dict(device="metal")):
        with pytest.raises(nvr.NvrError) as e:
            nvr.ModelRunner(nvr.Config(**base, **bad), _model_cfgs(mcfg))
        assert e.value.code == -10


def test_chunked_prefill_engine_parity():
    """Extension A-23 (enable_chunked_prefill): prompts longer than the token budget are cut into several prefill steps; later
    chunks reach the earlier ones through the block table (paged flash kernel).  Product vs the oracle's chunked engine:
    batch composition, chunk ranges (through the logits of the rows they feed), block tables, -1 tokens for unfinished prompts,
    logits within tolerance on every finishing row, greedy tokens equal outside near-ties; then chunked == unchunked tokens."""
    mcfg = mo.small()
    ecfg = dict(max_num_seqs=6, max_num_batched_tokens=48, max_model_len=256, kvcache_block_size=16, num_kvcache_blocks=80)
    prompts = [oracle.fill_tokens(n, 1, i, mcfg.vocab_size).tolist() for i, n in enumerate([130, 20, 75, 48, 3, 49])]
    sps = [dict(temperature=0.0, max_tokens=8, ignore_eos=True)] * len(prompts)

    def run(chunked, budget):
        eo.reset_sequence_counter(); nvr.lib().nvr_seq_reset_id_counter()
        cfg = dict(ecfg, max_num_batched_tokens=budget, enable_chunked_prefill=chunked)
        o = mo.OracleEngine(mcfg, eo.Config(**cfg), fp16=True, max_pos=cfg["max_model_len"])
        p = nvr.LLMEngine(nvr.Config(skip_block_size_check=1, **cfg), _model_cfgs(mcfg))
        for pr, sp in zip(prompts, sps):
            o.add_request(pr, eo.SamplingParams(**sp)); p.add_request(pr, nvr.SamplingParams(**sp))
        ties, partial_rows, steps = 0, 0, 0
        while not p.is_finished():
            rec = p.step()
            logits = p.model_runner.logits(rec["num_seqs"])
            orec = o.step(forced_tokens=rec["tokens"])
            assert orec["seq_ids"] == rec["seq_ids"] and orec["is_prefill"] == rec["is_prefill"], f"step {steps}"
            assert [t == -1 for t in rec["tokens"]] == [t == -1 for t in orec["tokens"]], f"step {steps}: unfinished prompts differ"
            srt = np.sort(orec["logits"], axis=1)
            for i, (a, b) in enumerate(zip(rec["tokens"], orec["tokens"])):
                if a == -1:
                    partial_rows += 1
                    continue
                assert np.abs(logits[i] - orec["logits"][i]).max() < LOGIT_TOL, f"step {steps} row {i}"
                if a != b:
                    assert srt[i, -1] - srt[i, -2] <= 2 * LOGIT_TOL
                    ties += 1
            steps += 1
            assert steps < 200
        assert o.scheduler.is_finished()
        return {s.seq_id: s.token_ids for s in p.take_finished()}, ties, partial_rows, steps
    fc, ties, partial, steps_c = run(True, 48)
    assert partial >= 4 and ties <= 2                      # 130 / 75 / 49-token prompts against a 48-token budget
    fu, ties_u, partial_u, steps_u = run(False, 512)
    assert partial_u == 0 and ties_u <= 2
    same = sum(fc[k] == fu[k] for k in fc)
    assert same >= len(fc) - 1, (same, len(fc))             # same greedy continuations whatever the chunking
    # a prompt longer than the budget is refused without chunking, accepted with it
    p2 = nvr.LLMEngine(nvr.Config(skip_block_size_check=1, **dict(ecfg, max_num_batched_tokens=48)), _model_cfgs(mcfg))
    with pytest.raises(nvr.NvrError, match="max_num_batched_tokens"):
        p2.add_request(prompts[0], nvr.SamplingParams(temperature=0.0, max_tokens=2))


@pytest.mark.parametrize("bs,eos,nblocks", [(16, None, 64), (16, 5, 64), (4, None, 40), (256, None, 6)])
def test_async_decode_is_transparent(bs, eos, nblocks):
    """nvr_config.async_decode = 1 (the next greedy decode step is enqueued before this step's tokens reach the host; ids go device
    to device) against async_decode = 0 on the same requests: every step reports the same batch, the same tokens, the same
    finished sequences and statistics — including steps around block boundaries (no launch-ahead there), sequences that stop on
    max_tokens at different steps, EOS-honouring sequences (never launched ahead), requests added between steps (the step in
    flight is cancelled so that the prefill comes first, as in the reference) and pool pressure with preemption."""
    mcfg = mo.small(seed=9)
    ecfg = dict(max_num_seqs=6, max_num_batched_tokens=512, max_model_len=400, kvcache_block_size=bs, num_kvcache_blocks=nblocks)
    if eos is not None:
        ecfg["eos_token_id"] = eos
    rng = np.random.default_rng(bs + (eos or 0))
    first = [(oracle.fill_tokens(int(n), 3, i, mcfg.vocab_size).tolist(), int(mt), bool(ig)) for i, (n, mt, ig) in
             enumerate([(9, 40, 1), (31, 17, 1), (5, 33, 1), (18, 40, eos is None)])]
    late = [(oracle.fill_tokens(12, 3, 9, mcfg.vocab_size).tolist(), 21, True), (oracle.fill_tokens(3, 3, 10, mcfg.vocab_size).tolist(), 9, True)]

    def run(async_on):
        nvr.lib().nvr_seq_reset_id_counter()
        p = nvr.LLMEngine(nvr.Config(skip_block_size_check=1, async_decode=async_on, **ecfg), _model_cfgs(mcfg))
        for pr, mt, ig in first:
            p.add_request(pr, nvr.SamplingParams(temperature=0.0, max_tokens=mt, ignore_eos=ig))
        trace, steps = [], 0
        while not p.is_finished():
            if steps in (7, 15):                                        # arrivals between steps
                pr, mt, ig = late[0 if steps == 7 else 1]
                p.add_request(pr, nvr.SamplingParams(temperature=0.0, max_tokens=mt, ignore_eos=ig))
            rec = p.step()
            st = p.get_stats()["scheduler"]
            trace.append((rec["is_prefill"], tuple(rec["seq_ids"]), tuple(rec["tokens"]), rec["num_finished"], st["decode_batches"],
                          st["prefill_batches"], st["finished_sequences"], st["preemptions"]))
            steps += 1
            assert steps < 400
        fin = {s.seq_id: s.token_ids for s in p.take_finished()}
        return trace, fin, p.get_stats()
    ta, fa, sa = run(1)
    ts, fs, ss = run(0)
    assert len(ta) == len(ts)
    for i, (a, b) in enumerate(zip(ta, ts)):
        assert a == b, f"step {i}: async {a} != sync {b}"
    assert fa == fs and sa == ss and len(fa) == 6


def test_async_decode_survives_a_full_graph_cache():
    """Launch-ahead error path (engine.cpp): the graph cache is limited to 2 entries (NVR_MAX_GRAPHS, read once when the runner is
    created) and holds (3 sequences, bucket 256) and (2 sequences, bucket 256) when the longest sequence grows past 256 tokens in the
    middle of a cache block (block size 48: launch-ahead is allowed there): the step behind the current one needs a third graph and
    cannot be enqueued ahead.  The current step must still deliver its tokens, the speculative schedule must be rolled back, and the next
    call must take the synchronous path (which flushes the cache): every step's batch, tokens, finished sets and statistics equal those
    of the synchronous engine, and nothing is left holding the placeholder token."""
    mcfg = mo.small(seed=21)
    ecfg = dict(max_num_seqs=8, max_num_batched_tokens=512, max_model_len=320, kvcache_block_size=48, num_kvcache_blocks=24)
    reqs = [(oracle.fill_tokens(int(n), 3, i, mcfg.vocab_size).tolist(), int(mt)) for i, (n, mt) in
            enumerate([(9, 6), (240, 45), (20, 45)])]
    late = {30: (oracle.fill_tokens(12, 3, 9, mcfg.vocab_size).tolist(), 25)}

    def run(async_on, max_graphs):
        os.environ["NVR_MAX_GRAPHS"] = str(max_graphs)
        try:
            nvr.lib().nvr_seq_reset_id_counter()
            p = nvr.LLMEngine(nvr.Config(skip_block_size_check=1, async_decode=async_on, **ecfg), _model_cfgs(mcfg))
        finally:
            os.environ.pop("NVR_MAX_GRAPHS", None)
        for pr, mt in reqs:
            p.add_request(pr, nvr.SamplingParams(temperature=0.0, max_tokens=mt, ignore_eos=True))
        trace, steps = [], 0
        while not p.is_finished():
            if steps in late:
                p.add_request(late[steps][0], nvr.SamplingParams(temperature=0.0, max_tokens=late[steps][1], ignore_eos=True))
            rec = p.step()
            assert all(t >= 0 for t in rec["tokens"]), rec
            st = p.get_stats()["scheduler"]
            trace.append((rec["is_prefill"], tuple(rec["seq_ids"]), tuple(rec["tokens"]), rec["num_finished"], st["decode_batches"],
                          st["finished_sequences"]))
            steps += 1
            assert steps < 300
        fin = {s.seq_id: s.token_ids for s in p.take_finished()}
        assert all(t >= 0 for toks in fin.values() for t in toks)
        return trace, fin, p.ahead_declined()
    ta, fa, declined = run(1, 2)
    ts, fs, _ = run(0, 256)
    assert declined > 0, "the scenario never filled the graph cache: the error path was not exercised"
    assert len(ta) == len(ts)
    for i, (a, b) in enumerate(zip(ta, ts)):
        assert a == b, f"step {i}: async {a} != sync {b}"
    assert fa == fs and len(fa) == 4


def test_tiled_weight_copies_do_not_change_a_bit():
    """Decode-sized steps stream tiled copies of the GEMM weights ([N/16][K/32][16][32]: 1 KiB contiguous per MFMA operand tile)
    instead of the row-major parameters: same values in the same summation order, so every step's logits are BIT-identical to a
    runner built with NVR_TILED_WEIGHTS=0 — including after checkpoint tensors were loaded (the copies are rebuilt)."""
    mcfg = mo.small(seed=4)
    ecfg = dict(max_num_seqs=6, max_num_batched_tokens=256, max_model_len=128, kvcache_block_size=16, num_kvcache_blocks=40)
    prompts = [oracle.fill_tokens(n, 2, i, mcfg.vocab_size).tolist() for i, n in enumerate([7, 30, 17, 1, 25])]
    rng = np.random.default_rng(12)
    newW = (rng.standard_normal((mcfg.hidden_size, mcfg.intermediate_size)) * 0.05).astype(np.float16)

    def run(tiled):
        os.environ["NVR_TILED_WEIGHTS"] = "1" if tiled else "0"
        try:
            nvr.lib().nvr_seq_reset_id_counter()
            p = nvr.LLMEngine(nvr.Config(skip_block_size_check=1, **ecfg), _model_cfgs(mcfg))
        finally:
            os.environ.pop("NVR_TILED_WEIGHTS", None)
        p.model_runner.load_tensor("layers.1.mlp.down_proj.weight", newW)
        for pr in prompts:
            p.add_request(pr, nvr.SamplingParams(temperature=0.0, max_tokens=10, ignore_eos=True))
        out = []
        while not p.is_finished():
            rec = p.step()
            out.append((rec["tokens"], p.model_runner.logits(rec["num_seqs"]).copy()))
        return out
    a, b = run(True), run(False)
    assert len(a) == len(b) > 8
    for (ta, la), (tb, lb) in zip(a, b):
        assert ta == tb and np.array_equal(la, lb)


@pytest.mark.parametrize("shape", ["small", "gqa2_d128"])
def test_qk_norm_checkpoint_end_to_end(tmp_path, shape):
    """A-27: a checkpoint with self_attn.q_norm / k_norm weights (real Qwen3 layout) runs with
    nvr_model_config.qk_norm = 1 — plain qkv GEMM, then one head-norm + RoPE + KV-store launch — in parity with the oracle that
    applies the reference's own RMSNorm to every q and k head before RoPE; graph and eager decode agree; without the flag the
    same tensors are refused (they are not part of the reference graph)."""
    from safetensors.numpy import save_file
    mcfg = mo.small(seed=5, qk_norm=True) if shape == "small" else \
        mo.small(seed=6, qk_norm=True, hidden_size=512, num_attention_heads=4, num_key_value_heads=2, head_dim=128, intermediate_size=768)
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
    r = _run_pair(mcfg, ecfg, prompts, sps, checkpoint=(sd, path))
    assert r["near_ties"] <= 2 and r["decode_steps"] >= 13, r
    e = _run_pair(mcfg, ecfg, prompts, sps, checkpoint=(sd, path), enforce_eager=True)
    assert e["finished"] == r["finished"]
    # norm weights of all ones (no checkpoint) are still a different graph from the reference's
    plain = _run_pair(mo.small(seed=5) if shape == "small" else
                      mo.small(seed=6, hidden_size=512, num_attention_heads=4, num_key_value_heads=2, head_dim=128, intermediate_size=768),
                      ecfg, prompts, sps)
    ones = _run_pair(mcfg, ecfg, prompts, sps)
    assert ones["finished"] != plain["finished"] and ones["finished"] != r["finished"]
    # the reference graph has no such parameters
    p = nvr.LLMEngine(nvr.Config(skip_block_size_check=1, **ecfg), _model_cfgs(mo.small(seed=5)))
    with pytest.warns(RuntimeWarning):
        assert sorted(p.model_runner.load_safetensors(path)) == sorted(sd)
    with pytest.raises(nvr.NvrError):
        p.model_runner.load_safetensors(path, strict=True)


@pytest.mark.parametrize("shape,dtype", [("small", "float16"), ("gqa2_d128", "float16"), ("small", "bfloat16")])
def test_use_bias_engine_parity_and_checkpoint(tmp_path, shape, dtype):
    """A-30: Qwen3Config::use_bias (qwen3.rs:54-55; qkv_proj :167, o_proj :178, gate_up_proj :276, down_proj :287).  candle's Linear is
    matmul, then broadcast_add — y = 16bit(16bit(x W^T) + b) —, so the product's use_bias graph runs plain GEMMs with a bias launch
    behind each (and SiluAndMul as its own launch on the biased halves).  Synthetic biases (generated like weights, keyed
    TID_BIAS + projection) and a checkpoint with *.bias tensors both run in parity with the oracle, graph and eager agree, the
    bias really changes the tokens, and without the flag the bias tensors are refused."""
    from safetensors.numpy import save_file
    kw = {} if shape == "small" else dict(hidden_size=512, num_attention_heads=4, num_key_value_heads=2, head_dim=128, intermediate_size=768)
    mcfg = mo.small(seed=11, use_bias=True, **kw)
    V, L = mcfg.vocab_size, mcfg.num_hidden_layers
    ecfg = dict(max_num_seqs=4, max_num_batched_tokens=256, max_model_len=128, kvcache_block_size=16, num_kvcache_blocks=40)
    prompts = [oracle.fill_tokens(n, 8, i, V).tolist() for i, n in enumerate([23, 70, 5])]
    sps = [dict(temperature=0.0, max_tokens=14, ignore_eos=True)] * 3
    r = _run_pair(mcfg, ecfg, prompts, sps, dtype=dtype)
    assert r["near_ties"] <= 2 and r["decode_steps"] >= 13, r
    e = _run_pair(mcfg, ecfg, prompts, sps, enforce_eager=True, dtype=dtype)
    assert e["finished"] == r["finished"]
    plain = _run_pair(mo.small(seed=11, **kw), ecfg, prompts, sps, dtype=dtype)
    assert plain["finished"] != r["finished"]
    if dtype != "float16":
        return
    # a checkpoint's biases (HF names, separate q / k / v and gate / up), mixed dtypes
    rng = np.random.default_rng(79)
    D, H, KVH, Hd, I = mcfg.hd(), mcfg.num_attention_heads, mcfg.num_key_value_heads, mcfg.hidden_size, mcfg.intermediate_size
    sd = {}
    for l in range(L):
        pre = f"model.layers.{l}."
        sd[pre + "self_attn.q_proj.bias"] = (0.1 * rng.standard_normal(H * D)).astype(np.float16)
        sd[pre + "self_attn.k_proj.bias"] = (0.1 * rng.standard_normal(KVH * D)).astype(np.float32)
        sd[pre + "self_attn.v_proj.bias"] = (0.1 * rng.standard_normal(KVH * D)).astype(np.float16)
        sd[pre + "self_attn.o_proj.bias"] = (0.05 * rng.standard_normal(Hd)).astype(np.float16)
        sd[pre + "mlp.gate_proj.bias"] = (0.1 * rng.standard_normal(I)).astype(np.float16)
        sd[pre + "mlp.up_proj.bias"] = (0.1 * rng.standard_normal(I)).astype(np.float32)
        sd[pre + "mlp.down_proj.bias"] = (0.05 * rng.standard_normal(Hd)).astype(np.float16)
    path = str(tmp_path / "bias.safetensors")
    save_file(sd, path)
    c = _run_pair(mcfg, ecfg, prompts, sps, checkpoint=(sd, path))
    assert c["near_ties"] <= 2 and c["finished"] != r["finished"], c
    # what landed: the packed local biases are the checkpoint's values in the runner's type
    nvr.lib().nvr_seq_reset_id_counter()
    p = nvr.LLMEngine(nvr.Config(skip_block_size_check=1, **ecfg), _model_cfgs(mcfg))
    assert p.model_runner.load_safetensors(path, strict=True) == []
    qkv_b = p.model_runner.weight("layers.1.qkv_b")
    want = np.concatenate([sd["model.layers.1.self_attn.q_proj.bias"].astype(np.float16), sd["model.layers.1.self_attn.k_proj.bias"].astype(np.float16),
                           sd["model.layers.1.self_attn.v_proj.bias"]])
    assert np.array_equal(qkv_b, want)
    assert np.array_equal(p.model_runner.weight("layers.0.down_b"), sd["model.layers.0.mlp.down_proj.bias"])
    with pytest.raises(nvr.NvrError):
        p.model_runner.load_tensor("model.layers.0.mlp.down_proj.bias", np.zeros(Hd + 8, np.float16))
    # the reference's default graph (use_bias false) has no such parameters
    q = nvr.LLMEngine(nvr.Config(skip_block_size_check=1, **ecfg), _model_cfgs(mo.small(seed=11, **kw)))
    with pytest.warns(RuntimeWarning):
        assert sorted(q.model_runner.load_safetensors(path)) == sorted(sd)


def test_split_kv_merges_ride_on_the_attention_launch_bit_identically():
    """r05: decode attention that is cut into partitions merges them without a second launch — plain split-KV on the last partition workgroup of a
    (sequence, kv head) to finish (tickets), and behind a batch-wide shared-prefix pass on the single own-partition workgroup (the shared partials come from
    the launch in front of it).  Same partials, same merge_partitions: per-step logits are BIT-identical to the runner that keeps the merge launch
    (NVR_ATTN_FUSED_MERGE=0, read when the runner is created), for a small batch (split-KV), a batch behind one system prompt (shared pass, whole batch)
    and the same with a stranger in the batch (grouped: keeps the merge launch either way)."""
    import os
    mcfg = mo.small(seed=9, hidden_size=512, num_attention_heads=4, num_key_value_heads=2, head_dim=128, intermediate_size=768)
    V = mcfg.vocab_size
    ecfg = dict(max_num_seqs=12, max_num_batched_tokens=2048, max_model_len=512, kvcache_block_size=64, num_kvcache_blocks=60, shared_prefix_min_seqs=4)
    system = oracle.fill_tokens(150, 4, 7, V).tolist()
    cases = {"split_kv": [oracle.fill_tokens(90 + 40 * i, 4, 300 + i, V).tolist() for i in range(3)],
             "shared_whole_batch": [system + oracle.fill_tokens(3 + 9 * i, 4, 100 + i, V).tolist() for i in range(9)],
             "shared_group_and_a_stranger": [system + oracle.fill_tokens(3 + 9 * i, 4, 100 + i, V).tolist() for i in range(8)] + [oracle.fill_tokens(70, 4, 999, V).tolist()]}

    def run(prompts, flag):
        os.environ["NVR_ATTN_FUSED_MERGE"] = flag
        try:
            nvr.lib().nvr_seq_reset_id_counter()
            p = nvr.LLMEngine(nvr.Config(skip_block_size_check=1, **ecfg), _model_cfgs(mcfg))
        finally:
            os.environ.pop("NVR_ATTN_FUSED_MERGE", None)
        for pr in prompts:
            p.add_request(pr, nvr.SamplingParams(temperature=0.0, max_tokens=10, ignore_eos=True))
        out = []
        while not p.is_finished():
            rec = p.step()
            out.append((rec["tokens"], p.model_runner.logits(rec["num_seqs"]).copy(), p.model_runner.last_shared_prefix_len() if not rec["is_prefill"] else -1))
        return out
    for name, prompts in cases.items():
        a, b = run(prompts, "1"), run(prompts, "0")
        assert len(a) == len(b) > 5
        if name != "split_kv":
            assert any(s > 0 for _, _, s in a), name                      # the shared pass did run
        for (ta, la, sa), (tb, lb, sb) in zip(a, b):
            assert ta == tb and sa == sb, name
            assert np.array_equal(la, lb), f"{name}: logits differ between the merge riding on the attention launch and the merge launch"


@pytest.mark.parametrize("nseq", [512, 300])
def test_shared_prefix_own_partitions_on_one_and_two_wave_workgroups_merge_bit_identically(nseq):
    """r05, BASELINE configs[4]'s geometry (8 kv heads): with >= 4096 (sequence, kv head) pairs behind the shared pass a ONE-wave workgroup streams a pair's
    own keys (>= 2048: two waves) and, as the last arriver by stream order, merges the pair with the shared partitions' partials (staged in LDS when the
    kernel starts).  Per-step logits are BIT-identical to the runner that keeps the merge launch (NVR_ATTN_FUSED_MERGE=0: the same one- / two-wave kernel,
    partials through the workspace, attn_merge_kernel), and the batch's tokens equal the plain paged path's (no shared pass) wherever the plain logits are
    not a near tie.  (The 512-sequence form is checked against the oracle through test_scale's configs[4] test, the kernel through test_kernels_gpu.)"""
    import os
    mcfg = mo.small(seed=21, hidden_size=256, num_attention_heads=16, num_key_value_heads=8, head_dim=128, intermediate_size=512, num_hidden_layers=2)
    V = mcfg.vocab_size
    ecfg = dict(max_num_seqs=nseq, max_num_batched_tokens=1 << 17, max_model_len=320, kvcache_block_size=64, num_kvcache_blocks=nseq * 2 + 8)
    system = oracle.fill_tokens(128, 4, 7, V).tolist()
    prompts = [system + oracle.fill_tokens(3 + (7 * i) % 50, 4, 100 + i, V).tolist() for i in range(nseq)]

    def run(flag, min_seqs):
        os.environ["NVR_ATTN_FUSED_MERGE"] = flag
        try:
            nvr.lib().nvr_seq_reset_id_counter()
            p = nvr.LLMEngine(nvr.Config(skip_block_size_check=1, shared_prefix_min_seqs=min_seqs, **ecfg), _model_cfgs(mcfg))
        finally:
            os.environ.pop("NVR_ATTN_FUSED_MERGE", None)
        for pr in prompts:
            p.add_request(pr, nvr.SamplingParams(temperature=0.0, max_tokens=6, ignore_eos=True))
        out = []
        while not p.is_finished():
            rec = p.step()
            if not rec["is_prefill"]:
                out.append((rec["tokens"], p.model_runner.logits(rec["num_seqs"]).copy(), p.model_runner.last_shared_prefix_len()))
        return out
    a, b, plain = run("1", 4), run("0", 4), run("1", -1)
    assert len(a) == len(b) == len(plain) >= 5
    for (ta, la, sa), (tb, lb, sb), (tp, lp, sp) in zip(a, b, plain):
        assert sa == sb == 128 and sp == 0
        assert ta == tb and np.array_equal(la, lb), "logits differ between the merge on the own-partition workgroup and the merge launch"
        top2 = np.sort(lp, axis=1)[:, -2:]
        clear = (top2[:, 1] - top2[:, 0]) > 2e-2
        assert clear.sum() > nseq // 2
        assert np.array_equal(np.asarray(ta)[clear], np.asarray(tp)[clear]), "shared pass vs plain paged attention: tokens differ off a near tie"
        if not np.array_equal(np.asarray(ta), np.asarray(tp)): break       # (token streams have parted at a near tie: later steps are not comparable)


@pytest.mark.parametrize("shape", ["d64_g2", "d128_g2"])
def test_shared_prefix_decode_attention_engine_parity(shape):
    """BASELINE configs[4] in small: every request starts with the same system prompt, BlockManager::allocate shares its full
    blocks (block_manager.rs:181-197), and decode batches of >= shared_prefix_min_seqs sequences send the shared keys through one
    MFMA pass (nvr_paged_attn_decode_shared).  Logits / tokens stay in parity with the oracle's plain paged attention, graph and
    eager and launch-ahead agree, the plain path (feature off) produces the same tokens, and the shared length follows the
    batch: a request WITHOUT the system prompt in the batch is attended to in full by the row kernel while the others keep the pass."""
    mcfg = mo.small(seed=9) if shape == "d64_g2" else \
        mo.small(seed=9, hidden_size=512, num_attention_heads=4, num_key_value_heads=2, head_dim=128, intermediate_size=768)
    V = mcfg.vocab_size
    ecfg = dict(max_num_seqs=12, max_num_batched_tokens=2048, max_model_len=512, kvcache_block_size=64, num_kvcache_blocks=60)
    system = oracle.fill_tokens(150, 4, 7, V).tolist()                    # 2 full blocks of 64 shared + 22 tokens recomputed per sequence
    prompts = [system + oracle.fill_tokens(3 + 9 * i, 4, 100 + i, V).tolist() for i in range(9)]
    sps = [dict(temperature=0.0, max_tokens=8 + 3 * i, ignore_eos=True) for i in range(9)]   # the batch shrinks below the threshold on the way
    on = dict(shared_prefix_min_seqs=4)
    r = _run_pair(mcfg, ecfg, prompts, sps, product_kw=on)
    assert r["near_ties"] <= 2 and r["decode_steps"] >= 30, r
    e = _run_pair(mcfg, ecfg, prompts, sps, product_kw=on, enforce_eager=True)
    off = _run_pair(mcfg, ecfg, prompts, sps, product_kw=dict(shared_prefix_min_seqs=-1))
    assert e["finished"] == r["finished"] == off["finished"]
    # launch-ahead (its logits accessor refers to the step launched ahead, so only the token streams are compared)
    # the path really is taken: 128 shared tokens while >= 4 sequences are alive, 0 afterwards and with the feature off
    nvr.lib().nvr_seq_reset_id_counter()
    p = nvr.LLMEngine(nvr.Config(skip_block_size_check=1, **ecfg, **on), _model_cfgs(mcfg))
    for pr, sp in zip(prompts, sps):
        p.add_request(pr, nvr.SamplingParams(**sp))
    seen = set()
    while not p.is_finished():
        rec = p.step()
        if not rec["is_prefill"]:
            seen.add((rec["num_seqs"] >= 4, p.model_runner.last_shared_prefix_len()))
    assert seen == {(True, 128), (False, 0)}, sorted(seen)
    gen = {}
    for is_prefill, seq_ids, tokens, _, _ in _run_product(mcfg, ecfg, prompts, sps, async_decode=1, **on):
        for sid, tok in zip(seq_ids, tokens):
            if tok >= 0:
                gen.setdefault(sid, []).append(tok)
    assert {sid: toks[-len(gen[sid]):] for sid, toks in r["finished"].items()} == gen
    # mixed batch: one request without the system prompt -> no common leading block
    prompts2 = prompts[:6] + [oracle.fill_tokens(70, 4, 55, V).tolist()]
    sps2 = [dict(temperature=0.0, max_tokens=10, ignore_eos=True)] * 7
    m = _run_pair(mcfg, ecfg, prompts2, sps2, product_kw=on)
    assert m["near_ties"] <= 2, m
    # ... and the six that do have it still take the shared pass as a group inside the batch of seven
    nvr.lib().nvr_seq_reset_id_counter()
    p = nvr.LLMEngine(nvr.Config(skip_block_size_check=1, **ecfg, **on), _model_cfgs(mcfg))
    for pr, sp in zip(prompts2, sps2):
        p.add_request(pr, nvr.SamplingParams(**sp))
    seen = set()
    while not p.is_finished():
        rec = p.step()
        if not rec["is_prefill"]:
            seen.add((rec["num_seqs"], p.model_runner.last_shared_prefix_len(), p.model_runner.last_shared_prefix_rows()))
    assert (7, 128, 6) in seen, seen


@pytest.mark.parametrize("seed", [11, 12, 13, 14])
def test_random_workloads_with_shared_prefix_pass(seed):
    """Randomised soak of the r02 decode extensions together: 64-token blocks, most requests behind one system prompt of 1-3 full
    blocks (some not: the batch-wide shared length then drops to 0 or to what the batch has in common), shared_prefix_min_seqs = 2,
    chunked prefill on odd seeds, preemption pressure from a small pool.  Same per-step comparison against the oracle as above."""
    rng = np.random.default_rng(seed)
    mcfg = mo.small(seed=seed % 5)
    V = mcfg.vocab_size
    nreq = int(rng.integers(6, 12))
    ecfg = dict(max_num_seqs=int(rng.integers(4, 9)), max_num_batched_tokens=int(rng.choice([512, 1024])), max_model_len=512,
                kvcache_block_size=64, num_kvcache_blocks=int(rng.integers(16, 30)), enable_chunked_prefill=bool(seed & 1))
    system = oracle.fill_tokens(64 * 3, seed, 4242, V).tolist()
    prompts, sps = [], []
    for i in range(nreq):
        own = oracle.fill_tokens(int(rng.integers(1, 90)), seed, i, V).tolist()
        pr = (system[:64 * int(rng.integers(1, 4))] + own) if rng.random() < 0.75 else own
        prompts.append(pr[:ecfg["max_num_batched_tokens"] - 1])
        sps.append(dict(temperature=0.0, max_tokens=int(rng.integers(2, 30)), ignore_eos=True))
    r = _run_pair(mcfg, ecfg, prompts, sps, max_steps=3000, product_kw=dict(shared_prefix_min_seqs=2))
    assert len(r["finished"]) == nreq
    assert r["near_ties"] <= 4, r
    off = _run_pair(mcfg, ecfg, prompts, sps, max_steps=3000, product_kw=dict(shared_prefix_min_seqs=-1))
    assert off["shared_steps"] == 0
    # both runs are in parity with the oracle step by step; their token streams can only part at a counted numerical near-tie
    assert off["finished"] == r["finished"] or r["near_ties"] + off["near_ties"] > 0, (r["near_ties"], off["near_ties"])
    _SOAK_SEEN["shared_steps"] = _SOAK_SEEN.get("shared_steps", 0) + r["shared_steps"]
    _SOAK_SEEN["shared_runs"] = _SOAK_SEEN.get("shared_runs", 0) + 1
    _SOAK_SEEN["shared_preemptions"] = _SOAK_SEEN.get("shared_preemptions", 0) + r["oracle"].scheduler.stats.preemptions


def test_random_workloads_did_take_the_shared_prefix_pass():
    if _SOAK_SEEN.get("shared_runs", 0) < 4:
        pytest.skip("runs after the four shared-prefix soak cases")
    assert _SOAK_SEEN["shared_steps"] > 20, _SOAK_SEEN


def test_shared_prefix_group_is_the_most_common_prompt():
    """Three system prompts in one batch (5 / 4 / 3 requests) + one request without any: no prompt has a majority, the group is the
    most common one (5 rows); everybody stays in parity with the oracle."""
    mcfg = mo.small(seed=10)
    V = mcfg.vocab_size
    ecfg = dict(max_num_seqs=16, max_num_batched_tokens=4096, max_model_len=512, kvcache_block_size=64, num_kvcache_blocks=90)
    systems = [oracle.fill_tokens(64 * n, 4, 70 + i, V).tolist() for i, n in enumerate([2, 1, 3])]
    prompts = []
    for i, cnt in enumerate([5, 4, 3]):
        prompts += [systems[i] + oracle.fill_tokens(5 + 3 * j, 4, 300 + 10 * i + j, V).tolist() for j in range(cnt)]
    prompts.append(oracle.fill_tokens(90, 4, 999, V).tolist())
    order = np.random.default_rng(4).permutation(len(prompts))
    prompts = [prompts[i] for i in order]
    sps = [dict(temperature=0.0, max_tokens=6, ignore_eos=True)] * len(prompts)
    r = _run_pair(mcfg, ecfg, prompts, sps, product_kw=dict(shared_prefix_min_seqs=4))
    assert r["near_ties"] <= 2 and r["shared_steps"] >= 4, r
    nvr.lib().nvr_seq_reset_id_counter()
    p = nvr.LLMEngine(nvr.Config(skip_block_size_check=1, shared_prefix_min_seqs=4, **ecfg), _model_cfgs(mcfg))
    for pr, sp in zip(prompts, sps):
        p.add_request(pr, nvr.SamplingParams(**sp))
    seen = set()
    while not p.is_finished():
        rec = p.step()
        if not rec["is_prefill"]:
            seen.add((rec["num_seqs"], p.model_runner.last_shared_prefix_len(), p.model_runner.last_shared_prefix_rows()))
    assert (13, 128, 5) in seen, seen


def test_prefill_reads_kv_from_the_caches_contiguous_rows_or_block_tables():
    """The flash prefill kernel takes K / V from the caches, never from the step's qkv buffer (the 256^2 qkv GEMM then writes them once and
    skips the k / v columns of the qkv buffer).  A whole-prompt prefill whose sequences all sit in consecutive cache blocks reads the cache
    rows in the kernel's contiguous form (nvr_runner_last_prefill_kv_source = 1); once the free list has been recycled the block tables are
    no longer consecutive and the kernel walks them (2), as it does behind a cached prefix (2).  The forms see the same bits: logits of the
    prefill and of the following decode steps are IDENTICAL whichever way K / V reached the attention kernel, and in parity with the
    oracle.  Shape: hidden 512, 4:2 heads x 128 (qkv width 1024, K 512: the 256^2 GEMM route for batches of >= 256 rows)."""
    mcfg = mo.small(seed=12, hidden_size=512, num_attention_heads=4, num_key_value_heads=2, head_dim=128, intermediate_size=768)
    V = mcfg.vocab_size
    # 64 blocks: the three prompts need 19 + 5 + 33; after the scrambling requests (24 blocks, freed interleaved) the allocations run past
    # the never-used blocks into the freed ones
    ecfg = dict(max_num_seqs=8, max_num_batched_tokens=2048, max_model_len=640, kvcache_block_size=16, num_kvcache_blocks=64)
    prompts = [oracle.fill_tokens(n, 6, i, V).tolist() for i, n in enumerate([300, 70, 513])]
    sps = [dict(temperature=0.0, max_tokens=4, ignore_eos=True)] * 3

    def run(scramble, **kw):
        nvr.lib().nvr_seq_reset_id_counter()
        p = nvr.LLMEngine(nvr.Config(skip_block_size_check=1, **ecfg, **kw), _model_cfgs(mcfg))
        if scramble:          # requests of different lengths that finish at different steps: their blocks return to the free list interleaved
            for i, (n, mt) in enumerate([(40, 3), (90, 9), (20, 5), (150, 2), (33, 7)]):
                p.add_request(oracle.fill_tokens(n, 7, 50 + i, V).tolist(), nvr.SamplingParams(temperature=0.0, max_tokens=mt, ignore_eos=True))
            while not p.is_finished():
                p.step()
            p.take_finished()
        for pr, sp in zip(prompts, sps):
            p.add_request(pr, nvr.SamplingParams(**sp))
        out, src, tables = [], None, None
        while not p.is_finished():
            rec = p.step()
            if rec["is_prefill"]:
                src = p.model_runner.last_prefill_kv_source()
                tables = [list(s.block_table) for s in p.last_batch()]
            else:
                assert p.model_runner.last_prefill_kv_source() == -1
            out.append((rec["is_prefill"], tuple(rec["tokens"]), p.model_runner.logits(rec["num_seqs"]).copy()))
        return out, src, tables
    fresh, src_f, tab_f = run(False)
    scr, src_s, tab_s = run(True)
    consecutive = lambda t: all(b == a + 1 for a, b in zip(t, t[1:]))
    assert src_f == 1 and all(consecutive(t) for t in tab_f), (src_f, tab_f)
    assert src_s == 2 and not all(consecutive(t) for t in tab_s), (src_s, tab_s)
    assert len(fresh) == len(scr) == 4
    for a, b in zip(fresh, scr):
        assert a[:2] == b[:2] and np.array_equal(a[2], b[2]), "K/V from contiguous cache rows and through the block tables must give identical logits"
    # the longest prompt again behind a cached prefix of 25 full blocks: through the block tables, same last-token logits
    nvr.lib().nvr_seq_reset_id_counter()
    p = nvr.LLMEngine(nvr.Config(skip_block_size_check=1, **ecfg), _model_cfgs(mcfg))
    p.add_request(prompts[2][:400], nvr.SamplingParams(temperature=0.0, max_tokens=6, ignore_eos=True))     # stays alive: its blocks keep their hashes
    assert p.step()["is_prefill"]
    p.add_request(prompts[2], nvr.SamplingParams(**sps[2]))
    rec = p.step()
    assert rec["is_prefill"] and rec["num_tokens"] == 513 - 400, rec
    assert p.model_runner.last_prefill_kv_source() == 2
    assert rec["tokens"][0] == fresh[0][1][2]
    # and against the oracle
    r = _run_pair(mcfg, ecfg, prompts, sps)
    assert r["near_ties"] <= 1, r


@pytest.mark.parametrize("shape", ["small", "gqa2_d128", "qk_norm_bias", "chunked"])
def test_float32_path_engine_parity(tmp_path, shape):
    """Config.dtype = "float32" (config.rs:51,113-116; the reference's own CPU tests run f32): the reference-precision path
    (kernels/f32_path.hip: every op on 4-byte storage, plain FMA kernels, eager, one GPU) against the oracle's f32 arithmetic — the SAME
    unrounded synthetic weights, logits within 2e-4 (summation order only), greedy tokens equal with NO near-tie allowance on these
    scenarios, a stochastic row too; prefix-sharing prompts, a preemption-sized pool; q/k norm + bias graph from a checkpoint; f32
    refuses tensor parallelism.  Cached prefixes are skipped and chunked prompts continue through the block tables (the f32 attention kernel's paged
    form), as on the 16-bit path."""
    from safetensors.numpy import save_file
    kw = dict(small={}, gqa2_d128=dict(hidden_size=512, num_attention_heads=4, num_key_value_heads=2, head_dim=128, intermediate_size=768),
              qk_norm_bias=dict(qk_norm=True, use_bias=True), chunked={})[shape]
    mcfg = mo.small(seed=13, **kw)
    V = mcfg.vocab_size
    ecfg = dict(max_num_seqs=4, max_num_batched_tokens=256, max_model_len=160, kvcache_block_size=16, num_kvcache_blocks=24)
    if shape == "chunked":                                # A-23 on the f32 path: prompts cut by a 48-token budget, later chunks through the block tables
        ecfg.update(max_num_batched_tokens=48, enable_chunked_prefill=True)
    shared = oracle.fill_tokens(40, 8, 99, V).tolist()
    prompts = [oracle.fill_tokens(23, 8, 0, V).tolist(), shared + oracle.fill_tokens(30, 8, 1, V).tolist(), shared + oracle.fill_tokens(5, 8, 2, V).tolist(),
               oracle.fill_tokens(60, 8, 3, V).tolist()]
    sps = [dict(temperature=0.0, max_tokens=20, ignore_eos=True)] * 3 + [dict(temperature=0.8, top_k=40, top_p=0.9, max_tokens=12, ignore_eos=True)]
    checkpoint = None
    if shape == "qk_norm_bias":
        rng = np.random.default_rng(80)
        D, Hd = mcfg.hd(), mcfg.hidden_size
        sd = {}
        for l in range(mcfg.num_hidden_layers):
            sd[f"model.layers.{l}.self_attn.q_norm.weight"] = (1 + 0.3 * rng.standard_normal(D)).astype(np.float32)
            sd[f"model.layers.{l}.self_attn.k_norm.weight"] = (1 + 0.3 * rng.standard_normal(D)).astype(np.float16)
            sd[f"model.layers.{l}.mlp.down_proj.bias"] = (0.05 * rng.standard_normal(Hd)).astype(np.float32)
        path = str(tmp_path / "f32.safetensors")
        save_file(sd, path)
        checkpoint = (sd, path)
    r = _run_pair(mcfg, ecfg, prompts, sps, dtype="float32", checkpoint=checkpoint)
    assert r["near_ties"] == 0 and r["decode_steps"] >= 19 and r["max_err"] < F32_TOL, (r["near_ties"], r["max_err"])
    # what the runner holds is the generator's unrounded f32 values (fp16 runners hold their fp16 roundings)
    p = nvr.LLMEngine(nvr.Config(skip_block_size_check=1, dtype="float32", **ecfg), _model_cfgs(mcfg))
    w = p.model_runner.weight("layers.1.gate_up")
    assert w.dtype == np.float32 and np.array_equal(w, r["oracle"].ranks[0].layers[1]["gate_up"])
    # a float32 tensor-parallel rank holds its slice of the same unrounded values (r05: tests/test_tp.py runs such ranks against the f32 oracle)
    p1 = nvr.LLMEngine(nvr.Config(skip_block_size_check=1, dtype="float32", tensor_parallel_size=2, tensor_parallel_rank=1, **ecfg), _model_cfgs(mcfg))
    I2 = mcfg.intermediate_size // 2
    full = r["oracle"].ranks[0].layers[1]["gate_up"]
    assert np.array_equal(p1.model_runner.weight("layers.1.gate_up"), np.concatenate([full[I2:2 * I2], full[3 * I2:]]))


@pytest.mark.parametrize("shape", ["gqa2_d128", "qk_norm_bias"])
def test_float32_decode_sized_steps_fuse_add_norm_into_their_gemvs_bit_identically(shape):
    """r05: on a float32 runner a decode-sized step (1..8 rows) forms the residual add + RMSNorm inside the workgroups of the GEMV that consumes it (qkv + RoPE +
    KV store, gate_up + SiluAndMul), the residual stream alternating between two buffers.  Per-step logits are BIT-identical to the runner that keeps the
    add + norm launches (NVR_F32_FUSED_NORM=0, read when the runner is created) — decode batches of 1, 3 and 4 rows, eager and as hipGraph replays; with q / k
    head norms the qkv side keeps its parts and only the MLP side fuses."""
    import os
    kw = dict(gqa2_d128=dict(hidden_size=512, num_attention_heads=4, num_key_value_heads=2, head_dim=128, intermediate_size=768), qk_norm_bias=dict(qk_norm=True, use_bias=True))[shape]
    mcfg = mo.small(seed=17, **kw)
    V = mcfg.vocab_size
    prompts = [oracle.fill_tokens(9 + 7 * i, 8, 40 + i, V).tolist() for i in range(4)]

    def run(flag, eager):
        os.environ["NVR_F32_FUSED_NORM"] = flag
        try:
            nvr.lib().nvr_seq_reset_id_counter()
            p = nvr.LLMEngine(nvr.Config(skip_block_size_check=1, dtype="float32", max_num_seqs=4, max_num_batched_tokens=256, max_model_len=128, kvcache_block_size=16,
                                         num_kvcache_blocks=40, enforce_eager=eager), _model_cfgs(mcfg))
        finally:
            os.environ.pop("NVR_F32_FUSED_NORM", None)
        for i, pr in enumerate(prompts):
            p.add_request(pr, nvr.SamplingParams(temperature=0.0, max_tokens=4 + 5 * i, ignore_eos=True))      # the batch shrinks 4 -> 3 -> 2 -> 1 rows
        out = []
        while not p.is_finished():
            rec = p.step()
            out.append((rec["is_prefill"], rec["num_seqs"], rec["tokens"], p.model_runner.logits(rec["num_seqs"]).copy()))
        return out
    ref = run("0", True)
    assert {n for pre, n, _, _ in ref if not pre} == {1, 2, 3, 4}
    for eager in (True, False):
        got = run("1", eager)
        assert len(got) == len(ref)
        for (pa, na, ta, la), (pb, nb, tb, lb) in zip(got, ref):
            assert (pa, na, ta) == (pb, nb, tb)
            assert np.array_equal(la.view(np.uint32), lb.view(np.uint32)), f"float32 logits differ between the fused and the unfused add + norm (eager={eager}, rows={na})"


def test_graft_entry_smoke_runs_in_a_fresh_process():
    """The driver's round-end check, as the driver runs it: __graft_entry__.smoke() in a process of its own — outside this suite's conftest, which creates
    engines with async_decode = 0 unless a test says otherwise (r05: smoke() read per-step logits from the DEFAULT engine, whose next decode step is
    already in flight, and nothing in the suite could see it)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0 and "smoke ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])
