"""Pins the CPU oracle against every known-answer value the reference's own in-file unit
tests hold for the hot path and against the public XXH64 vectors.
Each test cites the reference test it mirrors (file:line under /root/reference)."""
import math
import os
import struct

import numpy as np
import pytest

import oracle
from oracle import engine_oracle as eo

SP = eo.SamplingParams


# ---------------- B.3 XXH64 (public spec vectors; block_manager.rs:109-123 byte layout) -----------
def test_xxh64_spec_vectors():
    assert oracle.xxh64(b"") == 0xEF46DB3751D8E999
    assert oracle.xxh64(b"abc") == 0x44BC2CF5AD770999
    assert eo.xxh64(b"") == 0xEF46DB3751D8E999
    assert eo.xxh64(b"abc") == 0x44BC2CF5AD770999


@pytest.mark.parametrize("toks,prefix,expect", [
    ([1, 2, 3, 4, 5], None, 0xBC50EBD6BC8FA148),          # block_manager.rs:407-423 inputs
    ([1, 2, 3, 4, 6], None, 0x9D06FDE8C97293D0),
    ([1, 2, 3, 4, 5], 12345, 0xABC6D5998E62B562),
    ([1, 2, 3, 4], None, 0x73F859A04F669E6D),             # blocks of block_manager.rs:459-484
    ([5, 6, 7, 8], 0x73F859A04F669E6D, 0x087884241753DF3E),
    ([9, 10, 11, 12], 0x73F859A04F669E6D, 0x8FCDFAB13E7748AA),    # seq2's 2nd block, chained on [1,2,3,4]
    (list(range(256)), None, 0x486ADFCC62236EFE),
    (list(range(256, 512)), 0x486ADFCC62236EFE, 0x13BD65FA35AB695D),
])
def test_block_hash_vectors(toks, prefix, expect):
    assert oracle.block_hash(toks, prefix) == expect
    assert eo.BlockManager.compute_hash(toks, prefix) == expect


def test_xxh64_matches_python_xxhash_all_tail_lengths():
    xxhash = pytest.importorskip("xxhash")
    rng = np.random.default_rng(0)
    for n in list(range(0, 70)) + [255, 256, 257, 2048, 2056]:
        d = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        for seed in (0, 1, 2**63 + 5):
            ref = xxhash.xxh64_intdigest(d, seed)
            assert oracle.xxh64(d, seed) == ref
            assert eo.xxh64(d, seed) == ref


def test_hash_consistency_and_prefix():          # block_manager.rs:407-423
    h1 = eo.BlockManager.compute_hash([1, 2, 3, 4, 5])
    assert h1 == eo.BlockManager.compute_hash([1, 2, 3, 4, 5])
    assert h1 != eo.BlockManager.compute_hash([1, 2, 3, 4, 6])
    assert h1 != eo.BlockManager.compute_hash([1, 2, 3, 4, 5], 12345)


# ---------------- B.1 integer state machines --------------------------------------------------------
def test_block_new_and_refcount():               # block_manager.rs:369-395
    b = eo.Block(42)
    assert (b.block_id, b.ref_count, b.hash, b.is_free()) == (42, 0, None, True)
    b.ref_count += 1
    assert not b.is_free()
    b.reset()
    assert b.ref_count == 1                      # :44-48 reset() sets ref_count 1


def test_bm_basic_allocation():                  # block_manager.rs:426-438
    bm = eo.BlockManager(10, 4)
    s = eo.Sequence(list(range(1, 10)), SP(), block_size=4)
    assert bm.can_allocate(s)
    bm.allocate(s)
    assert s.block_table == [0, 1, 2]
    st = bm.get_stats()
    assert (st["free_blocks"], st["used_blocks"]) == (7, 3)


def test_bm_deallocation():                      # block_manager.rs:441-456
    bm = eo.BlockManager(10, 4)
    s = eo.Sequence(list(range(1, 10)), SP(), block_size=4)
    bm.allocate(s)
    bm.deallocate(s)
    assert s.block_table == [] and s.num_cached_tokens == 0
    assert bm.get_stats()["free_blocks"] == 10 and bm.get_stats()["used_blocks"] == 0
    assert list(bm.free_block_ids) == [3, 4, 5, 6, 7, 8, 9, 2, 1, 0]     # A-2: reverse dealloc, push_back


def test_bm_prefix_caching():                    # block_manager.rs:459-484
    bm = eo.BlockManager(10, 4)
    s1 = eo.Sequence([1, 2, 3, 4, 5, 6, 7, 8], SP(), block_size=4)
    s2 = eo.Sequence([1, 2, 3, 4, 9, 10, 11, 12], SP(), block_size=4)
    bm.allocate(s1)
    bm.allocate(s2)
    assert s2.num_cached_tokens == 4
    assert bm.blocks[s1.block_table[0]].ref_count == 2
    assert s1.block_table == [0, 1] and s2.block_table == [0, 2]


def test_bm_append_operations():                 # block_manager.rs:487-506
    bm = eo.BlockManager(10, 4)
    s = eo.Sequence([1, 2, 3], SP(), block_size=4)
    bm.allocate(s)
    assert len(s.block_table) == 1
    s.append_token(4)
    assert bm.can_append(s)
    bm.may_append(s)
    assert len(s.block_table) == 1
    assert bm.blocks[s.block_table[0]].hash == eo.BlockManager.compute_hash([1, 2, 3, 4])
    s.append_token(5)
    assert bm.can_append(s)
    bm.may_append(s)
    assert len(s.block_table) == 2


def test_bm_memory_exhaustion():                 # block_manager.rs:531-539
    bm = eo.BlockManager(2, 4)
    s = eo.Sequence(list(range(1, 13)), SP(), block_size=4)
    assert not bm.can_allocate(s)
    with pytest.raises(RuntimeError):
        bm.allocate(s)


def test_sequence_blocks_300_at_256():           # sequence.rs:288-303
    s = eo.Sequence(list(range(300)), SP())
    assert s.num_blocks() == 2 and s.last_block_num_tokens() == 44
    assert len(s.get_block_tokens(0)) == 256 and len(s.get_block_tokens(1)) == 44


def test_sequence_stop_rules_and_fsm():          # sequence.rs:306-362
    s = eo.Sequence([1, 2, 3], SP(max_tokens=2))
    assert not s.should_stop(None)
    s.append_token(4); assert not s.should_stop(None)
    s.append_token(5); assert s.should_stop(None)
    s = eo.Sequence([1, 2, 3], SP(max_tokens=10))
    s.append_token(2); assert s.should_stop(2)
    s = eo.Sequence([1, 2, 3], SP(max_tokens=10, ignore_eos=True))
    s.append_token(2); assert not s.should_stop(2)
    s = eo.Sequence([1, 2, 3], SP())
    assert s.status == eo.WAITING and s.can_schedule() and not s.is_finished()
    s.status = eo.RUNNING; assert not s.can_schedule()
    s.block_table = [1]; s.num_cached_tokens = 4
    s.preempt(); assert s.status == eo.PREEMPTED and s.can_schedule() and s.block_table == [] and s.num_cached_tokens == 0
    s.finish(); assert s.is_finished()


def _sched_cfg(**kw):                            # scheduler.rs:372-387
    d = dict(max_num_seqs=10, max_num_batched_tokens=1000, eos_token_id=2, kvcache_block_size=16,
             num_kvcache_blocks=100)
    d.update(kw)
    return eo.Config(**d)


def _mk(sched, toks, **sp):
    s = eo.Sequence(toks, SP(**sp), block_size=sched.block_manager.block_size)
    sched.add_sequence(s)
    return s


def test_scheduler_prefill_batch_of_three():     # scheduler.rs:417-439
    sc = eo.Scheduler(_sched_cfg())
    for p in ([1, 2, 3, 4, 5], [6, 7, 8], [9, 10, 11, 12]):
        _mk(sc, p, max_tokens=10)
    seqs, is_prefill = sc.schedule()
    assert is_prefill and len(seqs) == 3
    assert sc.get_queue_lengths() == (0, 3)


def test_scheduler_decode_after_prefill():       # scheduler.rs:442-465
    sc = eo.Scheduler(_sched_cfg())
    _mk(sc, [1, 2, 3, 4, 5], max_tokens=10)
    seqs, is_prefill = sc.schedule()
    assert is_prefill
    sc.postprocess(seqs, [6])
    seqs, is_prefill = sc.schedule()
    assert not is_prefill and len(seqs) == 1 and len(seqs[0]) == 6


def test_scheduler_finish_max_tokens_and_eos():  # scheduler.rs:468-505
    sc = eo.Scheduler(_sched_cfg())
    _mk(sc, [1, 2, 3], max_tokens=1)
    seqs, _ = sc.schedule()
    sc.postprocess(seqs, [4])
    assert sc.is_finished() and sc.stats.finished_sequences == 1
    sc = eo.Scheduler(_sched_cfg())
    _mk(sc, [1, 2, 3], max_tokens=10)
    seqs, _ = sc.schedule()
    sc.postprocess(seqs, [2])
    assert sc.is_finished()


def test_scheduler_batch_limits():               # scheduler.rs:508-528
    sc = eo.Scheduler(_sched_cfg(max_num_seqs=2, max_num_batched_tokens=10))
    for _ in range(5):
        _mk(sc, [1, 2, 3, 4, 5, 6], max_tokens=10)
    seqs, is_prefill = sc.schedule()
    assert is_prefill and len(seqs) == 1
    assert sc.get_queue_lengths() == (4, 1)


def test_scheduler_stats():                      # scheduler.rs:556-578
    sc = eo.Scheduler(_sched_cfg())
    for p in ([1, 2, 3], [4, 5, 6], [7, 8, 9]):
        _mk(sc, p, max_tokens=1)
    seqs, _ = sc.schedule()
    sc.postprocess(seqs, [10, 11, 12])
    st = sc.stats
    assert (st.total_sequences, st.finished_sequences, st.prefill_batches) == (3, 3, 1)
    assert st.avg_prefill_batch_size == 3.0
    assert st.finished_sequences / st.total_sequences == 1.0


def test_runner_decode_inputs():                 # model_runner.rs:501-520
    a = eo.Sequence([1, 2, 3], SP(), block_size=16); a.block_table = [0]
    b = eo.Sequence([4, 5], SP(), block_size=16); b.block_table = [1]
    p = eo.prepare_decode([a, b], 16)
    assert p["input_ids"] == [3, 5] and p["positions"] == [2, 1]
    assert p["context_lens"] == [3, 2] and p["slot_mapping"] == [2, 17]


def test_tp_vocab_mask_and_local_index():        # embed_head.rs:509-540
    m, _ = oracle.vocab_mask_local([10, 25, 60, 75], 0, 50)
    assert m.tolist() == [1, 1, 0, 0]
    m, loc = oracle.vocab_mask_local([55, 60, 75, 90], 50, 100)
    assert loc.tolist() == [5, 10, 25, 40] and m.tolist() == [1, 1, 1, 1]


def test_config_and_sampling_param_rules():      # config.rs:194-217, sampling_params.rs:127-168
    c = eo.Config()
    assert (c.max_num_batched_tokens, c.max_num_seqs, c.tensor_parallel_size) == (32768, 512, 1)
    with pytest.raises(ValueError):
        eo.Config(kvcache_block_size=100).validate()
    with pytest.raises(ValueError):
        eo.Config(tensor_parallel_size=10).validate()
    sp = SP()
    assert sp.temperature == 1.0 and sp.max_tokens == 64 and not sp.ignore_eos
    for bad in (dict(temperature=-1.0), dict(max_tokens=0), dict(top_p=1.5)):
        with pytest.raises(ValueError):
            SP(**bad).validate()
    assert SP(temperature=0.0).is_greedy() and not SP().is_greedy()


# ---------------- B.2 float ops ----------------------------------------------------------------------
def test_greedy_argmax_kats():                   # sampler.rs:333-356,423-443
    assert oracle.argmax([1.0, 2.0, 5.0, 1.5]) == 2
    assert oracle.sample([1.0, 2.0, 3.0], 0.0) == 2
    assert oracle.sample([1.0, 2.0, 3.0], 0.0, key=99) == 2
    assert oracle.argmax([3.0, 7.0, 7.0, 1.0]) == 1          # A-12: lowest index wins


def test_top_k_kat():                            # sampler.rs:359-374
    out = oracle.top_k([1.0, 5.0, 2.0, 4.0, 3.0], 3)
    assert out.tolist() == [-math.inf, 5.0, -math.inf, 4.0, 3.0]
    assert oracle.top_k([1.0, 2.0], 0).tolist() == [1.0, 2.0]   # A-18


def test_top_p_kat():                            # sampler.rs:377-389
    out = oracle.top_p([0.0, 10.0, 5.0, 1.0], 0.9)
    assert out[1] == 10.0 and np.isinf(out[[0, 2, 3]]).all()


def test_causal_prefill_matches_masked_softmax():   # attention.rs:470-485 mask + :449-457 scale
    assert abs(1.0 / math.sqrt(64) - 0.125) < 1e-6
    rng = np.random.default_rng(1)
    L, H, KVH, D = 3, 2, 1, 8
    q = rng.standard_normal((L, H, D)).astype(np.float32)
    k = rng.standard_normal((L, KVH, D)).astype(np.float32)
    v = rng.standard_normal((L, KVH, D)).astype(np.float32)
    mask = np.array([[0, -np.inf, -np.inf], [0, 0, -np.inf], [0, 0, 0]], np.float32)
    out = oracle.attn_prefill_varlen(q, k, v, [0, L], 0.125)
    for h in range(H):
        s = q[:, h] @ k[:, 0].T * 0.125 + mask
        p = np.exp(s - s.max(-1, keepdims=True)); p /= p.sum(-1, keepdims=True)
        np.testing.assert_allclose(out[:, h], p @ v[:, 0], rtol=1e-5, atol=1e-6)


def test_rmsnorm_kats():                         # layernorm.rs:195-221,277-311
    x = np.array([[1, 2, 3, 4], [5, 6, 7, 8]], np.float32)
    y = oracle.rmsnorm(x, np.ones(4, np.float32), 1e-6)
    rms = np.sqrt((y ** 2).mean(-1))
    assert np.all(np.abs(rms - 1.0) < 0.1)
    ref = x / np.sqrt((x.astype(np.float64) ** 2).mean(-1, keepdims=True) + 1e-6)
    np.testing.assert_allclose(y, ref, rtol=1e-6)
    for mag in (1e-10, 1e10):
        assert np.isfinite(oracle.rmsnorm(np.full((1, 4), mag, np.float32), np.ones(4, np.float32), 1e-8)).all()


def test_rope_kats():                            # rotary_embedding.rs:322-338,368-386,406-418,441-464
    cos, sin = oracle.rope_table(4, 3, 10000.0)
    assert cos.shape == (3, 2) and cos[0, 0] == 1.0 and sin[0, 0] == 0.0
    x = np.arange(2 * 1 * 4, dtype=np.float32).reshape(2, 1, 4)
    ident = oracle.rope_apply(x, [0, 0], cos, sin)            # position 0: cos=1, sin=0 => identity
    np.testing.assert_array_equal(ident, x)
    inv = 1.0 / 10000.0 ** (np.arange(0, 4, 2) / 4.0)         # on-demand vs table < 1e-5
    np.testing.assert_allclose(cos, np.cos(np.arange(3)[:, None] * inv[None]), atol=1e-5)
    np.testing.assert_allclose(sin, np.sin(np.arange(3)[:, None] * inv[None]), atol=1e-5)
    y = oracle.rope_apply(x, [1, 2], cos, sin)                # halves split at D/2
    c, s = cos[1], sin[1]
    np.testing.assert_allclose(y[0, 0], np.concatenate([x[0, 0, :2] * c - x[0, 0, 2:] * s,
                                                        x[0, 0, 2:] * c + x[0, 0, :2] * s]), rtol=1e-6)
    # test_rotary_embedding_with_scaling :406-418: new_with_scaling(4, 10, 10000.0, 2.0) has base 20000 and a [positions, D/2] table
    base, cs, sn = oracle.rope_table_with_scaling(4, 10, 10000.0, 2.0)
    assert base == 20000.0
    assert cs[[0, 1]].shape == (2, 2) and sn[[0, 1]].shape == (2, 2)
    inv2 = 1.0 / 20000.0 ** (np.arange(0, 4, 2) / 4.0)        # = the unscaled constructor at the scaled base (:131-133)
    np.testing.assert_allclose(cs, np.cos(np.arange(10)[:, None] * inv2[None]), atol=1e-5)
    np.testing.assert_allclose(sn, np.sin(np.arange(10)[:, None] * inv2[None]), atol=1e-5)
    c0, s0 = oracle.rope_table(4, 10, 20000.0)
    assert np.array_equal(cs, c0) and np.array_equal(sn, s0)


def test_silu_kats():                            # activation.rs:214-228,264-290,309-318
    x = np.array([[1, 2, 3, .5, 1.5, 2.5]], np.float32)
    y = oracle.silu_and_mul(x)
    silu = lambda t: t / (1 + np.exp(-t))
    np.testing.assert_allclose(y[0], silu(x[0, :3]) * x[0, 3:], rtol=1e-6)
    assert oracle.silu_and_mul(np.zeros((1, 2), np.float32))[0, 0] == 0.0
    g = np.array([-2, -1, 0, 1, 2], np.float32)
    m = oracle.silu_and_mul(np.concatenate([g, np.ones(5, np.float32)])[None])[0]
    assert np.all(np.diff(m[1:]) > 0)
    with pytest.raises(ValueError):
        oracle.silu_and_mul(np.zeros((1, 3), np.float32))


def test_activation_kats():                      # the rest of activation.rs: gelu / relu / GeluAndMul / Activation (:231-261, :293-334)
    g = np.array([[-2, -1, 0, 1, 2]], np.float32)
    y = oracle.activation("gelu", g)[0]
    assert abs(y[2]) < 1e-6 and np.all(np.diff(y[1:]) > 0)                      # GELU(0) = 0; increasing on [-1, 2] (:231-245 asserts it from -2: the tanh
    assert y[0] < 0 and y[0] > y[1]                                             # form dips: gelu(-2) = -0.0454 > gelu(-1) = -0.1588 — candle's own values)
    np.testing.assert_allclose(y, 0.5 * g[0] * (1 + np.tanh(np.sqrt(2 / np.pi) * (g[0] + 0.044715 * g[0] ** 3))), rtol=2e-6, atol=1e-7)
    assert oracle.activation("relu", g)[0].tolist() == [0.0, 0.0, 0.0, 1.0, 2.0]     # :248-261
    s = oracle.activation("silu", g)[0]
    assert s[2] == 0.0 and np.all(np.diff(s[1:]) > 0)                           # :214-228
    x = np.array([[1, 2, 3, .5, 1.5, 2.5]], np.float32)
    assert oracle.activation("gelu_and_mul", x).shape == (1, 3)                 # :293-307
    np.testing.assert_allclose(oracle.activation("gelu_and_mul", x)[0], oracle.activation("gelu", x[:, :3])[0] * x[0, 3:], rtol=1e-6)
    np.testing.assert_allclose(oracle.activation("silu_and_mul", x), oracle.silu_and_mul(x), rtol=1e-6)    # :264-290 (the Activation enum's arm, :321-334)
    for kind in ("silu_and_mul", "gelu_and_mul"):
        with pytest.raises(ValueError, match="must be even"):                   # :309-318
            oracle.activation(kind, np.zeros((1, 3), np.float32))


def test_linear_shape_rules():                   # linear.rs:476-559 (TP shape partitioning)
    assert 256 // 2 == 128                                    # column/row parallel split
    H, KVH, D = 8, 8, 64
    assert (H + 2 * KVH) * D == 1536 and (H * D, KVH * D, KVH * D) == (512, 512, 512)
    x = np.random.default_rng(0).standard_normal((3, 16)).astype(np.float32)
    W = np.random.default_rng(1).standard_normal((10, 16)).astype(np.float32)
    np.testing.assert_allclose(oracle.linear(x, W), x @ W.T, rtol=1e-5, atol=1e-5)


def test_top_p_one_is_no_filter():
    """Decision A-26.  Sampler::batch_sample (sampler.rs:233-240) gives rows without top_p the value 1.0 whenever another row of
    the batch sets top_p, and still runs apply_top_p on them.  In exact arithmetic p = 1.0 keeps every token; the f32 cumulative
    sum of :168-177 can reach 1.0 early and cut a tail.  The literal restatement (oracle.top_p / oracle.sample with p = 1.0)
    must then (i) only ever drop a tail of negligible total probability and (ii) sample the same token as no filter at all —
    which is how the product treats p >= 1.0."""
    rng = np.random.default_rng(99)
    dropped_mass, trials = 0.0, 0
    for V, scale in [(50, 3.0), (1000, 1.0), (1000, 6.0), (20000, 2.0), (151936, 0.3), (151936, 3.0)]:
        for rep in range(40 if V <= 20000 else 6):
            x = (rng.standard_normal(V) * scale).astype(np.float32)
            kept = np.isfinite(oracle.top_p(x, 1.0))
            p = np.exp(x.astype(np.float64) - x.max()); p /= p.sum()
            dropped_mass = max(dropped_mass, float(p[~kept].sum()))
            for temp in (1.0, 0.7):
                key = oracle.sample_key(5, rep, V % 97)
                assert oracle.sample(x, temp, 0, 1.0, key) == oracle.sample(x, temp, 0, None, key)
                trials += 1
    assert trials > 200 and dropped_mass < 1e-5, dropped_mass


def test_qk_norm_extension_is_off_by_default_and_changes_the_graph():
    """A-27: ModelConfig.qk_norm is False in every preset (the reference graph has no q/k norms); when on, the oracle applies
    the reference's RMSNorm over head_dim to q and k before RoPE and accepts the real checkpoints' q_norm / k_norm tensors."""
    from oracle import model_oracle as mo
    assert not mo.qwen3_0_6b().qk_norm and not mo.qwen3_8b().qk_norm and not mo.tiny().qk_norm
    rng = np.random.default_rng(3)
    ids = np.asarray([5, 9, 200, 31], np.int64)
    meta = dict(is_prefill=True, cu_seqlens_q=np.asarray([0, 4], np.int32), slot_mapping=np.arange(4, dtype=np.int32),
                block_tables=None, context_lens=None)
    out = {}
    for on in (False, True):
        m = mo.OracleModel(mo.tiny(qk_norm=on), num_blocks=2, block_size=16, fp16=True, max_pos=64)
        sd = {f"model.layers.{l}.self_attn.{n}_norm.weight": (1 + 0.3 * rng.standard_normal(m.D)).astype(np.float32)
              for l in range(2) for n in "qk"}
        skipped = m.load_state_dict(sd)
        assert (skipped == []) if on else (sorted(skipped) == sorted(sd))
        h = m.embed_tokens(ids)
        out[on] = m.attn_part(0, h, np.arange(4, dtype=np.int64), meta)
    assert np.abs(out[True] - out[False]).max() > 1e-3


def test_round_bf16_matches_torch_and_the_oracle_bf16_mode_rounds_everywhere():
    """The bf16-faithful oracle mode (Config.dtype "bfloat16", config.rs:51,113-116): oracle.round_bf16 is round-to-nearest-even to 8
    significant bits — pinned against torch's bfloat16 cast on random values, every tie case and the range ends — and an OracleModel
    built with bf16=True holds only bf16-representable weights and activations, distinct from the fp16 mode's."""
    import torch
    from oracle import model_oracle as mo
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.standard_normal(20000).astype(np.float32) * np.float32(10.0) ** rng.integers(-30, 30, 20000).astype(np.float32),
                        np.asarray([0.0, -0.0, 1.0, 1.0 + 2.0 ** -8, 1.0 + 3 * 2.0 ** -8, -(1.0 + 2.0 ** -8), 65504.0, 3.0e38, 1e-40, 2.0 ** -133,
                                    3.3895314e38], np.float32)])
    want = torch.from_numpy(x).to(torch.bfloat16).to(torch.float32).numpy()
    got = oracle.round_bf16(x)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(oracle.to_bf16_bits(x), (want.view(np.uint32) >> 16).astype(np.uint16))
    assert oracle.round_bf16(np.float32([1.0 + 2.0 ** -8, 1.0 + 3 * 2.0 ** -8])).tolist() == [1.0, 1.0 + 2.0 ** -6]     # ties to even, both ways
    ids = np.asarray([5, 9, 200, 31], np.int64)
    meta = dict(is_prefill=True, cu_seqlens_q=np.asarray([0, 4], np.int32), slot_mapping=np.arange(4, dtype=np.int32), block_tables=None, context_lens=None)
    outs = {}
    for mode in ("fp16", "bf16"):
        m = mo.OracleModel(mo.tiny(), num_blocks=2, block_size=16, fp16=(mode == "fp16"), bf16=(mode == "bf16"), max_pos=64)
        o = m.attn_part(0, m.embed_tokens(ids), np.arange(4, dtype=np.int64), meta)
        outs[mode] = o
        if mode == "bf16":
            for w in (m.embed, m.layers[0]["qkv"], m.layers[0]["down"], m.norm, o, m.k_cache[0][:1]):
                w = np.asarray(w, np.float32)
                assert np.array_equal(w, oracle.round_bf16(w))
    d = np.abs(outs["fp16"] - outs["bf16"]).max()
    assert 1e-5 < d < 5e-2, d


def test_use_bias_is_off_by_default_and_follows_the_reference_sharding():
    """A-30: Qwen3Config::use_bias (qwen3.rs:54-55, default false :82).  With it on, the oracle adds a bias after each of the four
    projections as a second rounded op (candle_nn::Linear: matmul, then broadcast_add); on tensor-parallel ranks the column-parallel
    biases are slices of the global vector and the row-parallel ones exist on rank 0 only (linear.rs:206), so that the all-reduced
    sum carries the bias once: the two-rank partial sums add up to the single-rank projection."""
    from oracle import model_oracle as mo
    assert not mo.qwen3_0_6b().use_bias and not mo.qwen3_8b().use_bias and not mo.tiny().use_bias
    ids = np.asarray([5, 9, 200, 31], np.int64)
    meta = lambda: dict(is_prefill=True, cu_seqlens_q=np.asarray([0, 4], np.int32), slot_mapping=np.arange(4, dtype=np.int32),
                        block_tables=None, context_lens=None)
    pos = np.arange(4, dtype=np.int64)
    one = mo.OracleModel(mo.tiny(use_bias=True), num_blocks=2, block_size=16, fp16=False, max_pos=64)
    off = mo.OracleModel(mo.tiny(), num_blocks=2, block_size=16, fp16=False, max_pos=64)
    h = one.embed_tokens(ids)
    full = one.attn_part(0, h, pos, meta())
    assert np.abs(full - off.attn_part(0, h, pos, meta())).max() > 1e-3
    ranks = [mo.OracleModel(mo.tiny(use_bias=True), num_blocks=2, block_size=16, fp16=False, max_pos=64, tp_size=2, tp_rank=r) for r in range(2)]
    assert ranks[1].layers[0]["o_b"] is None and ranks[1].layers[0]["down_b"] is None and ranks[0].layers[0]["o_b"] is not None
    np.testing.assert_array_equal(np.concatenate([ranks[0].layers[0]["gate_up_b"][:ranks[0].I], ranks[1].layers[0]["gate_up_b"][:ranks[1].I]]),
                                  one.layers[0]["gate_up_b"][:one.I])
    parts = sum(r.attn_part(0, h, pos, meta()) for r in ranks)
    np.testing.assert_allclose(parts, full, rtol=2e-5, atol=2e-6)
    # checkpoint names
    sd = {"model.layers.0.mlp.down_proj.bias": np.arange(one.Hd, dtype=np.float32) * 1e-3,
          "model.layers.1.self_attn.q_proj.bias": np.ones(one.H * one.D, np.float32)}
    assert one.load_state_dict(sd) == [] and sorted(off.load_state_dict(sd)) == sorted(sd)
    assert np.array_equal(one.layers[0]["down_b"], sd["model.layers.0.mlp.down_proj.bias"])
    assert np.array_equal(one.layers[1]["qkv_b"][:one.H * one.D], np.ones(one.H * one.D, np.float32))
