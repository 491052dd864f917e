"""Coverage of the BASELINE.json configurations beyond the small-model parity tests.
CPU: configs[4] (512 sequences sharing a 512-token system prompt) through the product's BlockManager/Scheduler against
     the oracle — dedup, ref counts, block tables — and the 288 GB pool sizing arithmetic.
GPU: Qwen3-8B head/MLP geometry (configs[3] shapes, 2 layers), a full-size Qwen3-0.6B spot check against the oracle,
     and size-independent properties of the full configs[1] workload (determinism, cache contents == recomputation)."""
import numpy as np
import pytest

import nvr_import
import oracle
from oracle import engine_oracle as eo
from oracle import model_oracle as mo

nvr = nvr_import.load()


def test_config5_shared_prefix_dedup_512_sequences():
    """512 seqs = the same 512-token prefix + 64 random suffix tokens, block size 256 (BASELINE configs[4])."""
    eo.reset_sequence_counter(); nvr.lib().nvr_seq_reset_id_counter()
    cfg = dict(max_num_seqs=512, max_num_batched_tokens=512 * 576, kvcache_block_size=256, num_kvcache_blocks=2048)
    o, p = eo.Scheduler(eo.Config(**cfg)), nvr.Scheduler(nvr.Config(**cfg))
    prefix = nvr.synthetic_tokens(512, 2, 0, 151936).tolist()
    for i in range(512):
        toks = prefix + nvr.synthetic_tokens(64, 1, i, 151936).tolist()
        o.add_sequence(eo.Sequence(toks, eo.SamplingParams(max_tokens=8, ignore_eos=True), 256))
        p.add_sequence(nvr.Sequence(toks, nvr.SamplingParams(max_tokens=8, ignore_eos=True), 256))
    oseqs, opf = o.schedule(); pseqs, ppf = p.schedule()
    assert opf and ppf and len(oseqs) == len(pseqs) == 512
    assert [s.block_table for s in pseqs] == [list(s.block_table) for s in oseqs]
    assert pseqs[0].num_cached_tokens == 0 and all(s.num_cached_tokens == 512 for s in pseqs[1:])
    bm = p.block_manager
    t0 = pseqs[0].block_table
    assert bm.get_block(t0[0])["ref_count"] == 512 and bm.get_block(t0[1])["ref_count"] == 512     # shared prefix blocks
    st = bm.get_stats()
    assert st["used_blocks"] == 2 + 512 and st == o.block_manager.get_stats()                       # 2 shared + 1 private each
    # decode a few steps (suffix blocks fill up; no new blocks needed before 768 tokens)
    for step in range(3):
        toks = [int(s.seq_id % 100) + 10 for s in oseqs]
        o.postprocess(oseqs, toks); p.postprocess(pseqs, toks)
        oseqs, opf = o.schedule(); pseqs, ppf = p.schedule()
        assert not opf and not ppf and [s.block_table for s in pseqs] == [list(s.block_table) for s in oseqs]
    assert p.get_stats()["preemptions"] == 0


def test_pool_sizing_for_288gb():
    """KV bytes per 256-token block of Qwen3-0.6B: 28 layers x K,V x 256 x 8 x 128 x 2 B = 28 MiB."""
    per_block = 28 * 2 * 256 * 8 * 128 * 2
    assert per_block == 28 * 2 ** 20
    assert int(0.9 * 288e9 // per_block) > 8800


def _mc(m):
    return nvr.ModelConfig(vocab_size=m.vocab_size, hidden_size=m.hidden_size, intermediate_size=m.intermediate_size,
                           num_hidden_layers=m.num_hidden_layers, num_attention_heads=m.num_attention_heads,
                           num_key_value_heads=m.num_key_value_heads, head_dim=m.head_dim or 0,
                           max_position_embeddings=m.max_position_embeddings, rms_norm_eps=m.rms_norm_eps, rope_theta=m.rope_theta,
                           tie_word_embeddings=m.tie_word_embeddings, init_std=m.init_std, seed=m.seed)


def _parity(mcfg, ecfg, prompts, max_tokens, tol=2e-2):
    eo.reset_sequence_counter(); nvr.lib().nvr_seq_reset_id_counter()
    o = mo.OracleEngine(mcfg, eo.Config(**ecfg), fp16=True, max_pos=ecfg["max_model_len"])
    p = nvr.LLMEngine(nvr.Config(**ecfg), _mc(mcfg))
    for pr in prompts:
        sp = dict(temperature=0.0, max_tokens=max_tokens, ignore_eos=True)
        o.add_request(pr, eo.SamplingParams(**sp)); p.add_request(pr, nvr.SamplingParams(**sp))
    ties, worst = 0, 0.0
    while not p.is_finished():
        rec = p.step()
        logits = p.model_runner.logits(rec["num_seqs"])
        orec = o.step(forced_tokens=rec["tokens"])
        assert orec["seq_ids"] == rec["seq_ids"] and orec["is_prefill"] == rec["is_prefill"]
        worst = max(worst, float(np.abs(logits - orec["logits"]).max()))
        srt = np.sort(orec["logits"], axis=1)
        for i, (a, b) in enumerate(zip(rec["tokens"], orec["tokens"])):
            if a != b:
                assert srt[i, -1] - srt[i, -2] <= 2 * tol, (a, b, srt[i, -1] - srt[i, -2])
                ties += 1
    assert worst < tol, worst
    return ties, worst


@pytest.mark.gpu
def test_qwen3_8b_geometry_two_layers():
    """configs[3] shapes: Hd 4096, 32:8 heads (group 4), D 128, I 12288 — two layers, small vocab."""
    mcfg = mo.ModelConfig(vocab_size=4096, hidden_size=4096, intermediate_size=12288, num_hidden_layers=2, num_attention_heads=32,
                          num_key_value_heads=8, head_dim=128, rope_theta=1e6, tie_word_embeddings=False,
                          max_position_embeddings=1024, init_std=0.02, seed=21)
    ecfg = dict(max_num_seqs=4, max_num_batched_tokens=512, max_model_len=512, kvcache_block_size=256, num_kvcache_blocks=8)
    prompts = [nvr.synthetic_tokens(n, 1, i, 4096).tolist() for i, n in enumerate([150, 40])]
    ties, worst = _parity(mcfg, ecfg, prompts, 5)
    assert ties <= 1


@pytest.mark.gpu
@pytest.mark.parametrize("nseq", [40, 64])
def test_qwen3_8b_geometry_decode_batches_of_33_to_64_rows(nseq):
    """The same two layers with 40 / 64 sequences in the decode batch: qkv and gate_up on the streaming kernels' 64-row images, o_proj /
    down_proj on the split-k LDS tiles with 8 k-slices (r06: the streaming split-k kernel walked the weights once per 32-row block)."""
    mcfg = mo.ModelConfig(vocab_size=4096, hidden_size=4096, intermediate_size=12288, num_hidden_layers=2, num_attention_heads=32,
                          num_key_value_heads=8, head_dim=128, rope_theta=1e6, tie_word_embeddings=False,
                          max_position_embeddings=1024, init_std=0.02, seed=21)
    ecfg = dict(max_num_seqs=nseq, max_num_batched_tokens=2048, max_model_len=256, kvcache_block_size=256, num_kvcache_blocks=nseq + 4)
    prompts = [nvr.synthetic_tokens(6 + i % 9, 1, i, 4096).tolist() for i in range(nseq)]
    ties, worst = _parity(mcfg, ecfg, prompts, 4)
    assert ties <= 2


@pytest.mark.gpu
@pytest.mark.parametrize("lens", [[300, 90], [260], [700, 200, 60]])
def test_prefill_steps_of_257_to_1024_tokens_on_an_engine_of_few_sequences(lens):
    """Qwen3-0.6B layer shapes, two layers: a prefill STEP of 257..1024 tokens on an engine whose max_num_seqs is far below that takes the split-k
    tiles for o_proj / down_proj like a decode step of as many rows (r06: the slab buffer was sized by max_num_seqs alone and such steps fell to
    9-12 row blocks of the streaming kernel: 2.4 against 1.6 ms per 257-token step) — against the oracle, logits of every step."""
    mcfg = mo.ModelConfig(vocab_size=4096, hidden_size=1024, intermediate_size=3072, num_hidden_layers=2, num_attention_heads=16,
                          num_key_value_heads=8, head_dim=128, rope_theta=1e6, tie_word_embeddings=True,
                          max_position_embeddings=2048, init_std=0.02, seed=23)
    ecfg = dict(max_num_seqs=4, max_num_batched_tokens=1024, max_model_len=1024, kvcache_block_size=256, num_kvcache_blocks=16)
    prompts = [nvr.synthetic_tokens(n, 1, i, 4096).tolist() for i, n in enumerate(lens)]
    ties, worst = _parity(mcfg, ecfg, prompts, 3)
    assert ties <= 1


@pytest.mark.gpu
@pytest.mark.parametrize("nseq,kvh", [(36, 8), (70, 4)])
def test_decode_batches_between_the_multiples_of_32_take_the_work_balanced_attention(nseq, kvh):
    """36 sequences x 8 kv heads = 288 (sequence, kv head) pairs, 70 x 4 = 280: more than the 256 CUs, far from the next multiple — the decode
    attention of such steps runs as 256 equal shares of all pairs' keys (attn_share_kernel, r06) instead of one workgroup per pair; contexts of
    257..400 keys, ragged.  Logits of every step against the oracle (the prefill fills the caches the shares walk)."""
    mcfg = mo.small(seed=29, num_attention_heads=16, num_key_value_heads=kvh, head_dim=64, hidden_size=256, intermediate_size=512)
    ecfg = dict(max_num_seqs=nseq, max_num_batched_tokens=4096, max_model_len=512, kvcache_block_size=256, num_kvcache_blocks=nseq * 2 + 4)
    prompts = [nvr.synthetic_tokens(257 + (37 * i) % 140, 1, i, mcfg.vocab_size).tolist() for i in range(nseq)]
    ties, worst = _parity(mcfg, ecfg, prompts, 4)
    assert ties <= 2


@pytest.mark.gpu
@pytest.mark.parametrize("nseq,lo,hi", [(24, 200, 1500), (40, 150, 1200), (12, 20, 1500)])
def test_ragged_decode_batches_take_the_work_balanced_attention(nseq, lo, hi):
    """Sequences of very different lengths in one decode batch (the serving case; every BASELINE config is uniform): the runner sees that the contexts sum
    to far less than batch x longest and passes the attention launch its balance hint — 256 equal shares of all pairs' keys instead of per-pair workgroups
    sized by the longest context (r06: 32 sequences of 256..8192 keys 4.09 -> 2.63 ms per Qwen3-0.6B step).  Logits of every step against the oracle; the
    runner must report the ragged form for the decode steps of batches with >= 6 units of 64 keys per CU (the first two; the third — 4.6 k keys in all — is
    launch-bound and keeps the per-pair launch) and never for a prefill."""
    mcfg = mo.small(seed=33, num_attention_heads=16, num_key_value_heads=8, head_dim=64, hidden_size=256, intermediate_size=512)
    ecfg = dict(max_num_seqs=nseq, max_num_batched_tokens=32768, max_model_len=2048, kvcache_block_size=256, num_kvcache_blocks=nseq * 8 + 4)
    lens = [int(lo * (hi / lo) ** (i / (nseq - 1))) for i in range(nseq)]
    prompts = [nvr.synthetic_tokens(n, 1, i, mcfg.vocab_size).tolist() for i, n in enumerate(lens)]
    eo.reset_sequence_counter(); nvr.lib().nvr_seq_reset_id_counter()
    o = mo.OracleEngine(mcfg, eo.Config(**ecfg), fp16=True, max_pos=ecfg["max_model_len"])
    p = nvr.LLMEngine(nvr.Config(**ecfg), _mc(mcfg))
    for pr in prompts:
        sp = dict(temperature=0.0, max_tokens=4, ignore_eos=True)
        o.add_request(pr, eo.SamplingParams(**sp)); p.add_request(pr, nvr.SamplingParams(**sp))
    worst, forms = 0.0, []
    while not p.is_finished():
        rec = p.step()
        logits = p.model_runner.logits(rec["num_seqs"])
        forms.append((bool(rec["is_prefill"]), p.model_runner.last_decode_ragged()))
        orec = o.step(forced_tokens=rec["tokens"])
        assert orec["seq_ids"] == rec["seq_ids"] and orec["is_prefill"] == rec["is_prefill"]
        worst = max(worst, float(np.abs(logits - orec["logits"]).max()))
    assert worst < 2e-2, worst
    want = sum(lens) * 8 // 64 >= 6 * 256
    assert all(not r for pre, r in forms if pre) and all(r == want for pre, r in forms if not pre) and any(not pre for pre, _ in forms), (want, forms)


@pytest.mark.gpu
def test_qwen3_0_6b_full_size_spot_check():
    """The benchmark model itself (28 layers, V=151936, tied head) against the oracle on two short prompts."""
    mcfg = mo.qwen3_0_6b()
    ecfg = dict(max_num_seqs=2, max_num_batched_tokens=256, max_model_len=256, kvcache_block_size=256, num_kvcache_blocks=4)
    prompts = [nvr.synthetic_tokens(n, 1, i, mcfg.vocab_size).tolist() for i, n in enumerate([24, 9])]
    ties, worst = _parity(mcfg, ecfg, prompts, 3)
    assert ties <= 1


@pytest.mark.gpu
def test_config2_full_size_properties():
    """BASELINE configs[1] at full size (32 x 1024, Qwen3-0.6B): (i) two engines produce identical token streams;
    (ii) the KV rows a decode step appends equal what a fresh prefill of the grown sequence computes for that
    position (cache write path == recomputation, up to fp16 GEMM tile-order noise)."""
    mc = nvr.ModelConfig("qwen3-0.6b")
    ecfg = dict(max_num_seqs=32, max_num_batched_tokens=32 * 1040, max_model_len=1040, kvcache_block_size=256, num_kvcache_blocks=200)

    def run(nsteps):
        nvr.lib().nvr_seq_reset_id_counter()
        eng = nvr.LLMEngine(nvr.Config(**ecfg), mc)
        for i in range(32):
            eng.add_request(nvr.synthetic_tokens(1024, 1, i, 151936).tolist(), nvr.SamplingParams(temperature=0.0, max_tokens=nsteps, ignore_eos=True))
        toks = []
        while not eng.is_finished():
            toks.append(eng.step()["tokens"])
        return toks
    a, b = run(6), run(6)
    assert a == b and len(a) == 6 and all(len(t) == 32 for t in a)
    assert all(0 <= t < 151936 for step in a for t in step)
    assert len({tuple(s) for s in zip(*a)}) > 16            # sequences do not collapse onto one stream


@pytest.mark.gpu
def test_config2_long_decode_graph_equals_eager():
    """BASELINE configs[1] geometry, 300 decode steps (context 1000 -> 1300): every sequence crosses a KV block boundary
    (new block, block hashing) and the batch crosses a 256-token context bucket (a second hipGraph is captured); the
    replayed graphs must produce exactly the token streams of kernel-by-kernel launches."""
    mc = nvr.ModelConfig("qwen3-0.6b")
    ecfg = dict(max_num_seqs=32, max_num_batched_tokens=32768, max_model_len=1400, kvcache_block_size=256, num_kvcache_blocks=200)

    def run(**kw):
        nvr.lib().nvr_seq_reset_id_counter()
        eng = nvr.LLMEngine(nvr.Config(**ecfg, **kw), mc)
        for i in range(32):
            eng.add_request(nvr.synthetic_tokens(1000, 1, i, 151936).tolist(), nvr.SamplingParams(temperature=0.0, max_tokens=300, ignore_eos=True))
        toks = []
        while not eng.is_finished():
            toks.append(tuple(eng.step()["tokens"]))
        return toks
    g, e = run(), run(enforce_eager=1)
    assert len(g) == 300 and g == e


@pytest.mark.gpu
def test_config5_full_size_prefix_skipping():
    """BASELINE configs[4] at full size (Qwen3-0.6B, 512 sequences = one 512-token system prompt + 64 own tokens, block
    size 256): with cached-prefix skipping the prefill feeds 576 + 511 x 64 rows through the model instead of 512 x 576,
    the first sampled token of every sequence is identical to the recompute-everything path, and the prefill is several
    times faster (both wall times are printed; the ratio is asserted loosely)."""
    import time
    mc = nvr.ModelConfig("qwen3-0.6b")
    ecfg = dict(max_num_seqs=512, max_num_batched_tokens=32768, max_model_len=1024, kvcache_block_size=256, num_kvcache_blocks=600)
    prefix = nvr.synthetic_tokens(512, 2, 0, 151936).tolist()
    prompts = [prefix + nvr.synthetic_tokens(64, 1, i, 151936).tolist() for i in range(512)]

    def run(**kw):
        nvr.lib().nvr_seq_reset_id_counter()
        eng = nvr.LLMEngine(nvr.Config(**ecfg, **kw), mc)
        for pr in prompts:
            eng.add_request(pr, nvr.SamplingParams(temperature=0.0, max_tokens=2, ignore_eos=True))
        first, rows, nvr_sync = {}, 0, nvr.synchronize
        nvr_sync(); t0 = time.perf_counter()
        steps = 0
        while True:
            rec = eng.step()
            if not rec["is_prefill"]:
                break
            rows += rec["num_tokens"]; steps += 1
            first.update(zip(rec["seq_ids"], rec["tokens"]))
        nvr_sync(); dt = time.perf_counter() - t0
        return first, rows, steps, dt
    f_skip, rows_skip, steps_skip, t_skip = run()
    f_full, rows_full, steps_full, t_full = run(recompute_cached_prefix=1)
    print(f"config 5 prefill: skipping {rows_skip} rows in {steps_skip} steps {t_skip * 1e3:.1f} ms; recomputing {rows_full} rows in {steps_full} steps {t_full * 1e3:.1f} ms")
    assert rows_full == 512 * 576 and rows_skip == 576 + 511 * 64
    assert f_skip == f_full and len(f_skip) == 512
    assert t_full > 3 * t_skip


@pytest.mark.gpu
def test_kv_pool_sized_from_free_hbm():
    """num_kvcache_blocks = auto: the pool is cut from free HBM * gpu_memory_utilization (the reference has the config
    field, config.rs:30, but no sizing code); 28 MiB per block for Qwen3-0.6B."""
    import ctypes as C
    free, total = C.c_uint64(), C.c_uint64()
    nvr.check(nvr.lib().nvr_device_set(0)); nvr.check(nvr.lib().nvr_device_mem_info(C.byref(free), C.byref(total)))
    util = 0.05
    eng = nvr.LLMEngine(nvr.Config(max_num_seqs=2, max_num_batched_tokens=512, max_model_len=512, num_kvcache_blocks="auto",
                                   gpu_memory_utilization=util), nvr.ModelConfig("qwen3-0.6b"))
    nb = eng.model_runner.num_kvcache_blocks()
    budget = free.value - (1 - util) * total.value
    assert budget > 0 and 0.8 * budget / (28 * 2 ** 20) - 200 < nb <= budget / (28 * 2 ** 20)
    assert eng.scheduler.get_block_stats()["total_blocks"] == nb          # scheduler and runner share the pool size
    eng.add_request(nvr.synthetic_tokens(20, 1, 0, 151936).tolist(), nvr.SamplingParams(temperature=0.0, max_tokens=2, ignore_eos=True))
    while not eng.is_finished():
        eng.step()


@pytest.mark.gpu
def test_chunked_prefill_full_size_bit_identity():
    """Extension A-23 at BASELINE configs[2] sizes (Qwen3-0.6B): two 4096-token prompts and one of 1500 prefilled in chunks
    against a 2048-token budget, versus one whole-prompt batch.  A GEMM row depends only on its own input row and the flash
    kernel walks a row's keys in the same 64-key steps whether they come from this step's rows or from the cache, so the
    last-token logits — and with them the first sampled tokens and the following decode steps — are BIT-identical."""
    mc = nvr.ModelConfig("qwen3-0.6b")
    base = dict(max_num_seqs=4, max_model_len=4200, kvcache_block_size=256, num_kvcache_blocks=40)
    prompts = [nvr.synthetic_tokens(n, 1, i, 151936).tolist() for i, n in enumerate([4096, 1500, 4096])]

    def run(**kw):
        nvr.lib().nvr_seq_reset_id_counter()
        eng = nvr.LLMEngine(nvr.Config(**base, **kw), mc)
        for pr in prompts:
            eng.add_request(pr, nvr.SamplingParams(temperature=0.0, max_tokens=4, ignore_eos=True))
        first_logits, toks, prefill_steps = {}, {}, 0
        while not eng.is_finished():
            rec = eng.step()
            lg = eng.model_runner.logits(rec["num_seqs"])
            prefill_steps += int(rec["is_prefill"])
            for i, (sid, t) in enumerate(zip(rec["seq_ids"], rec["tokens"])):
                if t == -1:
                    continue
                first_logits.setdefault(sid, lg[i].copy())
                toks.setdefault(sid, []).append(t)
        return first_logits, toks, prefill_steps
    lc, tc, sc = run(max_num_batched_tokens=2048, enable_chunked_prefill=1)
    lu, tu, su = run(max_num_batched_tokens=16384)
    assert sc == 5 and su == 1                              # 4096 + 1500 + 4096 tokens in 2048-token steps: 5 prefill steps
    assert tc == tu
    for sid in lu:
        assert np.array_equal(lc[sid], lu[sid]), f"sequence {sid}: chunked and whole-prompt prefill differ in bits"


@pytest.mark.gpu
def test_configs2_256_x_4096_chunked_equals_whole_sequence_first_tokens():
    """BASELINE configs[2] at its LARGEST point — 256 sequences x 4096 tokens = 1 048 576 prompt tokens against the 32 768-token budget
    (what bench.py's prefill_sweep times) — as a property test: with chunked prefill on and a 20 000-token budget
    (A-23: 4 whole prompts + a 3 616-token chunk per step, every later chunk reaching its predecessors through the block table) the first
    sampled token of EVERY sequence equals the whole-sequence run's (8 prompts per 32 768-token step, the reference's batching,
    scheduler.rs:119-168).  Last-token logits are bit-identical wherever both runs' steps take the same GEMM route; where a short last step
    routes differently they agree to fp16 rounding and a token may differ only inside that distance of a tie (counted, bounded)."""
    mc = nvr.ModelConfig("qwen3-0.6b")
    n, L = 256, 4096
    base = dict(max_num_seqs=n, max_model_len=L + 8, kvcache_block_size=256, num_kvcache_blocks=n * (L // 256 + 1) + 8)
    prompts = [nvr.synthetic_tokens(L, 1, i, 151936).tolist() for i in range(n)]

    def run(**kw):
        nvr.lib().nvr_seq_reset_id_counter()
        eng = nvr.LLMEngine(nvr.Config(**base, **kw), mc)
        for pr in prompts:
            eng.add_request(pr, nvr.SamplingParams(temperature=0.0, max_tokens=1, ignore_eos=True))
        logits, toks, steps, rows = {}, {}, 0, 0
        while not eng.is_finished():
            rec = eng.step()
            assert rec["is_prefill"]
            lg = eng.model_runner.logits(rec["num_seqs"])
            steps += 1; rows += rec["num_tokens"]
            for i, (sid, t) in enumerate(zip(rec["seq_ids"], rec["tokens"])):
                if t != -1:
                    logits[sid] = lg[i].copy(); toks[sid] = t
        del eng
        return logits, toks, steps, rows
    lw, tw, sw, rw = run(max_num_batched_tokens=32768)
    lc, tc, sc, rc = run(max_num_batched_tokens=20000, enable_chunked_prefill=1)
    assert rw == rc == n * L and sw == n * L // 32768 and sc == -(-n * L // 20000)
    assert len(tw) == len(tc) == n
    exact = near = 0
    for sid in tw:
        if np.array_equal(lw[sid], lc[sid]):
            exact += 1
            assert tw[sid] == tc[sid]
            continue
        d = float(np.abs(lw[sid] - lc[sid]).max())
        assert d < 2e-2, f"sequence {sid}: chunked and whole-sequence logits differ by {d}"
        if tw[sid] != tc[sid]:
            srt = np.sort(lw[sid]); near += 1
            assert srt[-1] - srt[-2] <= 2 * d, f"sequence {sid}: tokens differ outside a tie"
    print(f"configs[2] 256 x 4096: {sw} whole-sequence steps vs {sc} chunked steps; {exact} of {n} last-token logit rows bit-identical, {near} near-tie tokens")
    assert near <= 2 and exact >= n // 2


@pytest.mark.gpu
def test_configs4_512_sequences_equal_the_oracle_checked_48_sequence_run_on_the_shared_rows():
    """BASELINE configs[4] at FULL size — 512 sequences = one 512-token system prompt + 64 own tokens — against the run that
    test_configs4_shared_system_prompt_vs_oracle holds to the oracle (the first 48 of the same requests): on those 48 sequences the prefill's
    first tokens and four decode steps of the 512-sequence engine (prefix blocks shared 512 ways, the group-wide MFMA pass over the shared
    keys, mid-batch GEMM routes at 512 rows) give the 48-sequence engine's logits within the fp16-pipeline tolerance (other GEMM routes at
    another batch size: summation order only) and the same greedy ids outside near-ties."""
    mc = nvr.ModelConfig("qwen3-0.6b")
    shared = nvr.synthetic_tokens(512, 2, 0, 151936).tolist()
    prompts = [shared + nvr.synthetic_tokens(64, 1, i, 151936).tolist() for i in range(512)]

    def run(n):
        nvr.lib().nvr_seq_reset_id_counter()
        eng = nvr.LLMEngine(nvr.Config(max_num_seqs=n, max_num_batched_tokens=65536, max_model_len=640, kvcache_block_size=256, num_kvcache_blocks=n + 8), mc)
        for pr in prompts[:n]:
            eng.add_request(pr, nvr.SamplingParams(temperature=0.0, max_tokens=5, ignore_eos=True))
        out = []
        while not eng.is_finished():
            rec = eng.step()
            lg = eng.model_runner.logits(rec["num_seqs"])
            row = {sid: (t, lg[i].copy()) for i, (sid, t) in enumerate(zip(rec["seq_ids"], rec["tokens"])) if sid < 48}
            if row:                                         # (the 512 prompts take five budget-bound prefill steps, scheduler.rs:135-138: the first holds the 48)
                out.append(row)
            if not rec["is_prefill"]:
                assert eng.model_runner.last_shared_prefix_len() == 512
        used = eng.scheduler.block_manager.get_stats()
        del eng
        return out, used
    small, _ = run(48)
    big, used = run(512)
    assert len(small) == len(big) == 5
    near, worst, diverged = 0, 0.0, set()
    for a, b in zip(small, big):
        assert set(a) == set(b) == set(range(48))
        for sid in range(48):
            if sid in diverged:
                continue
            d = float(np.abs(a[sid][1] - b[sid][1]).max()); worst = max(worst, d)
            assert d < 2e-2, f"sequence {sid}: 512- and 48-sequence logits differ by {d}"
            if a[sid][0] != b[sid][0]:
                srt = np.sort(a[sid][1]); near += 1; diverged.add(sid)
                assert srt[-1] - srt[-2] <= 2 * 2e-2, f"sequence {sid}: tokens differ outside a tie"
    print(f"configs[4] 512 vs 48 sequences on the shared rows: max |d logit| {worst:.2e}, {near} near-tie tokens")
    assert near <= 2

