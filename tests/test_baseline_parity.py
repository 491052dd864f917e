"""Oracle parity ON the BASELINE.json workloads themselves (-m gpu).

Each test runs the product (C++ engine + HIP kernels, through the C ABI) and the CPU oracle on the SAME full-size
Qwen3-0.6B synthetic model and the same seeded prompts (token ids uniform in [0, V), seed 1, one stream per
sequence) and compares, per step: batch composition and block tables (bit-exact), f32 logits of every row
(|d| < LOGIT_TOL), greedy token ids (equal wherever the oracle's top-1/top-2 margin exceeds 2*LOGIT_TOL; every smaller
margin is a numerical tie between two fp16 pipelines, COUNTED, bounded by the assertion and written to the report).
The oracle is teacher-forced with the product's tokens so a tie cannot cascade.

  configs[0]  bs 1, 128-token prompt, 64 greedy steps       vs the fp16-faithful oracle AND vs the f32 "CPU path" oracle
  configs[1]  bs 32 x 1024-token prompts, prefill + 4 decode steps (hipGraph decode, 8-wave attention, V = 151 936 head)
  configs[2]  one 32 768-token prefill batch mixing 4 x 4096 ... 16 x 128 (42 sequences, the reference's token budget,
              config.rs:58) + one decode step; and 256 sequences x 200 tokens over two budget batches with chunked prefill on
  configs[3]  Qwen3-8B (36 layers, V = 151 936) on one GPU: product vs oracle at full depth on a reduced batch (2 x 256 + 2 decode
              steps), and the full 32 x 2048 workload as in-process tensor-parallel ranks (tp 8) against the single-rank product
A JSON summary of what was measured lands in gpurun_out/parity_r06.json (copied to profiles/ by hand)."""
import json
import os
import time

import numpy as np
import pytest

import nvr_import
import oracle
from oracle import engine_oracle as eo
from oracle import model_oracle as mo

nvr = nvr_import.load()
pytestmark = pytest.mark.gpu

LOGIT_TOL = 2e-2            # product fp16 pipeline vs fp16-faithful oracle (f32 logits, fp16 activations)
F32_TOL = 2e-4            # the f32 path vs the f32 oracle: summation order only (measured 7e-6 on Qwen3-0.6B)
CPU_PATH_TOL = 8e-2         # product fp16 pipeline vs the reference's f32 CPU path (weights and activations unrounded)
BF16_TOL = 1.6e-1          # Config.dtype = "bfloat16" against the bf16-faithful oracle: 8 x LOGIT_TOL (8 mantissa bits against 11); measured 4.5e-2 on Qwen3-0.6B
LOGIT_TOL_8B = 6e-2         # Qwen3-8B (36 layers, K = 4096 / 12 288): summation-order noise grows ~ sqrt(depth); the fp16-faithful oracle itself is
                            # 6.1e-2 from exact f32 arithmetic on the same weights, the product 4.3e-2 from the oracle (profiles/r03_parity_8b_stats.txt)
V = 151936


def _report(name, rec):
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        path = os.path.join(out, "parity_r06.json")
        cur = json.load(open(path)) if os.path.exists(path) else {}
        cur[name] = rec
        json.dump(cur, open(path, "w"), indent=1, sort_keys=True)
    except OSError:
        pass
    print(f"[parity] {name}: {json.dumps(rec)}")


def _pair(ecfg, prompts, max_tokens, fp16=True, tol=LOGIT_TOL, model="qwen3-0.6b", **kw):
    eo.reset_sequence_counter(); nvr.lib().nvr_seq_reset_id_counter()
    mcfg = mo.qwen3_0_6b() if model == "qwen3-0.6b" else mo.qwen3_8b()
    t0 = time.time()
    bf16 = kw.get("dtype") == "bfloat16"                 # the product's bf16 build against the oracle's bf16-faithful mode
    if kw.get("dtype") == "float32": fp16 = False        # the product's f32 path against the oracle's f32 arithmetic
    o = mo.OracleEngine(mcfg, eo.Config(**ecfg), fp16=fp16 and not bf16, bf16=bf16, max_pos=ecfg["max_model_len"], compact=(model != "qwen3-0.6b" and fp16))
    p = nvr.LLMEngine(nvr.Config(**ecfg, **kw), nvr.ModelConfig(model))
    t_build = time.time() - t0
    for pr in prompts:
        sp = dict(temperature=0.0, max_tokens=max_tokens, ignore_eos=True)
        o.add_request(pr, eo.SamplingParams(**sp)); p.add_request(pr, nvr.SamplingParams(**sp))
    st = dict(steps=0, prefill_steps=0, rows=0, near_ties=0, id_mismatch_outside_ties=0, max_abs_logit_err=0.0,
              min_margin=float("inf"), oracle_s=0.0, build_s=round(t_build, 1))
    while not p.is_finished():
        rec = p.step()
        logits = p.model_runner.logits(rec["num_seqs"])
        t0 = time.time()
        orec = o.step(forced_tokens=rec["tokens"])
        st["oracle_s"] += time.time() - t0
        assert orec["seq_ids"] == rec["seq_ids"] and orec["is_prefill"] == rec["is_prefill"], f"step {st['steps']}: batch differs"
        for ot, ps in zip(orec["block_tables"], p.last_batch()):          # a finished sequence has been deallocated (empty table)
            pt = list(ps.block_table)
            assert not pt or pt == ot, f"step {st['steps']}: block tables differ"
        assert [t == -1 for t in rec["tokens"]] == [t == -1 for t in orec["tokens"]], f"step {st['steps']}: unfinished prompts differ"
        live = [i for i, t in enumerate(rec["tokens"]) if t != -1]          # a prompt chunk that does not finish its prompt samples nothing (A-23)
        err = float(np.abs(logits[live] - orec["logits"][live]).max()) if live else 0.0
        st["max_abs_logit_err"] = max(st["max_abs_logit_err"], err)
        assert err < tol, f"step {st['steps']}: logits differ from the oracle by {err}"
        srt = np.sort(orec["logits"], axis=1)
        margin = srt[:, -1] - srt[:, -2]
        st["min_margin"] = min(st["min_margin"], float(margin.min()))
        for i, (a, b) in enumerate(zip(rec["tokens"], orec["tokens"])):
            if a != b:
                if margin[i] <= 2 * tol:
                    st["near_ties"] += 1
                else:
                    st["id_mismatch_outside_ties"] += 1
        st["steps"] += 1; st["prefill_steps"] += int(rec["is_prefill"]); st["rows"] += rec["num_seqs"]
    st["oracle_s"] = round(st["oracle_s"], 1)
    st["tol"] = tol
    assert st["id_mismatch_outside_ties"] == 0, st
    return st, o, p


def test_configs0_bs1_prompt128_greedy64_vs_oracle():
    """BASELINE configs[0] on the HIP path: one sequence, 128 prompt tokens, 64 greedy tokens, against the fp16-faithful
    oracle (tight) and against the f32 CPU-path oracle (the reference's device="cpu" arithmetic; looser: the weights
    themselves differ by fp16 rounding)."""
    ecfg = dict(max_num_seqs=1, max_num_batched_tokens=256, max_model_len=256, kvcache_block_size=256, num_kvcache_blocks=2)
    prompt = [nvr.synthetic_tokens(128, 1, 0, V).tolist()]
    st, o, p = _pair(ecfg, prompt, 64)
    assert st["steps"] == 64 and st["prefill_steps"] == 1
    assert st["near_ties"] <= 1, st
    ids_fp16 = list(next(iter(o.finished.values())).token_ids)
    _report("configs0_vs_fp16_oracle", st)
    del o, p
    st32, o32, p32 = _pair(ecfg, prompt, 64, fp16=False, tol=CPU_PATH_TOL)
    assert st32["steps"] == 64
    assert st32["near_ties"] <= 2, st32
    _report("configs0_vs_f32_cpu_path_oracle", st32)
    # same product both times: the two runs' token streams are the same stream
    assert list(next(iter(o32.finished.values())).token_ids) == ids_fp16
    ids_f32_oracle_forced = list(next(iter(o32.finished.values())).token_ids)
    del o32, p32
    # Config.dtype = "float32": the product's f32 path against the SAME f32 CPU-path oracle — the reference's own runnable configuration — at 2e-4 (F32_TOL):
    # 64 greedy ids with no near-tie allowance (the oracle is teacher-forced, and never needs to be: every id is its own arg-max)
    stf, of, pf = _pair(ecfg, prompt, 64, tol=F32_TOL, dtype="float32")
    assert stf["steps"] == 64 and stf["near_ties"] == 0, stf
    _report("configs0_float32_path_vs_f32_cpu_path_oracle", stf)
    assert len(list(next(iter(of.finished.values())).token_ids)) == len(ids_f32_oracle_forced)


def test_configs1_bs32_seq1024_full_size_vs_oracle():
    """BASELINE configs[1] at full size: 32 x 1024-token prompts through the 256^2 MFMA GEMMs and the flash prefill kernel
    over 28 layers, then 4 hipGraph decode steps at context ~1025 (8-wave paged attention, skinny GEMMs, fused LM head)."""
    ecfg = dict(max_num_seqs=32, max_num_batched_tokens=32768, max_model_len=1040, kvcache_block_size=256, num_kvcache_blocks=170)
    prompts = [nvr.synthetic_tokens(1024, 1, i, V).tolist() for i in range(32)]
    st, o, p = _pair(ecfg, prompts, 5)
    assert st["steps"] == 5 and st["prefill_steps"] == 1 and st["rows"] == 160
    assert st["near_ties"] <= 2, st
    _report("configs1_bs32_seq1024", st)


def _trace(ecfg, prompts, max_tokens, **kw):
    """(is_prefill, seq_ids, tokens, logits of the step as the logits accessor returns them AFTER nvr_engine_step) of every step + the engine"""
    nvr.lib().nvr_seq_reset_id_counter()
    e = nvr.LLMEngine(nvr.Config(**ecfg, **kw), nvr.ModelConfig("qwen3-0.6b"))
    for pr in prompts:
        e.add_request(pr, nvr.SamplingParams(temperature=0.0, max_tokens=max_tokens, ignore_eos=True))
    out = []
    while not e.is_finished():
        rec = e.step()
        out.append((rec["is_prefill"], rec["seq_ids"], rec["tokens"], e.model_runner.logits(rec["num_seqs"]).copy()))
    return out, e


def _default_equals_synchronous(ecfg, prompts, max_tokens, name, min_ahead):
    got, e = _trace(ecfg, prompts, max_tokens)                         # nvr_config_default: launch-ahead on
    assert e.config.c.async_decode == 1
    ahead = e.ahead_launched()
    assert ahead >= min_ahead, f"{name}: only {ahead} steps were launched ahead"
    del e
    ref, e0 = _trace(ecfg, prompts, max_tokens, async_decode=0)
    assert e0.ahead_launched() == 0
    del e0
    assert len(got) == len(ref)
    for k, (a, b) in enumerate(zip(got, ref)):
        assert a[:3] == b[:3], f"{name}: step {k}: batch / tokens differ between the default and the synchronous engine"
        assert np.array_equal(a[3].view(np.uint32), b[3].view(np.uint32)), f"{name}: step {k}: the logits read after the step differ in bits"
    _report(name, dict(steps=len(got), steps_launched_ahead=int(ahead), rows=int(sum(len(a[1]) for a in got)), logits_bit_identical=True))


def test_configs1_default_engine_returns_each_steps_own_logits_and_the_synchronous_stream():
    """BASELINE configs[1] (32 x 1024-token prompts) + 9 decode steps on the engine nvr_config_default builds (launch-ahead of
    greedy decode steps) against the synchronous engine (async_decode = 0): same batches, same tokens, and the logits read through the runner after
    every nvr_engine_step are THAT step's — bit for bit — although the next step has been enqueued behind it (ModelRunner::execute_model returns the
    step's own logits, model_runner.rs:105-128; LLMEngine::step, llm_engine.rs:155-197).  The default engine's stream itself is checked against the
    oracle by test_configs1_bs32_seq1024_full_size_vs_oracle above (the suite's engines are default engines since r06)."""
    ecfg = dict(max_num_seqs=32, max_num_batched_tokens=32768, max_model_len=1040, kvcache_block_size=256, num_kvcache_blocks=170)
    prompts = [nvr.synthetic_tokens(1024, 1, i, V).tolist() for i in range(32)]
    _default_equals_synchronous(ecfg, prompts, 10, "configs1_default_engine_vs_synchronous", min_ahead=6)


def test_configs4_default_engine_returns_each_steps_own_logits_and_the_synchronous_stream():
    """... and BASELINE configs[4]'s geometry (64 sequences = one 512-token system prompt + 64 own tokens; shared-prefix attention pass in every decode
    step, prefix blocks shared by BlockManager::allocate) over 10 decode steps."""
    n = 64
    ecfg = dict(max_num_seqs=n, max_num_batched_tokens=32768 + 4096, max_model_len=640, kvcache_block_size=256, num_kvcache_blocks=n + 8)
    shared = nvr.synthetic_tokens(512, 2, 0, V).tolist()
    prompts = [shared + nvr.synthetic_tokens(64, 1, i, V).tolist() for i in range(n)]
    _default_equals_synchronous(ecfg, prompts, 11, "configs4_default_engine_vs_synchronous", min_ahead=6)


def test_configs1_on_the_float32_path_vs_f32_cpu_path_oracle():
    """BASELINE configs[1]'s workload (32 x 1024-token prompts, then decode) on Config.dtype = "float32" against the oracle's f32 arithmetic — the
    reference's CPU-path numerics at the benchmark's size: logits within 2e-4 and EVERY greedy id equal (no near-tie allowance needed)."""
    ecfg = dict(max_num_seqs=32, max_num_batched_tokens=32768, max_model_len=1040, kvcache_block_size=256, num_kvcache_blocks=170)
    prompts = [nvr.synthetic_tokens(1024, 1, i, V).tolist() for i in range(32)]
    st, o, p = _pair(ecfg, prompts, 4, tol=F32_TOL, dtype="float32")
    assert st["steps"] == 4 and st["prefill_steps"] == 1 and st["rows"] == 128
    assert st["near_ties"] == 0, st
    _report("configs1_float32_path_vs_f32_cpu_path_oracle", st)


def test_configs2_mixed_length_32768_token_prefill_vs_oracle():
    """BASELINE configs[2]: the prefill sweep's lengths in ONE budget-bound batch — 4 x 4096, 2 x 2048, 4 x 1024, 8 x 512,
    8 x 256, 16 x 128 = 32 768 tokens = max_num_batched_tokens (config.rs:58), 42 sequences — last-token logits of every
    sequence against the oracle, then one decode step over the 42 ragged contexts."""
    lens = [4096] * 4 + [2048] * 2 + [1024] * 4 + [512] * 8 + [256] * 8 + [128] * 16
    assert sum(lens) == 32768
    nblk = sum((n + 1 + 255) // 256 for n in lens) + 2
    ecfg = dict(max_num_seqs=64, max_num_batched_tokens=32768, max_model_len=4100, kvcache_block_size=256, num_kvcache_blocks=nblk)
    prompts = [nvr.synthetic_tokens(n, 1, i, V).tolist() for i, n in enumerate(lens)]
    st, o, p = _pair(ecfg, prompts, 2)
    assert st["steps"] == 2 and st["prefill_steps"] == 1 and st["rows"] == 84
    assert st["near_ties"] <= 2, st
    _report("configs2_mixed_32768", st)


def test_configs2_256_sequences_over_budget_batches_chunked_vs_oracle():
    """BASELINE configs[2]'s OWN shape — 256 sequences of one length against the 32 768-token budget (scheduler.rs:119-168), several
    prefill batches — with chunked prefill on (A-23): 256 x 200 tokens = 51 200: step 1 takes 163 whole prompts + the first 168 tokens of
    the 164th, step 2 its last 32 tokens (attending to the first chunk through the block table) + the other 92 prompts; then one decode
    step over all 256.  Batch composition, block tables, which rows sample, logits and ids against the oracle's chunked engine.
    (r06, ADVICE r05: back at r03 / r04's prompts and near-tie bound — r05 had moved it to 256 x 150 with a bound of 8.)"""
    n, L = 256, 200
    ecfg = dict(max_num_seqs=n, max_num_batched_tokens=32768, max_model_len=256, kvcache_block_size=256, num_kvcache_blocks=n + 4,
                enable_chunked_prefill=True)
    prompts = [nvr.synthetic_tokens(L, 1, i, V).tolist() for i in range(n)]
    st, o, p = _pair(ecfg, prompts, 2)
    assert st["steps"] == 3 and st["prefill_steps"] == 2 and st["rows"] == 164 + 93 + 256
    assert st["near_ties"] <= 4, st
    sst = p.scheduler.get_stats()
    assert sst["prefill_batches"] == 2 and sst["decode_batches"] == 1
    _report("configs2_256seqs_x200_chunked", st)


def test_configs4_shared_system_prompt_vs_oracle():
    """BASELINE configs[4] geometry at the real model size (48 of the 512 sequences, so that the oracle finishes in a minute): every
    request = the same 512-token system prompt + 64 own tokens.  BlockManager::allocate must share the two prefix blocks exactly as the
    oracle's does (block tables compared every step), the prefill skips the cached prefix, and the decode steps take the shared-prefix
    attention pass (one MFMA pass over the 512 shared keys for the whole batch + the per-sequence remainder) — logits and greedy ids
    against the oracle's plain paged attention."""
    n = 48
    ecfg = dict(max_num_seqs=n, max_num_batched_tokens=32768, max_model_len=640, kvcache_block_size=256, num_kvcache_blocks=n + 8)
    shared = nvr.synthetic_tokens(512, 2, 0, V).tolist()
    prompts = [shared + nvr.synthetic_tokens(64, 1, i, V).tolist() for i in range(n)]
    st, o, p = _pair(ecfg, prompts, 4)
    assert st["steps"] == 4 and st["prefill_steps"] == 1 and st["rows"] == 4 * n
    assert st["near_ties"] <= 2, st
    assert p.model_runner.last_shared_prefix_len() == 512                   # the decode steps did take the shared pass
    bm = p.scheduler.block_manager.get_stats()
    st["kv_blocks_total"] = int(bm["total_blocks"])
    _report("configs4_shared_prefix_48seqs", st)


def test_configs2_and_configs4_on_the_float32_path_vs_f32_cpu_path_oracle():
    """The other two single-GPU BASELINE workloads on Config.dtype = "float32" against the oracle's f32 arithmetic, at the size of the fp16 tests above
    (r06, ADVICE r05: r05 had halved both): configs[2]'s mixed-length prefill batch (4096-token sequences down to 128, 32 768 tokens — its GEMMs on the
    f32 matrix cores since r06) + a decode step, and configs[4]'s shared 512-token system prompt (48 sequences: the prefix blocks shared by
    BlockManager::allocate, the cached prefix skipped through the f32 attention kernel's paged form) — logits within 2e-4, every greedy id equal."""
    lens = [4096] * 4 + [2048] * 2 + [1024] * 4 + [512] * 8 + [256] * 8 + [128] * 16
    assert sum(lens) == 32768
    nblk = sum((n + 1 + 255) // 256 for n in lens) + 2
    ecfg = dict(max_num_seqs=64, max_num_batched_tokens=32768, max_model_len=4100, kvcache_block_size=256, num_kvcache_blocks=nblk)
    prompts = [nvr.synthetic_tokens(n, 1, i, V).tolist() for i, n in enumerate(lens)]
    st, o, p = _pair(ecfg, prompts, 2, tol=F32_TOL, dtype="float32")
    assert st["steps"] == 2 and st["prefill_steps"] == 1 and st["rows"] == 2 * len(lens) and st["near_ties"] == 0, st
    _report("configs2_mixed_32768_float32_path_vs_f32_cpu_path_oracle", st)
    del o, p
    n = 48
    ecfg = dict(max_num_seqs=n, max_num_batched_tokens=32768, max_model_len=640, kvcache_block_size=256, num_kvcache_blocks=n + 8)
    shared = nvr.synthetic_tokens(512, 2, 0, V).tolist()
    prompts = [shared + nvr.synthetic_tokens(64, 1, i, V).tolist() for i in range(n)]
    st, o, p = _pair(ecfg, prompts, 4, tol=F32_TOL, dtype="float32")
    assert st["steps"] == 4 and st["prefill_steps"] == 1 and st["rows"] == 4 * n and st["near_ties"] == 0, st
    _report("configs4_shared_prefix_48seqs_float32_path_vs_f32_cpu_path_oracle", st)


def test_bfloat16_qwen3_0_6b_bs32_vs_bf16_oracle():
    """Config.dtype = "bfloat16" (config.rs:51,113-116) on the full-size model at BASELINE configs[1]'s batch AND length (r06, ADVICE r05: r05 ran it at
    32 x 256): the bf16 build of the 256^2 MFMA GEMMs, the flash prefill kernel, the weight-streaming decode GEMMs, the 8-wave paged attention and the
    fused LM head over 28 layers, against the oracle with bf16 at every 16-bit rounding point — 32 x 1024-token prompts + 4 hipGraph decode steps
    (r03 / r04 at this size: 5.2e-2, 5 near-ties of 160 rows)."""
    ecfg = dict(max_num_seqs=32, max_num_batched_tokens=32768, max_model_len=1040, kvcache_block_size=256, num_kvcache_blocks=170)
    prompts = [nvr.synthetic_tokens(1024, 1, i, V).tolist() for i in range(32)]
    st, o, p = _pair(ecfg, prompts, 5, tol=BF16_TOL, dtype="bfloat16")
    assert st["steps"] == 5 and st["prefill_steps"] == 1 and st["rows"] == 160
    # a token can differ from the oracle's only where the top-1 / top-2 margin is under twice the logit distance; with 8 x the fp16
    # build's distance about 8 x its count of such rows (1 of 160) is the expectation
    assert st["near_ties"] <= 12, st
    _report("bf16_bs32_seq1024", st)


def test_configs3_qwen3_8b_full_depth_vs_oracle():
    """BASELINE configs[3]'s model at FULL depth (Qwen3-8B shape: 36 layers, hidden 4096, 32:8 heads x 128, intermediate 12 288,
    V = 151 936, untied LM head; src/models/qwen3.rs:70-125 with the 8B numbers) on one GPU against the fp16-faithful oracle on a
    reduced batch: 2 x 256-token prompts + 2 decode steps (r03 / r04: 3; every step re-widens the oracle's 8.2 G fp16 weights: 10 s each) (the streaming decode GEMMs of 8B-class weights, the 256^2 prefill GEMMs
    at K = 4096 / 12 288, 36 layers of residual stream).  The oracle keeps its 8.2 G weights as fp16 in host memory (exact)."""
    ecfg = dict(max_num_seqs=2, max_num_batched_tokens=512, max_model_len=272, kvcache_block_size=256, num_kvcache_blocks=6)
    prompts = [nvr.synthetic_tokens(256, 1, i, V).tolist() for i in range(2)]
    st, o, p = _pair(ecfg, prompts, 3, model="qwen3-8b", tol=LOGIT_TOL_8B)
    assert st["steps"] == 3 and st["prefill_steps"] == 1 and st["rows"] == 6
    assert st["near_ties"] <= 1, st
    _report("configs3_qwen3_8b_full_depth_2x256", st)


def test_configs3_qwen3_8b_bs32_seq2048_tp8_in_process_equals_tp1():
    """BASELINE configs[3] at full size on ONE GPU: Qwen3-8B (36 layers, V 151 936), 32 x 2048-token prompts (two prefill batches under
    the reference's 32 768-token budget, config.rs:58) + 3 decode steps, run (a) by the single-rank product and (b) by 8 tensor-parallel
    ranks of this process (4 query heads + 1 kv head, 1536 MLP columns, 18 992 vocabulary rows per rank; linear.rs:228-239,300-304,
    421-433, embed_head.rs:57-59) exchanging at the reference's sites through the in-process communicator.  All ranks agree exactly;
    the concatenated shard logits equal the single-rank logits within the fp16-pipeline tolerance; tokens equal outside near-ties."""
    import threading
    mc = nvr.ModelConfig("qwen3-8b")
    nseq, plen, new = 32, 2048, 4
    ecfg = dict(max_num_seqs=nseq, max_num_batched_tokens=32768, max_model_len=plen + new + 8, kvcache_block_size=256,
                num_kvcache_blocks=nseq * ((plen + new + 8 + 255) // 256) + 2)
    prompts = [nvr.synthetic_tokens(plen, 1, i, V).tolist() for i in range(nseq)]
    sp = dict(temperature=0.0, max_tokens=new, ignore_eos=True)

    def drive(e, out):
        while not e.is_finished():
            rec = e.step(); rec["logits"] = e.model_runner.logits(rec["num_seqs"]).copy(); out.append(rec)
    t0 = time.time()
    nvr.lib().nvr_seq_reset_id_counter()
    e = nvr.LLMEngine(nvr.Config(**ecfg), mc)
    for pr in prompts:
        e.add_request(pr, nvr.SamplingParams(**sp))
    ref = []
    drive(e, ref)
    del e
    t_single = time.time() - t0
    assert [r["is_prefill"] for r in ref] == [True, True, False, False, False] and all(r["num_seqs"] == (16 if r["is_prefill"] else 32) for r in ref)

    tp = 8

    def run_ranks(p2p):
        group = nvr.LocalGroup(tp, p2p=p2p)
        engines = []
        for r in range(tp):
            e = nvr.LLMEngine(nvr.Config(tensor_parallel_size=tp, tensor_parallel_rank=r, **ecfg), mc)
            group.attach(e.model_runner)
            nvr.lib().nvr_seq_reset_id_counter()
            for pr in prompts:
                e.add_request(pr, nvr.SamplingParams(**sp))
            engines.append(e)
        traces, errors = [[] for _ in range(tp)], []

        def go(r):
            try:
                drive(engines[r], traces[r])
            except BaseException as ex:                                                 # noqa: BLE001
                errors.append((r, ex))
        threads = [threading.Thread(target=go, args=(r,)) for r in range(tp)]
        for t in threads: t.start()
        for t in threads: t.join(900)
        assert not errors, errors
        assert all(e.model_runner.p2p_active() == p2p for e in engines)
        return traces
    t0 = time.time()
    traces = run_ranks(False)                        # host-rendezvous collectives (RCCL-shaped data path: whole-buffer sums)
    t_tp = time.time() - t0
    assert all(len(tr) == len(ref) for tr in traces)
    st = dict(steps=len(ref), rows=0, near_ties=0, id_mismatch_outside_ties=0, max_abs_logit_err=0.0, single_s=round(t_single, 1), tp8_s=round(t_tp, 1))
    tol = LOGIT_TOL_8B                               # two fp16 pipelines with different summation trees over K = 4096 / 12 288, 36 layers
    diverged = set()
    for i, rec in enumerate(ref):
        step = [tr[i] for tr in traces]
        assert all(s["tokens"] == step[0]["tokens"] and s["seq_ids"] == step[0]["seq_ids"] and s["is_prefill"] == rec["is_prefill"] for s in step), \
            f"step {i}: ranks disagree"
        full = np.concatenate([s["logits"] for s in step], axis=1)
        assert full.shape == rec["logits"].shape
        srt = np.sort(rec["logits"], axis=1)
        margin = srt[:, -1] - srt[:, -2]
        for b, sid in enumerate(rec["seq_ids"]):
            if sid in diverged:
                continue                                                               # another history: not the same inputs any more
            err = float(np.abs(full[b] - rec["logits"][b]).max())
            st["max_abs_logit_err"] = max(st["max_abs_logit_err"], err)
            assert err < tol, f"step {i} row {b}: logits differ by {err}"
            st["rows"] += 1
            if step[0]["tokens"][b] != rec["tokens"][b]:
                diverged.add(sid)
                if margin[b] <= 2 * tol:
                    st["near_ties"] += 1
                else:
                    st["id_mismatch_outside_ties"] += 1
    assert st["id_mismatch_outside_ties"] == 0 and st["near_ties"] <= 2, st
    # ... and the same eight ranks on the one-shot peer-to-peer kernels (fused all-reduce + residual + RMSNorm in the decode steps, the arenas
    # slot by slot in the 32 768-row prefill steps incl. the chunk-overlapped exchange, device-side arg-max merge): rank-ordered sums in both
    # backends, so every rank's shard logits and tokens are the host-rendezvous run's, bit for bit, at FULL size
    t0 = time.time()
    p2p_traces = run_ranks(True)
    st["tp8_p2p_s"] = round(time.time() - t0, 1)
    for r in range(tp):
        assert len(p2p_traces[r]) == len(traces[r])
        for a, b in zip(p2p_traces[r], traces[r]):
            assert a["tokens"] == b["tokens"] and a["seq_ids"] == b["seq_ids"]
            assert np.array_equal(a["logits"], b["logits"]), "peer-to-peer and host-rendezvous collectives differ in bits at full size"
    st["p2p_equals_host_rendezvous_bitwise"] = True
    _report("configs3_qwen3_8b_bs32_seq2048_tp8_in_process_vs_tp1", st)
