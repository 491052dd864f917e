"""Tensor-parallel path.
CPU (-m "not gpu"): two gloo ranks, each holding its shard of the synthetic weights (heads / MLP columns /
vocab rows cut by the reference's shape rules, linear.rs:300-304,421-433,202, embed_head.rs:57-59), exchange
at exactly the reference's three TODO sites (all-reduce after o_proj and down_proj, vocab-shard gather for the
sampler) and must reproduce the single-rank token stream.  The arithmetic on the ranks is the oracle's: this
pins the sharding rule, the exchange points and the (max, argmax) merge with lowest-index ties that the HIP
runner implements with RCCL.
GPU (-m gpu): RCCL is loaded, a communicator is built and its all-reduce/all-gather are enqueued inside the
captured decode graph on the one GPU a test box has (NVR_TP_FORCE_COMM=1 keeps the collectives in the step)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import os, sys, json
sys.path.insert(0, os.environ["NVR_ROOT"])
import numpy as np, torch, torch.distributed as dist
import oracle
from oracle import engine_oracle as eo, model_oracle as mo
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
mcfg = mo.small(seed=9)
ecfg = dict(max_num_seqs=4, max_num_batched_tokens=128, max_model_len=128, kvcache_block_size=16, num_kvcache_blocks=16)
m = mo.OracleModel(mcfg, 16, 16, fp16=True, tp_rank=rank, tp_size=world, max_pos=128)
sched = eo.Scheduler(eo.Config(**ecfg))
eo.reset_sequence_counter()
for i, n in enumerate([5, 19, 33]):
    sched.add_sequence(eo.Sequence(oracle.fill_tokens(n, 3, i, mcfg.vocab_size).tolist(),
                                   eo.SamplingParams(temperature=0.0, max_tokens=10, ignore_eos=True), 16))
def allreduce(x):
    t = torch.from_numpy(np.ascontiguousarray(x)); dist.all_reduce(t); return oracle.round_f16(t.numpy())
out = []
while not sched.is_finished():
    seqs, pf = sched.schedule()
    ids, pos, meta = mo.build_meta(seqs, pf, 16)
    h = m.embed_tokens(ids)
    for l in range(mcfg.num_hidden_layers):
        h = oracle.add(h, allreduce(m.attn_part(l, h, pos, meta)), round16=True)      # linear.rs:236-238 (o_proj)
        h = oracle.add(h, allreduce(m.mlp_part(l, h)), round16=True)                  # linear.rs:236-238 (down_proj)
    logits = m.head_part(h, meta)                                                      # this rank's vocab shard
    val = torch.from_numpy(logits.max(1).astype(np.float32)); idx = torch.from_numpy(logits.argmax(1).astype(np.int64) + m.vocab_start)
    vals = [torch.empty_like(val) for _ in range(world)]; idxs = [torch.empty_like(idx) for _ in range(world)]
    dist.all_gather(vals, val); dist.all_gather(idxs, idx)                             # embed_head.rs:321-336
    toks = []
    for b in range(len(seqs)):
        bv, bi = float(vals[0][b]), int(idxs[0][b])
        for r in range(1, world):
            v, i = float(vals[r][b]), int(idxs[r][b])
            if v > bv or (v == bv and i < bi): bv, bi = v, i
        toks.append(bi)
    out.append(dict(pf=pf, ids=[s.seq_id for s in seqs], toks=toks, tables=[list(s.block_table) for s in seqs]))
    sched.postprocess(seqs, toks)
if rank == 0:
    json.dump(out, open(os.environ["NVR_OUT"], "w"))
dist.barrier(); dist.destroy_process_group()
'''


def test_tp2_gloo_matches_single_rank(tmp_path):
    sys.path.insert(0, ROOT)
    import oracle
    from oracle import engine_oracle as eo, model_oracle as mo
    script = tmp_path / "tp_worker.py"
    script.write_text(_WORKER)
    outp = tmp_path / "tp2.json"
    env = dict(os.environ, NVR_ROOT=ROOT, NVR_OUT=str(outp), OMP_NUM_THREADS="2")
    port = 29500 + os.getpid() % 2000
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                    "--master-port", str(port), str(script)], check=True, env=env, timeout=600, cwd=ROOT)
    import json
    got = json.load(open(outp))
    # single rank reference with the same requests
    mcfg = mo.small(seed=9)
    eo.reset_sequence_counter()
    eng = mo.OracleEngine(mcfg, eo.Config(max_num_seqs=4, max_num_batched_tokens=128, max_model_len=128, kvcache_block_size=16,
                                          num_kvcache_blocks=16), fp16=True, max_pos=128)
    for i, n in enumerate([5, 19, 33]):
        eng.add_request(oracle.fill_tokens(n, 3, i, mcfg.vocab_size).tolist(), eo.SamplingParams(temperature=0.0, max_tokens=10, ignore_eos=True))
    ref = eng.run()
    assert len(ref) == len(got)
    mism = 0
    for r, g in zip(ref, got):
        assert r["is_prefill"] == g["pf"] and r["seq_ids"] == g["ids"] and r["block_tables"] == g["tables"]
        srt = np.sort(r["logits"], axis=1)
        for i, (a, b) in enumerate(zip(r["tokens"], g["toks"])):
            if a != b:                                # only admissible at a numerical tie of the two summation orders
                assert srt[i, -1] - srt[i, -2] < 2e-2
                mism += 1
        if mism:
            break                                     # streams diverge after a tie; the prefix was identical
    assert mism <= 1


def test_oracle_tp_shards_reassemble_to_full_weights():
    """The per-rank slices cut by the reference's shape rules tile the tp=1 tensors exactly."""
    sys.path.insert(0, ROOT)
    from oracle import model_oracle as mo
    mcfg = mo.small(seed=2)
    full = mo.OracleModel(mcfg, 2, 16, fp16=True, max_pos=32)
    parts = [mo.OracleModel(mcfg, 2, 16, fp16=True, tp_rank=r, tp_size=2, max_pos=32) for r in range(2)]
    H, KVH, D, I = full.H, full.KVH, full.D, full.I
    for l in range(mcfg.num_hidden_layers):
        W, P = full.layers[l], [p.layers[l] for p in parts]
        q = np.concatenate([p["qkv"][:H // 2 * D] for p in P]); k = np.concatenate([p["qkv"][H // 2 * D:(H // 2 + KVH // 2) * D] for p in P])
        v = np.concatenate([p["qkv"][(H // 2 + KVH // 2) * D:] for p in P])
        assert np.array_equal(np.concatenate([q, k, v]), W["qkv"])
        assert np.array_equal(np.concatenate([p["o"] for p in P], 1), W["o"])
        assert np.array_equal(np.concatenate([P[0]["gate_up"][:I // 2], P[1]["gate_up"][:I // 2], P[0]["gate_up"][I // 2:], P[1]["gate_up"][I // 2:]]), W["gate_up"])
        assert np.array_equal(np.concatenate([p["down"] for p in P], 1), W["down"])
    assert np.array_equal(np.concatenate([p.lm_head for p in parts]), full.lm_head)


@pytest.mark.gpu
def test_rccl_collectives_inside_the_decode_graph_single_gpu():
    sys.path.insert(0, ROOT)
    import nvr_import
    import oracle
    from oracle import model_oracle as mo
    nvr = nvr_import.load()
    assert nvr.device_count() >= 1
    m = mo.small(seed=4)
    mc = nvr.ModelConfig(vocab_size=m.vocab_size, hidden_size=m.hidden_size, intermediate_size=m.intermediate_size,
                         num_hidden_layers=m.num_hidden_layers, num_attention_heads=m.num_attention_heads,
                         num_key_value_heads=m.num_key_value_heads, head_dim=m.head_dim, max_position_embeddings=m.max_position_embeddings,
                         rms_norm_eps=m.rms_norm_eps, rope_theta=m.rope_theta, tie_word_embeddings=m.tie_word_embeddings,
                         init_std=m.init_std, seed=m.seed)
    ecfg = dict(max_num_seqs=4, max_num_batched_tokens=128, max_model_len=128, kvcache_block_size=16, num_kvcache_blocks=16,
                skip_block_size_check=1)
    prompts = [oracle.fill_tokens(n, 5, i, m.vocab_size).tolist() for i, n in enumerate([7, 21])]

    def run(force):
        os.environ["NVR_TP_FORCE_COMM"] = "1" if force else "0"
        nvr.lib().nvr_seq_reset_id_counter()
        eng = nvr.LLMEngine(nvr.Config(**ecfg), mc)
        if force:
            eng.model_runner.init_comm(nvr.comm_unique_id())
            eng.model_runner.comm_selftest()
        for i, p in enumerate(prompts):          # one greedy row (pair merge), one stochastic row (logits gather + concat)
            sp = nvr.SamplingParams(temperature=0.0, max_tokens=8, ignore_eos=True) if i == 0 else \
                nvr.SamplingParams(temperature=0.9, top_k=30, top_p=0.9, max_tokens=8, ignore_eos=True)
            eng.add_request(p, sp)
        toks = []
        while not eng.is_finished():
            toks.append(eng.step()["tokens"])
        if force:
            eng.model_runner.comm_selftest()
        return toks
    try:
        assert run(True) == run(False)
    finally:
        os.environ.pop("NVR_TP_FORCE_COMM", None)


def _tp_ranks_vs_oracle(tp, m, ecfg, prompts, sps, product_kw=None, min_steps=5, dtype="float16", p2p=True):
    """tp in-process product ranks against each other and against the oracle's tensor-parallel engine (teacher-forced).  dtype = "bfloat16":
    the bf16 kernels and collectives against the oracle with bf16 at every 16-bit rounding point, at 8 x the fp16 tolerance."""
    import threading
    sys.path.insert(0, ROOT)
    import nvr_import
    from oracle import engine_oracle as eo, model_oracle as mo
    nvr = nvr_import.load()
    mc = nvr.ModelConfig(vocab_size=m.vocab_size, hidden_size=m.hidden_size, intermediate_size=m.intermediate_size,
                         num_hidden_layers=m.num_hidden_layers, num_attention_heads=m.num_attention_heads,
                         num_key_value_heads=m.num_key_value_heads, head_dim=m.head_dim, max_position_embeddings=m.max_position_embeddings,
                         rms_norm_eps=m.rms_norm_eps, rope_theta=m.rope_theta, tie_word_embeddings=m.tie_word_embeddings,
                         init_std=m.init_std, seed=m.seed, qk_norm=m.qk_norm, use_bias=m.use_bias)
    temps = [sp["temperature"] for sp in sps]
    product_kw = product_kw or {}
    group = nvr.LocalGroup(tp, p2p=p2p)
    bf16, f32 = dtype == "bfloat16", dtype == "float32"
    tol = 1.6e-1 if bf16 else 2e-4 if f32 else 2e-2
    engines = []
    for r in range(tp):
        e = nvr.LLMEngine(nvr.Config(skip_block_size_check=1, tensor_parallel_size=tp, tensor_parallel_rank=r, sample_seed=11, dtype=dtype,
                                     **ecfg, **product_kw), mc)
        group.attach(e.model_runner)
        engines.append(e)
    traces = [[] for _ in range(tp)]
    shared_seen = set()
    errors = []

    def drive(r):
        try:
            e = engines[r]
            while not e.is_finished():
                rec = e.step()
                rec["logits"] = e.model_runner.logits(rec["num_seqs"]).copy()           # this rank's vocab shard
                if not rec["is_prefill"]:
                    shared_seen.add(e.model_runner.last_shared_prefix_len())
                traces[r].append(rec)
        except BaseException as ex:                                                     # noqa: BLE001
            errors.append((r, ex))

    # identical requests on every rank, numbered 0.. on every rank as in the one-process-per-rank deployment (the sampling keys
    # are derived from the sequence id)
    for e in engines:
        nvr.lib().nvr_seq_reset_id_counter()
        for pr, sp in zip(prompts, sps):
            e.add_request(pr, nvr.SamplingParams(**sp))
    threads = [threading.Thread(target=drive, args=(r,)) for r in range(tp)]
    for t in threads: t.start()
    for t in threads: t.join(300)
    assert not errors, errors
    assert all(len(tr) == len(traces[0]) and len(tr) > min_steps for tr in traces)
    greedy_rows_only = all(t == 0.0 for t in temps)
    for step in zip(*traces):
        assert all(s["is_prefill"] == step[0]["is_prefill"] and s["num_seqs"] == step[0]["num_seqs"] for s in step)
        assert all(s["tokens"] == step[0]["tokens"] and s["seq_ids"] == step[0]["seq_ids"] for s in step), "ranks disagree on the sampled tokens"
    # the oracle's tensor-parallel engine, teacher-forced with rank 0's tokens
    eo.reset_sequence_counter()
    o = mo.OracleEngine(m, eo.Config(**ecfg), fp16=not bf16 and not f32, bf16=bf16, tp_size=tp, max_pos=ecfg["max_model_len"], sample_seed=11)
    for pr, sp in zip(prompts, sps):
        o.add_request(pr, eo.SamplingParams(**sp))
    near = 0
    Vl = m.vocab_size // tp
    for i, rec in enumerate(traces[0]):
        orec = o.step(forced_tokens=rec["tokens"])
        assert orec["is_prefill"] == rec["is_prefill"] and len(orec["seq_ids"]) == rec["num_seqs"]
        for r in range(tp):
            err = np.abs(traces[r][i]["logits"] - orec["logits"][:, r * Vl:(r + 1) * Vl]).max()
            assert err < tol, f"step {i} rank {r}: shard logits differ by {err}"
        if greedy_rows_only:
            srt = np.sort(orec["logits"], axis=1)
            for b, (tg, to) in enumerate(zip(rec["tokens"], orec["tokens"])):
                if tg != to:
                    assert srt[b, -1] - srt[b, -2] <= 2 * tol, f"step {i} row {b}: token {tg} != {to}"
                    near += 1
    assert near <= 2
    assert o.scheduler.is_finished()
    _tp_ranks_vs_oracle.ahead_launched = [e.ahead_launched() for e in engines]
    _tp_ranks_vs_oracle.last_ragged = [e.model_runner.last_decode_ragged() for e in engines]
    _tp_ranks_vs_oracle.rank0_tokens = [rec["tokens"] for rec in traces[0]]
    return shared_seen


@pytest.mark.gpu
@pytest.mark.parametrize("tp,temps", [(2, [0.0, 0.0, 0.0]), (2, [0.0, 0.8, 1.0]), (4, [0.0, 0.0, 0.0])])
def test_product_tensor_parallel_ranks_in_process_match_the_oracle(tp, temps):
    """The PRODUCT's tensor-parallel path end to end on one GPU: tp runners of this process (rank r holds its head / column /
    vocab shard, linear.rs:300-304,421-433,202, embed_head.rs:57-59), one host thread per rank, exchanging at the reference's
    sites (all-reduce after o_proj and down_proj, (max, argmax) pairs or logits shards for the sampler) through the in-process
    communicator — the same call sites as RCCL, with a host rendezvous instead of xGMI.  Every rank must schedule the same
    batches and sample the same tokens as the others, and they must be the oracle's tensor-parallel engine's (teacher-forced):
    shard logits within tolerance, greedy tokens equal outside numerical near-ties."""
    import oracle
    from oracle import model_oracle as mo
    m = mo.small(seed=6, num_attention_heads=8, num_key_value_heads=4, head_dim=64, hidden_size=256, intermediate_size=512)
    ecfg = dict(max_num_seqs=4, max_num_batched_tokens=256, max_model_len=128, kvcache_block_size=16, num_kvcache_blocks=24)
    prompts = [oracle.fill_tokens(n, 5, i, m.vocab_size).tolist() for i, n in enumerate([9, 40, 17])]
    sps = [dict(temperature=t, max_tokens=10, ignore_eos=True, **({} if t == 0.0 else dict(top_k=30))) for t in temps]
    _tp_ranks_vs_oracle(tp, m, ecfg, prompts, sps)


@pytest.mark.gpu
@pytest.mark.parametrize("tp", [2, 4])
def test_tensor_parallel_ranks_take_the_work_balanced_attention_for_ragged_batches(tp):
    """A ragged decode batch (44 sequences of 200..1500 keys) on tensor-parallel ranks: every rank holds 8 / tp kv heads of every sequence, sees the same
    ragged contexts, and takes the work-balanced attention launch (attn_share_kernel over its own pairs, inside the captured decode graph next to the one-shot
    collectives); ranks agree on every token, shard logits match the oracle's tensor-parallel engine."""
    import oracle
    from oracle import model_oracle as mo
    m = mo.small(seed=8, num_attention_heads=16, num_key_value_heads=8, head_dim=64, hidden_size=256, intermediate_size=512)
    ecfg = dict(max_num_seqs=44, max_num_batched_tokens=32768, max_model_len=2048, kvcache_block_size=256, num_kvcache_blocks=44 * 8 + 4)
    lens = [int(200 * (1500 / 200) ** (i / 43)) for i in range(44)]
    prompts = [oracle.fill_tokens(n, 5, i, m.vocab_size).tolist() for i, n in enumerate(lens)]
    sps = [dict(temperature=0.0, max_tokens=6, ignore_eos=True) for _ in lens]
    _tp_ranks_vs_oracle(tp, m, ecfg, prompts, sps, min_steps=4)
    # (the rule asks for six 64-key units of work per CU on the RANK: its 8 / tp kv heads of ~29 k keys — met at tp 2, not at tp 4: per-pair launches there)
    want = (sum(lens) + 44 * 6) * (8 // tp) // 64 >= 6 * 256
    assert _tp_ranks_vs_oracle.last_ragged == [want] * tp, _tp_ranks_vs_oracle.last_ragged


@pytest.mark.gpu
@pytest.mark.parametrize("tp,p2p,temps", [(2, True, [0.0, 0.0, 0.0]), (4, True, [0.0, 0.7, 0.0]), (2, False, [0.0, 0.0, 0.0])])
def test_bfloat16_tensor_parallel_ranks_match_the_bf16_oracle(tp, p2p, temps):
    """Config.dtype = "bfloat16" (config.rs:51,113-116) on tensor-parallel ranks: the bf16 build of every kernel AND of the exchange
    (one-shot peer-to-peer all-reduce + residual + RMSNorm rounding its sums to bf16; the host-rendezvous sum likewise) — ranks agree
    bit for bit, shard logits within the bf16 tolerance of the oracle's bf16 tensor-parallel engine; launch-ahead on (p2p) as well."""
    import oracle
    from oracle import model_oracle as mo
    m = mo.small(seed=6, num_attention_heads=8, num_key_value_heads=4, head_dim=64, hidden_size=256, intermediate_size=512)
    ecfg = dict(max_num_seqs=4, max_num_batched_tokens=256, max_model_len=128, kvcache_block_size=16, num_kvcache_blocks=24)
    prompts = [oracle.fill_tokens(n, 5, i, m.vocab_size).tolist() for i, n in enumerate([9, 40, 17])]
    sps = [dict(temperature=t, max_tokens=10, ignore_eos=True, **({} if t == 0.0 else dict(top_k=30))) for t in temps]
    _tp_ranks_vs_oracle(tp, m, ecfg, prompts, sps, dtype="bfloat16", p2p=p2p)


@pytest.mark.gpu
@pytest.mark.parametrize("tp,p2p,temps,bias", [(2, True, [0.0, 0.0, 0.0], False), (4, True, [0.0, 0.7, 0.0], False), (2, False, [0.0, 0.0, 0.0], True)])
def test_float32_tensor_parallel_ranks_match_the_f32_oracle(tp, p2p, temps, bias):
    """Config.dtype = "float32" on tensor-parallel ranks (r05; config.rs:51,113-116 with the dtype-agnostic TP classes of linear.rs:88-268): every
    rank's f32 partial sums are gathered (the one-shot arenas' all-gather form for decode-sized rows — bytes moved, never summed there —, slot by
    slot for prefill rows; the in-process rendezvous with p2p off) and summed in rank order in f32 by the add + RMSNorm kernel.  Ranks agree bit
    for bit; shard logits within 2e-4 of the f32 oracle's tensor-parallel engine (summation order only); greedy ids with the fp16 tests' tie rule
    at that tolerance.  The same requests on ONE float32 rank give the same greedy tokens."""
    import oracle
    from oracle import model_oracle as mo
    m = mo.small(seed=6, num_attention_heads=8, num_key_value_heads=4, head_dim=64, hidden_size=256, intermediate_size=512, use_bias=bias,
                 tie_word_embeddings=bias)                      # (third case: biases on rank 0 in front of the exchange, and the LM head = this rank's rows of the embedding)
    ecfg = dict(max_num_seqs=4, max_num_batched_tokens=256, max_model_len=128, kvcache_block_size=16, num_kvcache_blocks=24)
    prompts = [oracle.fill_tokens(n, 5, i, m.vocab_size).tolist() for i, n in enumerate([9, 40, 17])]
    sps = [dict(temperature=t, max_tokens=10, ignore_eos=True, **({} if t == 0.0 else dict(top_k=30))) for t in temps]
    _tp_ranks_vs_oracle(tp, m, ecfg, prompts, sps, dtype="float32", p2p=p2p)
    if all(t == 0.0 for t in temps):
        import nvr_import
        nvr = nvr_import.load()
        mc = nvr.ModelConfig(vocab_size=m.vocab_size, hidden_size=m.hidden_size, intermediate_size=m.intermediate_size, num_hidden_layers=m.num_hidden_layers,
                             num_attention_heads=m.num_attention_heads, num_key_value_heads=m.num_key_value_heads, head_dim=m.head_dim,
                             max_position_embeddings=m.max_position_embeddings, rms_norm_eps=m.rms_norm_eps, rope_theta=m.rope_theta,
                             tie_word_embeddings=m.tie_word_embeddings, init_std=m.init_std, seed=m.seed, qk_norm=m.qk_norm, use_bias=m.use_bias)
        nvr.lib().nvr_seq_reset_id_counter()
        one = nvr.LLMEngine(nvr.Config(skip_block_size_check=1, sample_seed=11, dtype="float32", **ecfg), mc)
        for pr, sp in zip(prompts, sps):
            one.add_request(pr, nvr.SamplingParams(**sp))
        single = []
        while not one.is_finished():
            single.append(one.step()["tokens"])
        assert single == _tp_ranks_vs_oracle.rank0_tokens, "tensor-parallel float32 ranks and the single float32 rank sample different tokens"


@pytest.mark.gpu
@pytest.mark.parametrize("feature", ["qk_norm", "shared_prefix", "use_bias"])
def test_tensor_parallel_ranks_with_r02_graph_extensions(feature):
    """The graph extensions on tensor-parallel ranks (tp = 2, in-process): q/k head norms (norm weights replicated, heads
    sharded), the shared-prefix decode attention pass (every rank sees the same block tables, so every rank takes it) and the
    projections' biases (use_bias)."""
    import oracle
    from oracle import model_oracle as mo
    V = 1024
    if feature == "qk_norm":
        m = mo.small(seed=7, num_attention_heads=8, num_key_value_heads=4, head_dim=64, hidden_size=256, intermediate_size=512, qk_norm=True)
        ecfg = dict(max_num_seqs=4, max_num_batched_tokens=256, max_model_len=128, kvcache_block_size=16, num_kvcache_blocks=24)
        prompts = [oracle.fill_tokens(n, 5, i, V).tolist() for i, n in enumerate([9, 40, 17])]
        sps = [dict(temperature=0.0, max_tokens=10, ignore_eos=True)] * 3
        assert _tp_ranks_vs_oracle(2, m, ecfg, prompts, sps) == {0}
    elif feature == "use_bias":                 # A-30: column-parallel biases are sharded with their rows, row-parallel ones live on rank 0 (linear.rs:206)
        m = mo.small(seed=12, num_attention_heads=8, num_key_value_heads=4, head_dim=64, hidden_size=256, intermediate_size=512, use_bias=True)
        ecfg = dict(max_num_seqs=4, max_num_batched_tokens=256, max_model_len=128, kvcache_block_size=16, num_kvcache_blocks=24)
        prompts = [oracle.fill_tokens(n, 5, i, V).tolist() for i, n in enumerate([9, 40, 17])]
        sps = [dict(temperature=0.0, max_tokens=10, ignore_eos=True)] * 3
        assert _tp_ranks_vs_oracle(2, m, ecfg, prompts, sps) == {0}
        assert _tp_ranks_vs_oracle(2, m, ecfg, prompts, sps, p2p=False) == {0}
    else:
        m = mo.small(seed=8, num_attention_heads=8, num_key_value_heads=4, head_dim=64, hidden_size=256, intermediate_size=512)
        ecfg = dict(max_num_seqs=8, max_num_batched_tokens=2048, max_model_len=384, kvcache_block_size=64, num_kvcache_blocks=40)
        system = oracle.fill_tokens(140, 4, 3, V).tolist()
        prompts = [system + oracle.fill_tokens(4 + 7 * i, 4, 50 + i, V).tolist() for i in range(6)]
        sps = [dict(temperature=0.0, max_tokens=9, ignore_eos=True)] * 6
        assert _tp_ranks_vs_oracle(2, m, ecfg, prompts, sps, product_kw=dict(shared_prefix_min_seqs=3)) == {128}


@pytest.mark.gpu
@pytest.mark.parametrize("tp,model", [(2, "qwen3-0.6b"), (8, "qwen3-0.6b"), (8, "qwen3-8b-2layers"), (4, "qwen3-8b-2layers")])
def test_full_size_tensor_parallel_in_process_equals_single_rank(tp, model):
    """The benchmark model at full size (and the Qwen3-8B layer geometry of BASELINE configs[3], two layers), sharded over tp in-process ranks (tp = 8: 2 query heads and 1 kv head, 384 MLP columns,
    18 992 vocabulary rows per rank — the narrow-shard launch rules, partitioned attention + merge, sharded LM head and the
    (max, argmax) gather), against the single-rank product on the same prompts: the concatenated shard logits agree with the
    single-rank logits to fp16-pipeline tolerance and the greedy tokens agree outside near-ties; all ranks agree exactly."""
    import threading
    sys.path.insert(0, ROOT)
    import nvr_import
    nvr = nvr_import.load()
    if model == "qwen3-0.6b":
        mc = nvr.ModelConfig("qwen3-0.6b")
    else:                                       # BASELINE configs[3] geometry (Hd 4096, 32:8 heads, D 128, I 12288), two layers, small vocabulary
        mc = nvr.ModelConfig(vocab_size=4096, hidden_size=4096, intermediate_size=12288, num_hidden_layers=2, num_attention_heads=32,
                             num_key_value_heads=8, head_dim=128, max_position_embeddings=1024, rms_norm_eps=1e-6, rope_theta=1e6,
                             tie_word_embeddings=False, init_std=0.02, seed=21)
    V = mc.c.vocab_size
    ecfg = dict(max_num_seqs=4, max_num_batched_tokens=1024, max_model_len=512, kvcache_block_size=256, num_kvcache_blocks=12)
    prompts = [nvr.synthetic_tokens(n, 1, i, V).tolist() for i, n in enumerate([24, 300, 9, 130])]
    sp = dict(temperature=0.0, max_tokens=6, ignore_eos=True)

    def run_single():
        nvr.lib().nvr_seq_reset_id_counter()
        e = nvr.LLMEngine(nvr.Config(**ecfg), mc)
        for pr in prompts:
            e.add_request(pr, nvr.SamplingParams(**sp))
        out = []
        while not e.is_finished():
            rec = e.step(); rec["logits"] = e.model_runner.logits(rec["num_seqs"]).copy(); out.append(rec)
        return out
    ref = run_single()

    # tp <= 4: the one-shot peer-to-peer collectives (ranks wait for each other inside kernels, one hardware queue each);
    # tp = 8: the host-rendezvous collectives (eight spinning kernels of one process would share the GPU's queues)
    group = nvr.LocalGroup(tp, p2p=tp <= 4)
    engines = []
    for r in range(tp):
        e = nvr.LLMEngine(nvr.Config(tensor_parallel_size=tp, tensor_parallel_rank=r, **ecfg), mc)
        group.attach(e.model_runner)
        nvr.lib().nvr_seq_reset_id_counter()
        for pr in prompts:
            e.add_request(pr, nvr.SamplingParams(**sp))
        engines.append(e)
    traces, errors = [[] for _ in range(tp)], []

    def drive(r):
        try:
            e = engines[r]
            while not e.is_finished():
                rec = e.step(); rec["logits"] = e.model_runner.logits(rec["num_seqs"]).copy(); traces[r].append(rec)
        except BaseException as ex:                                                     # noqa: BLE001
            errors.append((r, ex))
    threads = [threading.Thread(target=drive, args=(r,)) for r in range(tp)]
    for t in threads: t.start()
    for t in threads: t.join(600)
    assert not errors, errors
    assert all(len(tr) == len(ref) for tr in traces)
    near = 0
    for i, rec in enumerate(ref):
        step = [tr[i] for tr in traces]
        assert all(s["tokens"] == step[0]["tokens"] and s["is_prefill"] == rec["is_prefill"] for s in step)
        full = np.concatenate([s["logits"] for s in step], axis=1)
        assert full.shape == rec["logits"].shape
        if step[0]["tokens"] == rec["tokens"] or i == 0:                                # same history so far: same inputs
            assert np.abs(full - rec["logits"]).max() < 3e-2, (i, np.abs(full - rec["logits"]).max())
        srt = np.sort(rec["logits"], axis=1)
        for b, (tg, to) in enumerate(zip(step[0]["tokens"], rec["tokens"])):
            if tg != to:
                near += 1
                if i == 0:
                    assert srt[b, -1] - srt[b, -2] <= 6e-2
    assert near <= 2, near


@pytest.mark.gpu
@pytest.mark.parametrize("tp", [2, 4])
def test_one_shot_p2p_collectives_equal_host_rendezvous(tp):
    """The one-shot peer-to-peer all-reduce + residual + RMSNorm and the small all-gather (kernels/comm_p2p.hip: push into the
    peers' arenas, epoch flags, rank-ordered f32 sum) against the host-rendezvous collectives of the same in-process ranks:
    same summation order and rounding points, so the shard logits of every step and the token streams are BIT-identical; the
    p2p ranks replay captured hipGraphs (the collectives are plain kernel nodes), the rendezvous ranks launch eagerly.
    Also: the communicator self-test (all-reduce of ones, all-gather of rank ids) through the arenas."""
    import threading
    sys.path.insert(0, ROOT)
    import nvr_import
    import oracle
    from oracle import model_oracle as mo
    nvr = nvr_import.load()
    m = mo.small(seed=6, num_attention_heads=8, num_key_value_heads=4, head_dim=64, hidden_size=256, intermediate_size=512)
    mc = nvr.ModelConfig(vocab_size=m.vocab_size, hidden_size=m.hidden_size, intermediate_size=m.intermediate_size,
                         num_hidden_layers=m.num_hidden_layers, num_attention_heads=m.num_attention_heads,
                         num_key_value_heads=m.num_key_value_heads, head_dim=m.head_dim, max_position_embeddings=m.max_position_embeddings,
                         rms_norm_eps=m.rms_norm_eps, rope_theta=m.rope_theta, tie_word_embeddings=m.tie_word_embeddings,
                         init_std=m.init_std, seed=m.seed)
    ecfg = dict(max_num_seqs=8, max_num_batched_tokens=256, max_model_len=128, kvcache_block_size=16, num_kvcache_blocks=40)
    prompts = [oracle.fill_tokens(n, 5, i, m.vocab_size).tolist() for i, n in enumerate([9, 40, 17, 3, 26])]

    def run(p2p, eager):
        group = nvr.LocalGroup(tp, p2p=p2p)
        engines = []
        for r in range(tp):
            e = nvr.LLMEngine(nvr.Config(skip_block_size_check=1, tensor_parallel_size=tp, tensor_parallel_rank=r, enforce_eager=eager, **ecfg), mc)
            group.attach(e.model_runner)
            nvr.lib().nvr_seq_reset_id_counter()
            for pr in prompts:
                e.add_request(pr, nvr.SamplingParams(temperature=0.0, max_tokens=24, ignore_eos=True))
            engines.append(e)
        traces, errors = [[] for _ in range(tp)], []

        def drive(r):
            try:
                e = engines[r]
                e.model_runner.comm_selftest()
                while not e.is_finished():
                    rec = e.step(); rec["logits"] = e.model_runner.logits(rec["num_seqs"]).copy(); traces[r].append(rec)
                assert e.model_runner.p2p_active() == p2p
            except BaseException as ex:                                                 # noqa: BLE001
                errors.append((r, ex))
        threads = [threading.Thread(target=drive, args=(r,)) for r in range(tp)]
        for t in threads: t.start()
        for t in threads: t.join(300)
        assert not errors, errors
        return traces
    # eager on both sides for the bit comparison (a captured step partitions the attention by its 256-token context bucket,
    # an eager one by the real context: different split-KV merge order, same tokens)
    a, b = run(True, True), run(False, True)
    assert len(a[0]) == len(b[0]) > 20
    for r in range(tp):
        for sa, sb in zip(a[r], b[r]):
            assert sa["tokens"] == sb["tokens"] and sa["seq_ids"] == sb["seq_ids"]
            assert np.array_equal(sa["logits"], sb["logits"]), "p2p and host-rendezvous collectives differ in bits"
    g = run(True, False)                                 # p2p collectives inside replayed hipGraphs
    for step in zip(*g):
        assert all(s["tokens"] == step[0]["tokens"] for s in step)
    assert [s["tokens"] for s in g[0]] == [s["tokens"] for s in a[0]]


@pytest.mark.gpu
@pytest.mark.parametrize("tp,dtype", [(2, "float16"), (4, "float16"), (2, "bfloat16")])
def test_fenced_protocol_is_a_drop_in_for_the_fence_free_one_and_the_selftest_chain_falls_back(tp, dtype):
    """VERDICT r05 item 3 / ADVICE r05: the one-shot collectives exist in two protocols in ONE build (kernels/comm_p2p.hip): fence-free (default) and r04's
    fenced form (nvr_runner_p2p_set_fenced / NVR_P2P_FENCED=1) — the fallback BETWEEN fence-free and RCCL.  In-process ranks: (1) the communicator
    self-test passes under both, including its back-to-back part (the largest decode message, 32 collectives on consecutive epochs with no host
    synchronisation, fused and plain launches alternating, every round's payload its own); (2) a fenced run's shard logits and tokens are the
    fence-free run's bit for bit, in replayed decode graphs; (3) the control plane's chain: a self-test that FAILS under fence-free (injected on
    every rank by NVR_SELFTEST_INJECT=1) -> p2p_reset -> set_fenced(1) -> the self-test passes -> the engines serve, fenced, with the same stream."""
    import threading
    sys.path.insert(0, ROOT)
    import nvr_import
    import oracle
    from oracle import model_oracle as mo
    nvr = nvr_import.load()
    m = mo.small(seed=6, num_attention_heads=8, num_key_value_heads=4, head_dim=64, hidden_size=256, intermediate_size=512)
    mc = nvr.ModelConfig(vocab_size=m.vocab_size, hidden_size=m.hidden_size, intermediate_size=m.intermediate_size,
                         num_hidden_layers=m.num_hidden_layers, num_attention_heads=m.num_attention_heads,
                         num_key_value_heads=m.num_key_value_heads, head_dim=m.head_dim, max_position_embeddings=m.max_position_embeddings,
                         rms_norm_eps=m.rms_norm_eps, rope_theta=m.rope_theta, tie_word_embeddings=m.tie_word_embeddings,
                         init_std=m.init_std, seed=m.seed)
    ecfg = dict(max_num_seqs=8, max_num_batched_tokens=512, max_model_len=128, kvcache_block_size=16, num_kvcache_blocks=40, dtype=dtype)
    prompts = [oracle.fill_tokens(n, 5, i, m.vocab_size).tolist() for i, n in enumerate([9, 40, 17, 3, 26])]

    def run(mode):
        """mode: 'free' | 'fenced' | 'chain' (fence-free self-test fails by injection, the ranks fall back to fenced)"""
        if mode == "chain": os.environ["NVR_SELFTEST_INJECT"] = "1"
        try:
            group = nvr.LocalGroup(tp, p2p=True)
            engines = []
            for r in range(tp):
                e = nvr.LLMEngine(nvr.Config(skip_block_size_check=1, tensor_parallel_size=tp, tensor_parallel_rank=r, **ecfg), mc)
                group.attach(e.model_runner)
                nvr.lib().nvr_seq_reset_id_counter()
                for pr in prompts:
                    e.add_request(pr, nvr.SamplingParams(temperature=0.0, max_tokens=24, ignore_eos=True))
                engines.append(e)
        finally:
            os.environ.pop("NVR_SELFTEST_INJECT", None)
        traces, errors, backend = [[] for _ in range(tp)], [], [None] * tp
        gate = threading.Barrier(tp)

        def drive(r):
            try:
                e = engines[r]; mr = e.model_runner
                assert not mr.p2p_fenced()
                if mode == "fenced":
                    mr.p2p_set_fenced(True)
                    gate.wait(60)
                    mr.comm_selftest()
                elif mode == "chain":
                    failed = False
                    try:
                        mr.comm_selftest()
                    except nvr.NvrError as ex:
                        failed = True
                        assert ex.code == -9 and "injected" in str(ex)
                    assert failed
                    gate.wait(60)                                                # (the control plane's all_ok: every rank saw the failure)
                    mr.p2p_reset(); mr.p2p_set_fenced(True)
                    gate.wait(60)
                    mr.comm_selftest()                                           # passes: the ranks go on, fenced
                else:
                    mr.comm_selftest()
                backend[r] = "fenced" if mr.p2p_fenced() else "fence-free"
                while not e.is_finished():
                    rec = e.step(); rec["logits"] = mr.logits(rec["num_seqs"]).copy(); traces[r].append(rec)
                assert mr.p2p_active()
            except BaseException as ex:                                                 # noqa: BLE001
                errors.append((r, ex))
                gate.abort()
        threads = [threading.Thread(target=drive, args=(r,)) for r in range(tp)]
        for t in threads: t.start()
        for t in threads: t.join(300)
        assert not errors, errors
        return traces, backend
    free, b0 = run("free")
    assert b0 == ["fence-free"] * tp and len(free[0]) > 20
    for mode in ("fenced", "chain"):
        got, b1 = run(mode)
        assert b1 == ["fenced"] * tp
        for r in range(tp):
            assert len(got[r]) == len(free[r])
            for sa, sb in zip(got[r], free[r]):
                assert sa["tokens"] == sb["tokens"] and sa["seq_ids"] == sb["seq_ids"]
                assert np.array_equal(sa["logits"].view(np.uint32), sb["logits"].view(np.uint32)), f"{mode}: the fenced protocol changed bits"


@pytest.mark.gpu
@pytest.mark.parametrize("tp,p2p,heads,inter,shared", [(2, True, 32, 2048, False), (4, True, 64, 4096, False), (2, False, 32, 2048, False), (2, True, 32, 2048, True)])
def test_tensor_parallel_prefill_exchange_overlaps_on_a_second_stream(tp, p2p, heads, inter, shared):
    """Row g, prefill side (linear.rs:228-239 with its all-reduce :236-238): on tensor-parallel ranks a prefill step of >= 1024 rows is cut into
    token chunks (multiples of the GEMM's 256-row tile) and the all-reduce of chunk i runs on a second HIP stream under the GEMM of chunk i + 1
    (events between the streams; residual add + RMSNorm of a chunk behind its reduce) — csrc/model_runner.cpp row_parallel_norm.  Same bits as
    the serial form (GEMM -> all-reduce -> add + norm on one stream, nvr_runner_set_tp_prefill_overlap(0)): every rank's shard logits of the
    prefill step and of the decode steps behind it, and the token streams; with the one-shot peer-to-peer kernels (the arenas chunk by chunk:
    messages beyond a slot) and with the host-rendezvous backend (which blocks the host inside the reduce: the next GEMM is already queued)."""
    import threading
    sys.path.insert(0, ROOT)
    import nvr_import
    import oracle
    from oracle import model_oracle as mo
    nvr = nvr_import.load()
    # (shapes for which a rank's o_proj / down_proj take the 256x256 GEMM — K = 1024 per rank, hidden 1024, > 8192 rows — like the BASELINE models' prefills)
    m = mo.small(seed=26, num_attention_heads=heads, num_key_value_heads=8, head_dim=64, hidden_size=1024, intermediate_size=inter)
    mc = nvr.ModelConfig(vocab_size=m.vocab_size, hidden_size=m.hidden_size, intermediate_size=m.intermediate_size,
                         num_hidden_layers=m.num_hidden_layers, num_attention_heads=m.num_attention_heads,
                         num_key_value_heads=m.num_key_value_heads, head_dim=m.head_dim, max_position_embeddings=2048,
                         rms_norm_eps=m.rms_norm_eps, rope_theta=m.rope_theta, tie_word_embeddings=m.tie_word_embeddings,
                         init_std=m.init_std, seed=m.seed)
    ecfg = dict(max_num_seqs=8, max_num_batched_tokens=8704, max_model_len=2048, kvcache_block_size=64, num_kvcache_blocks=160)
    lens = [1700, 1513, 1300, 1900, 1257, 833]                                             # one prefill step of 8503 rows: 4 chunks of 2304 (the last 1591)
    prompts = [oracle.fill_tokens(n, 5, i, m.vocab_size).tolist() for i, n in enumerate(lens)]
    if shared:                                                                             # sequences 4 and 5 start with the first 640 tokens (10 cache blocks) of
        prompts[4] = prompts[0][:640] + prompts[4][640:]                                   # sequence 0: their cached prefixes are skipped, the step's attention goes
        prompts[5] = prompts[0][:640] + prompts[5][640:]                                   # through the block tables, and in mode 2 they sit in the OTHER micro-batch

    def run(overlap):
        group = nvr.LocalGroup(tp, p2p=p2p)
        engines = []
        for r in range(tp):
            e = nvr.LLMEngine(nvr.Config(skip_block_size_check=1, tensor_parallel_size=tp, tensor_parallel_rank=r, enforce_eager=1, **ecfg), mc)
            group.attach(e.model_runner)
            e.model_runner.set_tp_prefill_overlap(overlap)
            nvr.lib().nvr_seq_reset_id_counter()
            for pr in prompts:
                e.add_request(pr, nvr.SamplingParams(temperature=0.0, max_tokens=4, ignore_eos=True))
            engines.append(e)
        traces, errors, chunks = [[] for _ in range(tp)], [], [0] * tp

        def drive(r):
            try:
                e = engines[r]
                while not e.is_finished():
                    rec = e.step(); rec["logits"] = e.model_runner.logits(rec["num_seqs"]).copy(); traces[r].append(rec)
                    if rec["is_prefill"]:
                        chunks[r] = e.model_runner.last_overlap_chunks()
            except BaseException as ex:                                                 # noqa: BLE001
                errors.append((r, ex))
        threads = [threading.Thread(target=drive, args=(r,)) for r in range(tp)]
        for t in threads: t.start()
        for t in threads: t.join(300)
        assert not errors, errors
        return traces, chunks
    (a, ca), (b, cb), (c, cc) = run(1), run(0), run(2)
    two = 2 if (heads // tp) // (8 // tp) in (1, 2, 4) else 4          # (mode 2 needs the MFMA attention kernel's tiles; GQA group 8 falls back to the chunks)
    print(f"[overlap forms] tp={tp} p2p={p2p} shared={shared}: chunks {ca}, serial {cb}, micro-batches {cc}")
    if shared:                                                          # 7 223 rows: the cost rule routes this step's o / down GEMMs to the 128-row kernel: its chunks
        assert cb == [0] * tp and ca == [4] * tp and cc == [2] * tp, (ca, cb, cc)
    else:
        assert ca == [4] * tp and cb == [0] * tp and cc == [two] * tp, (ca, cb, cc)
    for r in range(tp):                                                                    # mode 2: two micro-batches of whole sequences, same bits again
        for sc, sb in zip(c[r], b[r]):
            assert sc["tokens"] == sb["tokens"] and np.array_equal(sc["logits"], sb["logits"]), "two-micro-batch and serial prefill differ in bits"
    assert len(a[0]) == len(b[0]) == 4 and a[0][0]["is_prefill"] and a[0][0]["num_tokens"] == sum(lens) - (2 * 640 if shared else 0)
    for r in range(tp):
        for sa, sb in zip(a[r], b[r]):
            assert sa["tokens"] == sb["tokens"] and sa["seq_ids"] == sb["seq_ids"]
            assert np.array_equal(sa["logits"], sb["logits"]), "overlapped and serial prefill exchange differ in bits"
    for step in zip(*a):                                                                   # and the ranks agree with each other
        assert all(s["tokens"] == step[0]["tokens"] for s in step)


@pytest.mark.gpu
@pytest.mark.parametrize("async_on", [0, 1])
def test_peer_that_never_arrives_fails_the_step_and_the_group_recovers(async_on):
    """Negative path of the one-shot collectives (kernels/comm_p2p.hip): rank 1 never enters a decode step that rank 0 runs.  Rank 0's
    reduce workgroups wait a bounded time (NVR_P2P_TIMEOUT_MS, read when the runner is created), set the error word and finish —
    the GPU does not hang — and the step fails with NVR_ERR_RCCL at the ABI instead of returning sums with zeros in place of the
    peer's (the batch is aborted on the failing rank).  Recovery as nvr.h documents it: every rank aborts that batch
    (nvr_engine_abort_last_batch), resets its arena words (nvr_runner_p2p_reset), the control plane barriers, and the SAME engines
    serve new requests: both ranks sample the same tokens, equal to the single-rank product's.  async_on = 1 (the default engine): the step both
    ranks launched ahead completes normally on rank 0's next call; the one only rank 0 launches behind it is the step that fails, one call later."""
    import threading
    import oracle
    from oracle import model_oracle as mo
    sys.path.insert(0, ROOT)
    import nvr_import
    nvr = nvr_import.load()
    m = mo.small(seed=16, num_attention_heads=8, num_key_value_heads=4, head_dim=64, hidden_size=256, intermediate_size=512)
    mc = nvr.ModelConfig(vocab_size=m.vocab_size, hidden_size=m.hidden_size, intermediate_size=m.intermediate_size,
                         num_hidden_layers=m.num_hidden_layers, num_attention_heads=m.num_attention_heads,
                         num_key_value_heads=m.num_key_value_heads, head_dim=m.head_dim, max_position_embeddings=m.max_position_embeddings,
                         rms_norm_eps=m.rms_norm_eps, rope_theta=m.rope_theta, tie_word_embeddings=m.tie_word_embeddings,
                         init_std=m.init_std, seed=m.seed)
    ecfg = dict(max_num_seqs=4, max_num_batched_tokens=256, max_model_len=128, kvcache_block_size=16, num_kvcache_blocks=24, skip_block_size_check=1)
    first = [oracle.fill_tokens(n, 5, i, m.vocab_size).tolist() for i, n in enumerate([9, 21])]
    second = [oracle.fill_tokens(n, 5, 10 + i, m.vocab_size).tolist() for i, n in enumerate([13, 6, 30])]
    sp = dict(temperature=0.0, max_tokens=8, ignore_eos=True)

    os.environ["NVR_P2P_TIMEOUT_MS"] = "150"
    try:
        group = nvr.LocalGroup(2)
        engines = []
        for r in range(2):
            e = nvr.LLMEngine(nvr.Config(tensor_parallel_size=2, tensor_parallel_rank=r, async_decode=async_on, **ecfg), mc)
            group.attach(e.model_runner)
            engines.append(e)
    finally:
        os.environ.pop("NVR_P2P_TIMEOUT_MS", None)

    def both(fn):
        out, errs = [None, None], []

        def go(r):
            try:
                out[r] = fn(engines[r])
            except BaseException as ex:                                                 # noqa: BLE001
                errs.append((r, ex))
        ts = [threading.Thread(target=go, args=(r,)) for r in range(2)]
        for t in ts: t.start()
        for t in ts: t.join(120)
        assert not errs, errs
        return out

    for e in engines:
        nvr.lib().nvr_seq_reset_id_counter()
        for pr in first:
            e.add_request(pr, nvr.SamplingParams(**sp))
    a, b = both(lambda e: [e.step()["tokens"] for _ in range(3)])       # prefill + 2 decode steps together
    assert a == b
    # rank 1 stays away from the next step
    t0 = __import__("time").perf_counter()
    if async_on:                                                         # the step BOTH ranks had enqueued behind the third one: complete, and correct
        assert engines[0].ahead_launched() == 3 and engines[1].ahead_launched() == 3
        assert len(engines[0].step()["tokens"]) == 2
    with pytest.raises(nvr.NvrError) as ei:
        engines[0].step()
    assert ei.value.code == -9, ei.value                                 # NVR_ERR_RCCL
    assert "did not arrive" in str(ei.value)
    assert __import__("time").perf_counter() - t0 < 30.0                 # bounded: a handful of collectives x 150 ms
    assert engines[0].is_finished()                                      # the failing rank aborted its batch
    # recovery, driven by the caller's control plane
    engines[1].abort_last_batch()
    assert engines[1].is_finished()
    for e in engines:
        e.take_finished()
        e.model_runner.p2p_reset()

    def serve(e):
        for pr in second:
            e.add_request(pr, nvr.SamplingParams(**sp))
        recs = []
        while not e.is_finished():
            recs.append(e.step()["tokens"])
        return recs
    nvr.lib().nvr_seq_reset_id_counter()
    a, b = both(serve)
    assert a == b and len(a) >= 8
    nvr.lib().nvr_seq_reset_id_counter()
    single = nvr.LLMEngine(nvr.Config(**ecfg), mc)
    ref = serve(single)
    assert len(ref) == len(a)
    # (a numerical near-tie between the sharded and the single-rank sums may send ONE sequence down another path)
    diverged = {col for ra, rr in zip(a, ref) for col, (x, y) in enumerate(zip(ra, rr)) if x != y}
    assert len(diverged) <= 1, (a, ref)


@pytest.mark.gpu
def test_failed_token_wait_without_a_step_in_flight_aborts_the_batch():
    """csrc/engine.cpp step_async, the tail that is NOT launching ahead (the step may be a sequence's last: max_tokens reached): when the
    stream-ordered sampling never delivers (the peer stayed away: bounded wait -> NVR_ERR_RCCL) the batch is aborted — blocks returned,
    nothing left in the running queue — exactly as the synchronous step and the launch-ahead branch do; a batch left scheduled would be
    rescheduled against peers whose epochs differ.  Also: nvr_engine_abort_last_batch after the step's finished sequences were taken
    AND destroyed by the caller (the natural order of an external control plane) touches no dead handle."""
    import oracle
    from oracle import model_oracle as mo
    sys.path.insert(0, ROOT)
    import nvr_import
    nvr = nvr_import.load()
    m = mo.small(seed=16, num_attention_heads=8, num_key_value_heads=4, head_dim=64, hidden_size=256, intermediate_size=512)
    mc = nvr.ModelConfig(vocab_size=m.vocab_size, hidden_size=m.hidden_size, intermediate_size=m.intermediate_size,
                         num_hidden_layers=m.num_hidden_layers, num_attention_heads=m.num_attention_heads,
                         num_key_value_heads=m.num_key_value_heads, head_dim=m.head_dim, max_position_embeddings=m.max_position_embeddings,
                         rms_norm_eps=m.rms_norm_eps, rope_theta=m.rope_theta, tie_word_embeddings=m.tie_word_embeddings,
                         init_std=m.init_std, seed=m.seed)
    ecfg = dict(max_num_seqs=4, max_num_batched_tokens=256, max_model_len=128, kvcache_block_size=16, num_kvcache_blocks=24,
                skip_block_size_check=1, async_decode=1)
    prompts = [oracle.fill_tokens(n, 5, i, m.vocab_size).tolist() for i, n in enumerate([9, 21])]
    os.environ["NVR_P2P_TIMEOUT_MS"] = "150"
    try:
        group = nvr.LocalGroup(2)
        engines = []
        for r in range(2):
            e = nvr.LLMEngine(nvr.Config(tensor_parallel_size=2, tensor_parallel_rank=r, **ecfg), mc)
            group.attach(e.model_runner)
            engines.append(e)
    finally:
        os.environ.pop("NVR_P2P_TIMEOUT_MS", None)
    import threading
    # Sequence 1 stops with the prefill's token: the prefill is not followed by a step launched ahead (can_launch_ahead: "this token
    # could be the last"), so the decode step after it is scheduled and executed by ITS OWN call; sequence 0 stops after that step's
    # token, so that step does not launch ahead either and ends in the plain sample_wait tail.
    sps = [dict(temperature=0.0, max_tokens=2, ignore_eos=True), dict(temperature=0.0, max_tokens=1, ignore_eos=True)]
    for e in engines:
        nvr.lib().nvr_seq_reset_id_counter()
        for pr, sp in zip(prompts, sps):
            e.add_request(pr, nvr.SamplingParams(**sp))
    out = [None, None]
    ts = [threading.Thread(target=lambda r=r: out.__setitem__(r, engines[r].step()["tokens"])) for r in range(2)]
    for t in ts: t.start()
    for t in ts: t.join(120)
    assert out[0] == out[1] and out[0] is not None                       # the prefill, together
    free_before = engines[0].scheduler.get_block_stats()["free_blocks"]
    with pytest.raises(nvr.NvrError) as ei:                              # rank 1 stays away from the decode step
        engines[0].step()
    assert ei.value.code == -9, ei.value                                 # NVR_ERR_RCCL
    assert engines[0].ahead_launched() == 0
    assert engines[0].is_finished(), "the batch stayed scheduled after the failed wait"
    st = engines[0].scheduler.get_block_stats()
    assert st["free_blocks"] > free_before and st["free_blocks"] == st["total_blocks"]
    # control plane on the failing rank, in the order take -> destroy -> abort
    fin = engines[0].take_finished()
    assert len(fin) == 2
    del fin
    import gc; gc.collect()
    engines[0].abort_last_batch()                                        # nothing of that batch is live: a no-op that touches no handle
    engines[1].abort_last_batch()
    assert engines[1].is_finished()


@pytest.mark.gpu
@pytest.mark.parametrize("tp,hidden", [(2, 256), (4, 256), (2, 4096)])       # hidden 4096: the K-chunked fused LM head of 8B-class models
def test_tensor_parallel_launch_ahead_is_transparent(tp, hidden):
    """Row g: launch-ahead (nvr_config.async_decode) on tensor-parallel ranks.  The vocabulary-sharded greedy tokens are merged on the
    DEVICE (every rank's (max, arg-max) records all-gathered through the peer arenas, rank-ordered merge, ids straight into the next
    step's device-side input ids), so the next decode step is enqueued before the host has seen the current tokens — on every rank, from
    the same scheduler state.  Per step: the same batch, tokens and finished counts on every rank and the same as the SYNCHRONOUS
    tensor-parallel ranks (async_decode = 0; those are checked against the oracle's tensor-parallel engine by the tests above), with a
    request arriving in the middle (cancels the step in flight on every rank); and steps WERE launched ahead on every rank."""
    import threading
    import oracle
    from oracle import model_oracle as mo
    sys.path.insert(0, ROOT)
    import nvr_import
    nvr = nvr_import.load()
    m = mo.small(seed=26, num_attention_heads=8, num_key_value_heads=4, head_dim=64, hidden_size=hidden, intermediate_size=512)
    mc = nvr.ModelConfig(vocab_size=m.vocab_size, hidden_size=m.hidden_size, intermediate_size=m.intermediate_size,
                         num_hidden_layers=m.num_hidden_layers, num_attention_heads=m.num_attention_heads,
                         num_key_value_heads=m.num_key_value_heads, head_dim=m.head_dim, max_position_embeddings=m.max_position_embeddings,
                         rms_norm_eps=m.rms_norm_eps, rope_theta=m.rope_theta, tie_word_embeddings=m.tie_word_embeddings,
                         init_std=m.init_std, seed=m.seed)
    ecfg = dict(max_num_seqs=4, max_num_batched_tokens=256, max_model_len=160, kvcache_block_size=64, num_kvcache_blocks=16, skip_block_size_check=1)
    reqs = [(oracle.fill_tokens(n, 5, i, m.vocab_size).tolist(), mt) for i, (n, mt) in enumerate([(9, 30), (40, 22), (17, 30)])]
    late = (oracle.fill_tokens(11, 5, 9, m.vocab_size).tolist(), 12)

    def run(async_on):
        group = nvr.LocalGroup(tp)
        engines = []
        for r in range(tp):
            e = nvr.LLMEngine(nvr.Config(tensor_parallel_size=tp, tensor_parallel_rank=r, async_decode=async_on, **ecfg), mc)
            group.attach(e.model_runner)
            nvr.lib().nvr_seq_reset_id_counter()
            for pr, mt in reqs:
                e.add_request(pr, nvr.SamplingParams(temperature=0.0, max_tokens=mt, ignore_eos=True))
            engines.append(e)
        traces, errors = [[] for _ in range(tp)], []

        def drive(r):
            try:
                e, steps = engines[r], 0
                while not e.is_finished():
                    if steps == 9:
                        e.add_request(late[0], nvr.SamplingParams(temperature=0.0, max_tokens=late[1], ignore_eos=True))
                    rec = e.step()
                    traces[r].append((rec["is_prefill"], len(rec["seq_ids"]), tuple(rec["tokens"]), rec["num_finished"]))   # (ids of the late request depend on thread timing: one global counter in this process)
                    steps += 1
            except BaseException as ex:                                                 # noqa: BLE001
                errors.append((r, ex))
        threads = [threading.Thread(target=drive, args=(r,)) for r in range(tp)]
        for t in threads: t.start()
        for t in threads: t.join(300)
        assert not errors, errors
        assert all(tr == traces[0] for tr in traces), "ranks disagree"
        return traces[0], [e.ahead_launched() for e in engines]
    ta, launched = run(1)
    ts, none = run(0)
    assert ta == ts and len(ta) > 25
    assert all(n >= 10 for n in launched) and len(set(launched)) == 1 and not any(none), (launched, none)


_IPC_WORKER = r"""
import os, sys, json
sys.path.insert(0, os.environ["NVR_ROOT"])
import numpy as np
import nvr_import
nvr = nvr_import.load()
import oracle
from oracle import model_oracle as mo
g = nvr_import.load_ctrl().SocketGroup(timeout=120.0)      # control plane: the TCP rendezvous on MASTER_ADDR / MASTER_PORT (one ROCm stack per process)
rank, world = g.rank, g.world
dtype = os.environ.get("NVR_TEST_DTYPE", "float16")
scenario = os.environ.get("NVR_TEST_SCENARIO", "decode")

def attach(eng):
    gathered = g.all_gather((eng.model_runner.p2p_export(), 0))            # hipIpc handle of my arena; every rank sits on device 0
    eng.model_runner.p2p_attach([x[0] for x in gathered], [x[1] for x in gathered])
    g.barrier()
    eng.model_runner.comm_selftest()
    g.barrier()

def model(**kw):
    m = mo.small(**kw)
    return m, nvr.ModelConfig(vocab_size=m.vocab_size, hidden_size=m.hidden_size, intermediate_size=m.intermediate_size, num_hidden_layers=m.num_hidden_layers,
                              num_attention_heads=m.num_attention_heads, num_key_value_heads=m.num_key_value_heads, head_dim=m.head_dim,
                              max_position_embeddings=2048, rms_norm_eps=m.rms_norm_eps, rope_theta=m.rope_theta,
                              tie_word_embeddings=m.tie_word_embeddings, init_std=m.init_std, seed=m.seed)

if scenario == "decode":
    m, mc = model(seed=31, num_attention_heads=16, num_key_value_heads=8, head_dim=64, hidden_size=256, intermediate_size=512)
    eng = nvr.LLMEngine(nvr.Config(skip_block_size_check=1, max_num_seqs=4, max_num_batched_tokens=256, max_model_len=160, kvcache_block_size=64, num_kvcache_blocks=16,
                                   tensor_parallel_size=world, tensor_parallel_rank=rank, device_ordinal=0, async_decode=1, dtype=dtype), mc)
    attach(eng)
    for i, (n, mt) in enumerate([(9, 30), (40, 22), (17, 30)]):
        eng.add_request(oracle.fill_tokens(n, 5, i, m.vocab_size).tolist(), nvr.SamplingParams(temperature=0.0, max_tokens=mt, ignore_eos=True))
    trace = []
    while not eng.is_finished():
        rec = eng.step()
        trace.append([bool(rec["is_prefill"]), len(rec["seq_ids"]), list(rec["tokens"])])
    out = g.all_gather((trace, eng.ahead_launched()))
else:
    # prefill exchange on a second stream (row g): overlapped == serial, bit for bit, across process boundaries
    m, mc = model(seed=26, num_attention_heads=32, num_key_value_heads=8, head_dim=64, hidden_size=1024, intermediate_size=2048)
    res = []
    for overlap in (1, 0, 2):
        nvr.lib().nvr_seq_reset_id_counter()
        eng = nvr.LLMEngine(nvr.Config(skip_block_size_check=1, max_num_seqs=8, max_num_batched_tokens=8704, max_model_len=2048, kvcache_block_size=64,
                                       num_kvcache_blocks=160, tensor_parallel_size=world, tensor_parallel_rank=rank, device_ordinal=0, enforce_eager=1, dtype=dtype), mc)
        attach(eng)
        eng.model_runner.set_tp_prefill_overlap(overlap)
        for i, n in enumerate([1700, 1513, 1300, 1900, 1257, 833]):
            eng.add_request(oracle.fill_tokens(n, 5, i, m.vocab_size).tolist(), nvr.SamplingParams(temperature=0.0, max_tokens=3, ignore_eos=True))
        steps, chunks = [], 0
        while not eng.is_finished():
            rec = eng.step()
            if rec["is_prefill"]: chunks = eng.model_runner.last_overlap_chunks()
            steps.append((list(rec["tokens"]), eng.model_runner.logits(rec["num_seqs"]).copy()))
        res.append((steps, chunks))
        g.barrier()
        del eng
    same = all(ta == tb and np.array_equal(la, lb) for (ta, la), (tb, lb) in zip(res[0][0], res[1][0]))
    same2 = all(ta == tb and np.array_equal(la, lb) for (ta, la), (tb, lb) in zip(res[2][0], res[1][0]))
    out = g.all_gather(([t for t, _ in res[0][0]], [res[0][1], res[1][1], bool(same), len(res[0][0]), res[2][1], bool(same2)]))
if rank == 0:
    json.dump(out, open(os.environ["NVR_OUT"], "w"))
g.barrier()
g.close()
"""


def _run_ipc_workers(tmp_path, world, dtype, scenario):
    import json
    script = tmp_path / "ipc_worker.py"
    script.write_text(_IPC_WORKER)
    outp = tmp_path / "ipc.json"
    env = dict(os.environ, NVR_ROOT=ROOT, NVR_OUT=str(outp), NVR_TEST_DTYPE=dtype, NVR_TEST_SCENARIO=scenario, OMP_NUM_THREADS="2")
    port = 29700 + os.getpid() % 2000
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                    "--master-port", str(port), str(script)], check=True, env=env, timeout=420, cwd=ROOT)
    return json.load(open(outp))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,world", [("float16", 2), ("bfloat16", 2), ("float16", 4), ("float16", 8)])
def test_processes_sharing_the_gpu_exchange_over_hipipc(tmp_path, dtype, world):
    """The DEPLOYMENT form of tensor parallel — one PROCESS per rank, arenas exported with hipIpcGetMemHandle and mapped with
    hipIpcOpenMemHandle (nvr_runner_p2p_export / _attach), control plane over the TCP rendezvous of nano-vllm-rs_amd/ctrl.py — on the one GPU a
    test box has: 2, 4 and 8 ranks on device 0.  The one-shot collectives, the device-side arg-max merge and launch-ahead then run across
    process boundaries; all ranks must report the same per-step batches and tokens, steps must have been launched ahead, and the token
    streams must be bit-identical to the in-process group's (same kernels, same rank-ordered sums: only the way the arenas were mapped differs)."""
    import threading
    sys.path.insert(0, ROOT)
    import nvr_import
    import oracle
    from oracle import model_oracle as mo
    nvr = nvr_import.load()
    got = _run_ipc_workers(tmp_path, world, dtype, "decode")
    t0, a0 = got[0]
    assert len(got) == world and all(t == t0 and a == a0 for t, a in got) and len(t0) > 25, "the processes disagree"
    assert a0 >= 10, a0
    # the same ranks as runners of THIS process
    m = mo.small(seed=31, num_attention_heads=16, num_key_value_heads=8, head_dim=64, hidden_size=256, intermediate_size=512)
    mc = nvr.ModelConfig(vocab_size=m.vocab_size, hidden_size=m.hidden_size, intermediate_size=m.intermediate_size, num_hidden_layers=m.num_hidden_layers,
                         num_attention_heads=m.num_attention_heads, num_key_value_heads=m.num_key_value_heads, head_dim=m.head_dim,
                         max_position_embeddings=2048, rms_norm_eps=m.rms_norm_eps, rope_theta=m.rope_theta,
                         tie_word_embeddings=m.tie_word_embeddings, init_std=m.init_std, seed=m.seed)
    group = nvr.LocalGroup(world)
    engines, traces, errors = [], [[] for _ in range(world)], []
    for r in range(world):
        e = nvr.LLMEngine(nvr.Config(skip_block_size_check=1, max_num_seqs=4, max_num_batched_tokens=256, max_model_len=160, kvcache_block_size=64,
                                     num_kvcache_blocks=16, tensor_parallel_size=world, tensor_parallel_rank=r, async_decode=1, dtype=dtype), mc)
        group.attach(e.model_runner)
        nvr.lib().nvr_seq_reset_id_counter()
        for i, (n, mt) in enumerate([(9, 30), (40, 22), (17, 30)]):
            e.add_request(oracle.fill_tokens(n, 5, i, m.vocab_size).tolist(), nvr.SamplingParams(temperature=0.0, max_tokens=mt, ignore_eos=True))
        engines.append(e)

    def drive(r):
        try:
            while not engines[r].is_finished():
                rec = engines[r].step()
                traces[r].append([bool(rec["is_prefill"]), len(rec["seq_ids"]), list(rec["tokens"])])
        except BaseException as ex:                                                     # noqa: BLE001
            errors.append((r, ex))
    ths = [threading.Thread(target=drive, args=(r,)) for r in range(world)]
    for t in ths: t.start()
    for t in ths: t.join(120)
    assert not errors, errors
    assert all(tr == t0 for tr in traces), "processes over hipIpc and in-process ranks must sample the same tokens"


@pytest.mark.gpu
def test_two_processes_overlap_the_prefill_exchange_bit_identically(tmp_path):
    """Row g across process boundaries: two rank processes (hipIpc arenas on one GPU) run a 8503-token prefill with the all-reduce of each token
    chunk on the second stream (4 chunks), serially, and as two micro-batches whose exchanges run under each other's compute: same tokens, same shard
    logits in every step, on both ranks."""
    got = _run_ipc_workers(tmp_path, 2, "float16", "overlap")
    assert len(got) == 2 and got[0][0] == got[1][0]
    for toks, (chunks_on, chunks_off, same, nsteps, chunks_two, same_two) in got:
        assert chunks_on == 4 and chunks_off == 0 and same and nsteps == 3, (chunks_on, chunks_off, same, nsteps)
        assert chunks_two == 2 and same_two, (chunks_two, same_two)          # two micro-batches of whole sequences: the same bits again
