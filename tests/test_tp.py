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
