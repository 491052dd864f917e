"""Child of tests/test_sanitizers.py: runs in an interpreter with libasan preloaded and NVO_ORACLE_LIB pointing at the
-fsanitize=address,undefined build of oracle/nvr_oracle.c.  Touches every exported op at small, ragged and empty sizes; any
out-of-bounds access, misaligned or overflowing arithmetic aborts the process (the parent asserts exit code 0)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle                                                            # noqa: E402
from oracle import engine_oracle as eo, model_oracle as mo             # noqa: E402

assert "asan" in (os.environ.get("NVO_ORACLE_LIB") or "")
rng = np.random.default_rng(0)
# hashes (all tail lengths of XXH64), weights, tokens
for n in list(range(0, 40)) + [255, 256, 257]:
    oracle.lib().nvo_xxh64(bytes(range(256)) * 2, n, 0)
    oracle.block_hash(list(range(n)), 12345 if n % 2 else None)
oracle.fill_weight(7, 33, 33, 3, 5, oracle.weight_key(1, 2), oracle.weight_scale(0.02), True)
oracle.fill_tokens(17, 1, 2, 1000)
# float ops: ragged shapes, T not a multiple of 4, K not a multiple of 8 / 32, single rows
for T, K, N in [(1, 8, 3), (3, 20, 7), (4, 64, 2), (5, 33, 5), (9, 256, 17), (70, 128, 33)]:
    x = rng.standard_normal((T, K)).astype(np.float32); W = rng.standard_normal((N, K)).astype(np.float32)
    y = oracle.linear(x, W)
    assert np.allclose(y, x @ W.T, rtol=1e-4, atol=1e-4)
    oracle.rmsnorm(x, np.ones(K, np.float32), 1e-6); oracle.round_f16(x); oracle.add(x, x, round16=True)
    if K % 2 == 0:
        oracle.silu_and_mul(x)
for D in (8, 64, 128):
    cos, sin = oracle.rope_table(D, 50, 1e4)
    H, KVH, bs, nb = 4, 2, 4, 12
    for lens in ([1], [5, 1, 9], [4, 4], [13]):
        T = sum(lens); cu = np.cumsum([0] + lens).astype(np.int32)
        q = rng.standard_normal((T, H, D)).astype(np.float32); k = rng.standard_normal((T, KVH, D)).astype(np.float32); v = rng.standard_normal((T, KVH, D)).astype(np.float32)
        pos = np.concatenate([np.arange(n) for n in lens]).astype(np.int64)
        q = oracle.rope_apply(q, pos, cos, sin)
        oracle.attn_prefill_varlen(q, k, v, cu, 0.125)
        kc = np.zeros((nb, bs, KVH, D), np.float32); vc = np.zeros_like(kc)
        tables, slots, used = [], [], 0
        for n in lens:
            nblk = (n + bs - 1) // bs
            tables.append(list(range(used, used + nblk))); used += nblk
            slots += [tables[-1][p // bs] * bs + p % bs for p in range(n)]
        oracle.kv_store(k, v, np.asarray(slots, np.int32), kc, vc)
        mb = max(len(t) for t in tables)
        bt = np.asarray([t + [-1] * (mb - len(t)) for t in tables], np.int32)
        oracle.attn_paged(q, cu, kc, vc, bt, np.asarray(lens, np.int32), 0.125)                       # chunk-style: nq = ctx
        oracle.attn_paged(q[cu[1:] - 1], np.arange(len(lens) + 1, dtype=np.int32), kc, vc, bt, np.asarray(lens, np.int32), 0.125)   # decode
# sampler: greedy, top-k incl. k > V, top-p incl. 1.0, ties
for V in (1, 2, 5, 1000):
    x = rng.standard_normal(V).astype(np.float32)
    oracle.sample(x, 0.0); oracle.sample(x, 0.7, V + 3, 0.9, 5); oracle.sample(x, 1.0, 1, 1.0, 6); oracle.top_k(x, max(1, V // 2)); oracle.top_p(x, 0.5)
    oracle.sample(np.zeros(V, np.float32), 1.0, 0, 0.5, 7)
# engine: tiny model through prefill, chunked prefill, decode, preemption
for chunked in (False, True):
    eo.reset_sequence_counter()
    e = mo.OracleEngine(mo.tiny(), eo.Config(max_num_seqs=4, max_num_batched_tokens=24, kvcache_block_size=4, num_kvcache_blocks=14,
                                             max_model_len=64, enable_chunked_prefill=chunked), fp16=True, max_pos=64)
    for i, n in enumerate([9, 21 if chunked else 20, 3, 17]):
        e.add_request(oracle.fill_tokens(n, 1, i, 256).tolist(), eo.SamplingParams(temperature=0.0 if i % 2 else 0.8, max_tokens=6, ignore_eos=True))
    steps = e.run(200)
    assert e.scheduler.is_finished() and len(steps) > 6
print("asan oracle smoke ok")
