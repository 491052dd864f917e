"""The stateless ops under nvr_ops_set_dtype("float32") (kernels/f32_path.hip: the reference-precision path) against the oracle's f32 ops:
same arithmetic up to the order of f32 sums (1e-5 relative), bit-exact where no sum is involved (embedding, RoPE products, cache rows,
SiluAndMul up to libm's expf)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nvr_import  # noqa: E402
import oracle  # noqa: E402

nvr = nvr_import.load()
pytestmark = pytest.mark.gpu
F32 = np.float32
_KEEP = []


@pytest.fixture(autouse=True)
def _f32_ops():
    assert nvr.device_count() >= 1
    nvr.check(nvr.lib().nvr_device_set(0))
    nvr.check(nvr.lib().nvr_ops_set_dtype(b"float32"))
    assert nvr.lib().nvr_ops_dtype() == b"float32"
    yield
    nvr.synchronize()
    _KEEP.clear()
    nvr.check(nvr.lib().nvr_ops_set_dtype(b"float16"))


def dev(a):
    b = nvr.DeviceBuffer.from_numpy(np.ascontiguousarray(a))
    _KEEP.append(b)
    return b


def close(got, ref, rtol=2e-5, atol=2e-6, what=""):
    got = np.asarray(got, F32); ref = np.asarray(ref, F32)
    bad = np.abs(got - ref) > atol + rtol * np.abs(ref)
    assert not bad.any(), f"{what}: {bad.sum()} / {bad.size} off, max err {np.abs(got - ref).max()}"


def test_embedding_norms_and_select():
    rng = np.random.default_rng(1)
    V, Hd, T = 300, 256, 37
    E = rng.standard_normal((V, Hd)).astype(F32); ids = rng.integers(0, V, T).astype(np.int64)
    out = nvr.DeviceBuffer(T * Hd * 4)
    nvr.check(nvr.lib().nvr_embedding(dev(ids).ptr, T, dev(E).ptr, Hd, out.ptr, None))
    x = out.to_numpy((T, Hd), F32)
    assert np.array_equal(x, E[ids])
    w = (1 + 0.2 * rng.standard_normal(Hd)).astype(F32)
    o2 = nvr.DeviceBuffer(T * Hd * 4)
    nvr.check(nvr.lib().nvr_rmsnorm(out.ptr, dev(w).ptr, 1e-6, T, Hd, o2.ptr, None))
    close(o2.to_numpy((T, Hd), F32), oracle.rmsnorm(x, w, 1e-6), what="rmsnorm")
    y = rng.standard_normal((T, Hd)).astype(F32)
    d_h = dev(x.copy())
    nvr.check(nvr.lib().nvr_add_rmsnorm(d_h.ptr, dev(y).ptr, dev(w).ptr, 1e-6, T, Hd, o2.ptr, None))
    assert np.array_equal(d_h.to_numpy((T, Hd), F32), x + y)
    close(o2.to_numpy((T, Hd), F32), oracle.rmsnorm(x + y, w, 1e-6), what="add_rmsnorm")
    cu = np.asarray([0, 5, 6, 37], np.int32)
    o3 = nvr.DeviceBuffer(3 * Hd * 4)
    nvr.check(nvr.lib().nvr_select_last_tokens(out.ptr, dev(cu).ptr, 3, Hd, o3.ptr, None))
    assert np.array_equal(o3.to_numpy((3, Hd), F32), x[[4, 5, 36]])


@pytest.mark.parametrize("T,K,N", [(1, 1024, 4096), (7, 512, 48), (8, 96, 100), (9, 96, 100), (130, 1000, 200),
                                   # r06: K % 32 == 0 and more than 8 rows -> the matrix-core kernel (64 x 64 tiles; ragged rows and columns)
                                   (130, 1024, 200), (64, 64, 64), (300, 2048, 6144), (33, 3072, 1028)])
def test_linear_gemv_and_tiles(T, K, N):
    rng = np.random.default_rng(T + K)
    x = rng.standard_normal((T, K)).astype(F32); W = (rng.standard_normal((N, K)) * 0.1).astype(F32)
    y = nvr.DeviceBuffer(T * N * 4)
    nvr.check(nvr.lib().nvr_linear(dev(x).ptr, K, dev(W).ptr, T, K, N, y.ptr, 1, None))
    # (f32 sums of K products of magnitude ~0.1 against exact sums: the rounding noise grows with sqrt(K))
    close(y.to_numpy((T, N), F32), x.astype(np.float64) @ W.astype(np.float64).T, rtol=2e-5, atol=2e-5 * max(1.0, (K / 1024) ** 0.5) * 2, what="linear")


def test_silu_and_mul():
    rng = np.random.default_rng(3)
    T, I = 11, 96
    x = (rng.standard_normal((T, 2 * I)) * 3).astype(F32)
    out = nvr.DeviceBuffer(T * I * 4)
    nvr.check(nvr.lib().nvr_silu_and_mul(dev(x).ptr, T, I, out.ptr, None))
    close(out.to_numpy((T, I), F32), oracle.silu_and_mul(x), rtol=1e-6, atol=1e-7, what="silu_and_mul")


@pytest.mark.parametrize("kind", ["silu", "gelu", "relu", "silu_and_mul", "gelu_and_mul"])
def test_activation_types(kind):
    """nvr_activation under nvr_ops_set_dtype("float32") (Activation::forward, activation.rs:147-159) against the oracle's f32 arithmetic"""
    rng = np.random.default_rng(4)
    T, cols = 11, 192
    x = (rng.standard_normal((T, cols)) * 3).astype(F32)
    k = oracle.ACTIVATION_TYPES[kind]
    co = cols // 2 if k >= 3 else cols
    out = nvr.DeviceBuffer(T * co * 4)
    nvr.check(nvr.lib().nvr_activation(k, dev(x).ptr, T, cols, out.ptr, None))
    close(out.to_numpy((T, co), F32), oracle.activation(kind, x), rtol=2e-6, atol=2e-7, what=kind)


@pytest.mark.parametrize("H,KVH,D", [(4, 2, 64), (8, 1, 128)])
def test_rope_store_and_attention(H, KVH, D):
    rng = np.random.default_rng(H * D)
    bs, lens = 16, [21, 3, 40]
    T = sum(lens); cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    qkv = rng.standard_normal((T, (H + 2 * KVH) * D)).astype(F32)
    pos = np.concatenate([np.arange(n) for n in lens]).astype(np.int64)
    nb = sum((n + bs - 1) // bs for n in lens) + 2
    tables, nxt = [], 1
    for n in lens:
        k = (n + bs - 1) // bs; tables.append(list(range(nxt, nxt + k))); nxt += k
    slots = np.concatenate([[tables[b][p // bs] * bs + p % bs for p in range(n)] for b, n in enumerate(lens)]).astype(np.int32)
    cos, sin = oracle.rope_table(D, 64, 10000.0)
    d_cos, d_sin = dev(cos), dev(sin)
    d_qkv = dev(qkv.copy()); kc = nvr.DeviceBuffer(nb * bs * KVH * D * 4); vc = nvr.DeviceBuffer(nb * bs * KVH * D * 4)
    nvr.check(nvr.lib().nvr_fill_const(kc.ptr, nb * bs * KVH * D, 0.0, None)); nvr.check(nvr.lib().nvr_fill_const(vc.ptr, nb * bs * KVH * D, 0.0, None))
    nvr.check(nvr.lib().nvr_rope_store_kv(d_qkv.ptr, dev(pos).ptr, dev(slots).ptr, T, H, KVH, D, d_cos.ptr, d_sin.ptr, kc.ptr, vc.ptr, None))
    got = d_qkv.to_numpy((T, (H + 2 * KVH) * D), F32)
    q = oracle.rope_apply(qkv[:, :H * D].reshape(T, H, D), pos, cos, sin)
    k = oracle.rope_apply(qkv[:, H * D:(H + KVH) * D].reshape(T, KVH, D), pos, cos, sin)
    v = qkv[:, (H + KVH) * D:].reshape(T, KVH, D)
    close(got[:, :H * D].reshape(T, H, D), q, rtol=1e-6, atol=1e-6, what="rope q")
    kcn = kc.to_numpy((nb * bs, KVH, D), F32); vcn = vc.to_numpy((nb * bs, KVH, D), F32)
    close(kcn[slots], k, rtol=1e-6, atol=1e-6, what="k cache rows"); assert np.array_equal(vcn[slots], v)
    scale = float(1 / np.sqrt(D))
    # varlen causal prefill over the step's own rows
    meta = nvr.AttnMetaC(); meta.is_prefill = 1; d_cu = dev(cu); meta.cu_seqlens_q = d_cu.ptr; meta.cu_seqlens_k = d_cu.ptr
    meta.max_seqlen_q = max(lens); meta.max_seqlen_k = max(lens); meta.batch = len(lens)
    out = nvr.DeviceBuffer(T * H * D * 4)
    QKV = (H + 2 * KVH) * D
    nvr.check(nvr.lib().nvr_attn_prefill_varlen(d_qkv.ptr, d_qkv.ptr + H * D * 4, d_qkv.ptr + (H + KVH) * D * 4, QKV, C_byref(meta), T, H, KVH, D, scale, out.ptr, None))
    ref = oracle.attn_prefill_varlen(got[:, :H * D].reshape(T, H, D), got[:, H * D:(H + KVH) * D].reshape(T, KVH, D), v, cu, scale)
    close(out.to_numpy((T, H, D), F32), ref, rtol=2e-5, atol=2e-6, what="varlen prefill attention")
    # paged decode: one query per sequence over its cached rows
    B = len(lens)
    qd = rng.standard_normal((B, H * D)).astype(F32)
    mb = max(len(t) for t in tables)
    bt = -np.ones((B, mb), np.int32)
    for b, t in enumerate(tables): bt[b, :len(t)] = t
    m2 = nvr.AttnMetaC(); d_ctx = dev(np.asarray(lens, np.int32)); d_bt = dev(bt)
    m2.context_lens = d_ctx.ptr; m2.block_tables = d_bt.ptr; m2.max_blocks = mb; m2.batch = B; m2.max_context_len = max(lens)
    o2 = nvr.DeviceBuffer(B * H * D * 4)
    nvr.check(nvr.lib().nvr_paged_attn_decode(dev(qd).ptr, H * D, kc.ptr, vc.ptr, C_byref(m2), H, KVH, D, bs, scale, o2.ptr, None, None))
    ref2 = oracle.attn_decode(qd.reshape(B, H, D), kcn.reshape(nb, bs, KVH, D), vcn.reshape(nb, bs, KVH, D), bt, np.asarray(lens, np.int32), scale)
    close(o2.to_numpy((B, H, D), F32), ref2, rtol=2e-5, atol=2e-6, what="paged decode attention")


def C_byref(x):
    import ctypes
    return ctypes.byref(x)


@pytest.mark.parametrize("T,H,KVH,D,K", [(1, 16, 8, 128, 1024), (3, 4, 2, 64, 256), (8, 8, 1, 128, 512)])
def test_decode_sized_fused_forms_equal_their_parts_bit_for_bit(T, H, KVH, D, K):
    """r05: the float32 path fuses what a decode-sized step (1..8 rows) launches back to back — the qkv projection with RoPE + KV store
    (linear.rs:354-356, rotary_embedding.rs:23-48, attention.rs:150-174) and the gate_up projection with SiluAndMul (linear.rs:437-439,
    activation.rs:46-63).  Same loads, same FMA chains, the rotation written with contraction off in both kernels: the fused launches give the BITS of
    their parts, and the parts are checked against float64 / the oracle."""
    rng = np.random.default_rng(T * 100 + D)
    N = (H + 2 * KVH) * D
    x = rng.standard_normal((T, K)).astype(F32); W = (rng.standard_normal((N, K)) * 0.05).astype(F32)
    pos = rng.integers(0, 60, T).astype(np.int64); slots = (np.arange(T) * 3 + 1).astype(np.int32); slots[-1] = -1 if T > 1 else slots[-1]
    cos, sin = oracle.rope_table(D, 64, 10000.0)
    d_x, d_W, d_pos, d_slots, d_cos, d_sin = dev(x), dev(W), dev(pos), dev(slots), dev(cos), dev(sin)
    nslots = int(T * 3 + 2)

    def caches():
        kc, vc = nvr.DeviceBuffer(nslots * KVH * D * 4), nvr.DeviceBuffer(nslots * KVH * D * 4)
        nvr.check(nvr.lib().nvr_fill_const(kc.ptr, nslots * KVH * D, 0.0, None)); nvr.check(nvr.lib().nvr_fill_const(vc.ptr, nslots * KVH * D, 0.0, None))
        return kc, vc
    ya, kca, vca = nvr.DeviceBuffer(T * N * 4), *caches()
    nvr.check(nvr.lib().nvr_linear(d_x.ptr, K, d_W.ptr, T, K, N, ya.ptr, 1, None))
    nvr.check(nvr.lib().nvr_rope_store_kv(ya.ptr, d_pos.ptr, d_slots.ptr, T, H, KVH, D, d_cos.ptr, d_sin.ptr, kca.ptr, vca.ptr, None))
    yb, kcb, vcb = nvr.DeviceBuffer(T * N * 4), *caches()
    nvr.check(nvr.lib().nvr_linear_qkv_rope_store(d_x.ptr, K, d_W.ptr, T, K, H, KVH, D, d_pos.ptr, d_slots.ptr, d_cos.ptr, d_sin.ptr, yb.ptr, kcb.ptr, vcb.ptr, None))
    a, b = ya.to_numpy((T, N), np.uint32), yb.to_numpy((T, N), np.uint32)
    assert np.array_equal(a, b), "qkv + RoPE: fused launch differs from gemv followed by rope_store"
    assert np.array_equal(kca.to_numpy((nslots, KVH * D), np.uint32), kcb.to_numpy((nslots, KVH * D), np.uint32))
    assert np.array_equal(vca.to_numpy((nslots, KVH * D), np.uint32), vcb.to_numpy((nslots, KVH * D), np.uint32))
    ref = (x.astype(np.float64) @ W.astype(np.float64).T).astype(F32)
    q = oracle.rope_apply(ref[:, :H * D].reshape(T, H, D), pos, cos, sin)
    close(yb.to_numpy((T, N), F32)[:, :H * D].reshape(T, H, D), q, rtol=3e-5, atol=3e-5, what="fused qkv + RoPE vs float64 GEMM + oracle rotation")
    kcn = kcb.to_numpy((nslots, KVH, D), F32)
    k = oracle.rope_apply(ref[:, H * D:(H + KVH) * D].reshape(T, KVH, D), pos, cos, sin)
    live = slots >= 0
    close(kcn[slots[live]], k[live], rtol=3e-5, atol=3e-5, what="cache rows of the fused launch")
    assert not kcn[np.setdiff1d(np.arange(nslots), slots[live])].any()                                  # nothing else written (the row with slot -1 included)
    # gate_up + SiluAndMul
    I = 3 * D
    Wg = (rng.standard_normal((2 * I, K)) * 0.05).astype(F32); d_Wg = dev(Wg)
    gu, act_a, act_b = nvr.DeviceBuffer(T * 2 * I * 4), nvr.DeviceBuffer(T * I * 4), nvr.DeviceBuffer(T * I * 4)
    nvr.check(nvr.lib().nvr_linear(d_x.ptr, K, d_Wg.ptr, T, K, 2 * I, gu.ptr, 1, None))
    nvr.check(nvr.lib().nvr_silu_and_mul(gu.ptr, T, I, act_a.ptr, None))
    nvr.check(nvr.lib().nvr_linear_silu_mul(d_x.ptr, K, d_Wg.ptr, T, K, I, act_b.ptr, None))
    assert np.array_equal(act_a.to_numpy((T, I), np.uint32), act_b.to_numpy((T, I), np.uint32)), "gate_up + SiluAndMul: fused launch differs from its parts"
    # steps that are not decode-sized keep the unfused parts
    assert nvr.lib().nvr_linear_silu_mul(d_x.ptr, K, d_Wg.ptr, 9, K, I, act_b.ptr, None) == -10
    assert nvr.lib().nvr_linear_qkv_rope_store(d_x.ptr, K, d_W.ptr, 9, K, H, KVH, D, d_pos.ptr, d_slots.ptr, d_cos.ptr, d_sin.ptr, yb.ptr, kcb.ptr, vcb.ptr, None) == -10


def test_fused_16_bit_ops_say_so_and_weights_are_unrounded():
    assert nvr.lib().nvr_linear_silu_mul(None, 0, None, 0, 0, 0, None, None) == -10
    out = nvr.DeviceBuffer(8 * 16 * 4)
    nvr.check(nvr.lib().nvr_fill_weight(out.ptr, 8, 16, 16, 16, 0, 0, 1234, 0.01, None))
    assert np.array_equal(out.to_numpy((8, 16), F32), oracle.fill_weight(8, 16, 16, 0, 0, 1234, 0.01, round16=False))
