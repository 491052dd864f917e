"""GPU parity tests (-m gpu) of the bfloat16 BUILD of the kernels (namespace nvr::kb; Config.dtype = "bfloat16", reference
src/config.rs:51,113-116), called through the stateless C ABI after nvr_ops_set_dtype("bfloat16"), against the CPU oracle with its results
rounded to bf16 (oracle.round_bf16, pinned against torch in tests/test_oracle_kat.py) at the fp16 build's rounding points.

Same bar as tests/test_kernels_gpu.py with the unit of the 16-bit type: copies, RoPE, residual adds and the weight generator are
BIT-EXACT (no-contraction f32 math on both sides, one rounding); GEMM / attention / norm outputs may differ by the rounding of f32
results that differ in summation order (<= 1-2 bf16 ulp = 2^-7 relative).  One test per kernel family and GEMM route."""
import ctypes as C

import numpy as np
import pytest

import nvr_import
import oracle

nvr = nvr_import.load()
pytestmark = pytest.mark.gpu
U16 = np.uint16
_KEEP = []


@pytest.fixture(autouse=True)
def _bf16_ops():
    assert nvr.device_count() >= 1
    nvr.check(nvr.lib().nvr_device_set(0))
    nvr.check(nvr.lib().nvr_ops_set_dtype(b"bfloat16"))
    assert nvr.lib().nvr_ops_dtype() == b"bfloat16"
    yield
    nvr.synchronize()
    _KEEP.clear()
    nvr.check(nvr.lib().nvr_ops_set_dtype(b"float16"))


def dev(a):
    b = nvr.DeviceBuffer.from_numpy(np.ascontiguousarray(a))
    _KEEP.append(b)
    return b


def b16(a):
    """bf16-representable f32 array + its bf16 bit patterns"""
    f = oracle.round_bf16(np.asarray(a, np.float32))
    return f, oracle.to_bf16_bits(f)


def out16(buf, shape):
    """device buffer of bf16 elements -> f32 values"""
    return (buf.to_numpy(shape, U16).astype(np.uint32) << 16).view(np.float32)


def assert_close_bf16(got, ref, ulps=2, atol=1e-3, what=""):
    got = np.asarray(got, np.float32); ref = np.asarray(ref, np.float32)
    tol = atol + ulps * np.abs(ref) * 2.0 ** -7
    bad = np.abs(got - ref) > tol
    assert not bad.any(), f"{what}: {bad.sum()} / {bad.size} elements off, max err {np.abs(got - ref).max()}"


def test_dtype_switch_is_validated_and_thread_local():
    assert nvr.lib().nvr_ops_set_dtype(b"int8") == -10 and nvr.lib().nvr_ops_dtype() == b"bfloat16"   # ("float32" selects the f32 ops since r04: tests/test_kernels_f32_gpu.py)
    import threading
    seen = []
    t = threading.Thread(target=lambda: seen.append(nvr.lib().nvr_ops_dtype()))
    t.start(); t.join()
    assert seen == [b"float16"]                      # another thread still has the default


def test_embedding_fill_weight_and_copies_exact():
    rng = np.random.default_rng(0)
    V, Hd, T = 1000, 256, 37
    E, Eb = b16(rng.standard_normal((V, Hd)))
    ids = rng.integers(0, V, T).astype(np.int64)
    d_out = nvr.DeviceBuffer(T * Hd * 2)
    nvr.check(nvr.lib().nvr_embedding(dev(ids).ptr, T, dev(Eb).ptr, Hd, d_out.ptr, None))
    assert np.array_equal(d_out.to_numpy((T, Hd), U16), Eb[ids])
    # synthetic weights: generated in f32, rounded ONCE to bf16 (the oracle's unrounded generator + round_bf16)
    rows, cols, gcols = 37, 96, 512
    key, sc = oracle.weight_key(7, 1234), oracle.weight_scale(0.02)
    d = nvr.DeviceBuffer(rows * cols * 2)
    nvr.check(nvr.lib().nvr_fill_weight(d.ptr, rows, cols, cols, gcols, 5, 64, key, sc, None))
    assert np.array_equal(out16(d, (rows, cols)), oracle.round_bf16(oracle.fill_weight(rows, cols, gcols, 5, 64, key, sc, False)))
    nvr.check(nvr.lib().nvr_fill_const(d.ptr, rows * cols, 1.0, None))
    assert (d.to_numpy((rows * cols,), U16) == 0x3F80).all()                 # bf16 1.0
    cu = np.asarray([0, 5, 6, 37], np.int32)
    h, hb = b16(rng.standard_normal((37, 64)))
    d_l = nvr.DeviceBuffer(3 * 64 * 2)
    nvr.check(nvr.lib().nvr_select_last_tokens(dev(hb).ptr, dev(cu).ptr, 3, 64, d_l.ptr, None))
    assert np.array_equal(d_l.to_numpy((3, 64), U16), hb[[4, 5, 36]])


@pytest.mark.parametrize("T,Hd", [(1, 64), (5, 1024), (33, 4096), (32, 1024)])
def test_rmsnorm_and_add_rmsnorm(T, Hd):
    rng = np.random.default_rng(1)
    x, xb = b16(rng.standard_normal((T, Hd)) * 3)
    y, yb = b16(rng.standard_normal((T, Hd)))
    w, wb = b16(1 + 0.1 * rng.standard_normal(Hd))
    d_out = nvr.DeviceBuffer(T * Hd * 2)
    nvr.check(nvr.lib().nvr_rmsnorm(dev(xb).ptr, dev(wb).ptr, 1e-6, T, Hd, d_out.ptr, None))
    assert_close_bf16(out16(d_out, (T, Hd)), oracle.round_bf16(oracle.rmsnorm(x, w, 1e-6)), ulps=1, atol=1e-6, what="rmsnorm")
    d_h = dev(xb)
    nvr.check(nvr.lib().nvr_add_rmsnorm(d_h.ptr, dev(yb).ptr, dev(wb).ptr, 1e-6, T, Hd, d_out.ptr, None))
    hn = oracle.round_bf16(oracle.add(x, y, round16=False))
    assert np.array_equal(out16(d_h, (T, Hd)), hn), "residual add must be bit-exact"
    assert_close_bf16(out16(d_out, (T, Hd)), oracle.round_bf16(oracle.rmsnorm(hn, w, 1e-6)), ulps=1, atol=1e-6, what="add_rmsnorm")


@pytest.mark.parametrize("T,K,N,f32", [(1, 64, 64, False), (32, 1024, 4096, False), (32, 2048, 1024, False), (16, 1024, 6144, False),
                                       (4, 256, 1024, True), (100, 512, 512, False), (300, 2048, 1024, False), (1000, 3072, 1024, False),
                                       (513, 192, 768, False), (32, 4096, 6144, False)])
def test_linear_routes(T, K, N, f32):
    """weight-streaming (skinny / stream), 128x128 LDS-tiled and 256x256 MFMA GEMMs with bf16 operands (v_mfma_f32_16x16x32_bf16)"""
    rng = np.random.default_rng(3)
    x, xb = b16(rng.standard_normal((T, K)))
    W, Wb = b16(rng.standard_normal((N, K)) * 0.05)
    d_y = nvr.DeviceBuffer(T * N * (4 if f32 else 2))
    nvr.check(nvr.lib().nvr_linear(dev(xb).ptr, K, dev(Wb).ptr, T, K, N, d_y.ptr, int(f32), None))
    ref = oracle.linear(x, W)
    if f32:
        np.testing.assert_allclose(d_y.to_numpy((T, N), np.float32), ref, rtol=2e-5, atol=2e-4)
    else:
        assert_close_bf16(out16(d_y, (T, N)), oracle.round_bf16(ref), ulps=1, atol=1e-4, what="linear")


@pytest.mark.parametrize("T,K,H,KVH,D", [(32, 1024, 16, 8, 128), (7, 256, 4, 2, 64), (130, 1024, 16, 8, 128), (600, 1024, 16, 8, 128),
                                         (32, 4096, 32, 8, 128)])
def test_qkv_rope_store_fused_and_unfused(T, K, H, KVH, D):
    """rope_store_kv is bit-exact with the oracle; the fused qkv GEMM + RoPE + cache store equals linear followed by rope_store_kv bit
    for bit (decode, tiled, 256^2 and large-weight streaming routes)."""
    rng = np.random.default_rng(21)
    NB, bs, max_pos = max(24, T // 16 + 2), 16, 300
    QKV = (H + 2 * KVH) * D
    x, xb = b16(rng.standard_normal((T, K)))
    W, Wb = b16(rng.standard_normal((QKV, K)) * 0.05)
    pos = rng.integers(0, max_pos, T).astype(np.int64)
    slots = rng.permutation(NB * bs)[:T].astype(np.int32)
    slots[T // 2] = -1
    cos, sin = oracle.rope_table(D, max_pos, 1e6)
    d_qkv, d_qkv2 = nvr.DeviceBuffer(T * QKV * 2), nvr.DeviceBuffer(T * QKV * 2)
    caches = [nvr.DeviceBuffer(NB * bs * KVH * D * 2) for _ in range(4)]
    for c in caches: c.zero()
    nvr.check(nvr.lib().nvr_linear_qkv_rope_store(dev(xb).ptr, K, dev(Wb).ptr, T, K, H, KVH, D, dev(pos).ptr, dev(slots).ptr,
                                                  dev(cos).ptr, dev(sin).ptr, d_qkv.ptr, caches[0].ptr, caches[1].ptr, None))
    nvr.check(nvr.lib().nvr_linear(dev(xb).ptr, K, dev(Wb).ptr, T, K, QKV, d_qkv2.ptr, 0, None))
    lin = out16(d_qkv2, (T, QKV)).copy()                                      # the product's own GEMM output (bf16)
    nvr.check(nvr.lib().nvr_rope_store_kv(d_qkv2.ptr, dev(pos).ptr, dev(slots).ptr, T, H, KVH, D, dev(cos).ptr, dev(sin).ptr,
                                          caches[2].ptr, caches[3].ptr, None))
    assert np.array_equal(d_qkv.to_numpy((T, QKV), U16), d_qkv2.to_numpy((T, QKV), U16)), "fused qkv+rope differs from linear + rope_store_kv"
    for a, b in ((0, 2), (1, 3)):
        assert np.array_equal(caches[a].to_numpy((NB * bs, KVH * D), U16), caches[b].to_numpy((NB * bs, KVH * D), U16))
    # RoPE + store of the product's GEMM output against the oracle: bit-exact
    q = oracle.round_bf16(oracle.rope_apply(lin[:, :H * D].reshape(T, H, D), pos, cos, sin)).reshape(T, H * D)
    kk = oracle.round_bf16(oracle.rope_apply(lin[:, H * D:(H + KVH) * D].reshape(T, KVH, D), pos, cos, sin))
    vv = np.ascontiguousarray(lin[:, (H + KVH) * D:].reshape(T, KVH, D))
    got = out16(d_qkv, (T, QKV))
    assert np.array_equal(got[:, :H * D], q)
    kc, vc = np.zeros((NB, bs, KVH, D), np.float32), np.zeros((NB, bs, KVH, D), np.float32)
    oracle.kv_store(kk, vv, slots, kc, vc)
    assert np.array_equal(out16(caches[0], kc.shape), kc) and np.array_equal(out16(caches[1], vc.shape), vc)
    # and the GEMM itself against the oracle
    assert_close_bf16(lin, oracle.round_bf16(oracle.linear(x, W)), ulps=1, atol=1e-4, what="qkv GEMM")


@pytest.mark.parametrize("T,K,I", [(32, 1024, 3072), (5, 256, 64), (200, 1024, 3072), (600, 512, 1024), (32, 4096, 12288)])
def test_silu_mul_fused_and_plain(T, K, I):
    rng = np.random.default_rng(10)
    x, xb = b16(rng.standard_normal((T, K)))
    W, Wb = b16(rng.standard_normal((2 * I, K)) * 0.05)
    d_act, d_gu, d_act2 = nvr.DeviceBuffer(T * I * 2), nvr.DeviceBuffer(T * 2 * I * 2), nvr.DeviceBuffer(T * I * 2)
    nvr.check(nvr.lib().nvr_linear_silu_mul(dev(xb).ptr, K, dev(Wb).ptr, T, K, I, d_act.ptr, None))
    nvr.check(nvr.lib().nvr_linear(dev(xb).ptr, K, dev(Wb).ptr, T, K, 2 * I, d_gu.ptr, 0, None))
    nvr.check(nvr.lib().nvr_silu_and_mul(d_gu.ptr, T, I, d_act2.ptr, None))
    if K < 2048:        # same GEMM kernel family on both sides: bit-identical.  (Large weights: the fused kernel is the streaming kernel with
        #                 its own k split — another f32 summation order — and is held to the oracle below.)
        assert np.array_equal(d_act.to_numpy((T, I), U16), d_act2.to_numpy((T, I), U16)), "fused gate_up+SiLU differs from linear + silu_and_mul"
    ref = oracle.round_bf16(oracle.silu_and_mul(oracle.round_bf16(oracle.linear(x, W))))
    # a 1-ulp difference of a gate / up value passes through silu(g) * u: 3 bf16 ulp of an O(1) product
    assert_close_bf16(out16(d_act, (T, I)), ref, ulps=3, atol=2e-3, what="fused gate_up + SiLU vs oracle")
    gu = out16(d_gu, (T, 2 * I))
    assert_close_bf16(out16(d_act2, (T, I)), oracle.round_bf16(oracle.silu_and_mul(gu)), ulps=1, atol=1e-6, what="silu_and_mul")


@pytest.mark.parametrize("T,K,N,S", [(32, 2048, 1024, 4), (32, 3072, 1024, 4), (40, 512, 256, 4), (300, 3072, 1024, 4)])
def test_splitk_slab_norm_and_residual_gemm(T, K, N, S):
    rng = np.random.default_rng(22)
    x, xb = b16(rng.standard_normal((T, K)))
    W, Wb = b16(rng.standard_normal((N, K)) * 0.05)
    h, hb = b16(rng.standard_normal((T, N)))
    w, wb = b16(1 + 0.1 * rng.standard_normal(N))
    d_slabs = nvr.DeviceBuffer(S * T * N * 4)
    nvr.check(nvr.lib().nvr_linear_splitk(dev(xb).ptr, K, dev(Wb).ptr, T, K, N, S, d_slabs.ptr, None))
    slabs = d_slabs.to_numpy((S, T, N), np.float32)
    np.testing.assert_allclose(slabs.sum(0), oracle.linear(x, W), rtol=2e-5, atol=3e-4)
    d_h, d_out = dev(hb), nvr.DeviceBuffer(T * N * 2)
    nvr.check(nvr.lib().nvr_add_rmsnorm_slabs(d_h.ptr, d_slabs.ptr, S, dev(wb).ptr, 1e-6, T, N, d_out.ptr, None))
    y = slabs[0].copy()
    for z in range(1, S):
        y = y + slabs[z]
    hn = oracle.round_bf16(oracle.add(h, oracle.round_bf16(y), round16=False))
    assert np.array_equal(out16(d_h, (T, N)), hn)
    assert_close_bf16(out16(d_out, (T, N)), oracle.round_bf16(oracle.rmsnorm(hn, w, 1e-6)), ulps=1, atol=1e-6)


def _paged_case(rng, B, KVH, D, bs, ctx_lens, NB):
    max_blocks = max((c + bs - 1) // bs for c in ctx_lens) + 1
    kc, kcb = b16(rng.standard_normal((NB, bs, KVH, D)))
    vc, vcb = b16(rng.standard_normal((NB, bs, KVH, D)))
    bt = -np.ones((B, max_blocks), np.int32)
    perm = rng.permutation(NB); o = 0
    for b, c in enumerate(ctx_lens):
        nb = (c + bs - 1) // bs
        bt[b, :nb] = perm[o:o + nb]; o += nb
    return kc, kcb, vc, vcb, bt, max_blocks


@pytest.mark.parametrize("B,H,KVH,D,bs,ctxs", [
    (3, 4, 2, 64, 16, [1, 17, 40]), (4, 16, 8, 128, 256, [1, 255, 256, 700]), (2, 32, 8, 128, 256, [513, 1024]),
    (32, 16, 8, 128, 256, [1024] * 32), (1, 16, 8, 128, 256, [3000]), (3, 8, 4, 64, 8, [700, 64, 1])])
def test_paged_attn_decode(B, H, KVH, D, bs, ctxs):
    rng = np.random.default_rng(6)
    NB = sum((c + bs - 1) // bs for c in ctxs) + 3
    kc, kcb, vc, vcb, bt, max_blocks = _paged_case(rng, B, KVH, D, bs, ctxs, NB)
    q, qb = b16(rng.standard_normal((B, H, D)))
    ctx = np.asarray(ctxs, np.int32)
    scale = float(np.float32(1.0) / np.sqrt(np.float32(D)))
    meta = nvr.AttnMetaC()
    d_ctx, d_bt = dev(ctx), dev(bt)
    meta.is_prefill, meta.context_lens, meta.block_tables = 0, d_ctx.ptr, d_bt.ptr
    meta.max_blocks, meta.batch, meta.max_context_len = max_blocks, B, int(max(ctxs))
    ws = nvr.DeviceBuffer(nvr.lib().nvr_paged_attn_workspace_bytes(B, H, D, int(max(ctxs))))
    d_out = nvr.DeviceBuffer(B * H * D * 2)
    nvr.check(nvr.lib().nvr_paged_attn_decode(dev(qb).ptr, H * D, dev(kcb).ptr, dev(vcb).ptr, C.byref(meta), H, KVH, D, bs,
                                              scale, d_out.ptr, ws.ptr, None))
    ref = oracle.round_bf16(oracle.attn_decode(q, kc, vc, bt, ctx, scale))
    assert_close_bf16(out16(d_out, (B, H, D)), ref, ulps=2, atol=1e-3, what="paged decode attention (bf16)")


@pytest.mark.parametrize("H,KVH,D,lens", [(4, 2, 64, [1, 5, 33]), (16, 8, 128, [70, 129]), (8, 2, 128, [100, 3, 64]), (4, 2, 128, [1500, 129, 64, 1]),
                                          (8, 2, 64, [260])])
def test_attn_prefill_varlen(H, KVH, D, lens):
    rng = np.random.default_rng(9)
    T = sum(lens)
    QKV = (H + 2 * KVH) * D
    qkv, qkvb = b16(rng.standard_normal((T, QKV)))
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    scale = float(np.float32(1.0) / np.sqrt(np.float32(D)))
    d_qkv, d_cu, d_out = dev(qkvb), dev(cu), nvr.DeviceBuffer(T * H * D * 2)
    meta = nvr.AttnMetaC()
    meta.is_prefill, meta.cu_seqlens_q, meta.cu_seqlens_k = 1, d_cu.ptr, d_cu.ptr
    meta.max_seqlen_q = meta.max_seqlen_k = int(max(lens)); meta.batch = len(lens)
    nvr.check(nvr.lib().nvr_attn_prefill_varlen(d_qkv.ptr, d_qkv.ptr + H * D * 2, d_qkv.ptr + (H + KVH) * D * 2, QKV, C.byref(meta),
                                                T, H, KVH, D, scale, d_out.ptr, None))
    q = np.ascontiguousarray(qkv[:, :H * D].reshape(T, H, D))
    k = np.ascontiguousarray(qkv[:, H * D:(H + KVH) * D].reshape(T, KVH, D))
    v = np.ascontiguousarray(qkv[:, (H + KVH) * D:].reshape(T, KVH, D))
    ref = oracle.round_bf16(oracle.attn_prefill_varlen(q, k, v, cu, scale))
    # MFMA path: P is rounded to bf16 before P.V (relative 2^-8 per term): 2 bf16 ulp + 1.6e-2 absolute on O(1) outputs (8 x the fp16 build's 2e-3)
    assert_close_bf16(out16(d_out, (T, H, D)), ref, ulps=2, atol=1.6e-2, what="varlen prefill attention (bf16)")


@pytest.mark.parametrize("T,K,N", [(32, 1024, 151936), (7, 1024, 18992), (16, 2048, 4096), (128, 1024, 151936), (33, 4096, 2064), (32, 4096, 151936),
                                   (5, 6144, 2064)])
def test_lm_head_logits_and_argmax_partials(T, K, N):
    rng = np.random.default_rng(31)
    x, xb = b16(rng.standard_normal((T, K)))
    W, Wb = b16(rng.standard_normal((N, K)) * 0.05)
    W[N // 3] = W[5]; Wb[N // 3] = Wb[5]
    x[0], xb[0] = b16(W[5] * 8)
    d_y = nvr.DeviceBuffer(T * N * 4)
    P = 2048
    d_pv, d_pi = nvr.DeviceBuffer(P * T * 4), nvr.DeviceBuffer(P * T * 4)
    nparts = C.c_int32(0)
    nvr.check(nvr.lib().nvr_lm_head(dev(xb).ptr, K, dev(Wb).ptr, T, K, N, d_y.ptr, d_pv.ptr, d_pi.ptr, C.byref(nparts), None))
    y = d_y.to_numpy((T, N), np.float32)
    np.testing.assert_allclose(y, oracle.linear(x, W), rtol=2e-5, atol=3e-4)
    d_tok, d_val = nvr.DeviceBuffer(T * 8), nvr.DeviceBuffer(T * 4)
    nvr.check(nvr.lib().nvr_argmax_partials(d_pv.ptr, d_pi.ptr, nparts.value, T, d_tok.ptr, d_val.ptr, 0, None))
    want = np.asarray([oracle.argmax(y[t]) for t in range(T)])
    assert d_tok.to_numpy((T,), np.int64).tolist() == want.tolist() and want[0] == 5
    assert np.array_equal(d_val.to_numpy((T,), np.float32), y[np.arange(T), want])


def test_add_bias_bf16():
    """A-30 in bfloat16: y <- bf16(y + b), exact against the oracle's rounding."""
    rng = np.random.default_rng(41)
    T, N = 33, 1024
    y, yb = b16(rng.standard_normal((T, N)) * 3)
    b, bb = b16(rng.standard_normal(N))
    d_y = dev(yb)
    nvr.check(nvr.lib().nvr_add_bias(d_y.ptr, dev(bb).ptr, T, N, None))
    assert np.array_equal(out16(d_y, (T, N)), oracle.round_bf16(y + b[None, :]))
