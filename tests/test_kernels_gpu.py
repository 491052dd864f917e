"""GPU parity tests (-m gpu): every HIP kernel of the hot path, called through the C ABI with raw
device pointers, against the CPU oracle on the same seeded inputs.  fp16 storage on both sides
(oracle in fp16-faithful mode), f32 math inside an op.  Tolerances are stated per test: integer
outputs (token ids, cache contents of pure copies) are bit-exact; fp16 outputs may differ by
fp16 rounding of f32 results that differ in summation order (<= 1-2 fp16 ulp)."""
import ctypes as C
import os

import numpy as np
import pytest

import nvr_import
import oracle

nvr = nvr_import.load()
pytestmark = pytest.mark.gpu

F16 = np.float16


@pytest.fixture(scope="module", autouse=True)
def _device():
    assert nvr.device_count() >= 1, "no HIP device visible: the GPU tests must run on an MI355X"
    nvr.check(nvr.lib().nvr_device_set(0))
    yield
    nvr.synchronize()


_KEEP = []          # device buffers stay alive until the test ends (a temporary would be hipFree'd
                    # as soon as its .ptr has been read, before the asynchronous kernel runs)


def dev(a):
    b = nvr.DeviceBuffer.from_numpy(np.ascontiguousarray(a))
    _KEEP.append(b)
    return b


@pytest.fixture(autouse=True)
def _release_buffers():
    yield
    nvr.synchronize()
    _KEEP.clear()


def h16(a):
    """fp16-representable f32 array + its fp16 bits"""
    b = np.asarray(a, dtype=np.float32).astype(F16)
    return b.astype(np.float32), b


def assert_close_f16(got, ref, ulps=2, atol=1e-3, what=""):
    got = np.asarray(got, np.float32); ref = np.asarray(ref, np.float32)
    tol = atol + ulps * np.abs(ref) * 2.0 ** -10
    bad = np.abs(got - ref) > tol
    assert not bad.any(), f"{what}: {bad.sum()} / {bad.size} elements off, max err {np.abs(got - ref).max()}"


# ------------------------------------------------------------------------------------------- K1
def test_embedding_exact():
    rng = np.random.default_rng(0)
    V, Hd, T = 1000, 256, 37
    E, Eb = h16(rng.standard_normal((V, Hd)))
    ids = rng.integers(0, V, T).astype(np.int64)
    d_out = nvr.DeviceBuffer(T * Hd * 2)
    nvr.check(nvr.lib().nvr_embedding(dev(ids).ptr, T, dev(Eb).ptr, Hd, d_out.ptr, None))
    got = d_out.to_numpy((T, Hd), F16)
    assert np.array_equal(got.view(np.uint16), Eb[ids].view(np.uint16))


# ------------------------------------------------------------------------------------------- K2 / K11
@pytest.mark.parametrize("T,Hd", [(1, 64), (5, 1024), (33, 4096), (32, 1024)])
def test_rmsnorm(T, Hd):
    rng = np.random.default_rng(1)
    x, xb = h16(rng.standard_normal((T, Hd)) * 3)
    w, wb = h16(1 + 0.1 * rng.standard_normal(Hd))
    d_out = nvr.DeviceBuffer(T * Hd * 2)
    nvr.check(nvr.lib().nvr_rmsnorm(dev(xb).ptr, dev(wb).ptr, 1e-6, T, Hd, d_out.ptr, None))
    ref = oracle.round_f16(oracle.rmsnorm(x, w, 1e-6))
    assert_close_f16(d_out.to_numpy((T, Hd), F16), ref, ulps=1, atol=1e-6, what="rmsnorm")


def test_rmsnorm_extreme_magnitudes_finite():        # layernorm.rs:277-311
    for mag in (6e-5, 6e4):
        xb = np.full((2, 64), mag, F16)
        d_out = nvr.DeviceBuffer(2 * 64 * 2)
        nvr.check(nvr.lib().nvr_rmsnorm(dev(xb).ptr, dev(np.ones(64, F16)).ptr, 1e-8, 2, 64, d_out.ptr, None))
        assert np.isfinite(d_out.to_numpy((2, 64), F16).astype(np.float32)).all()


@pytest.mark.parametrize("T,Hd", [(3, 64), (32, 1024)])
def test_add_rmsnorm(T, Hd):
    rng = np.random.default_rng(2)
    h, hb = h16(rng.standard_normal((T, Hd)))
    y, yb = h16(rng.standard_normal((T, Hd)))
    w, wb = h16(1 + 0.1 * rng.standard_normal(Hd))
    d_h, d_out = dev(hb), nvr.DeviceBuffer(T * Hd * 2)
    nvr.check(nvr.lib().nvr_add_rmsnorm(d_h.ptr, dev(yb).ptr, dev(wb).ptr, 1e-6, T, Hd, d_out.ptr, None))
    hn = oracle.add(h, y, round16=True)
    assert np.array_equal(d_h.to_numpy((T, Hd), F16).astype(np.float32), hn), "residual add must be bit-exact"
    assert_close_f16(d_out.to_numpy((T, Hd), F16), oracle.round_f16(oracle.rmsnorm(hn, w, 1e-6)), ulps=1, atol=1e-6)


# ------------------------------------------------------------------------------------------- linear
@pytest.mark.parametrize("T,K,N,f32", [(1, 64, 64, False), (7, 128, 48, False), (32, 1024, 4096, False),
                                       (32, 2048, 1024, False), (33, 1024, 256, False), (32, 3072, 1024, False),
                                       (16, 1024, 6144, False), (4, 256, 1024, True), (32, 1024, 16 * 1187, True),
                                       (100, 512, 512, False), (128, 1024, 4096, False), (300, 2048, 1024, False),
                                       (1000, 3072, 1024, False), (257, 64, 48, False), (256, 128, 256, False),
                                       (700, 1024, 512, False), (513, 192, 768, False)])
def test_linear(T, K, N, f32):
    rng = np.random.default_rng(3)
    x, xb = h16(rng.standard_normal((T, K)))
    W, Wb = h16(rng.standard_normal((N, K)) * 0.05)
    d_y = nvr.DeviceBuffer(T * N * (4 if f32 else 2))
    nvr.check(nvr.lib().nvr_linear(dev(xb).ptr, K, dev(Wb).ptr, T, K, N, d_y.ptr, int(f32), None))
    ref = oracle.linear(x, W)
    if f32:
        got = d_y.to_numpy((T, N), np.float32)
        np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-4)     # f32 accumulate, different order
    else:
        assert_close_f16(d_y.to_numpy((T, N), F16), oracle.round_f16(ref), ulps=1, atol=2e-4, what="linear")


def test_linear_strided_input_and_errors():
    rng = np.random.default_rng(4)
    T, K, N, ld = 5, 64, 32, 192
    buf, bb = h16(rng.standard_normal((T, ld)))
    W, Wb = h16(rng.standard_normal((N, K)) * 0.1)
    d_y = nvr.DeviceBuffer(T * N * 2)
    d_x = dev(bb)
    nvr.check(nvr.lib().nvr_linear(d_x.ptr + 64 * 2, ld, dev(Wb).ptr, T, K, N, d_y.ptr, 0, None))
    assert_close_f16(d_y.to_numpy((T, N), F16), oracle.round_f16(oracle.linear(buf[:, 64:128], W)), ulps=1, atol=2e-4)
    assert nvr.lib().nvr_linear(d_x.ptr, ld, dev(Wb).ptr, T, 40, N, d_y.ptr, 0, None) == -10   # K % 32 != 0


# ------------------------------------------------------------------------------------------- K5 + K6
@pytest.mark.parametrize("H,KVH,D,T", [(4, 2, 64, 9), (16, 8, 128, 32), (2, 2, 128, 3)])
def test_rope_store_kv(H, KVH, D, T):
    rng = np.random.default_rng(5)
    NB, bs, max_pos = 6, 16, 200
    QKV = (H + 2 * KVH) * D
    qkv, qkvb = h16(rng.standard_normal((T, QKV)))
    pos = rng.integers(0, max_pos, T).astype(np.int64)
    slots = rng.permutation(NB * bs)[:T].astype(np.int32)
    slots[0] = -1                                                   # skipped token
    cos, sin = oracle.rope_table(D, max_pos, 1e6)
    d_cos, d_sin = nvr.DeviceBuffer(cos.nbytes), nvr.DeviceBuffer(sin.nbytes)
    nvr.check(nvr.lib().nvr_rope_table(D, max_pos, 1e6, d_cos.ptr, d_sin.ptr))
    assert np.array_equal(d_cos.to_numpy(cos.shape, np.float32), cos), "RoPE tables must be bit-identical to the oracle's"
    assert np.array_equal(d_sin.to_numpy(sin.shape, np.float32), sin)
    d_qkv = dev(qkvb)
    d_k, d_v = nvr.DeviceBuffer(NB * bs * KVH * D * 2), nvr.DeviceBuffer(NB * bs * KVH * D * 2)
    d_k.zero(); d_v.zero()
    nvr.check(nvr.lib().nvr_rope_store_kv(d_qkv.ptr, dev(pos).ptr, dev(slots).ptr, T, H, KVH, D, d_cos.ptr, d_sin.ptr,
                                          d_k.ptr, d_v.ptr, None))
    q = oracle.round_f16(oracle.rope_apply(qkv[:, :H * D].reshape(T, H, D), pos, cos, sin))
    k = oracle.round_f16(oracle.rope_apply(qkv[:, H * D:(H + KVH) * D].reshape(T, KVH, D), pos, cos, sin))
    v = np.ascontiguousarray(qkv[:, (H + KVH) * D:].reshape(T, KVH, D))
    kc, vc = np.zeros((NB, bs, KVH, D), np.float32), np.zeros((NB, bs, KVH, D), np.float32)
    oracle.kv_store(k, v, slots, kc, vc)
    got = d_qkv.to_numpy((T, QKV), F16).astype(np.float32)
    # no-contraction f32 math on both sides: rope is bit-exact
    assert np.array_equal(got[:, :H * D].reshape(T, H, D), q)
    assert np.array_equal(got[:, H * D:(H + KVH) * D].reshape(T, KVH, D), k)
    assert np.array_equal(d_k.to_numpy(kc.shape, F16).astype(np.float32), kc)
    assert np.array_equal(d_v.to_numpy(vc.shape, F16).astype(np.float32), vc)


# ------------------------------------------------------------------------------------------- K9
def _paged_case(rng, B, H, KVH, D, bs, ctx_lens, NB):
    max_blocks = max((c + bs - 1) // bs for c in ctx_lens) + 1
    kc, kcb = h16(rng.standard_normal((NB, bs, KVH, D)))
    vc, vcb = h16(rng.standard_normal((NB, bs, KVH, D)))
    bt = -np.ones((B, max_blocks), np.int32)
    perm = rng.permutation(NB)
    o = 0
    for b, c in enumerate(ctx_lens):
        nb = (c + bs - 1) // bs
        bt[b, :nb] = perm[o:o + nb]; o += nb
    return kc, kcb, vc, vcb, bt, max_blocks


@pytest.mark.parametrize("B,H,KVH,D,bs,ctxs", [
    (3, 4, 2, 64, 16, [1, 17, 40]),                     # partial last block, GQA 2:1, D=64
    (4, 16, 8, 128, 256, [1, 255, 256, 700]),           # Qwen3-0.6B head shape, block size 256
    (2, 32, 8, 128, 256, [513, 1024]),                  # Qwen3-8B group of 4
    (5, 8, 8, 128, 16, [5, 16, 31, 32, 33]),            # MHA (group 1)
    (32, 16, 8, 128, 256, [1024] * 32),                 # BASELINE config 2 decode shape (one layer)
    (1, 16, 8, 128, 256, [3000]),                       # long context, single sequence: many partitions
    (3, 4, 2, 128, 16, [2500, 1030, 7]),                # > 64 blocks per sequence: the register copy of the block table is reloaded
    (40, 16, 8, 128, 16, [1100 + 3 * i for i in range(40)]),   # same, on the one-workgroup-per-pair (direct) path
    (2, 16, 8, 128, 256, [20000, 300]),                 # 79 blocks of 256 tokens
    (3, 8, 4, 64, 8, [700, 64, 1]),                     # D=64 (8-token row groups), block size = one row group
    (512, 16, 8, 128, 64, [1 + (37 * i) % 120 for i in range(512)]),    # r05: >= 2048 (sequence, kv head) pairs — 4096 pairs, context bound 256: ONE-wave workgroups
    (300, 16, 8, 128, 64, [1 + (41 * i) % 250 for i in range(300)]),    # 2400 pairs: two waves
    (260, 16, 8, 128, 64, [1 + (43 * i) % 500 for i in range(260)]),    # context bound 512: four waves
    (520, 8, 4, 64, 32, [1 + (29 * i) % 200 for i in range(520)]),      # D=64, 2080 pairs
    (33, 16, 8, 128, 256, [1 + (389 * i) % 2000 for i in range(33)]),   # r06: 264 pairs on 256 CUs -> the work-balanced form (fused entry), ragged contexts
    (37, 16, 8, 128, 64, [900 + (53 * i) % 300 for i in range(37)]),    # 296 pairs, blocks of 64
    (70, 16, 8, 128, 256, [600 + i for i in range(70)]),                # 560 pairs: three rounds' worth on two and a bit
    (48, 32, 8, 128, 256, [1500 - 7 * i for i in range(48)]),           # group of 4 (Qwen3-8B heads), 384 pairs
    (40, 8, 8, 64, 32, [1 + (61 * i) % 700 for i in range(40)]),        # D=64, MHA, 320 pairs, some one-key contexts
])
def test_paged_attn_decode(B, H, KVH, D, bs, ctxs):
    rng = np.random.default_rng(6)
    NB = sum((c + bs - 1) // bs for c in ctxs) + 3
    kc, kcb, vc, vcb, bt, max_blocks = _paged_case(rng, B, H, KVH, D, bs, ctxs, NB)
    q, qb = h16(rng.standard_normal((B, H, D)))
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
    ref = oracle.round_f16(oracle.attn_decode(q, kc, vc, bt, ctx, scale))
    # tolerance: f32 online softmax vs two-pass softmax; outputs are O(1): 2 fp16 ulp + 1e-3 abs
    got = d_out.to_numpy((B, H, D), F16)
    assert_close_f16(got, ref, ulps=2, atol=1e-3, what="paged decode attention")
    _fused_merge_equals_two_launches(qb, kcb, vcb, meta, B, H, KVH, D, bs, scale, ws, got)


def _fused_merge_equals_two_launches(qb, kcb, vcb, meta, B, H, KVH, D, bs, scale, ws, two_launch_out):
    """nvr_paged_attn_decode_fused (the merge of a pair's split-KV partitions on its last partition workgroup to finish) == the partition
    launch + merge launch, bit for bit; called twice on one ticket array (the kernel re-arms the counters) with the workspace poisoned in between."""
    tickets = dev(np.zeros(B * KVH, np.uint32))
    # r06: pair counts that leave the last round of one-workgroup-per-pair launches mostly empty take the WORK-BALANCED form through the fused entry
    # (attn_share_kernel: 256 workgroups, each an equal share of all pairs' 64-key units): a pair cut by a share boundary is merged from partials whose key ranges
    # are not the two-launch form's — equal within the rounding of one f32 merge, not bit for bit; its own launches must agree bit for bit (slot-ordered merge)
    pairs, mc = B * KVH, int(meta.max_context_len)
    balanced = (256 < pairs < 2048 and -(-pairs // 256) * 256 * 100 >= pairs * 115 and mc >= 256 and B <= 1024 and (H // KVH) * D <= 512
                and bs >= 8 and bs & (bs - 1) == 0 and os.environ.get("NVR_ATTN_SHARE", "1") != "0")
    first = None
    for rep in range(2):
        d_o = nvr.DeviceBuffer(B * H * D * 2); _KEEP.append(d_o)
        nvr.check(nvr.lib().nvr_paged_attn_decode_fused(dev(qb).ptr, H * D, dev(kcb).ptr, dev(vcb).ptr, C.byref(meta), H, KVH, D, bs,
                                                        scale, d_o.ptr, ws.ptr, tickets.ptr, None))
        f = d_o.to_numpy((B, H, D), F16)
        if balanced:
            assert_close_f16(f, np.asarray(two_launch_out), ulps=2, atol=1e-3, what="work-balanced form vs the two-launch form")
            first = f.copy() if first is None else first
            assert np.array_equal(f.view(np.uint16), first.view(np.uint16)), "two launches of the work-balanced form differ"
        else:
            assert np.array_equal(f.view(np.uint16), np.asarray(two_launch_out).view(np.uint16)), f"fused merge differs from the merge launch (call {rep})"
        assert not tickets.to_numpy((B * KVH,), np.uint32).any(), "arrival counters not re-armed"
        nvr.check(nvr.lib().nvr_fill_const(ws.ptr, ws.nbytes // 2, C.c_float(float("nan")), None))


def test_work_balanced_attention_writes_zero_rows_for_queries_without_keys():
    """A query whose context is empty owns no unit of the work-balanced launch (attn_share_kernel): its output row must still be written — zeros, as the
    per-pair form writes them — and the other rows must equal the per-pair form's within the rounding of one f32 merge."""
    rng = np.random.default_rng(91)
    l = nvr.lib()
    B, H, KVH, D, bs = 36, 16, 8, 128, 64
    ctxs = [0 if i % 9 == 4 else 300 + 11 * i for i in range(B)]
    NB = sum((c + bs - 1) // bs for c in ctxs) + 3
    kc, kcb, vc, vcb, bt, max_blocks = _paged_case(rng, B, H, KVH, D, bs, ctxs, NB)
    q, qb = h16(rng.standard_normal((B, H, D)))
    scale = float(np.float32(1.0) / np.sqrt(np.float32(D)))
    meta = nvr.AttnMetaC()
    d_ctx, d_bt = dev(np.asarray(ctxs, np.int32)), dev(bt)
    meta.context_lens, meta.block_tables, meta.max_blocks, meta.batch, meta.max_context_len = d_ctx.ptr, d_bt.ptr, max_blocks, B, int(max(ctxs))
    ws = nvr.DeviceBuffer(l.nvr_paged_attn_workspace_bytes(B, H, D, int(max(ctxs))))
    d_q, d_k, d_v = dev(qb), dev(kcb), dev(vcb)
    d_ref, d_out = nvr.DeviceBuffer(B * H * D * 2), nvr.DeviceBuffer(B * H * D * 2)
    nvr.check(l.nvr_fill_const(d_out.ptr, B * H * D, C.c_float(float("nan")), None))               # (poison: a row nobody writes would show)
    nvr.check(l.nvr_paged_attn_decode(d_q.ptr, H * D, d_k.ptr, d_v.ptr, C.byref(meta), H, KVH, D, bs, scale, d_ref.ptr, ws.ptr, None))
    tickets = dev(np.zeros(B * KVH, np.uint32))
    nvr.check(l.nvr_paged_attn_decode_fused(d_q.ptr, H * D, d_k.ptr, d_v.ptr, C.byref(meta), H, KVH, D, bs, scale, d_out.ptr, ws.ptr, tickets.ptr, None))
    ref, got = d_ref.to_numpy((B, H, D), F16), d_out.to_numpy((B, H, D), F16)
    empty = [i for i, c in enumerate(ctxs) if c == 0]
    assert len(empty) == 4 and not got[empty].view(np.uint16).any() and not ref[empty].view(np.uint16).any()
    assert_close_f16(got, ref, ulps=2, atol=1e-3, what="work-balanced form vs the per-pair form")
    assert not tickets.to_numpy((B * KVH,), np.uint32).any()


def test_fused_split_kv_merge_under_uneven_load():
    """The last-arriver merge of split-KV attention (sc1 partials + one ticket per workgroup, no acquire fence: cdna guide Guideline 16) held to
    the guide's own test rule — uneven load, every word checked, many repetitions: while a second stream keeps the memory system busy with large
    fills (the partition workgroups of a pair then finish far apart and in changing order), 60 back-to-back fused launches over ragged contexts
    (1 .. 3000 keys: 1 .. 12 partitions per pair) must each reproduce the two-launch result bit for bit and leave every arrival counter at zero."""
    rng = np.random.default_rng(77)
    l = nvr.lib()
    B, H, KVH, D, bs = 6, 8, 2, 128, 64
    ctxs = [1, 63, 700, 3000, 1025, 2047]
    NB = sum((c + bs - 1) // bs for c in ctxs) + 3
    kc, kcb, vc, vcb, bt, max_blocks = _paged_case(rng, B, H, KVH, D, bs, ctxs, NB)
    q, qb = h16(rng.standard_normal((B, H, D)))
    ctx = np.asarray(ctxs, np.int32)
    scale = float(np.float32(1.0) / np.sqrt(np.float32(D)))
    meta = nvr.AttnMetaC()
    d_ctx, d_bt = dev(ctx), dev(bt)
    meta.context_lens, meta.block_tables, meta.max_blocks, meta.batch, meta.max_context_len = d_ctx.ptr, d_bt.ptr, max_blocks, B, int(max(ctxs))
    ws = nvr.DeviceBuffer(l.nvr_paged_attn_workspace_bytes(B, H, D, int(max(ctxs)))); _KEEP.append(ws)
    d_q, d_k, d_v = dev(qb), dev(kcb), dev(vcb)
    d_ref = nvr.DeviceBuffer(B * H * D * 2); _KEEP.append(d_ref)
    nvr.check(l.nvr_paged_attn_decode(d_q.ptr, H * D, d_k.ptr, d_v.ptr, C.byref(meta), H, KVH, D, bs, scale, d_ref.ptr, ws.ptr, None))
    ref = d_ref.to_numpy((B, H, D), np.uint16)
    assert_close_f16(ref.view(F16), oracle.round_f16(oracle.attn_decode(q, kc, vc, bt, ctx, scale)), ulps=2, atol=1e-3, what="two-launch reference")
    sa, sb = C.c_void_p(), C.c_void_p()
    nvr.check(l.nvr_stream_create(C.byref(sa))); nvr.check(l.nvr_stream_create(C.byref(sb)))
    hog = nvr.DeviceBuffer(512 << 20); _KEEP.append(hog)
    tickets = dev(np.zeros(B * KVH, np.uint32))
    outs = [nvr.DeviceBuffer(B * H * D * 2) for _ in range(60)]; _KEEP.extend(outs)
    try:
        for i, d_o in enumerate(outs):
            if i % 3 == 0:                                   # bursts of competing traffic on the other stream: load that comes and goes
                nvr.check(l.nvr_fill_weight(hog.ptr, 16384, 16384, 16384, 16384, 0, 0, 9 + i, 1e-6, sb))
            nvr.check(l.nvr_paged_attn_decode_fused(d_q.ptr, H * D, d_k.ptr, d_v.ptr, C.byref(meta), H, KVH, D, bs, scale, d_o.ptr, ws.ptr, tickets.ptr, sa))
        nvr.check(l.nvr_stream_synchronize(sa)); nvr.check(l.nvr_stream_synchronize(sb))
    finally:
        l.nvr_stream_destroy(sa); l.nvr_stream_destroy(sb)
    for i, d_o in enumerate(outs):
        assert np.array_equal(d_o.to_numpy((B, H, D), np.uint16), ref), f"fused launch {i} differs from the two-launch result"
    assert not tickets.to_numpy((B * KVH,), np.uint32).any()


def test_work_balanced_attention_under_uneven_load():
    """attn_share_kernel's hand-over (sc1 partials of a pair cut by share boundaries + one ticket per share, merged by the last share to finish) under the same
    rule as the split-KV form above: 60 back-to-back launches over ragged contexts while a second stream keeps the memory system busy — the shares of a pair then
    finish far apart and in changing order.  The merge is slot-ordered, so every launch must reproduce the first one bit for bit, stay within the rounding of one
    f32 merge of the per-pair form, and leave every arrival counter at zero."""
    rng = np.random.default_rng(78)
    l = nvr.lib()
    B, H, KVH, D, bs = 40, 16, 8, 128, 64
    ctxs = [1 + (977 * i) % 2600 for i in range(B)]
    NB = sum((c + bs - 1) // bs for c in ctxs) + 3
    kc, kcb, vc, vcb, bt, max_blocks = _paged_case(rng, B, H, KVH, D, bs, ctxs, NB)
    q, qb = h16(rng.standard_normal((B, H, D)))
    scale = float(np.float32(1.0) / np.sqrt(np.float32(D)))
    meta = nvr.AttnMetaC()
    d_ctx, d_bt = dev(np.asarray(ctxs, np.int32)), dev(bt)
    meta.context_lens, meta.block_tables, meta.max_blocks, meta.batch, meta.max_context_len = d_ctx.ptr, d_bt.ptr, max_blocks, B, int(max(ctxs))
    ws = nvr.DeviceBuffer(l.nvr_paged_attn_workspace_bytes(B, H, D, int(max(ctxs)))); _KEEP.append(ws)
    d_q, d_k, d_v = dev(qb), dev(kcb), dev(vcb)
    d_ref = nvr.DeviceBuffer(B * H * D * 2); _KEEP.append(d_ref)
    nvr.check(l.nvr_paged_attn_decode(d_q.ptr, H * D, d_k.ptr, d_v.ptr, C.byref(meta), H, KVH, D, bs, scale, d_ref.ptr, ws.ptr, None))
    sa, sb = C.c_void_p(), C.c_void_p()
    nvr.check(l.nvr_stream_create(C.byref(sa))); nvr.check(l.nvr_stream_create(C.byref(sb)))
    hog = nvr.DeviceBuffer(512 << 20); _KEEP.append(hog)
    tickets = dev(np.zeros(B * KVH, np.uint32))
    outs = [nvr.DeviceBuffer(B * H * D * 2) for _ in range(60)]; _KEEP.extend(outs)
    try:
        for i, d_o in enumerate(outs):
            if i % 3 == 0:
                nvr.check(l.nvr_fill_weight(hog.ptr, 16384, 16384, 16384, 16384, 0, 0, 9 + i, 1e-6, sb))
            nvr.check(l.nvr_paged_attn_decode_fused(d_q.ptr, H * D, d_k.ptr, d_v.ptr, C.byref(meta), H, KVH, D, bs, scale, d_o.ptr, ws.ptr, tickets.ptr, sa))
        nvr.check(l.nvr_stream_synchronize(sa)); nvr.check(l.nvr_stream_synchronize(sb))
    finally:
        l.nvr_stream_destroy(sa); l.nvr_stream_destroy(sb)
    first = outs[0].to_numpy((B, H, D), np.uint16)
    assert_close_f16(first.view(F16), d_ref.to_numpy((B, H, D), F16), ulps=2, atol=1e-3, what="work-balanced form vs the per-pair form")
    for i, d_o in enumerate(outs[1:], 1):
        assert np.array_equal(d_o.to_numpy((B, H, D), np.uint16), first), f"work-balanced launch {i} differs from launch 0"
    assert not tickets.to_numpy((B * KVH,), np.uint32).any()


def test_paged_attn_ignores_garbage_beyond_context():
    """A-8: exactly context_lens[b] keys are visible; poison everything else (incl. -1 padded table slots)."""
    rng = np.random.default_rng(7)
    B, H, KVH, D, bs = 2, 4, 2, 64, 16
    ctxs = [19, 33]
    NB = 8
    kc, kcb, vc, vcb, bt, max_blocks = _paged_case(rng, B, H, KVH, D, bs, ctxs, NB)
    q, qb = h16(rng.standard_normal((B, H, D)))
    kcb2, vcb2 = kcb.copy(), vcb.copy()
    for b, c in enumerate(ctxs):
        blk, off = bt[b, c // bs], c % bs
        if blk >= 0:
            kcb2[blk, off:] = F16(6e4); vcb2[blk, off:] = F16(6e4)
    used = set(bt[bt >= 0].tolist())
    for blk in range(NB):
        if blk not in used:
            kcb2[blk] = F16(np.nan); vcb2[blk] = F16(np.nan)
    scale = 0.125
    outs = []
    for kb, vb in ((kcb, vcb), (kcb2, vcb2)):
        meta = nvr.AttnMetaC()
        d_ctx, d_bt = dev(np.asarray(ctxs, np.int32)), dev(bt)
        meta.context_lens, meta.block_tables, meta.max_blocks, meta.batch, meta.max_context_len = d_ctx.ptr, d_bt.ptr, max_blocks, B, 33
        ws = nvr.DeviceBuffer(nvr.lib().nvr_paged_attn_workspace_bytes(B, H, D, 33))
        d_out = nvr.DeviceBuffer(B * H * D * 2)
        nvr.check(nvr.lib().nvr_paged_attn_decode(dev(qb).ptr, H * D, dev(kb).ptr, dev(vb).ptr, C.byref(meta), H, KVH, D, bs, scale,
                                                  d_out.ptr, ws.ptr, None))
        outs.append(d_out.to_numpy((B, H, D), F16))
    assert np.array_equal(outs[0].view(np.uint16), outs[1].view(np.uint16))


def test_online_softmax_rescale_branch_forced():
    """cdna guide rule 26: force the running-max rescale with a spiked key late in the context."""
    rng = np.random.default_rng(8)
    B, H, KVH, D, bs, c = 1, 2, 1, 128, 16, 400
    NB = c // bs + 2
    kc, kcb, vc, vcb, bt, max_blocks = _paged_case(rng, B, H, KVH, D, bs, [c], NB)
    q, qb = h16(rng.standard_normal((B, H, D)))
    for tok in (70, 333):                               # keys aligned with q -> score jumps by ~ +|q|^2
        blk, off = bt[0, tok // bs], tok % bs
        kcb[blk, off, 0] = (q[0, 0] * (2.0 if tok == 333 else 1.0)).astype(F16)
    kc = kcb.astype(np.float32)
    scale = float(1 / np.sqrt(np.float32(D)))
    meta = nvr.AttnMetaC()
    d_ctx, d_bt = dev(np.asarray([c], np.int32)), dev(bt)
    meta.context_lens, meta.block_tables, meta.max_blocks, meta.batch, meta.max_context_len = d_ctx.ptr, d_bt.ptr, max_blocks, B, c
    ws = nvr.DeviceBuffer(nvr.lib().nvr_paged_attn_workspace_bytes(B, H, D, c))
    d_out = nvr.DeviceBuffer(B * H * D * 2)
    nvr.check(nvr.lib().nvr_paged_attn_decode(dev(qb).ptr, H * D, dev(kcb).ptr, dev(vcb).ptr, C.byref(meta), H, KVH, D, bs, scale,
                                              d_out.ptr, ws.ptr, None))
    ref = oracle.round_f16(oracle.attn_decode(q, kc, vc, bt, np.asarray([c], np.int32), scale))
    assert_close_f16(d_out.to_numpy((B, H, D), F16), ref, ulps=2, atol=1e-3)


# ------------------------------------------------------------------------------------------- K7
@pytest.mark.parametrize("H,KVH,D,lens", [(4, 2, 64, [1, 5, 33]), (16, 8, 128, [70, 129]), (8, 8, 128, [17]), (8, 2, 128, [100, 3, 64]),
                                          (16, 8, 128, [300]), (16, 2, 128, [40]),
                                          (4, 2, 128, [1500, 129, 64, 1]),      # many 64-key steps, tiles of every fill level
                                          (2, 2, 64, [700, 65]), (8, 2, 64, [260])])   # G=1 / G=4 at head_dim 64
def test_attn_prefill_varlen(H, KVH, D, lens):
    rng = np.random.default_rng(9)
    T = sum(lens)
    QKV = (H + 2 * KVH) * D
    qkv, qkvb = h16(rng.standard_normal((T, QKV)))
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
    ref = oracle.round_f16(oracle.attn_prefill_varlen(q, k, v, cu, scale))
    # MFMA path: P is rounded to fp16 before P·V (relative 2^-11 per term): 2 fp16 ulp + 2e-3 absolute on O(1) outputs
    assert_close_f16(d_out.to_numpy((T, H, D), F16), ref, ulps=2, atol=2e-3, what="varlen prefill attention")


@pytest.mark.parametrize("T,N", [(1, 8), (7, 2560), (300, 3072)])
def test_add_bias_is_the_second_rounding_of_a_linear_with_bias(T, N):
    """nvr_add_bias (A-30): y <- fp16(y + b) in place, bit for bit the oracle's add of the already-rounded matmul result."""
    rng = np.random.default_rng(40 + T)
    y, yb = h16(rng.standard_normal((T, N)) * 3)
    b, bb = h16(rng.standard_normal(N))
    d_y = dev(yb)
    nvr.check(nvr.lib().nvr_add_bias(d_y.ptr, dev(bb).ptr, T, N, None))
    assert np.array_equal(d_y.to_numpy((T, N), F16), oracle.add(y, np.broadcast_to(b, y.shape).copy(), round16=True).astype(F16))
    assert nvr.lib().nvr_add_bias(d_y.ptr, dev(bb).ptr, T, 12, None) == -10


# ------------------------------------------------------------------------------------------- K13 / K15
def test_silu_and_mul():
    rng = np.random.default_rng(10)
    T, I = 32, 3072
    x, xb = h16(rng.standard_normal((T, 2 * I)) * 2)
    d_out = nvr.DeviceBuffer(T * I * 2)
    nvr.check(nvr.lib().nvr_silu_and_mul(dev(xb).ptr, T, I, d_out.ptr, None))
    # device __expf vs host expf: <= 1 fp16 ulp after rounding
    assert_close_f16(d_out.to_numpy((T, I), F16), oracle.round_f16(oracle.silu_and_mul(x)), ulps=1, atol=1e-6)


@pytest.mark.parametrize("kind", ["silu", "gelu", "relu", "silu_and_mul", "gelu_and_mul"])
def test_activation_types(kind):
    """nvr_activation = Activation::forward (activation.rs:147-159) for every ActivationType against the oracle: f32 inside, one fp16 rounding at the store
    (device expf / tanhf against the host's: <= 1 fp16 ulp); relu exactly; the reference's KAT inputs; an odd width of the fused types is refused with the
    reference's message (:50-52, :88-90)."""
    rng = np.random.default_rng(12)
    T, cols = 33, 2 * 3072
    raw = rng.standard_normal((T, cols)) * 2
    raw[0, :5] = [-2, -1, 0, 1, 2]                                               # the inputs of the reference's own tests (activation.rs:214-261)
    x, xb = h16(raw)
    k = oracle.ACTIVATION_TYPES[kind]
    co = cols // 2 if k >= 3 else cols
    d_out = nvr.DeviceBuffer(T * co * 2)
    nvr.check(nvr.lib().nvr_activation(k, dev(xb).ptr, T, cols, d_out.ptr, None))
    got, ref = d_out.to_numpy((T, co), F16), oracle.round_f16(oracle.activation(kind, x))
    if kind == "relu":
        assert np.array_equal(got.astype(np.float32), ref)
        assert got[0, :5].tolist() == [0.0, 0.0, 0.0, 1.0, 2.0]
    else:
        assert_close_f16(got, ref, ulps=1, atol=1e-6)
    if kind in ("silu", "gelu"):
        assert got[0, 2] == 0.0
    if k >= 3:
        assert nvr.lib().nvr_activation(k, dev(xb).ptr, T, 4097, d_out.ptr, None) == -7
        assert f"Input dimension must be even for {'SiluAndMul' if k == 3 else 'GeluAndMul'}, got 4097" in nvr.last_error()
    assert nvr.lib().nvr_activation(7, dev(xb).ptr, T, cols, d_out.ptr, None) == -7


def test_select_last_tokens_exact():
    rng = np.random.default_rng(11)
    lens, Hd = [3, 1, 7], 128
    h, hb = h16(rng.standard_normal((sum(lens), Hd)))
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    d_out = nvr.DeviceBuffer(3 * Hd * 2)
    nvr.check(nvr.lib().nvr_select_last_tokens(dev(hb).ptr, dev(cu).ptr, 3, Hd, d_out.ptr, None))
    assert np.array_equal(d_out.to_numpy((3, Hd), F16).view(np.uint16), hb[cu[1:] - 1].view(np.uint16))


# ------------------------------------------------------------------------------------------- K17 / K18
def test_argmax_exact_with_ties():
    rng = np.random.default_rng(12)
    B, V = 33, 151936
    x = rng.standard_normal((B, V)).astype(np.float32)
    x[0, [5, 77777, 151935]] = 9.0                     # tie -> lowest index (A-12)
    x[1, V - 1] = 50.0
    x[2, 0] = 50.0
    x[3] = -np.inf; x[3, 1234] = -1e30
    d_out = nvr.DeviceBuffer(B * 8)
    nvr.check(nvr.lib().nvr_argmax(dev(x).ptr, B, V, d_out.ptr, None))
    got = d_out.to_numpy((B,), np.int64)
    assert got.tolist() == [oracle.argmax(x[b]) for b in range(B)]
    assert got[0] == 5 and got[1] == V - 1 and got[2] == 0 and got[3] == 1234
    for kat, want in (([1.0, 2.0, 5.0, 1.5], 2), ([1.0, 2.0, 3.0], 2)):      # sampler.rs:333-356
        nvr.check(nvr.lib().nvr_argmax(dev(np.asarray(kat, np.float32)).ptr, 1, len(kat), d_out.ptr, None))
        assert d_out.to_numpy((1,), np.int64)[0] == want


@pytest.mark.parametrize("T,K,N", [(32, 1024, 151936), (7, 1024, 18992), (16, 2048, 4096), (17, 256, 48), (1, 512, 16), (32, 1024, 100000),
                                   # more than 32 rows (large decode batches, many-sequence prefills): 128x128 tiles, one partial per tile
                                   (128, 1024, 151936), (100, 512, 5008), (300, 256, 1000 * 16), (33, 4096, 2064),
                                   # r06: >= 256 rows go through the 256x256 tiles (one partial per 256-column tile; the vocabulary's last tile ragged:
                                   # 151 936 = 593 x 256 + 128, 18 992 = 74 x 256 + 48 (a tp-8 shard), 2064 = 8 x 256 + 16; ragged row blocks: 300, 257, 700)
                                   (512, 1024, 151936), (256, 1024, 18992), (257, 4096, 2064), (700, 128, 4096), (1024, 1024, 75968), (200, 1024, 18992), (192, 512, 4112),
                                   # K > 2048 (Qwen3-8B: hidden 4096): the activation block goes through LDS in 2048-column chunks, every wave
                                   # keeps its tiles' accumulators in registers across the chunks
                                   (32, 4096, 151936), (7, 4096, 18992), (16, 6144, 4096), (1, 8192, 48), (32, 4096, 16 * 2048 * 8), (8, 5120, 4096), (20, 2560, 1024)])
def test_lm_head_logits_and_argmax_partials(T, K, N):
    """lm_head: f32 logits == oracle linear (f32 sums in another order), and the arg-max that rides along in the epilogue
    is EXACTLY the lowest-index arg-max of the logits the kernel itself wrote (ties included)."""
    rng = np.random.default_rng(31)
    x, xb = h16(rng.standard_normal((T, K)))
    W, Wb = h16(rng.standard_normal((N, K), dtype=np.float32) * np.float32(0.05))   # (f32 draws: the 262 144 x 4096 case holds a billion of them)
    W[N // 3] = W[5]; Wb[N // 3] = Wb[5]                   # identical rows: equal logits -> exact ties
    W[N - 1] = W[5]; Wb[N - 1] = Wb[5]
    x[0], xb[0] = h16(W[5] * 8)                            # make that tied logit the row maximum of token 0
    d_x, d_W = dev(xb), dev(Wb)
    d_y = nvr.DeviceBuffer(T * N * 4)
    P = 2048                                               # NVR_LM_HEAD_MAX_PARTS
    d_pv, d_pi = nvr.DeviceBuffer(P * T * 4), nvr.DeviceBuffer(P * T * 4)
    nparts = C.c_int32(0)
    nvr.check(nvr.lib().nvr_lm_head(d_x.ptr, K, d_W.ptr, T, K, N, d_y.ptr, d_pv.ptr, d_pi.ptr, C.byref(nparts), None))
    assert 1 <= nparts.value <= P
    y = d_y.to_numpy((T, N), np.float32)
    np.testing.assert_allclose(y, oracle.linear(x, W), rtol=2e-5, atol=3e-4)
    d_tok, d_val = nvr.DeviceBuffer(T * 8), nvr.DeviceBuffer(T * 4)
    nvr.check(nvr.lib().nvr_argmax_partials(d_pv.ptr, d_pi.ptr, nparts.value, T, d_tok.ptr, d_val.ptr, 1000, None))
    tok, val = d_tok.to_numpy((T,), np.int64), d_val.to_numpy((T,), np.float32)
    want = np.asarray([oracle.argmax(y[t]) for t in range(T)])
    assert (tok - 1000).tolist() == want.tolist()
    assert np.array_equal(val, y[np.arange(T), want])
    if N > 16:
        assert want[0] == 5                                # three-way tie at rows 5, N//3, N-1 -> lowest index
    # the plain arg-max kernel over the same logits agrees
    d_t2 = nvr.DeviceBuffer(T * 8)
    nvr.check(nvr.lib().nvr_argmax(d_y.ptr, T, N, d_t2.ptr, None))
    assert d_t2.to_numpy((T,), np.int64).tolist() == want.tolist()
    # logits == NULL (what a greedy step asks for): the same partials without the f32 stores
    d_pv2, d_pi2 = nvr.DeviceBuffer(P * T * 4), nvr.DeviceBuffer(P * T * 4)
    np2 = C.c_int32(0)
    nvr.check(nvr.lib().nvr_lm_head(d_x.ptr, K, d_W.ptr, T, K, N, None, d_pv2.ptr, d_pi2.ptr, C.byref(np2), None))
    assert np2.value == nparts.value
    n = nparts.value * T
    assert np.array_equal(d_pv2.to_numpy((P * T,), np.float32)[:n].view(np.uint32), d_pv.to_numpy((P * T,), np.float32)[:n].view(np.uint32))
    assert np.array_equal(d_pi2.to_numpy((P * T,), np.int32)[:n], d_pi.to_numpy((P * T,), np.int32)[:n])
    # shapes neither kernel takes are refused, not approximated: N not a multiple of 16
    assert nvr.lib().nvr_lm_head(d_x.ptr, K, d_W.ptr, T, K, N - 8, d_y.ptr, d_pv.ptr, d_pi.ptr, C.byref(nparts), None) == -10


def _gpu_sample(x, temps, top_k, top_p, keys):
    B, V = x.shape
    ws = nvr.DeviceBuffer(nvr.lib().nvr_sample_workspace_bytes(B, V))
    d_out = nvr.DeviceBuffer(B * 8)
    nvr.check(nvr.lib().nvr_sample(dev(x).ptr, B, V, dev(np.asarray(temps, np.float32)).ptr,
                                   dev(np.asarray(top_k, np.int64)).ptr, dev(np.asarray(top_p, np.float32)).ptr,
                                   dev(np.asarray(keys, np.uint64)).ptr, d_out.ptr, ws.ptr, None))
    return d_out.to_numpy((B,), np.int64), ws.to_numpy((B, V), np.float32)


def test_sampler_filters_match_reference_kats():
    # top-k=3 of [1,5,2,4,3] -> [-inf,5,-inf,4,3] (sampler.rs:359-374); top-p 0.9 of [0,10,5,1] keeps idx 1 (:377-389)
    x = np.asarray([[1, 5, 2, 4, 3]], np.float32)
    tok, w = _gpu_sample(x, [1.0], [3], [-1.0], [1])
    assert w[0].tolist() == [-np.inf, 5.0, -np.inf, 4.0, 3.0] and tok[0] in (1, 3, 4)
    x = np.asarray([[0, 10, 5, 1]], np.float32)
    tok, w = _gpu_sample(x, [1.0], [0], [0.9], [1])
    assert w[0, 1] == 10.0 and np.isinf(w[0, [0, 2, 3]]).all() and tok[0] == 1
    # ties at the top-k threshold keep the lowest indices (stable sort, A-19)
    x = np.asarray([[2, 7, 7, 7, 1, 7]], np.float32)
    _, w = _gpu_sample(x, [1.0], [2], [-1.0], [1])
    assert w[0].tolist() == [-np.inf, 7.0, 7.0, -np.inf, -np.inf, -np.inf]
    assert w[0].tolist() == oracle.top_k(x[0], 2).tolist()


@pytest.mark.parametrize("V,reps,scale", [(1000, 1, 3.0), (151936, 1, 3.0), (151936, 1, 0.05), (20000, 1, 3.0), (50001, 4, 1.0), (4097, 6, 3.0),
                                          (151936, 4, 3.0), (151936, 8, 1.0)])   # 48 rows: 4 sharers x 38 elements per thread; 96 rows: one workgroup per row
def test_sampler_matches_oracle(V, reps, scale):
    """Rows of every filter combination; reps tiles them to B = 12 * reps rows (1..8 workgroups share a row depending on B and V;
    scale 0.05 = the nearly flat rows of a random-init model, where top-p keeps most of the vocabulary)."""
    rng = np.random.default_rng(13)
    B = 12 * reps
    x = (rng.standard_normal((B, V)) * scale).astype(np.float32)
    temps = [0.0, 1.0, 0.7, 1.3, 1.0, 0.5, 1.0, 2.0, 1.0, 0.9, 1.0, 0.0] * reps
    top_k = [0, 0, 50, 0, 5, 0, 1, 0, 200, 40, V + 5, 7] * reps
    top_p = [-1, -1, -1, 0.9, 0.5, 0.95, -1, 0.3, 0.8, 1.0, -1, 0.5] * reps
    keys = [oracle.sample_key(123, b, 4) for b in range(B)]
    assert [nvr.lib().nvr_sample_key(123, b, 4) for b in range(B)] == keys
    tok, w = _gpu_sample(x, temps, top_k, top_p, keys)
    exact = 0
    for b in range(B):
        ref = oracle.sample(x[b], temps[b], top_k[b], None if top_p[b] < 0 else top_p[b], keys[b])
        if temps[b] == 0.0:
            assert tok[b] == ref                         # greedy: bit-exact
            continue
        # the kept set must equal the oracle's, up to the single marginal element where an f32
        # cumulative sum lands within rounding of p (A-19/A-20: stochastic path is tolerance-tested)
        a = x[b] / np.float32(temps[b]) if temps[b] != 1.0 else x[b]
        if top_k[b] > 0:
            a = oracle.top_k(a, top_k[b])
        if top_p[b] >= 0:
            a = oracle.top_p(a, top_p[b])
        kept_ref, kept_gpu = np.isfinite(a), np.isfinite(w[b])
        # (a nearly flat row keeps ~1e5 elements of ~equal probability: the oracle's sequential f32 cumulative sum drifts by a few
        # elements' worth of mass against the kernel's 2^-40 fixed-point sums - bounded at 2e-4 of the kept set)
        slack = max(1, int(2e-4 * kept_ref.sum()))
        assert (kept_ref != kept_gpu).sum() <= slack, f"row {b}: kept sets differ in {(kept_ref != kept_gpu).sum()} places"
        exact += int(tok[b] == ref)
        assert kept_gpu[tok[b]]
    assert exact >= B - (3 + 2) * reps                   # Gumbel argmax over (nearly) identical sets


@pytest.mark.parametrize("B,V", [(5, 151936), (3, 1000), (40, 30000)])
def test_sampler_ties_at_the_threshold_keep_the_lowest_indices(B, V):
    """Quantised logits: hundreds of elements share the k-th value, spread over every slice of a shared row.  top-k keeps
    the first `need` of them in index order (a stable descending sort, sampler.rs:115-148, A-19) - exactly the oracle's set;
    top-p on the same rows stays within the one marginal element of the f32 cumulative sum."""
    rng = np.random.default_rng(31)
    x = (np.round(rng.standard_normal((B, V)) * 4) / 4).astype(np.float32)       # 0.25 steps: heavy ties
    temps = [1.0] * B
    keys = [oracle.sample_key(7, b, 1) for b in range(B)]
    for k, p in [(50, -1.0), (1000, -1.0), (0, 0.6), (300, 0.9)]:
        tok, w = _gpu_sample(x, temps, [k] * B, [p] * B, keys)
        for b in range(B):
            a = x[b]
            if k > 0:
                a = oracle.top_k(a, k)
            if p >= 0:
                a = oracle.top_p(a, p)
            kept_ref, kept_gpu = np.isfinite(a), np.isfinite(w[b])
            if p < 0:
                assert np.array_equal(kept_ref, kept_gpu), f"k={k} row {b}: kept sets differ"
                assert kept_gpu.sum() == k
            else:
                # the reference accumulates the sorted probabilities sequentially in f32 (sampler.rs:168-177) and so does the
                # oracle; the kernel sums them exactly (2^-40 fixed point).  Over the 1e3..1e5 kept elements the f32 running sum
                # is good to ~1e-4 of the mass: the two kept sets may differ by that much mass, and only in elements tied at
                # the boundary value
                d = np.flatnonzero(kept_ref != kept_gpu)
                if d.size:
                    a0 = oracle.top_k(x[b], k).astype(np.float64) if k > 0 else x[b].astype(np.float64)
                    pr = np.exp(a0 - a0[np.isfinite(a0)].max()); pr[~np.isfinite(a0)] = 0.0; pr /= pr.sum()
                    assert pr[d].sum() <= 2e-4, f"k={k} p={p} row {b}: kept sets differ by {pr[d].sum():.2e} of the mass ({d.size} elements)"
                    assert np.unique(x[b][d]).size == 1
            assert kept_gpu[tok[b]]


def test_fill_weight_bit_exact_with_oracle():
    rows, cols, gcols = 37, 96, 512
    key = oracle.weight_key(7, 1234)
    assert nvr.lib().nvr_weight_key(7, 1234) == key
    sc = oracle.weight_scale(0.02)
    assert nvr.lib().nvr_weight_scale(0.02) == np.float32(sc)
    d = nvr.DeviceBuffer(rows * cols * 2)
    nvr.check(nvr.lib().nvr_fill_weight(d.ptr, rows, cols, cols, gcols, 5, 64, key, sc, None))
    ref = oracle.fill_weight(rows, cols, gcols, 5, 64, key, sc, True)
    assert np.array_equal(d.to_numpy((rows, cols), F16).astype(np.float32), ref)
    assert abs(ref.std() - 0.02) < 0.003


@pytest.mark.parametrize("T", [32, 9, 48, 64, 33])
def test_decode_gemms_over_large_weights(T):
    """linear_stream.hip (Qwen3-8B-class shapes: >= 24 MiB of weights, K >= 2048): plain, SiLU and RoPE+store epilogues
    against the oracle compositions; several K chunks (K = 6144 -> 3 chunks of 2048 at T > 16) and several tiles per
    workgroup (N = 8192 -> 512 tiles on 256 workgroups).  r06: K = 2048 / 4096 take the double-buffered LDS-DMA image (two chunks of weight
    pieces in flight), and 33..64 rows stay on the streaming kernels there (64-row images of 512 columns) instead of two 32-row skinny blocks."""
    rng = np.random.default_rng(61)
    # plain, hidden 4096 and 2048 (the IMG instantiations: NC = 4 / 2, 8 / 4 at more than 32 rows): 512 and 1024 tiles -> one or two per workgroup
    for K, N in ((4096, 8192), (2048, 8192)):
        x, xb = h16(rng.standard_normal((T, K)))
        W, Wb = h16(rng.standard_normal((N, K)) * 0.02)
        d_y = nvr.DeviceBuffer(T * N * 2)
        nvr.check(nvr.lib().nvr_linear(dev(xb).ptr, K, dev(Wb).ptr, T, K, N, d_y.ptr, 0, None))
        assert_close_f16(d_y.to_numpy((T, N), F16), oracle.round_f16(oracle.linear(x, W)), ulps=1, atol=3e-4, what=f"stream plain K={K}")
    # plain: K = 6144, N = 8192 (512 tiles: two per workgroup)
    K, N = 6144, 8192
    x, xb = h16(rng.standard_normal((T, K)))
    W, Wb = h16(rng.standard_normal((N, K)) * 0.02)
    d_y = nvr.DeviceBuffer(T * N * 2)
    nvr.check(nvr.lib().nvr_linear(dev(xb).ptr, K, dev(Wb).ptr, T, K, N, d_y.ptr, 0, None))
    assert_close_f16(d_y.to_numpy((T, N), F16), oracle.round_f16(oracle.linear(x, W)), ulps=1, atol=3e-4, what="stream plain")
    # SiLU: K = 4096, I = 6144 (2 x 6144 rows = 96 MiB), 384 tile pairs
    K, I = 4096, 6144
    x, xb = h16(rng.standard_normal((T, K)))
    W, Wb = h16(rng.standard_normal((2 * I, K)) * 0.02)
    d_o = nvr.DeviceBuffer(T * I * 2)
    nvr.check(nvr.lib().nvr_linear_silu_mul(dev(xb).ptr, K, dev(Wb).ptr, T, K, I, d_o.ptr, None))
    ref = oracle.round_f16(oracle.silu_and_mul(oracle.round_f16(oracle.linear(x, W))))
    assert_close_f16(d_o.to_numpy((T, I), F16), ref, ulps=2, atol=3e-4, what="stream silu")
    # qkv + RoPE + store: Qwen3-8B heads (H=32, KVH=8, D=128, hidden 4096): 384 tiles
    H, KVH, D, K = 32, 8, 128, 4096
    QKV = (H + 2 * KVH) * D
    NB, bs, max_pos = 8, 16, 300
    x, xb = h16(rng.standard_normal((T, K)))
    W, Wb = h16(rng.standard_normal((QKV, K)) * 0.02)
    pos = rng.integers(0, max_pos, T).astype(np.int64)
    slots = rng.permutation(NB * bs)[:T].astype(np.int32)
    slots[T // 2] = -1
    cos, sin = oracle.rope_table(D, max_pos, 1e6)
    d_qkv = nvr.DeviceBuffer(T * QKV * 2)
    d_k, d_v = nvr.DeviceBuffer(NB * bs * KVH * D * 2), nvr.DeviceBuffer(NB * bs * KVH * D * 2)
    d_k.zero(); d_v.zero()
    nvr.check(nvr.lib().nvr_linear_qkv_rope_store(dev(xb).ptr, K, dev(Wb).ptr, T, K, H, KVH, D, dev(pos).ptr, dev(slots).ptr,
                                                  dev(cos).ptr, dev(sin).ptr, d_qkv.ptr, d_k.ptr, d_v.ptr, None))
    got = d_qkv.to_numpy((T, QKV), F16).astype(np.float32)
    qkv = oracle.round_f16(oracle.linear(x, W))
    q = oracle.round_f16(oracle.rope_apply(qkv[:, :H * D].reshape(T, H, D), pos, cos, sin)).reshape(T, H * D)
    kk = oracle.round_f16(oracle.rope_apply(qkv[:, H * D:(H + KVH) * D].reshape(T, KVH, D), pos, cos, sin))
    vv = np.ascontiguousarray(qkv[:, (H + KVH) * D:].reshape(T, KVH, D))
    assert_close_f16(got[:, :H * D], q, ulps=2, atol=6e-3, what="stream fused q")
    kc, vc = np.zeros((NB, bs, KVH, D), np.float32), np.zeros((NB, bs, KVH, D), np.float32)
    oracle.kv_store(kk, vv, slots, kc, vc)
    assert_close_f16(d_k.to_numpy(kc.shape, F16), kc, ulps=2, atol=6e-3, what="stream fused k cache")
    assert_close_f16(d_v.to_numpy(vc.shape, F16), vc, ulps=2, atol=4e-4, what="stream fused v cache")


def test_prefill_gemm_large_repeatable():
    """The 256x256 eight-wave kernel keeps LDS-DMA loads in flight across barriers: screen it for races — many runs
    of a many-workgroup shape must be bit-identical to each other and match the oracle on sampled rows."""
    rng = np.random.default_rng(41)
    T, K, N = 8192, 1024, 2048                            # 8 x 32 = 256 tiles: routed to the 256x256 kernel (> 128 tiles)
    x, xb = h16(rng.standard_normal((T, K)))
    W, Wb = h16(rng.standard_normal((N, K)) * 0.05)
    d_x, d_W, d_y = dev(xb), dev(Wb), nvr.DeviceBuffer(T * N * 2)
    first = None
    for rep in range(12):
        nvr.check(nvr.lib().nvr_linear(d_x.ptr, K, d_W.ptr, T, K, N, d_y.ptr, 0, None))
        got = d_y.to_numpy((T, N), np.uint16)
        if first is None: first = got
        else: assert np.array_equal(got, first), rep
    rows = rng.choice(T, 256, replace=False)
    assert_close_f16(first.view(F16)[rows], oracle.round_f16(oracle.linear(x[rows], W)), ulps=1, atol=2e-4, what="gemm256")
    # same screen for the SiLU epilogue on a persistent grid with more tiles than workgroups (T=8192, I=1024: 8 x 32 tiles x ...)
    T2, I = 8192, 1024
    x2, x2b = h16(rng.standard_normal((T2, K)))
    W2, W2b = h16(rng.standard_normal((2 * I, K)) * 0.05)
    d_x2, d_W2, d_o = dev(x2b), dev(W2b), nvr.DeviceBuffer(T2 * I * 2)
    first = None
    for rep in range(8):
        nvr.check(nvr.lib().nvr_linear_silu_mul(d_x2.ptr, K, d_W2.ptr, T2, K, I, d_o.ptr, None))
        got = d_o.to_numpy((T2, I), np.uint16)
        if first is None: first = got
        else: assert np.array_equal(got, first), rep
    rows = rng.choice(T2, 128, replace=False)
    ref = oracle.round_f16(oracle.silu_and_mul(oracle.round_f16(oracle.linear(x2[rows], W2))))
    assert_close_f16(first.view(F16)[rows], ref, ulps=2, atol=3e-4, what="gemm256 silu")
    # the 128x128 kernel's 4-buffer ring (grids smaller than the chip: counted waits, LDS-DMA in flight across barriers) and
    # its k-split: the same screen at a 48-tile shape
    T3, K3, N3 = 384, 2048, 2048
    x3, x3b = h16(rng.standard_normal((T3, K3)))
    W3, W3b = h16(rng.standard_normal((N3, K3)) * 0.05)
    d_x3, d_W3, d_y3 = dev(x3b), dev(W3b), nvr.DeviceBuffer(T3 * N3 * 2)
    d_sl = nvr.DeviceBuffer(4 * T3 * N3 * 4)
    first = first_sl = None
    for rep in range(12):
        nvr.check(nvr.lib().nvr_linear(d_x3.ptr, K3, d_W3.ptr, T3, K3, N3, d_y3.ptr, 0, None))
        nvr.check(nvr.lib().nvr_linear_splitk(d_x3.ptr, K3, d_W3.ptr, T3, K3, N3, 4, d_sl.ptr, None))
        got, gsl = d_y3.to_numpy((T3, N3), np.uint16), d_sl.to_numpy((4, T3, N3), np.float32)
        if first is None: first, first_sl = got, gsl
        else: assert np.array_equal(got, first) and np.array_equal(gsl, first_sl), rep
    ref3 = oracle.linear(x3, W3)
    assert_close_f16(first.view(F16), oracle.round_f16(ref3), ulps=1, atol=3e-4, what="gemm_tiled ring")
    np.testing.assert_allclose(first_sl.sum(0), ref3, rtol=2e-5, atol=6e-4)


# ------------------------------------------------------------------------------------------- fused epilogues
@pytest.mark.parametrize("T,K,I", [(32, 1024, 3072), (5, 256, 64), (40, 512, 128), (200, 1024, 3072), (129, 256, 64),
                                   (300, 1024, 3072), (517, 512, 128), (256, 128, 384),
                                   # shards of a tensor-parallel rank (few column tiles): 16-token workgroups, 8 / 16 waves
                                   (32, 1024, 384), (23, 4096, 1536), (32, 2048, 96)])
def test_linear_silu_mul_fused_equals_unfused_graph(T, K, I):
    """gate_up GEMM + SiluAndMul in one launch must equal linear -> fp16 -> silu_and_mul -> fp16 (oracle order)."""
    rng = np.random.default_rng(20)
    x, xb = h16(rng.standard_normal((T, K)))
    W, Wb = h16(rng.standard_normal((2 * I, K)) * 0.05)
    d_out = nvr.DeviceBuffer(T * I * 2)
    nvr.check(nvr.lib().nvr_linear_silu_mul(dev(xb).ptr, K, dev(Wb).ptr, T, K, I, d_out.ptr, None))
    ref = oracle.round_f16(oracle.silu_and_mul(oracle.round_f16(oracle.linear(x, W))))
    assert_close_f16(d_out.to_numpy((T, I), F16), ref, ulps=2, atol=3e-4, what="linear+silu_mul")


@pytest.mark.parametrize("T,K,H,KVH,D", [(32, 1024, 16, 8, 128), (7, 256, 4, 2, 64), (33, 512, 2, 2, 128), (130, 1024, 16, 8, 128),
                                         (300, 256, 4, 2, 64), (128, 512, 2, 2, 128), (600, 1024, 16, 8, 128), (257, 128, 2, 1, 128), (8192, 512, 16, 8, 128),
                                         # shards of a tensor-parallel rank: 16-token workgroups, 8 / 16 waves
                                         (32, 1024, 2, 1, 128), (29, 4096, 4, 1, 128), (32, 2048, 4, 2, 64)])
def test_linear_qkv_rope_store_fused(T, K, H, KVH, D):
    rng = np.random.default_rng(21)
    NB, bs, max_pos = max(24, T // 16 + 2), 16, 300          # the 8192-token case has >= 256 output tiles: XCD-aware tile order
    QKV = (H + 2 * KVH) * D
    x, xb = h16(rng.standard_normal((T, K)))
    W, Wb = h16(rng.standard_normal((QKV, K)) * 0.05)
    pos = rng.integers(0, max_pos, T).astype(np.int64)
    slots = rng.permutation(NB * bs)[:T].astype(np.int32)
    slots[T // 2] = -1
    cos, sin = oracle.rope_table(D, max_pos, 1e6)
    d_qkv = nvr.DeviceBuffer(T * QKV * 2)
    d_k, d_v = nvr.DeviceBuffer(NB * bs * KVH * D * 2), nvr.DeviceBuffer(NB * bs * KVH * D * 2)
    d_k.zero(); d_v.zero()
    nvr.check(nvr.lib().nvr_linear_qkv_rope_store(dev(xb).ptr, K, dev(Wb).ptr, T, K, H, KVH, D, dev(pos).ptr, dev(slots).ptr,
                                                  dev(cos).ptr, dev(sin).ptr, d_qkv.ptr, d_k.ptr, d_v.ptr, None))
    got = d_qkv.to_numpy((T, QKV), F16).astype(np.float32)
    # the unfused kernels on the same inputs are the bit-exact twin (same GEMM tile order, same rope math)
    d_qkv2 = nvr.DeviceBuffer(T * QKV * 2)
    d_k2, d_v2 = nvr.DeviceBuffer(NB * bs * KVH * D * 2), nvr.DeviceBuffer(NB * bs * KVH * D * 2)
    d_k2.zero(); d_v2.zero()
    nvr.check(nvr.lib().nvr_linear(dev(xb).ptr, K, dev(Wb).ptr, T, K, QKV, d_qkv2.ptr, 0, None))
    nvr.check(nvr.lib().nvr_rope_store_kv(d_qkv2.ptr, dev(pos).ptr, dev(slots).ptr, T, H, KVH, D, dev(cos).ptr, dev(sin).ptr,
                                          d_k2.ptr, d_v2.ptr, None))
    got2 = d_qkv2.to_numpy((T, QKV), F16).astype(np.float32)
    assert np.array_equal(got, got2), "fused qkv+rope differs from linear followed by rope_store_kv"
    assert np.array_equal(d_k.to_numpy((NB * bs, KVH * D), F16).view(np.uint16), d_k2.to_numpy((NB * bs, KVH * D), F16).view(np.uint16))
    assert np.array_equal(d_v.to_numpy((NB * bs, KVH * D), F16).view(np.uint16), d_v2.to_numpy((NB * bs, KVH * D), F16).view(np.uint16))
    # and against the oracle composition
    qkv = oracle.round_f16(oracle.linear(x, W))
    q = oracle.round_f16(oracle.rope_apply(qkv[:, :H * D].reshape(T, H, D), pos, cos, sin)).reshape(T, H * D)
    kk = oracle.round_f16(oracle.rope_apply(qkv[:, H * D:(H + KVH) * D].reshape(T, KVH, D), pos, cos, sin))
    vv = np.ascontiguousarray(qkv[:, (H + KVH) * D:].reshape(T, KVH, D))
    # a 1-ulp fp16 difference in a GEMM output (f32 summation order) passes through the rotation, where the
    # result can be much smaller than its inputs: the tolerance is absolute, one fp16 ulp of the inputs (|x| < 4)
    assert_close_f16(got[:, :H * D], q, ulps=2, atol=4e-3, what="fused q")
    kc, vc = np.zeros((NB, bs, KVH, D), np.float32), np.zeros((NB, bs, KVH, D), np.float32)
    oracle.kv_store(kk, vv, slots, kc, vc)
    assert_close_f16(d_k.to_numpy(kc.shape, F16), kc, ulps=2, atol=4e-3, what="fused k cache")
    assert_close_f16(d_v.to_numpy(vc.shape, F16), vc, ulps=2, atol=3e-4, what="fused v cache")


def _retiled(d_W, N, K, mode=0, H=0, KVH=0, D=0):
    """Device pointer of the nvr_retile_weight copy of d_W [N, K]."""
    d_T = nvr.DeviceBuffer(N * K * 2)
    _KEEP.append(d_T)
    nvr.check(nvr.lib().nvr_retile_weight(d_W.ptr, d_T.ptr, N, K, mode, H, KVH, D, None))
    return d_T.ptr


@pytest.mark.parametrize("T,K,N,S", [(32, 2048, 1024, 4), (32, 3072, 1024, 4), (5, 256, 64, 2), (40, 512, 256, 4),
                                     (32, 4096, 4096, 4), (9, 12288, 4096, 4), (32, 4096, 8192, 2),   # large weights: 64-column workgroups, wide rows
                                     (130, 2048, 1024, 4), (300, 3072, 1024, 4), (70, 512, 256, 2), (512, 1024, 2048, 4),   # > 64 rows: k-split of the 128x128 kernel
                                     (64, 4096, 4096, 8), (40, 12288, 4096, 8), (48, 4096, 4096, 4),     # 33..64 rows over large weights: the tiles too (r06), 8 slices
                                     (40, 1024, 256, 8), (33, 2048, 2048, 8)])                           # 8 slices on the streaming kernel, every slab-norm width
def test_linear_splitk_and_slab_norm(T, K, N, S):
    """split-k slabs + add_rmsnorm_slabs == linear -> fp16 -> add -> rmsnorm (the unfused graph order)."""
    rng = np.random.default_rng(22)
    x, xb = h16(rng.standard_normal((T, K)))
    W, Wb = h16(rng.standard_normal((N, K)) * 0.05)
    h, hb = h16(rng.standard_normal((T, N)))
    w, wb = h16(1 + 0.1 * rng.standard_normal(N))
    d_slabs = nvr.DeviceBuffer(S * T * N * 4)
    nvr.check(nvr.lib().nvr_linear_splitk(dev(xb).ptr, K, dev(Wb).ptr, T, K, N, S, d_slabs.ptr, None))
    slabs = d_slabs.to_numpy((S, T, N), np.float32)
    ref = oracle.linear(x, W)
    np.testing.assert_allclose(slabs.sum(0), ref, rtol=2e-5, atol=3e-4)
    d_h, d_out = dev(hb), nvr.DeviceBuffer(T * N * 2)
    nvr.check(nvr.lib().nvr_add_rmsnorm_slabs(d_h.ptr, d_slabs.ptr, S, dev(wb).ptr, 1e-6, T, N, d_out.ptr, None))
    y = slabs[0].copy()
    for z in range(1, S):
        y = y + slabs[z]                                    # same f32 order as the kernel
    hn = oracle.add(h, oracle.round_f16(y), round16=True)
    assert np.array_equal(d_h.to_numpy((T, N), F16).astype(np.float32), hn)
    assert_close_f16(d_out.to_numpy((T, N), F16), oracle.round_f16(oracle.rmsnorm(hn, w, 1e-6)), ulps=1, atol=1e-6)
    assert nvr.lib().nvr_linear_splitk(dev(xb).ptr, K, dev(Wb).ptr, T, K, N, 5, d_slabs.ptr, None) == -10


# ------------------------------------------------------------------------------------------- K8
@pytest.mark.parametrize("H,KVH,D,bs,cases", [
    (16, 8, 128, 16, [(40, 9), (100, 100), (33, 1)]),      # (context_len, new tokens): cached prefix + new, all new, decode-like
    (4, 2, 64, 16, [(70, 6), (16, 16)]),
    (8, 2, 128, 256, [(300, 44)]),
    (4, 2, 64, 64, [(150, 22), (150, 150), (200, 60), (64, 1)]),   # block size = the kernel's key step, several sequences (register copy of the table per tile)
    (16, 8, 128, 128, [(700, 130), (129, 129)]),
])
def test_attn_prefill_paged_prefix(H, KVH, D, bs, cases):
    """Prefix-cached prefill (attention.rs:211-222): queries are the last nq tokens of each context, K/V via block table."""
    rng = np.random.default_rng(23)
    B = len(cases)
    ctxs = [c for c, _ in cases]; nqs = [n for _, n in cases]
    NB = sum((c + bs - 1) // bs for c in ctxs) + 2
    kc, kcb, vc, vcb, bt, max_blocks = _paged_case(rng, B, H, KVH, D, bs, ctxs, NB)
    T = sum(nqs)
    q, qb = h16(rng.standard_normal((T, H, D)))
    cu = np.concatenate([[0], np.cumsum(nqs)]).astype(np.int32)
    ctx = np.asarray(ctxs, np.int32)
    scale = float(np.float32(1.0) / np.sqrt(np.float32(D)))
    meta = nvr.AttnMetaC()
    d_cu, d_ctx, d_bt = dev(cu), dev(ctx), dev(bt)
    meta.is_prefill, meta.cu_seqlens_q, meta.context_lens, meta.block_tables = 1, d_cu.ptr, d_ctx.ptr, d_bt.ptr
    meta.max_blocks, meta.batch, meta.max_context_len = max_blocks, B, int(max(ctxs))
    d_out = nvr.DeviceBuffer(T * H * D * 2)
    nvr.check(nvr.lib().nvr_attn_prefill_paged(dev(qb).ptr, H * D, dev(kcb).ptr, dev(vcb).ptr, C.byref(meta), T, H, KVH, D, bs, scale,
                                               d_out.ptr, None))
    ref = oracle.round_f16(oracle.attn_paged(q, cu, kc, vc, bt, ctx, scale))
    assert_close_f16(d_out.to_numpy((T, H, D), F16), ref, ulps=2, atol=2e-3, what="paged prefix prefill attention")


# ------------------------------------------------------------------------------------------- randomized sweeps
def test_paged_attn_decode_random_geometries():
    """40 random (batch, context, block size incl. non powers of two, head grouping, head dim) cases, ragged contexts."""
    rng = np.random.default_rng(99)
    for case in range(40):
        D = int(rng.choice([64, 128])); G = int(rng.choice([1, 2, 4, 8])); KVH = int(rng.choice([1, 2, 8])); H = G * KVH
        bs = int(rng.choice([16, 48, 256, 64])); B = int(rng.integers(1, 40))
        ctxs = rng.integers(1, int(rng.choice([40, 300, 1500])), B).tolist()
        NB = sum((c + bs - 1) // bs for c in ctxs) + 2
        kc, kcb, vc, vcb, bt, max_blocks = _paged_case(rng, B, H, KVH, D, bs, ctxs, NB)
        q, qb = h16(rng.standard_normal((B, H, D)))
        ctx = np.asarray(ctxs, np.int32)
        scale = float(np.float32(1.0) / np.sqrt(np.float32(D)))
        meta = nvr.AttnMetaC()
        d_ctx, d_bt = dev(ctx), dev(bt)
        meta.context_lens, meta.block_tables, meta.max_blocks, meta.batch, meta.max_context_len = d_ctx.ptr, d_bt.ptr, max_blocks, B, int(max(ctxs))
        ws = nvr.DeviceBuffer(nvr.lib().nvr_paged_attn_workspace_bytes(B, H, D, int(max(ctxs)))); _KEEP.append(ws)
        d_out = nvr.DeviceBuffer(B * H * D * 2); _KEEP.append(d_out)
        nvr.check(nvr.lib().nvr_paged_attn_decode(dev(qb).ptr, H * D, dev(kcb).ptr, dev(vcb).ptr, C.byref(meta), H, KVH, D, bs, scale,
                                                  d_out.ptr, ws.ptr, None))
        ref = oracle.round_f16(oracle.attn_decode(q, kc, vc, bt, ctx, scale))
        got = d_out.to_numpy((B, H, D), F16)
        assert_close_f16(got, ref, ulps=2, atol=1e-3, what=f"case {case}: B={B} H={H} KVH={KVH} D={D} bs={bs}")
        _fused_merge_equals_two_launches(qb, kcb, vcb, meta, B, H, KVH, D, bs, scale, ws, got)
        _KEEP.clear()


def test_linear_random_shapes():
    rng = np.random.default_rng(98)
    for case in range(30):
        T = int(rng.choice([1, 3, 16, 17, 32, 33, 64, 65, 127, 128, 129, 300]))
        K = int(rng.choice([32, 64, 96, 256, 384, 1024, 2048, 3072])); N = 16 * int(rng.integers(1, 80))
        x, xb = h16(rng.standard_normal((T, K)))
        W, Wb = h16(rng.standard_normal((N, K)) * 0.05)
        d_y = nvr.DeviceBuffer(T * N * 2); _KEEP.append(d_y)
        rc = nvr.lib().nvr_linear(dev(xb).ptr, K, dev(Wb).ptr, T, K, N, d_y.ptr, 0, None)
        assert rc == 0, nvr.last_error()
        assert_close_f16(d_y.to_numpy((T, N), F16), oracle.round_f16(oracle.linear(x, W)), ulps=1, atol=3e-4, what=f"case {case}: T={T} K={K} N={N}")
        _KEEP.clear()


@pytest.mark.parametrize("N,K,mode,H,KVH,D", [(64, 128, 0, 0, 0, 0), (1024, 3072, 0, 0, 0, 0), (4096, 1024, 1, 16, 8, 128), (512, 256, 1, 4, 2, 64)])
def test_retile_weight_layout(N, K, mode, H, KVH, D):
    """nvr_retile_weight against a numpy restatement of the tiled layout [N/16][K/32][16][32] (mode 1: the qkv row order of the
    RoPE epilogue: first-half / second-half rotation partners of a head share a tile)."""
    rng = np.random.default_rng(3)
    W = rng.integers(0, 65535, (N, K)).astype(np.uint16)
    d_src, d_dst = dev(W), nvr.DeviceBuffer(N * K * 2)
    _KEEP.append(d_dst)
    nvr.check(nvr.lib().nvr_retile_weight(d_src.ptr, d_dst.ptr, N, K, mode, H, KVH, D, None))
    got = d_dst.to_numpy((N // 16, K // 32, 16, 32), np.uint16)
    rows = np.arange(N).reshape(N // 16, 16)
    if mode == 1:
        tph = D // 16
        for t in range(N // 16):
            head, c = t // tph, t % tph
            if head < H + KVH:
                rows[t] = [head * D + (c * 8 + r if r < 8 else D // 2 + c * 8 + r - 8) for r in range(16)]
    ref = W[rows].reshape(N // 16, 16, K // 32, 32).transpose(0, 2, 1, 3)
    assert np.array_equal(got, ref)
    assert nvr.lib().nvr_retile_weight(d_src.ptr, d_dst.ptr, N, K + 8, mode, H, KVH, D, None) == -10


@pytest.mark.parametrize("T", [1, 32, 64, 200])
def test_tiled_entry_points_match_row_major(T):
    """Every *_tiled entry point against its row-major twin on the Qwen3-0.6B shapes: identical bits (the tiled copy changes where
    a weight byte lives, never which bytes meet in an MFMA or in which order they are summed); T = 200 takes the tile GEMMs, which
    read the row-major W and ignore Wt."""
    rng = np.random.default_rng(17 + T)
    l = nvr.lib()
    Hd, H, KVH, D, I = 1024, 16, 8, 128, 3072
    QKV = (H + 2 * KVH) * D

    def tiled(Wb, N, K, mode):
        d_W, d_T = dev(Wb), nvr.DeviceBuffer(N * K * 2)
        _KEEP.append(d_T)
        nvr.check(l.nvr_retile_weight(d_W.ptr, d_T.ptr, N, K, mode, H, KVH, D, None))
        return d_W, d_T
    # plain linear and split-k (o_proj shape)
    _, xb = h16(rng.standard_normal((T, H * D)) * 0.3)
    _, Wb = h16(rng.standard_normal((Hd, H * D)) * 0.05)
    d_x = dev(xb); d_W, d_T = tiled(Wb, Hd, H * D, 0)
    ya, yb = nvr.DeviceBuffer(T * Hd * 2), nvr.DeviceBuffer(T * Hd * 2)
    nvr.check(l.nvr_linear(d_x.ptr, H * D, d_W.ptr, T, H * D, Hd, ya.ptr, 0, None))
    nvr.check(l.nvr_linear_tiled(d_x.ptr, H * D, d_W.ptr, d_T.ptr, T, H * D, Hd, yb.ptr, 0, None))
    assert np.array_equal(ya.to_numpy((T, Hd), np.uint16), yb.to_numpy((T, Hd), np.uint16))
    S = l.nvr_decode_splitk_slices(T, H * D, Hd)
    if S > 0:
        sa, sb = nvr.DeviceBuffer(S * T * Hd * 4), nvr.DeviceBuffer(S * T * Hd * 4)
        nvr.check(l.nvr_linear_splitk(d_x.ptr, H * D, d_W.ptr, T, H * D, Hd, S, sa.ptr, None))
        nvr.check(l.nvr_linear_splitk_tiled(d_x.ptr, H * D, d_W.ptr, d_T.ptr, T, H * D, Hd, S, sb.ptr, None))
        assert np.array_equal(sa.to_numpy((S, T, Hd), np.uint32), sb.to_numpy((S, T, Hd), np.uint32))
    # gate_up + SiLU·mul
    _, xb = h16(rng.standard_normal((T, Hd)) * 0.5)
    _, Wb = h16(rng.standard_normal((2 * I, Hd)) * 0.05)
    d_x = dev(xb); d_W, d_T = tiled(Wb, 2 * I, Hd, 0)
    oa, ob = nvr.DeviceBuffer(T * I * 2), nvr.DeviceBuffer(T * I * 2)
    nvr.check(l.nvr_linear_silu_mul(d_x.ptr, Hd, d_W.ptr, T, Hd, I, oa.ptr, None))
    nvr.check(l.nvr_linear_silu_mul_tiled(d_x.ptr, Hd, d_W.ptr, d_T.ptr, T, Hd, I, ob.ptr, None))
    assert np.array_equal(oa.to_numpy((T, I), np.uint16), ob.to_numpy((T, I), np.uint16))
    # qkv + RoPE + KV store (mode 1 copy)
    _, Wb = h16(rng.standard_normal((QKV, Hd)) * 0.05)
    d_W, d_T = tiled(Wb, QKV, Hd, 1)
    pos = rng.integers(0, 512, T).astype(np.int64); slots = rng.permutation(256)[:T].astype(np.int32)
    cos, sin = oracle.rope_table(D, 512, 1e6)
    d_pos, d_slots, d_cos, d_sin = dev(pos), dev(slots), dev(cos), dev(sin)
    outs = []
    for fn, extra in ((l.nvr_linear_qkv_rope_store, ()), (l.nvr_linear_qkv_rope_store_tiled, (d_T.ptr,))):
        q, kc, vc = nvr.DeviceBuffer(T * QKV * 2), dev(np.zeros((256, KVH, D), np.uint16)), dev(np.zeros((256, KVH, D), np.uint16))
        nvr.check(fn(d_x.ptr, Hd, d_W.ptr, *extra, T, Hd, H, KVH, D, d_pos.ptr, d_slots.ptr, d_cos.ptr, d_sin.ptr, q.ptr, kc.ptr, vc.ptr, None))
        outs.append((q.to_numpy((T, QKV), np.uint16), kc.to_numpy((256, KVH, D), np.uint16), vc.to_numpy((256, KVH, D), np.uint16)))
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
    # LM head (+ arg-max partials)
    if T <= 64:
        N = 151936 // 4
        _, Wb = h16(rng.standard_normal((N, Hd)) * 0.05)
        d_W, d_T = tiled(Wb, N, Hd, 0)
        res = []
        for fn, extra in ((l.nvr_lm_head, ()), (l.nvr_lm_head_tiled, (d_T.ptr,))):
            y, pv, pi = nvr.DeviceBuffer(T * N * 4), nvr.DeviceBuffer(2048 * T * 4), nvr.DeviceBuffer(2048 * T * 4)
            nparts = C.c_int32(0)
            nvr.check(fn(d_x.ptr, Hd, d_W.ptr, *extra, T, Hd, N, y.ptr, pv.ptr, pi.ptr, C.byref(nparts), None))
            tok = nvr.DeviceBuffer(T * 8)
            nvr.check(l.nvr_argmax_partials(pv.ptr, pi.ptr, nparts.value, T, tok.ptr, None, 0, None))
            res.append((y.to_numpy((T, N), np.uint32), tok.to_numpy((T,), np.int64)))
        assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])


@pytest.mark.parametrize("T,H,KVH,D", [(5, 4, 2, 64), (33, 16, 8, 128), (1, 2, 2, 128), (300, 4, 4, 64)])
def test_qk_norm_rope_store(T, H, KVH, D):
    """nvr_qk_norm_rope_store_kv (A-27): RMSNorm over head_dim on the q and k heads, then RoPE and the cache store, against
    the oracle ops in that order (fp16 between them); v rows and slots as in nvr_rope_store_kv."""
    rng = np.random.default_rng(41)
    N = (H + 2 * KVH) * D
    x, xb = h16(rng.standard_normal((T, N)) * 1.5)
    qw, qwb = h16(1 + 0.3 * rng.standard_normal(D)); kw, kwb = h16(1 + 0.3 * rng.standard_normal(D))
    pos = rng.integers(0, 400, T).astype(np.int64)
    nslots = 2 * T + 4
    slots = rng.permutation(nslots)[:T].astype(np.int32)
    if T > 2:
        slots[2] = -1
    cos, sin = oracle.rope_table(D, 512, 1e6)
    d_x, d_kc, d_vc = dev(xb.copy()), nvr.DeviceBuffer(nslots * KVH * D * 2), nvr.DeviceBuffer(nslots * KVH * D * 2)
    _KEEP.extend([d_kc, d_vc]); d_kc.zero(); d_vc.zero()
    nvr.check(nvr.lib().nvr_qk_norm_rope_store_kv(d_x.ptr, dev(pos).ptr, dev(slots).ptr, T, H, KVH, D, dev(cos).ptr, dev(sin).ptr,
                                                  dev(qwb).ptr, dev(kwb).ptr, 1e-6, d_kc.ptr, d_vc.ptr, None))
    got = d_x.to_numpy((T, N), F16)
    q = oracle.round_f16(oracle.rmsnorm(x[:, :H * D].reshape(T * H, D), qw, 1e-6)).reshape(T, H, D)
    k = oracle.round_f16(oracle.rmsnorm(x[:, H * D:(H + KVH) * D].reshape(T * KVH, D), kw, 1e-6)).reshape(T, KVH, D)
    q = oracle.round_f16(oracle.rope_apply(q, pos, cos, sin)).reshape(T, H * D)
    k = oracle.round_f16(oracle.rope_apply(k, pos, cos, sin)).reshape(T, KVH * D)
    ref = np.concatenate([q, k, x[:, (H + KVH) * D:]], 1)
    assert_close_f16(got, ref, ulps=3, atol=2e-3, what="norm + rope")
    assert np.array_equal(got.view(np.uint16)[:, (H + KVH) * D:], xb[:, (H + KVH) * D:].view(np.uint16))      # v untouched
    kc, vc = d_kc.to_numpy((nslots, KVH * D), np.uint16), d_vc.to_numpy((nslots, KVH * D), np.uint16)
    gb = got.view(np.uint16)
    seen = np.zeros(nslots, bool)
    for t in range(T):
        if slots[t] >= 0:
            assert np.array_equal(kc[slots[t]], gb[t, H * D:(H + KVH) * D]) and np.array_equal(vc[slots[t]], gb[t, (H + KVH) * D:])
            seen[slots[t]] = True
    assert not kc[~seen].any() and not vc[~seen].any()
    assert nvr.lib().nvr_qk_norm_rope_store_kv(d_x.ptr, dev(pos).ptr, dev(slots).ptr, T, H, KVH, D, dev(cos).ptr, dev(sin).ptr,
                                               None, dev(kwb).ptr, 1e-6, d_kc.ptr, d_vc.ptr, None) == -7


@pytest.mark.parametrize("B,H,KVH,D,bs,P,own", [
    (70, 16, 8, 128, 256, 2, [int(x) for x in np.random.default_rng(1).integers(0, 300, 70)]),   # configs[4] shape: 512 shared tokens
    (5, 4, 2, 64, 64, 1, [0, 1, 63, 64, 130]),          # nobody / somebody without own tokens, partial blocks, D=64
    (130, 8, 8, 128, 64, 3, [(7 * i) % 200 for i in range(130)]),   # group 1: 128 sequences per workgroup + a ragged last tile
    (33, 8, 2, 128, 128, 1, [5 + i for i in range(33)]),            # group 4
    (2, 16, 8, 128, 256, 1, [0, 0]),                    # only the shared partition exists
    (512, 16, 8, 128, 64, 1, [(11 * i) % 120 for i in range(512)]),   # 4096 (sequence, kv head) pairs: ONE-wave workgroups stream the own keys (r05)
    (300, 16, 8, 128, 64, 1, [(13 * i) % 125 for i in range(300)]),   # 2400 pairs: two-wave workgroups
])
def test_paged_attn_decode_shared_prefix(B, H, KVH, D, bs, P, own):
    """nvr_paged_attn_decode_shared: every sequence's first P blocks are the SAME cache blocks (prefix-cache hits); the shared
    keys go through the MFMA kernel once for the batch, the rest through the row kernel, merged as split-KV partials — against
    the oracle's plain paged attention and against nvr_paged_attn_decode on the same inputs."""
    rng = np.random.default_rng(12)
    S = P * bs
    ctxs = [S + o for o in own]
    own_blocks = [(o + bs - 1) // bs for o in own]
    NB = P + sum(own_blocks) + 2
    max_blocks = P + max(own_blocks) + 1
    kc, kcb = h16(rng.standard_normal((NB, bs, KVH, D)))
    vc, vcb = h16(rng.standard_normal((NB, bs, KVH, D)))
    perm = rng.permutation(NB)
    bt = -np.ones((B, max_blocks), np.int32)
    o = P
    for b in range(B):
        bt[b, :P] = perm[:P]
        bt[b, P:P + own_blocks[b]] = perm[o:o + own_blocks[b]]; o += own_blocks[b]
    q, qb = h16(rng.standard_normal((B, H, D)))
    ctx = np.asarray(ctxs, np.int32)
    scale = float(np.float32(1.0) / np.sqrt(np.float32(D)))
    meta = nvr.AttnMetaC()
    d_ctx, d_bt = dev(ctx), dev(bt)
    meta.is_prefill, meta.context_lens, meta.block_tables = 0, d_ctx.ptr, d_bt.ptr
    meta.max_blocks, meta.batch, meta.max_context_len = max_blocks, B, int(max(ctxs))
    ws = nvr.DeviceBuffer(nvr.lib().nvr_paged_attn_workspace_bytes(B, H, D, int(max(ctxs)) + bs))
    d_out, d_plain = nvr.DeviceBuffer(B * H * D * 2), nvr.DeviceBuffer(B * H * D * 2)
    d_q, d_k, d_v = dev(qb), dev(kcb), dev(vcb)
    nvr.check(nvr.lib().nvr_paged_attn_decode_shared(d_q.ptr, H * D, d_k.ptr, d_v.ptr, C.byref(meta), H, KVH, D, bs, scale, S,
                                                     None, None, None, d_out.ptr, ws.ptr, None))
    nvr.check(nvr.lib().nvr_paged_attn_decode(d_q.ptr, H * D, d_k.ptr, d_v.ptr, C.byref(meta), H, KVH, D, bs, scale, d_plain.ptr, ws.ptr, None))
    got = d_out.to_numpy((B, H, D), F16)
    ref = oracle.round_f16(oracle.attn_decode(q, kc, vc, bt, ctx, scale))
    assert_close_f16(got, ref, ulps=2, atol=1e-3, what="shared-prefix decode attention vs oracle")
    assert_close_f16(got, d_plain.to_numpy((B, H, D), F16), ulps=3, atol=1e-3, what="shared-prefix vs plain kernel")
    # shared_len that is no multiple of the block size is refused
    assert nvr.lib().nvr_paged_attn_decode_shared(d_q.ptr, H * D, d_k.ptr, d_v.ptr, C.byref(meta), H, KVH, D, bs, scale, S + 8,
                                                  None, None, None, d_out.ptr, ws.ptr, None) == -7


@pytest.mark.parametrize("T", [8, 32, 40, 64])
def test_tiled_entry_points_match_row_major_stream_shapes(T):
    """The same on Qwen3-8B shapes, where linear / silu / qkv route to linear_stream_kernel (>= 24 MiB of weights, K = 4096): the
    tiled copy must give the bits of the row-major weights there too."""
    rng = np.random.default_rng(23 + T)
    l = nvr.lib()
    Hd, H, KVH, D, I = 4096, 32, 8, 128, 12288
    QKV = (H + 2 * KVH) * D

    def tiled(Wb, N, K, mode):
        d_W, d_T = dev(Wb), nvr.DeviceBuffer(N * K * 2)
        _KEEP.append(d_T)
        nvr.check(l.nvr_retile_weight(d_W.ptr, d_T.ptr, N, K, mode, H, KVH, D, None))
        return d_W, d_T
    _, xb = h16(rng.standard_normal((T, Hd)) * 0.5)
    d_x = dev(xb)
    Wb = (rng.standard_normal((QKV, Hd)) * 0.03).astype(F16)
    # plain
    d_W, d_T = tiled(Wb, QKV, Hd, 0)
    ya, yb = nvr.DeviceBuffer(T * QKV * 2), nvr.DeviceBuffer(T * QKV * 2)
    nvr.check(l.nvr_linear(d_x.ptr, Hd, d_W.ptr, T, Hd, QKV, ya.ptr, 0, None))
    nvr.check(l.nvr_linear_tiled(d_x.ptr, Hd, d_W.ptr, d_T.ptr, T, Hd, QKV, yb.ptr, 0, None))
    assert np.array_equal(ya.to_numpy((T, QKV), np.uint16), yb.to_numpy((T, QKV), np.uint16))
    # qkv + RoPE + KV store (mode 1 copy)
    d_W, d_T = tiled(Wb, QKV, Hd, 1)
    pos = rng.integers(0, 512, T).astype(np.int64); slots = rng.permutation(64)[:T].astype(np.int32)
    cos, sin = oracle.rope_table(D, 512, 1e6)
    d_pos, d_slots, d_cos, d_sin = dev(pos), dev(slots), dev(cos), dev(sin)
    outs = []
    for fn, extra in ((l.nvr_linear_qkv_rope_store, ()), (l.nvr_linear_qkv_rope_store_tiled, (d_T.ptr,))):
        q, kc, vc = nvr.DeviceBuffer(T * QKV * 2), dev(np.zeros((64, KVH, D), np.uint16)), dev(np.zeros((64, KVH, D), np.uint16))
        nvr.check(fn(d_x.ptr, Hd, d_W.ptr, *extra, T, Hd, H, KVH, D, d_pos.ptr, d_slots.ptr, d_cos.ptr, d_sin.ptr, q.ptr, kc.ptr, vc.ptr, None))
        outs.append((q.to_numpy((T, QKV), np.uint16), kc.to_numpy((64, KVH, D), np.uint16), vc.to_numpy((64, KVH, D), np.uint16)))
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
    # gate_up + SiLU·mul
    Wg = (rng.standard_normal((2 * I, Hd)) * 0.03).astype(F16)
    d_W, d_T = tiled(Wg, 2 * I, Hd, 0)
    oa, ob = nvr.DeviceBuffer(T * I * 2), nvr.DeviceBuffer(T * I * 2)
    nvr.check(l.nvr_linear_silu_mul(d_x.ptr, Hd, d_W.ptr, T, Hd, I, oa.ptr, None))
    nvr.check(l.nvr_linear_silu_mul_tiled(d_x.ptr, Hd, d_W.ptr, d_T.ptr, T, Hd, I, ob.ptr, None))
    assert np.array_equal(oa.to_numpy((T, I), np.uint16), ob.to_numpy((T, I), np.uint16))
    # (and against the oracle, a few columns)
    ref = oracle.round_f16(oracle.silu_and_mul(oracle.round_f16(oracle.linear(xb.astype(np.float32), np.concatenate([Wg[:64], Wg[I:I + 64]]).astype(np.float32)))))
    assert_close_f16(ob.to_numpy((T, I), F16)[:, :64], ref, ulps=2, atol=3e-4, what="stream silu tiled vs oracle")


@pytest.mark.parametrize("T", [70, 100, 200, 352, 400, 512, 600])
def test_gemm_tiled_tile_heights(T):
    """Decode batches of 65..512 rows (and beyond) take the LDS-tiled GEMM with 32-, 64- or 128-token tiles (the largest that reaches ~192
    workgroups: T = 70 / 100 -> 32-token tiles for qkv, T = 512 -> 64-token tiles; gate_up at 385..512 rows -> the 96-row x 128-token
    SiLU tiles of r05, 256 workgroups instead of 192; T = 400: a ragged last token tile; r06: 257..511 ring workgroups of 64 tokens become
    half as many 128-token tiles — T = 352: gate_up on three 96-row tiles, the last one ragged; T = 600: qkv and split-k on five 128-token
    tiles): plain, SiLU, RoPE + KV store and split-k
    epilogues on the Qwen3-0.6B shapes against the oracle, and the fused gate_up + SiluAndMul against the plain GEMM followed by
    nvr_silu_and_mul bit for bit (every tiling keeps the K order of an output)."""
    rng = np.random.default_rng(60 + T)
    l = nvr.lib()
    Hd, H, KVH, D, I = 1024, 16, 8, 128, 3072
    QKV = (H + 2 * KVH) * D
    x, xb = h16(rng.standard_normal((T, Hd)) * 0.5)
    d_x = dev(xb)
    # plain
    W, Wb = h16(rng.standard_normal((QKV, Hd)) * 0.05)
    d_W = dev(Wb)
    d_y = nvr.DeviceBuffer(T * QKV * 2)
    nvr.check(l.nvr_linear(d_x.ptr, Hd, d_W.ptr, T, Hd, QKV, d_y.ptr, 0, None))
    lin = oracle.round_f16(oracle.linear(x, W))
    assert_close_f16(d_y.to_numpy((T, QKV), F16), lin, ulps=1, atol=3e-4, what="plain")
    # qkv + RoPE + store
    pos = rng.integers(0, 500, T).astype(np.int64)
    nslots = T + 9
    slots = rng.permutation(nslots)[:T].astype(np.int32)
    cos, sin = oracle.rope_table(D, 512, 1e6)
    d_q, d_kc, d_vc = nvr.DeviceBuffer(T * QKV * 2), nvr.DeviceBuffer(nslots * KVH * D * 2), nvr.DeviceBuffer(nslots * KVH * D * 2)
    _KEEP.extend([d_q, d_kc, d_vc]); d_kc.zero(); d_vc.zero()
    nvr.check(l.nvr_linear_qkv_rope_store(d_x.ptr, Hd, d_W.ptr, T, Hd, H, KVH, D, dev(pos).ptr, dev(slots).ptr, dev(cos).ptr, dev(sin).ptr,
                                          d_q.ptr, d_kc.ptr, d_vc.ptr, None))
    q = oracle.round_f16(oracle.rope_apply(lin[:, :H * D].reshape(T, H, D), pos, cos, sin)).reshape(T, H * D)
    k = oracle.round_f16(oracle.rope_apply(lin[:, H * D:(H + KVH) * D].reshape(T, KVH, D), pos, cos, sin)).reshape(T, KVH * D)
    got = d_q.to_numpy((T, QKV), F16)
    assert_close_f16(got, np.concatenate([q, k, lin[:, (H + KVH) * D:]], 1), ulps=2, atol=3e-3, what="rope")
    kc, gb = d_kc.to_numpy((nslots, KVH * D), np.uint16), got.view(np.uint16)
    vc = d_vc.to_numpy((nslots, KVH * D), np.uint16)
    for t in range(T):
        assert np.array_equal(kc[slots[t]], gb[t, H * D:(H + KVH) * D]) and np.array_equal(vc[slots[t]], gb[t, (H + KVH) * D:])
    # gate_up + SiLU * up
    Wg, Wgb = h16(rng.standard_normal((2 * I, Hd)) * 0.05)
    d_o = nvr.DeviceBuffer(T * I * 2)
    nvr.check(l.nvr_linear_silu_mul(d_x.ptr, Hd, dev(Wgb).ptr, T, Hd, I, d_o.ptr, None))
    ref = oracle.round_f16(oracle.silu_and_mul(oracle.round_f16(oracle.linear(x, Wg))))
    assert_close_f16(d_o.to_numpy((T, I), F16), ref, ulps=2, atol=3e-4, what="silu")
    d_gu, d_o2 = nvr.DeviceBuffer(T * 2 * I * 2), nvr.DeviceBuffer(T * I * 2)
    _KEEP.extend([d_gu, d_o2])
    nvr.check(l.nvr_linear(d_x.ptr, Hd, dev(Wgb).ptr, T, Hd, 2 * I, d_gu.ptr, 0, None))
    nvr.check(l.nvr_silu_and_mul(d_gu.ptr, T, I, d_o2.ptr, None))
    assert np.array_equal(d_o.to_numpy((T, I), np.uint16), d_o2.to_numpy((T, I), np.uint16)), "fused gate_up + SiLU differs from GEMM, then SiluAndMul"
    # split-k slabs (o_proj shape)
    a, ab = h16(rng.standard_normal((T, H * D)) * 0.3)
    Wo, Wob = h16(rng.standard_normal((Hd, H * D)) * 0.05)
    d_sl = nvr.DeviceBuffer(4 * T * Hd * 4)
    nvr.check(l.nvr_linear_splitk(dev(ab).ptr, H * D, dev(Wob).ptr, T, H * D, Hd, 4, d_sl.ptr, None))
    np.testing.assert_allclose(d_sl.to_numpy((4, T, Hd), np.float32).sum(0), oracle.linear(a, Wo), rtol=2e-5, atol=3e-4)


@pytest.mark.parametrize("T,K,N", [(16384, 2048, 1024), (16384, 3072, 1024), (1000, 1024, 512)])
def test_linear_add_residual_prefill(T, K, N):
    """nvr_linear_add_residual (256x256 GEMM with the residual add in its epilogue) == nvr_linear into a scratch tensor followed by
    nvr_add_rmsnorm's add, bit for bit (same fp16 rounding points), and within GEMM tolerance of the oracle; shapes the 256x256
    kernel does not take are refused."""
    rng = np.random.default_rng(90)
    x, xb = h16(rng.standard_normal((T, K)) * 0.3)
    W, Wb = h16(rng.standard_normal((N, K)) * 0.05)
    h, hb = h16(rng.standard_normal((T, N)))
    w1 = dev(h16(np.ones(N))[1])
    d_x, d_W = dev(xb), dev(Wb)
    d_h1, d_h2, d_y, d_n = dev(hb.copy()), dev(hb.copy()), nvr.DeviceBuffer(T * N * 2), nvr.DeviceBuffer(T * N * 2)
    _KEEP.extend([d_y, d_n])
    rc = nvr.lib().nvr_linear_add_residual(d_x.ptr, K, d_W.ptr, T, K, N, d_h1.ptr, None)
    if T < 2048:                                       # small tile counts go to the 128x128 kernel: not this entry point's business
        assert rc == -10
        return
    nvr.check(rc)
    nvr.check(nvr.lib().nvr_linear(d_x.ptr, K, d_W.ptr, T, K, N, d_y.ptr, 0, None))
    nvr.check(nvr.lib().nvr_add_rmsnorm(d_h2.ptr, d_y.ptr, w1.ptr, 1e-6, T, N, d_n.ptr, None))
    assert np.array_equal(d_h1.to_numpy((T, N), np.uint16), d_h2.to_numpy((T, N), np.uint16))
    ref = oracle.add(h, oracle.round_f16(oracle.linear(x[:64], W)), round16=True) if False else None
    part = oracle.round_f16(oracle.linear(x[:64], W))
    assert_close_f16(d_h1.to_numpy((T, N), F16)[:64], oracle.add(h[:64], part, round16=True), ulps=2, atol=2e-3, what="h + x W^T")


@pytest.mark.parametrize("seed", range(8))
def test_paged_attn_decode_shared_prefix_random_geometries(seed):
    """Random geometries of the shared-prefix decode attention: head_dim 64 / 128, group 1 / 2 / 4, block size 64 / 128 / 256, 1-4 shared
    blocks, 2-200 sequences with 0-3 blocks of own tokens (some with none), against the plain kernel on the same inputs."""
    rng = np.random.default_rng(700 + seed)
    D = int(rng.choice([64, 128])); G = int(rng.choice([1, 2, 4])); KVH = int(rng.choice([1, 2, 4])); H = G * KVH
    bs = int(rng.choice([64, 128, 256])); P = int(rng.integers(1, 5)); B = int(rng.integers(2, 201))
    own = [int(x) for x in rng.integers(0, 3 * bs, B)]
    for i in rng.integers(0, B, max(1, B // 8)):
        own[int(i)] = 0
    S = P * bs
    ctxs = [S + o for o in own]
    own_blocks = [(o + bs - 1) // bs for o in own]
    NB = P + sum(own_blocks) + 2
    max_blocks = P + max(own_blocks) + 1
    kcb = (rng.standard_normal((NB, bs, KVH, D))).astype(F16); vcb = (rng.standard_normal((NB, bs, KVH, D))).astype(F16)
    perm = rng.permutation(NB)
    bt = -np.ones((B, max_blocks), np.int32)
    o = P
    for b in range(B):
        bt[b, :P] = perm[:P]
        bt[b, P:P + own_blocks[b]] = perm[o:o + own_blocks[b]]; o += own_blocks[b]
    qb = (rng.standard_normal((B, H, D))).astype(F16)
    scale = float(np.float32(1.0) / np.sqrt(np.float32(D)))
    meta = nvr.AttnMetaC()
    d_ctx, d_bt = dev(np.asarray(ctxs, np.int32)), dev(bt)
    meta.is_prefill, meta.context_lens, meta.block_tables = 0, d_ctx.ptr, d_bt.ptr
    meta.max_blocks, meta.batch, meta.max_context_len = max_blocks, B, int(max(ctxs))
    ws = nvr.DeviceBuffer(nvr.lib().nvr_paged_attn_workspace_bytes(B, H, D, int(max(ctxs)) + bs))
    d_out, d_plain = nvr.DeviceBuffer(B * H * D * 2), nvr.DeviceBuffer(B * H * D * 2)
    d_q, d_k, d_v = dev(qb), dev(kcb), dev(vcb)
    nvr.check(nvr.lib().nvr_paged_attn_decode_shared(d_q.ptr, H * D, d_k.ptr, d_v.ptr, C.byref(meta), H, KVH, D, bs, scale, S, None, None, None, d_out.ptr, ws.ptr, None))
    nvr.check(nvr.lib().nvr_paged_attn_decode(d_q.ptr, H * D, d_k.ptr, d_v.ptr, C.byref(meta), H, KVH, D, bs, scale, d_plain.ptr, ws.ptr, None))
    assert_close_f16(d_out.to_numpy((B, H, D), F16), d_plain.to_numpy((B, H, D), F16), ulps=3, atol=1e-3,
                     what=f"D={D} G={G} KVH={KVH} bs={bs} P={P} B={B}")


@pytest.mark.parametrize("seed", range(6))
def test_paged_attn_decode_shared_prefix_group(seed):
    """Only SOME sequences of the batch share the prefix (rows / kv0 / count arrays): members scattered over the batch take the shared
    pass + their own remainder, the others (different first blocks, any context length, even shorter than the shared length) are
    attended to in full by the row kernel; everything against the plain kernel."""
    rng = np.random.default_rng(900 + seed)
    D = int(rng.choice([64, 128])); G = int(rng.choice([1, 2, 4])); KVH = int(rng.choice([1, 2])); H = G * KVH
    bs = int(rng.choice([64, 256])); P = int(rng.integers(1, 4)); B = int(rng.integers(4, 150))
    S = P * bs
    member = rng.random(B) < 0.7
    member[int(rng.integers(0, B))] = True
    ctxs, nblk = [], []
    for b in range(B):
        c = S + int(rng.integers(0, 2 * bs)) if member[b] else int(rng.integers(1, S + 2 * bs))
        ctxs.append(c); nblk.append((c + bs - 1) // bs)
    NB = P + sum(nblk) + 2
    max_blocks = max(nblk) + 1
    kcb = rng.standard_normal((NB, bs, KVH, D)).astype(F16); vcb = rng.standard_normal((NB, bs, KVH, D)).astype(F16)
    perm = rng.permutation(NB)
    bt = -np.ones((B, max_blocks), np.int32)
    o = P
    for b in range(B):
        if member[b]:
            bt[b, :P] = perm[:P]
            bt[b, P:nblk[b]] = perm[o:o + nblk[b] - P]; o += nblk[b] - P
        else:
            bt[b, :nblk[b]] = perm[o:o + nblk[b]]; o += nblk[b]
    rows = np.flatnonzero(member).astype(np.int32)
    rows_pad = np.concatenate([rows, np.zeros(B - len(rows), np.int32)])
    kv0 = np.where(member, S, 0).astype(np.int32)
    qb = rng.standard_normal((B, H, D)).astype(F16)
    scale = float(np.float32(1.0) / np.sqrt(np.float32(D)))
    meta = nvr.AttnMetaC()
    d_ctx, d_bt = dev(np.asarray(ctxs, np.int32)), dev(bt)
    meta.is_prefill, meta.context_lens, meta.block_tables = 0, d_ctx.ptr, d_bt.ptr
    meta.max_blocks, meta.batch, meta.max_context_len = max_blocks, B, int(max(ctxs))
    ws = nvr.DeviceBuffer(nvr.lib().nvr_paged_attn_workspace_bytes(B, H, D, int(max(ctxs)) + S + bs))
    d_out, d_plain = nvr.DeviceBuffer(B * H * D * 2), nvr.DeviceBuffer(B * H * D * 2)
    d_q, d_k, d_v = dev(qb), dev(kcb), dev(vcb)
    nvr.check(nvr.lib().nvr_paged_attn_decode_shared(d_q.ptr, H * D, d_k.ptr, d_v.ptr, C.byref(meta), H, KVH, D, bs, scale, S,
                                                     dev(rows_pad).ptr, dev(kv0).ptr, dev(np.asarray([len(rows)], np.int32)).ptr,
                                                     d_out.ptr, ws.ptr, None))
    nvr.check(nvr.lib().nvr_paged_attn_decode(d_q.ptr, H * D, d_k.ptr, d_v.ptr, C.byref(meta), H, KVH, D, bs, scale, d_plain.ptr, ws.ptr, None))
    assert_close_f16(d_out.to_numpy((B, H, D), F16), d_plain.to_numpy((B, H, D), F16), ulps=3, atol=1e-3,
                     what=f"D={D} G={G} KVH={KVH} bs={bs} P={P} B={B} members={len(rows)}")


# ------------------------------------------------------------------------------------------- K12 + K13 + K14 as one persistent launch
