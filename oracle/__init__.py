"""CPU ORACLE package (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

numpy-facing bindings to oracle/libnvr_oracle.so (nvr_oracle.c).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libnvr_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "nvr_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libnvr_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def usable_cores(cap: int = 32) -> int:
    """CPUs this process may actually run on (affinity + cgroup quota), capped: an OpenMP team wider
    than that only spins (the GPU box reports 256 logical CPUs to a container with far fewer)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, cap))


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        path = os.environ.get("NVO_ORACLE_LIB") or _LIB_PATH          # NVO_ORACLE_LIB: the sanitizer build (tests/test_sanitizers.py)
        if path == _LIB_PATH and not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(path)
        _declare(_lib)
        _lib.nvo_set_num_threads(usable_cores())
    return _lib


_f32p = C.POINTER(C.c_float)
_i64p = C.POINTER(C.c_int64)
_i32p = C.POINTER(C.c_int32)
_u16p = C.POINTER(C.c_uint16)


def _declare(l: C.CDLL) -> None:
    l.nvo_round_f16.restype = C.c_float
    l.nvo_round_f16.argtypes = [C.c_float]
    l.nvo_round_f16_array.argtypes = [_f32p, C.c_size_t]
    l.nvo_round_f16_copy.argtypes = [_f32p, _f32p, C.c_size_t]
    l.nvo_round_bf16_copy.argtypes = [_f32p, _f32p, C.c_size_t]
    l.nvo_f32_to_f16.argtypes = [_f32p, _u16p, C.c_size_t]
    l.nvo_f16_to_f32.argtypes = [_u16p, _f32p, C.c_size_t]
    l.nvo_num_threads.restype = C.c_int
    l.nvo_set_num_threads.argtypes = [C.c_int]
    l.nvo_xxh64.restype = C.c_uint64
    l.nvo_xxh64.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64]
    l.nvo_block_hash.restype = C.c_uint64
    l.nvo_block_hash.argtypes = [_i64p, C.c_size_t, C.c_int, C.c_uint64]
    l.nvo_splitmix64.restype = C.c_uint64
    l.nvo_splitmix64.argtypes = [C.c_uint64]
    l.nvo_weight_key.restype = C.c_uint64
    l.nvo_weight_key.argtypes = [C.c_uint64, C.c_uint64]
    l.nvo_weight_scale.restype = C.c_float
    l.nvo_weight_scale.argtypes = [C.c_double]
    l.nvo_fill_weight.argtypes = [_f32p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                  C.c_int64, C.c_uint64, C.c_float, C.c_int]
    l.nvo_fill_tokens.argtypes = [_i64p, C.c_int64, C.c_uint64, C.c_uint64, C.c_int64]
    l.nvo_embedding.argtypes = [_i64p, C.c_int64, _f32p, C.c_int64, _f32p]
    l.nvo_vocab_mask_local.argtypes = [_i64p, C.c_int64, C.c_int64, C.c_int64, _i32p, _i64p]
    l.nvo_rmsnorm.argtypes = [_f32p, _f32p, C.c_float, C.c_int64, C.c_int64, _f32p]
    l.nvo_add.argtypes = [_f32p, _f32p, C.c_size_t, _f32p, C.c_int]
    l.nvo_linear.argtypes = [_f32p, _f32p, _f32p, C.c_int64, C.c_int64, C.c_int64, _f32p]
    l.nvo_rope_table.argtypes = [C.c_int64, C.c_int64, C.c_double, _f32p, _f32p]
    l.nvo_rope_apply.argtypes = [_f32p, _i64p, C.c_int64, C.c_int64, C.c_int64, _f32p, _f32p]
    l.nvo_kv_store.argtypes = [_f32p, _f32p, _i32p, C.c_int64, C.c_int64, _f32p, _f32p]
    l.nvo_attn_prefill_varlen.argtypes = [_f32p, _f32p, _f32p, _i32p, C.c_int64, C.c_int64,
                                          C.c_int64, C.c_int64, C.c_float, _f32p]
    l.nvo_attn_paged.argtypes = [_f32p, _i32p, _f32p, _f32p, _i32p, C.c_int64, _i32p, C.c_int64,
                                 C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_float, _f32p]
    l.nvo_silu_and_mul.argtypes = [_f32p, C.c_int64, C.c_int64, _f32p]
    l.nvo_activation.argtypes = [C.c_int, _f32p, C.c_int64, C.c_int64, _f32p]
    l.nvo_argmax.restype = C.c_int64
    l.nvo_argmax.argtypes = [_f32p, C.c_int64]
    l.nvo_top_k.argtypes = [_f32p, C.c_int64, C.c_int64, _f32p]
    l.nvo_top_p.argtypes = [_f32p, C.c_int64, C.c_float, _f32p]
    l.nvo_sample_key.restype = C.c_uint64
    l.nvo_sample_key.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
    l.nvo_gumbel.restype = C.c_float
    l.nvo_gumbel.argtypes = [C.c_uint64, C.c_int64]
    l.nvo_sample.restype = C.c_int64
    l.nvo_sample.argtypes = [_f32p, C.c_int64, C.c_float, C.c_int64, C.c_float, C.c_int, C.c_uint64]


def _f(a: np.ndarray):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_f32p)


def _i64(a: np.ndarray):
    assert a.dtype == np.int64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_i64p)


def _i32(a: np.ndarray):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_i32p)


def f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


# ---- thin numpy wrappers ---------------------------------------------------
def round_f16(a: np.ndarray) -> np.ndarray:
    a = f32(a)
    out = np.empty_like(a)
    lib().nvo_round_f16_copy(_f(a), _f(out), a.size)
    return out


def round_bf16(a: np.ndarray) -> np.ndarray:
    """f32 -> bfloat16 -> f32, round to nearest even (the rounding points of the product's bfloat16 build, Config.dtype = "bfloat16",
    reference src/config.rs:51,113-116).  Finite inputs only.  (u + 0x7FFF + lowest kept bit) with the low half cleared; pinned against
    torch's bfloat16 cast in tests/test_oracle_kat.py."""
    a = f32(a)
    out = np.empty_like(a)
    lib().nvo_round_bf16_copy(_f(a), _f(out), a.size)
    return out


def to_bf16_bits(a: np.ndarray) -> np.ndarray:
    return (round_bf16(a).view(np.uint32) >> 16).astype(np.uint16)


def to_f16_bits(a: np.ndarray) -> np.ndarray:
    a = f32(a)
    out = np.empty(a.shape, dtype=np.uint16)
    lib().nvo_f32_to_f16(_f(a), out.ctypes.data_as(_u16p), a.size)
    return out


def from_f16_bits(a: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint16)
    out = np.empty(a.shape, dtype=np.float32)
    lib().nvo_f16_to_f32(a.ctypes.data_as(_u16p), _f(out), a.size)
    return out


def xxh64(data: bytes, seed: int = 0) -> int:
    buf = C.create_string_buffer(data, len(data))
    return int(lib().nvo_xxh64(buf, len(data), seed))


def block_hash(tokens, prefix=None) -> int:
    t = np.ascontiguousarray(tokens, dtype=np.int64)
    return int(lib().nvo_block_hash(_i64(t), t.size, 0 if prefix is None else 1, 0 if prefix is None else prefix))


def weight_key(seed: int, tensor_id: int) -> int:
    return int(lib().nvo_weight_key(seed, tensor_id))


def weight_scale(std: float) -> float:
    return float(lib().nvo_weight_scale(std))


def fill_weight(rows, cols, global_cols, row0, col0, key, scale, round16=True) -> np.ndarray:
    out = np.empty((rows, cols), dtype=np.float32)
    lib().nvo_fill_weight(_f(out), rows, cols, cols, global_cols, row0, col0, key, scale, int(round16))
    return out


def fill_tokens(n, seed, stream, vocab) -> np.ndarray:
    out = np.empty(n, dtype=np.int64)
    lib().nvo_fill_tokens(_i64(out), n, seed, stream, vocab)
    return out


def embedding(ids, E) -> np.ndarray:
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    E = f32(E)
    out = np.empty((ids.size, E.shape[1]), dtype=np.float32)
    lib().nvo_embedding(_i64(ids), ids.size, _f(E), E.shape[1], _f(out))
    return out


def vocab_mask_local(ids, start, end):
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    mask = np.empty(ids.size, dtype=np.int32)
    local = np.empty(ids.size, dtype=np.int64)
    lib().nvo_vocab_mask_local(_i64(ids), ids.size, start, end, _i32(mask), _i64(local))
    return mask, local


def rmsnorm(x, w, eps) -> np.ndarray:
    x, w = f32(x), f32(w)
    out = np.empty_like(x)
    lib().nvo_rmsnorm(_f(x), _f(w), eps, x.shape[0], x.shape[1], _f(out))
    return out


def add(a, b, round16=False) -> np.ndarray:
    a, b = f32(a), f32(b)
    out = np.empty_like(a)
    lib().nvo_add(_f(a), _f(b), a.size, _f(out), int(round16))
    return out


def linear(x, W, bias=None) -> np.ndarray:
    x, W = f32(x), f32(W)
    T, K = x.shape
    N = W.shape[0]
    assert W.shape[1] == K
    out = np.empty((T, N), dtype=np.float32)
    b = None if bias is None else _f(f32(bias))
    lib().nvo_linear(_f(x), _f(W), b, T, K, N, _f(out))
    return out


def rope_table(D, max_pos, theta):
    c = np.empty((max_pos, D // 2), dtype=np.float32)
    s = np.empty((max_pos, D // 2), dtype=np.float32)
    lib().nvo_rope_table(D, max_pos, theta, _f(c), _f(s))
    return c, s


def rope_table_with_scaling(D, max_pos, base, scaling_factor):
    """RotaryEmbedding::new_with_scaling, reference src/layers/rotary_embedding.rs:121-134: the table of `new` for base * scaling_factor.
    Returns (scaled base, cos, sin)."""
    scaled = float(base) * float(scaling_factor)
    return (scaled,) + rope_table(D, max_pos, scaled)


def rope_apply(x, pos, cos_t, sin_t) -> np.ndarray:
    x = f32(x).copy()
    pos = np.ascontiguousarray(pos, dtype=np.int64)
    T, nh, D = x.shape
    lib().nvo_rope_apply(_f(x), _i64(pos), T, nh, D, _f(cos_t), _f(sin_t))
    return x


def kv_store(k, v, slots, k_cache, v_cache) -> None:
    """In place on k_cache/v_cache ([NB, bs, KVH, D] f32)."""
    k, v = f32(k), f32(v)
    slots = np.ascontiguousarray(slots, dtype=np.int32)
    row = k.shape[1] * k.shape[2]
    lib().nvo_kv_store(_f(k), _f(v), _i32(slots), k.shape[0], row, _f(k_cache), _f(v_cache))


def attn_prefill_varlen(q, k, v, cu_seqlens, scale) -> np.ndarray:
    q, k, v = f32(q), f32(k), f32(v)
    cu = np.ascontiguousarray(cu_seqlens, dtype=np.int32)
    out = np.empty_like(q)
    lib().nvo_attn_prefill_varlen(_f(q), _f(k), _f(v), _i32(cu), cu.size - 1, q.shape[1], k.shape[1],
                                  q.shape[2], scale, _f(out))
    return out


def attn_paged(q, cu_seqlens_q, k_cache, v_cache, block_tables, context_lens, scale) -> np.ndarray:
    q = f32(q)
    cu = np.ascontiguousarray(cu_seqlens_q, dtype=np.int32)
    bt = np.ascontiguousarray(block_tables, dtype=np.int32)
    ctx = np.ascontiguousarray(context_lens, dtype=np.int32)
    NB, bs, KVH, D = k_cache.shape
    out = np.empty_like(q)
    lib().nvo_attn_paged(_f(q), _i32(cu), _f(k_cache), _f(v_cache), _i32(bt), bt.shape[1], _i32(ctx),
                         ctx.size, q.shape[1], KVH, D, bs, scale, _f(out))
    return out


def attn_decode(q, k_cache, v_cache, block_tables, context_lens, scale) -> np.ndarray:
    B = np.asarray(context_lens).size
    return attn_paged(q, np.arange(B + 1, dtype=np.int32), k_cache, v_cache, block_tables, context_lens, scale)


def silu_and_mul(x) -> np.ndarray:
    x = f32(x)
    if x.shape[-1] % 2 != 0:   # activation.rs:50-52
        raise ValueError(f"Input dimension must be even for SiluAndMul, got {x.shape[-1]}")
    T, I2 = x.shape
    out = np.empty((T, I2 // 2), dtype=np.float32)
    lib().nvo_silu_and_mul(_f(x), T, I2 // 2, _f(out))
    return out


ACTIVATION_TYPES = {"silu": 0, "gelu": 1, "relu": 2, "silu_and_mul": 3, "gelu_and_mul": 4}      # ActivationType, src/layers/activation.rs:111-117


def activation(kind, x) -> np.ndarray:
    """Activation::forward (src/layers/activation.rs:147-159): silu :12-15, gelu :20-22 (candle's tanh form), relu :25-27 on [T, cols]; SiluAndMul :46-63 and
    GeluAndMul :74-100 on [T, 2 I] -> [T, I].  An odd last dimension of the two fused types is the reference's error (:50-52, :88-90)."""
    k = ACTIVATION_TYPES[kind] if isinstance(kind, str) else int(kind)
    x = f32(x)
    T, cols = x.shape
    if k >= 3 and cols % 2:
        raise ValueError(f"Input dimension must be even for {'SiluAndMul' if k == 3 else 'GeluAndMul'}, got {cols}")
    out = np.empty((T, cols // 2 if k >= 3 else cols), np.float32)
    lib().nvo_activation(k, _f(x), T, cols, _f(out))
    return out


def argmax(x) -> int:
    x = f32(x)
    return int(lib().nvo_argmax(_f(x), x.size))


def top_k(x, k) -> np.ndarray:
    x = f32(x)
    out = np.empty_like(x)
    lib().nvo_top_k(_f(x), x.size, k, _f(out))
    return out


def top_p(x, p) -> np.ndarray:
    x = f32(x)
    out = np.empty_like(x)
    lib().nvo_top_p(_f(x), x.size, p, _f(out))
    return out


def sample_key(seed, seq_id, step) -> int:
    return int(lib().nvo_sample_key(seed, seq_id, step))


def gumbel(key, v) -> float:
    return float(lib().nvo_gumbel(key, v))


def sample(logits, temperature, top_k_=0, top_p_=None, key=0) -> int:
    x = f32(logits)
    return int(lib().nvo_sample(_f(x), x.size, temperature, top_k_ or 0,
                                0.0 if top_p_ is None else top_p_, 0 if top_p_ is None else 1, key))
