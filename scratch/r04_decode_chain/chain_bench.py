"""Decode GEMM/norm chain of Qwen3-0.6B (no attention) as ONE captured hipGraph of 28 layers with their own weights
(880 MB: HBM-cold every replay), replayed back to back: microseconds per layer for the four-launch chain
(kernels/linear_decode.hip), the r01 six-launch chain and each kernel type alone.  Run on the GPU box:
    python tools/chain_bench.py [T]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nvr_import

nvr = nvr_import.load(); l = nvr.lib(); nvr.check(l.nvr_device_set(0))
T = int(sys.argv[1]) if len(sys.argv) > 1 else 32
HOT = os.environ.get("CHAIN_HOT") == "1"          # every layer uses weight set 0 (31 MB: L2 / MALL resident) instead of its own
Hd, H, KVH, D, I, L = 1024, 16, 8, 128, 3072, 28
QKV = (H + 2 * KVH) * D
st = C.c_void_p(); nvr.check(l.nvr_stream_create(C.byref(st)))
e0, e1 = C.c_void_p(), C.c_void_p(); l.nvr_event_create(C.byref(e0)); l.nvr_event_create(C.byref(e1))
rng = np.random.default_rng(0)
keep = []


def buf(nbytes):
    b = nvr.DeviceBuffer(nbytes); keep.append(b); return b


def arr(a):
    b = nvr.DeviceBuffer.from_numpy(np.ascontiguousarray(a)); keep.append(b); return b


def weights(rows, cols):
    ws = [buf(rows * cols * 2) for _ in range(L)]
    for i, w in enumerate(ws):
        nvr.check(l.nvr_fill_weight(w.ptr, rows, cols, cols, cols, 0, 0, 5 + i, 1e-6, None))
    return ws


Wqkv, Wo, Wgu, Wd = weights(QKV, Hd), weights(Hd, H * D), weights(2 * I, Hd), weights(Hd, I)


def tiled(ws, rows, cols, mode):                      # the runner's tiled copies [N/16][K/32][16][32]
    ts = [buf(rows * cols * 2) for _ in ws]
    for w, t in zip(ws, ts):
        nvr.check(l.nvr_retile_weight(w.ptr, t.ptr, rows, cols, mode, H, KVH, D, None))
    return ts


Tqkv, To, Tgu, Td = tiled(Wqkv, QKV, Hd, 1), tiled(Wo, Hd, H * D, 0), tiled(Wgu, 2 * I, Hd, 0), tiled(Wd, Hd, I, 0)
h = arr(rng.standard_normal((T, Hd)).astype(np.float16)); n = buf(T * Hd * 2)
g = arr(np.ones(Hd, np.float16))
qkv, attn, act = buf(T * QKV * 2), arr(rng.standard_normal((T, H * D)).astype(np.float16) * 0.1), buf(T * I * 2)
slabs = buf(4 * T * Hd * 4); cnt = arr(np.zeros(4096, np.uint32))
pos = arr(np.arange(T, dtype=np.int64) + 1000); slots = arr(np.arange(T, dtype=np.int32))
cos = arr(np.ones((2048, D // 2), np.float32)); sin = arr(np.zeros((2048, D // 2), np.float32))
kc, vc = buf(64 * KVH * D * 2), buf(64 * KVH * D * 2)
So, Sd = l.nvr_decode_splitk_slices(T, H * D, Hd), l.nvr_decode_splitk_slices(T, I, Hd)

ops = {
    "qkv_normed": lambda i: l.nvr_linear_qkv_rope_store_normed(h.ptr, Hd, g.ptr, 1e-6, Wqkv[i].ptr, None, T, Hd, H, KVH, D, pos.ptr, slots.ptr, cos.ptr, sin.ptr, qkv.ptr, kc.ptr, vc.ptr, st),
    "resid_o": lambda i: l.nvr_linear_resid(attn.ptr, H * D, Wo[i].ptr, None, T, H * D, Hd, So, slabs.ptr, cnt.ptr, h.ptr, st),
    "silu_normed": lambda i: l.nvr_linear_silu_mul_normed(h.ptr, Hd, g.ptr, 1e-6, Wgu[i].ptr, None, T, Hd, I, act.ptr, st),
    "resid_down": lambda i: l.nvr_linear_resid(act.ptr, I, Wd[i].ptr, None, T, I, Hd, Sd, slabs.ptr, cnt.ptr, h.ptr, st),
    "qkv": lambda i: l.nvr_linear_qkv_rope_store(n.ptr, Hd, Wqkv[i].ptr, T, Hd, H, KVH, D, pos.ptr, slots.ptr, cos.ptr, sin.ptr, qkv.ptr, kc.ptr, vc.ptr, st),
    "splitk_o": lambda i: l.nvr_linear_splitk(attn.ptr, H * D, Wo[i].ptr, T, H * D, Hd, So, slabs.ptr, st),
    "slabnorm": lambda i: l.nvr_add_rmsnorm_slabs(h.ptr, slabs.ptr, 4, g.ptr, 1e-6, T, Hd, n.ptr, st),
    "silu": lambda i: l.nvr_linear_silu_mul(n.ptr, Hd, Wgu[i].ptr, T, Hd, I, act.ptr, st),
    "splitk_down": lambda i: l.nvr_linear_splitk(act.ptr, I, Wd[i].ptr, T, I, Hd, Sd, slabs.ptr, st),
    "qkv_normed_t": lambda i: l.nvr_linear_qkv_rope_store_normed(h.ptr, Hd, g.ptr, 1e-6, Wqkv[i].ptr, Tqkv[i].ptr, T, Hd, H, KVH, D, pos.ptr, slots.ptr, cos.ptr, sin.ptr, qkv.ptr, kc.ptr, vc.ptr, st),
    "resid_o_t": lambda i: l.nvr_linear_resid(attn.ptr, H * D, Wo[i].ptr, To[i].ptr, T, H * D, Hd, So, slabs.ptr, cnt.ptr, h.ptr, st),
    "silu_normed_t": lambda i: l.nvr_linear_silu_mul_normed(h.ptr, Hd, g.ptr, 1e-6, Wgu[i].ptr, Tgu[i].ptr, T, Hd, I, act.ptr, st),
    "resid_down_t": lambda i: l.nvr_linear_resid(act.ptr, I, Wd[i].ptr, Td[i].ptr, T, I, Hd, Sd, slabs.ptr, cnt.ptr, h.ptr, st),
    "resid_o_ht": lambda i: l.nvr_linear_resid(attn.ptr, H * D, Wo[i].ptr, To[i].ptr, T, H * D, Hd, 0, None, None, h.ptr, st),
    "resid_down_ht": lambda i: l.nvr_linear_resid(act.ptr, I, Wd[i].ptr, Td[i].ptr, T, I, Hd, 0, None, None, h.ptr, st),
    "qkv_t": lambda i: l.nvr_linear_qkv_rope_store_tiled(n.ptr, Hd, Wqkv[i].ptr, Tqkv[i].ptr, T, Hd, H, KVH, D, pos.ptr, slots.ptr, cos.ptr, sin.ptr, qkv.ptr, kc.ptr, vc.ptr, st),
    "splitk_o_t": lambda i: l.nvr_linear_splitk_tiled(attn.ptr, H * D, Wo[i].ptr, To[i].ptr, T, H * D, Hd, So, slabs.ptr, st),
    "silu_t": lambda i: l.nvr_linear_silu_mul_tiled(n.ptr, Hd, Wgu[i].ptr, Tgu[i].ptr, T, Hd, I, act.ptr, st),
    "splitk_down_t": lambda i: l.nvr_linear_splitk_tiled(act.ptr, I, Wd[i].ptr, Td[i].ptr, T, I, Hd, Sd, slabs.ptr, st),
    "rmsnorm": lambda i: l.nvr_rmsnorm(h.ptr, g.ptr, 1e-6, T, Hd, n.ptr, st),
}
chains = {
    "c6t (c6 reading the tiled weight copies: the product's default)": ["qkv_t", "splitk_o_t", "slabnorm", "silu_t", "splitk_down_t", "slabnorm"],
    "c4t (four launches, tiled, split-k + last arriver)": ["qkv_normed_t", "resid_o_t", "silu_normed_t", "resid_down_t"],
    "c4ht (four launches, tiled, 8-row tiles without k split)": ["qkv_normed_t", "resid_o_ht", "silu_normed_t", "resid_down_ht"],
    "c4 (qkv_normed, resid_o, silu_normed, resid_down)": ["qkv_normed", "resid_o", "silu_normed", "resid_down"],
    "c6 (qkv, splitk_o, slabnorm, silu, splitk_down, slabnorm)": ["qkv", "splitk_o", "slabnorm", "silu", "splitk_down", "slabnorm"],
}
for k in ops:
    chains[k + " alone"] = [k]


def measure(seq, reps=30):
    ge = C.c_void_p()
    nvr.check(l.nvr_graph_capture_begin(st))
    for i in range(L):
        for name in seq:
            nvr.check(ops[name](0 if HOT else i))
    nvr.check(l.nvr_graph_capture_end(st, C.byref(ge)))
    for _ in range(3):
        nvr.check(l.nvr_graph_launch(ge, st))
    nvr.check(l.nvr_stream_synchronize(st))
    best = 1e9
    for rnd in range(3):
        l.nvr_event_record(e0, st)
        for _ in range(reps):
            nvr.check(l.nvr_graph_launch(ge, st))
        l.nvr_event_record(e1, st)
        ms = C.c_float(); nvr.check(l.nvr_event_elapsed_ms(e0, e1, C.byref(ms)))
        best = min(best, ms.value * 1e3 / reps / L)
    nvr.check(l.nvr_graph_destroy(ge))
    return best


print(f"T={T}  S(o)={So} S(down)={Sd}   us per layer (28 layers per graph, {L} weight sets, best of 3 x 30 replays)")
for name, seq in chains.items():
    print(f"{measure(seq):8.2f}  {name}", flush=True)
