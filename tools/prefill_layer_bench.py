"""The kernels of one Qwen3-0.6B prefill layer at 32 x 1024 tokens (BASELINE configs[1] prefill / configs[2]), each timed alone with HIP
events on one stream (median of 10): qkv + RoPE + KV store, flash prefill attention, o_proj + residual, RMSNorm, gate_up + SiLU,
down_proj + residual, plus the plain GEMMs.  Run on the GPU box: python tools/prefill_layer_bench.py [seq_len]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nvr_import
nvr = nvr_import.load(); l = nvr.lib(); nvr.check(l.nvr_device_set(0))
T = 32768
st = C.c_void_p(); l.nvr_stream_create(C.byref(st))
def timeit(fn, n=12):
    evs = []
    for rep in range(n):
        a, b = C.c_void_p(), C.c_void_p(); l.nvr_event_create(C.byref(a)); l.nvr_event_create(C.byref(b))
        l.nvr_event_record(a, st); fn(); l.nvr_event_record(b, st); evs.append((a, b))
    nvr.check(l.nvr_stream_synchronize(st)); ts = []
    for a, b in evs:
        ms = C.c_float(); nvr.check(l.nvr_event_elapsed_ms(a, b, C.byref(ms))); ts.append(ms.value * 1e3)
    ts = sorted(ts[2:]); return ts[len(ts) // 2]
rng = np.random.default_rng(0)
def buf(n): 
    b = nvr.DeviceBuffer(n * 2); nvr.check(l.nvr_fill_weight(b.ptr, 1, n, n, n, 0, 0, 7, 0.01, None)); return b
x1024, x2048, x3072 = buf(T * 1024), buf(T * 2048), buf(T * 3072)
for name, K, N, x in [("plain N=4096 K=1024", 1024, 4096, x1024), ("plain N=6144 K=1024", 1024, 6144, x1024), ("plain N=1024 K=2048", 2048, 1024, x2048), ("plain N=1024 K=3072", 3072, 1024, x3072)]:
    W = buf(N * K); y = nvr.DeviceBuffer(T * N * 2)
    us = timeit(lambda: nvr.check(l.nvr_linear(x.ptr, K, W.ptr, T, K, N, y.ptr, 0, st)))
    print(f"{name:24s} {us:8.1f} us  {2*T*K*N/us/1e6:7.1f} TF/s")
W = buf(6144 * 1024); y = nvr.DeviceBuffer(T * 3072 * 2)
us = timeit(lambda: nvr.check(l.nvr_linear_silu_mul(x1024.ptr, 1024, W.ptr, T, 1024, 3072, y.ptr, st)))
print(f"{'silu  I=3072 K=1024':24s} {us:8.1f} us  {2*T*1024*6144/us/1e6:7.1f} TF/s")
H, KVH, D = 16, 8, 128
W = buf(4096 * 1024); y = nvr.DeviceBuffer(T * 4096 * 2)
pos = nvr.DeviceBuffer.from_numpy((np.arange(T) % 1024).astype(np.int64)); slots = nvr.DeviceBuffer.from_numpy(np.arange(T, dtype=np.int32))
cos = nvr.DeviceBuffer(1024 * 64 * 4); sin = nvr.DeviceBuffer(1024 * 64 * 4); nvr.check(l.nvr_rope_table(D, 1024, 1e6, cos.ptr, sin.ptr))
kc = nvr.DeviceBuffer(T * KVH * D * 2); vc = nvr.DeviceBuffer(T * KVH * D * 2)
us = timeit(lambda: nvr.check(l.nvr_linear_qkv_rope_store(x1024.ptr, 1024, W.ptr, T, 1024, H, KVH, D, pos.ptr, slots.ptr, cos.ptr, sin.ptr, y.ptr, kc.ptr, vc.ptr, st)))
print(f"{'rope  N=4096 K=1024':24s} {us:8.1f} us  {2*T*1024*4096/us/1e6:7.1f} TF/s")

# residual-epilogue GEMMs (o_proj K = 2048, down_proj K = 3072) and the norm that follows them
h = buf(T * 1024); n = nvr.DeviceBuffer(T * 1024 * 2); g = nvr.DeviceBuffer.from_numpy(np.ones(1024, np.float16))
for name, K, x in [("resid N=1024 K=2048 (o)", 2048, x2048), ("resid N=1024 K=3072 (down)", 3072, x3072)]:
    W = buf(1024 * K)
    us = timeit(lambda: nvr.check(l.nvr_linear_add_residual(x.ptr, K, W.ptr, T, K, 1024, h.ptr, st)))
    print(f"{name:24s} {us:8.1f} us  {2*T*K*1024/us/1e6:7.1f} TF/s")
us = timeit(lambda: nvr.check(l.nvr_rmsnorm(h.ptr, g.ptr, 1e-6, T, 1024, n.ptr, st)))
print(f"{'rmsnorm [T,1024]':24s} {us:8.1f} us  {T*1024*4/us/1e6:7.2f} TB/s")
# flash prefill over the packed qkv of the RoPE launch above: B sequences of L tokens
L = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
B = T // L
cu = nvr.DeviceBuffer.from_numpy((np.arange(B + 1) * L).astype(np.int32))
meta = nvr.AttnMetaC(); meta.is_prefill = 1; meta.cu_seqlens_q = cu.ptr; meta.cu_seqlens_k = cu.ptr; meta.max_seqlen_q = L; meta.max_seqlen_k = L; meta.batch = B
out = nvr.DeviceBuffer(T * H * D * 2)
qp = y.ptr; kp = y.ptr + H * D * 2; vp = y.ptr + (H + KVH) * D * 2
us = timeit(lambda: nvr.check(l.nvr_attn_prefill_varlen(qp, kp, vp, 4096, C.byref(meta), T, H, KVH, D, float(1 / np.sqrt(D)), out.ptr, st)))
fl = 4 * H * D * B * (L * (L + 1) // 2)
print(f"{'flash ' + str(B) + ' x ' + str(L):24s} {us:8.1f} us  {fl/us/1e6:7.1f} TF/s causal")
