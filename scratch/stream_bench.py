"""Decode GEMMs at Qwen3-8B shapes, T=32: stream kernel (default) vs skinny kernel (NVR_STREAM_GEMM=0, separate process)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nvr_import
nvr = nvr_import.load(); l = nvr.lib(); nvr.check(l.nvr_device_set(0))
T = 32
st = C.c_void_p(); l.nvr_stream_create(C.byref(st)); e0, e1 = C.c_void_p(), C.c_void_p(); l.nvr_event_create(C.byref(e0)); l.nvr_event_create(C.byref(e1))
def timeit(fn):
    best = 1e9
    for rnd in range(3):
        for i in range(8): fn(i)
        nvr.check(l.nvr_stream_synchronize(st)); l.nvr_event_record(e0, st)
        for i in range(16): fn(i)
        l.nvr_event_record(e1, st); nvr.check(l.nvr_stream_synchronize(st))
        ms = C.c_float(); l.nvr_event_elapsed_ms(e0, e1, C.byref(ms)); best = min(best, ms.value * 1e3 / 16)
    return best
rng = np.random.default_rng(0)
for K, N in ((4096, 4096), (12288, 4096)):
    Ws = [nvr.DeviceBuffer(N * K * 2) for _ in range(4)]
    for w in Ws: nvr.check(l.nvr_fill_weight(w.ptr, N, K, K, K, 0, 0, 5, 1e-3, None))
    x = nvr.DeviceBuffer.from_numpy(rng.standard_normal((T, K)).astype(np.float16)); y = nvr.DeviceBuffer(T * N * 2)
    us = timeit(lambda i: nvr.check(l.nvr_linear(x.ptr, K, Ws[i % 4].ptr, T, K, N, y.ptr, 0, st)))
    print(f"plain K={K:5d} N={N:5d}: {us:8.2f} us  {N * K * 2 / us / 1e3:7.1f} GB/s", flush=True)
    del Ws
K, I = 4096, 12288
Ws = [nvr.DeviceBuffer(2 * I * K * 2) for _ in range(3)]
for w in Ws: nvr.check(l.nvr_fill_weight(w.ptr, 2 * I, K, K, K, 0, 0, 5, 1e-3, None))
x = nvr.DeviceBuffer.from_numpy(rng.standard_normal((T, K)).astype(np.float16)); y = nvr.DeviceBuffer(T * I * 2)
us = timeit(lambda i: nvr.check(l.nvr_linear_silu_mul(x.ptr, K, Ws[i % 3].ptr, T, K, I, y.ptr, st)))
print(f"silu  K={K:5d} I={I:5d}: {us:8.2f} us  {2 * I * K * 2 / us / 1e3:7.1f} GB/s", flush=True)
del Ws
H, KVH, D = 32, 8, 128
QKV = (H + 2 * KVH) * D
Ws = [nvr.DeviceBuffer(QKV * K * 2) for _ in range(4)]
for w in Ws: nvr.check(l.nvr_fill_weight(w.ptr, QKV, K, K, K, 0, 0, 5, 1e-3, None))
qkv = nvr.DeviceBuffer(T * QKV * 2)
pos = nvr.DeviceBuffer.from_numpy((np.arange(T) % 64).astype(np.int64)); slots = nvr.DeviceBuffer.from_numpy(np.arange(T, dtype=np.int32))
cos = nvr.DeviceBuffer(64 * 64 * 4); sin = nvr.DeviceBuffer(64 * 64 * 4); nvr.check(l.nvr_rope_table(D, 64, 1e6, cos.ptr, sin.ptr))
kc = nvr.DeviceBuffer(64 * KVH * D * 2); vc = nvr.DeviceBuffer(64 * KVH * D * 2)
us = timeit(lambda i: nvr.check(l.nvr_linear_qkv_rope_store(x.ptr, K, Ws[i % 4].ptr, T, K, H, KVH, D, pos.ptr, slots.ptr, cos.ptr, sin.ptr, qkv.ptr, kc.ptr, vc.ptr, st)))
print(f"rope  K={K:5d} N={QKV:5d}: {us:8.2f} us  {QKV * K * 2 / us / 1e3:7.1f} GB/s", flush=True)
