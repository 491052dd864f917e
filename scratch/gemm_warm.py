"""Decode GEMM launch time with weights HBM-cold (8 buffers cycled + 512 MB flush) vs cache-warm (same buffer)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nvr_import
nvr = nvr_import.load(); l = nvr.lib(); nvr.check(l.nvr_device_set(0))
T = 32
st = C.c_void_p(); l.nvr_stream_create(C.byref(st))
F = nvr.DeviceBuffer(512 << 20)
def timeit(fn, n=40, flush_every=0):
    evs = []
    for rep in range(n):
        if flush_every and rep % flush_every == 0: nvr.check(l.nvr_fill_const(F.ptr, 256 << 20, float(rep), st))
        a, b = C.c_void_p(), C.c_void_p(); l.nvr_event_create(C.byref(a)); l.nvr_event_create(C.byref(b))
        l.nvr_event_record(a, st); fn(rep); l.nvr_event_record(b, st); evs.append((a, b))
    nvr.check(l.nvr_stream_synchronize(st)); ts = []
    for a, b in evs:
        ms = C.c_float(); nvr.check(l.nvr_event_elapsed_ms(a, b, C.byref(ms))); ts.append(ms.value * 1e3)
    ts = sorted(ts[8:]); return ts[len(ts) // 2]
for (K, N, name) in [(1024, 4096, "qkv-like plain"), (1024, 6144, "gate_up-like plain"), (2048, 1024, "o_proj")]:
    Ws = [nvr.DeviceBuffer(N * K * 2) for _ in range(16)]
    for w in Ws: nvr.check(l.nvr_fill_weight(w.ptr, N, K, K, K, 0, 0, 5, 1e-6, None))
    x = nvr.DeviceBuffer.from_numpy(np.random.default_rng(0).standard_normal((T, K)).astype(np.float16)); y = nvr.DeviceBuffer(T * N * 4)
    cold = timeit(lambda r: nvr.check(l.nvr_linear(x.ptr, K, Ws[r % 16].ptr, T, K, N, y.ptr, 0, st)), flush_every=8)
    warm = timeit(lambda r: nvr.check(l.nvr_linear(x.ptr, K, Ws[0].ptr, T, K, N, y.ptr, 0, st)))
    print(f"{name:20s} K={K} N={N}: cold {cold:6.2f} us   warm {warm:6.2f} us")
empty = timeit(lambda r: nvr.check(l.nvr_fill_const(F.ptr, 64, 0.0, st)))
print(f"tiny kernel (event overhead floor): {empty:6.2f} us")
