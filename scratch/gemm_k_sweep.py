"""gemm256 plain at T = 32768, N = 4096, K swept: per-tile time = KT * t_k + e (profiles/r03_prefill_gemm_overheads.txt)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nvr_import
nvr = nvr_import.load(); l = nvr.lib(); nvr.check(l.nvr_device_set(0))
T = int(os.environ.get("T", "32768")); N = int(os.environ.get("N", "4096"))
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
def buf(n):
    b = nvr.DeviceBuffer(n * 2); nvr.check(l.nvr_fill_weight(b.ptr, 1, n, n, n, 0, 0, 7, 0.01, None)); return b
y = nvr.DeviceBuffer(T * N * 2)
tiles = (T // 256) * (N // 256)
for K in (256, 512, 1024, 2048, 4096):
    x = buf(T * K); W = buf(N * K)
    us = timeit(lambda: nvr.check(l.nvr_linear(x.ptr, K, W.ptr, T, K, N, y.ptr, 0, st)))
    per_tile = us / (tiles / 256)
    print(f"K={K:5d}: {us:8.1f} us  {2*T*K*N/us/1e6:7.1f} TF/s  per tile {per_tile:6.2f} us ({K // 64} K-tiles)", flush=True)
    del x, W
