"""Mid-batch decode GEMMs (65-512 rows): time of ONE launch of nvr_linear as a function of K at fixed T x N, weights cycled through more
buffers than the caches hold (as in a decode step, where every layer's weights come from HBM), launches captured in one graph.
slope = per-CU operand intake (L2 -> LDS), intercept = the fixed cost of a launch that is one wave of workgroups.
    python scratch/mid_k_sweep.py        (profiles/r04_mid_batch_gemm.txt)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nvr_import
nvr = nvr_import.load(); l = nvr.lib(); nvr.check(l.nvr_device_set(0))
st = C.c_void_p(); l.nvr_stream_create(C.byref(st))

def buf(n, key):
    b = nvr.DeviceBuffer(n * 2); nvr.check(l.nvr_fill_weight(b.ptr, 1, n, n, n, 0, 0, key, 0.01, None)); return b

def graph_us(calls, reps=5):
    nvr.check(l.nvr_graph_capture_begin(st))
    for f in calls: f()
    g = C.c_void_p(); nvr.check(l.nvr_graph_capture_end(st, C.byref(g)))
    nvr.check(l.nvr_graph_launch(g, st)); nvr.check(l.nvr_stream_synchronize(st))
    best = 1e9
    for _ in range(reps):
        a, b = C.c_void_p(), C.c_void_p(); l.nvr_event_create(C.byref(a)); l.nvr_event_create(C.byref(b))
        l.nvr_event_record(a, st); nvr.check(l.nvr_graph_launch(g, st)); l.nvr_event_record(b, st); nvr.check(l.nvr_stream_synchronize(st))
        ms = C.c_float(); nvr.check(l.nvr_event_elapsed_ms(a, b, C.byref(ms))); best = min(best, ms.value * 1e3 / len(calls))
    l.nvr_graph_destroy(g)
    return best

shapes = [(512, 4096), (512, 1024), (512, 6144), (256, 4096), (256, 1024), (128, 4096), (128, 1024)]
for T, N in shapes:
    rows = []
    for K in (256, 512, 1024, 2048, 4096):
        nb = max(4, min(96, (700 << 20) // (N * K * 2)))
        Ws = [buf(N * K, 10 + i) for i in range(nb)]
        x = buf(T * K, 3); y = nvr.DeviceBuffer(T * N * 2)
        calls = [(lambda W=W: nvr.check(l.nvr_linear(x.ptr, K, W.ptr, T, K, N, y.ptr, 0, st))) for W in Ws] * (2 if nb < 48 else 1)
        us = graph_us(calls)
        rows.append((K, us))
        print(f"T={T:4d} N={N:5d} K={K:5d}: {us:7.2f} us  ({nb} weight buffers; W {N*K*2/1e6:5.1f} MB -> {N*K*2/us/1e6:5.2f} TB/s; {2*T*N*K/us/1e6:6.1f} TF/s)", flush=True)
        del Ws, x, y
    (k0, u0), (k1, u1) = rows[1], rows[3]
    slope = (u1 - u0) / (k1 - k0)
    print(f"   slope {slope*1024:6.2f} us per 1024 of K, intercept {u0 - slope*k0:5.2f} us", flush=True)
