import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nvr_import
nvr = nvr_import.load(); l = nvr.lib(); nvr.check(l.nvr_device_set(0))
T = 32
st = C.c_void_p(); l.nvr_stream_create(C.byref(st))
F = nvr.DeviceBuffer(512 << 20)
for (K, N) in [(1024, 4096), (1024, 6144), (2048, 1024)]:
    Ws = [nvr.DeviceBuffer(N * K * 2) for _ in range(8)]
    for w in Ws: nvr.check(l.nvr_fill_weight(w.ptr, N, K, K, K, 0, 0, 5, 1e-6, None))
    x = nvr.DeviceBuffer.from_numpy(np.random.default_rng(0).standard_normal((T, K)).astype(np.float16)); y = nvr.DeviceBuffer(T * N * 4)
    for ab in ["", "0", "1", "2"]:
        if ab: os.environ["NVR_LIN_ABLATE"] = ab
        else: os.environ.pop("NVR_LIN_ABLATE", None)
        evs = []
        for rep in range(24):
            if rep % 8 == 0: nvr.check(l.nvr_fill_const(F.ptr, 256 << 20, float(rep), st))
            a, b = C.c_void_p(), C.c_void_p(); l.nvr_event_create(C.byref(a)); l.nvr_event_create(C.byref(b))
            l.nvr_event_record(a, st); nvr.check(l.nvr_linear(x.ptr, K, Ws[rep % 8].ptr, T, K, N, y.ptr, 0, st)); l.nvr_event_record(b, st)
            evs.append((a, b))
        nvr.check(l.nvr_stream_synchronize(st)); ts = []
        for a, b in evs:
            ms = C.c_float(); nvr.check(l.nvr_event_elapsed_ms(a, b, C.byref(ms))); ts.append(ms.value * 1e3)
        ts = sorted(ts[4:]); print(f"K={K} N={N} ablate={ab or 'prod':5s} median {ts[len(ts)//2]:6.2f} us  min {ts[0]:6.2f}")
