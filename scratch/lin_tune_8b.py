"""Plain decode GEMM variants (NT x WAVES) at Qwen3-8B shapes, T=32, weights cycled over 4 buffers (HBM-cold)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nvr_import
nvr = nvr_import.load(); l = nvr.lib(); nvr.check(l.nvr_device_set(0))
T = 32
st = C.c_void_p(); l.nvr_stream_create(C.byref(st)); e0, e1 = C.c_void_p(), C.c_void_p(); l.nvr_event_create(C.byref(e0)); l.nvr_event_create(C.byref(e1))
for K, N in ((4096, 4096), (12288, 4096), (4096, 6144), (4096, 24576)):
    Ws = [nvr.DeviceBuffer(N * K * 2) for _ in range(4)]
    for w in Ws: nvr.check(l.nvr_fill_weight(w.ptr, N, K, K, K, 0, 0, 5, 1e-3, None))
    x = nvr.DeviceBuffer.from_numpy(np.random.default_rng(0).standard_normal((T, K)).astype(np.float16)); y = nvr.DeviceBuffer(T * N * 2)
    for v in ["1,16", "1,8", "2,16", "2,8", "2,4", "4,16", "4,8", "4,4"]:
        os.environ["NVR_LIN_TUNE"] = v
        best = 1e9
        for rnd in range(3):
            for i in range(8): nvr.check(l.nvr_linear(x.ptr, K, Ws[i % 4].ptr, T, K, N, y.ptr, 0, st))
            nvr.check(l.nvr_stream_synchronize(st)); l.nvr_event_record(e0, st)
            for i in range(16): nvr.check(l.nvr_linear(x.ptr, K, Ws[i % 4].ptr, T, K, N, y.ptr, 0, st))
            l.nvr_event_record(e1, st); nvr.check(l.nvr_stream_synchronize(st))
            ms = C.c_float(); l.nvr_event_elapsed_ms(e0, e1, C.byref(ms)); best = min(best, ms.value * 1e3 / 16)
        print(f"K={K:5d} N={N:5d} NTxW={v:5s} {best:8.2f} us  {N * K * 2 / best / 1e3:7.1f} GB/s", flush=True)
    del Ws
