"""Split-k slab GEMM variants at the Qwen3-8B o/down shapes (N=4096), T=32: tile columns per workgroup (NT) x k-slices (S)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nvr_import
nvr = nvr_import.load(); l = nvr.lib(); nvr.check(l.nvr_device_set(0))
T, N = 32, 4096
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
for K in (4096, 12288):
    Ws = [nvr.DeviceBuffer(N * K * 2) for _ in range(4)]
    for w in Ws: nvr.check(l.nvr_fill_weight(w.ptr, N, K, K, K, 0, 0, 5, 1e-3, None))
    x = nvr.DeviceBuffer.from_numpy(np.random.default_rng(0).standard_normal((T, K)).astype(np.float16))
    slabs = nvr.DeviceBuffer(8 * T * N * 4)
    for nt in ("4", "2", "28", "48", "1"):
        for S in (1, 2, 4, 8):
            if K % (32 * S): continue
            if nt == "1": os.environ.pop("NVR_SPLITK_NT", None)
            else: os.environ["NVR_SPLITK_NT"] = nt
            us = timeit(lambda i: nvr.check(l.nvr_linear_splitk(x.ptr, K, Ws[i % 4].ptr, T, K, N, S, slabs.ptr, st)))
            print(f"K={K:5d} NT/waves code {nt:>2s} S={S}: {us:7.2f} us  {N * K * 2 / us / 1e3:7.1f} GB/s", flush=True)
    del Ws
