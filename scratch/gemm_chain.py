"""Cost of the o_proj / down_proj step of a decode layer as back-to-back launches (one event pair around 64 of them,
weights cycled over 16 buffers): split-k slabs + slab norm (current) vs one non-split launch."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nvr_import
nvr = nvr_import.load(); l = nvr.lib(); nvr.check(l.nvr_device_set(0))
T, N = 32, 1024
st = C.c_void_p(); l.nvr_stream_create(C.byref(st))
e0, e1 = C.c_void_p(), C.c_void_p(); l.nvr_event_create(C.byref(e0)); l.nvr_event_create(C.byref(e1))
def chain(fn, n=64):
    best = 1e9
    for rnd in range(5):
        for i in range(n): fn(i)
        nvr.check(l.nvr_stream_synchronize(st))
        l.nvr_event_record(e0, st)
        for i in range(n): fn(i)
        l.nvr_event_record(e1, st); nvr.check(l.nvr_stream_synchronize(st))
        ms = C.c_float(); nvr.check(l.nvr_event_elapsed_ms(e0, e1, C.byref(ms))); best = min(best, ms.value * 1e3 / n)
    return best
mode = sys.argv[1] if len(sys.argv) > 1 else "all"
for K in (2048, 3072):
    Ws = [nvr.DeviceBuffer(N * K * 2) for _ in range(16)]
    for w in Ws: nvr.check(l.nvr_fill_weight(w.ptr, N, K, K, K, 0, 0, 5, 1e-6, None))
    rng = np.random.default_rng(0)
    x = nvr.DeviceBuffer.from_numpy(rng.standard_normal((T, K)).astype(np.float16))
    h = nvr.DeviceBuffer.from_numpy(rng.standard_normal((T, N)).astype(np.float16))
    g = nvr.DeviceBuffer.from_numpy(np.ones(N, np.float16)); n_out = nvr.DeviceBuffer(T * N * 2); y = nvr.DeviceBuffer(T * N * 2)
    slabs = nvr.DeviceBuffer(4 * T * N * 4)
    def pair(i):
        nvr.check(l.nvr_linear_splitk(x.ptr, K, Ws[i % 16].ptr, T, K, N, 4, slabs.ptr, st))
        nvr.check(l.nvr_add_rmsnorm_slabs(h.ptr, slabs.ptr, 4, g.ptr, 1e-6, T, N, n_out.ptr, st))
    def single(i):
        nvr.check(l.nvr_linear(x.ptr, K, Ws[i % 16].ptr, T, K, N, y.ptr, 0, st))
    def slab_only(i):
        nvr.check(l.nvr_linear_splitk(x.ptr, K, Ws[i % 16].ptr, T, K, N, 4, slabs.ptr, st))
    print(f"K={K}: splitk+slabnorm {chain(pair):6.2f} us   splitk only {chain(slab_only):6.2f} us   single launch ({'MT1 grid 64x2' if os.environ.get('NVR_LIN_MT1') else 'MT2 grid 64'}) {chain(single):6.2f} us", flush=True)
