import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nvr_import
nvr = nvr_import.load(); l = nvr.lib(); nvr.check(l.nvr_device_set(0))
T, K, N = 32, 1024, 151936
W = nvr.DeviceBuffer(N * K * 2); nvr.check(l.nvr_fill_weight(W.ptr, N, K, K, K, 0, 0, l.nvr_weight_key(1, 5), l.nvr_weight_scale(0.02), None))
x = nvr.DeviceBuffer.from_numpy(np.random.default_rng(0).standard_normal((T, K)).astype(np.float16))
y = nvr.DeviceBuffer(T * N * 4)
# a second big buffer to flush caches between launches
F = nvr.DeviceBuffer(512 << 20)
st = C.c_void_p(); l.nvr_stream_create(C.byref(st)); e0, e1 = C.c_void_p(), C.c_void_p(); l.nvr_event_create(C.byref(e0)); l.nvr_event_create(C.byref(e1))
pv, pi, npart = nvr.DeviceBuffer(1024 * 32 * 4), nvr.DeviceBuffer(1024 * 32 * 4), C.c_int32(0)
def run(t):
    if t and t != "old": os.environ["NVR_LMHEAD_TUNE"] = t
    else: os.environ.pop("NVR_LMHEAD_TUNE", None)
    evs = []
    for rep in range(12):                       # keep the GPU busy: flush (512 MB fill) then the GEMM, back to back
        nvr.check(l.nvr_fill_const(F.ptr, 256 << 20, float(rep), st))
        a, b = C.c_void_p(), C.c_void_p(); l.nvr_event_create(C.byref(a)); l.nvr_event_create(C.byref(b))
        l.nvr_event_record(a, st)
        if t == "old": nvr.check(l.nvr_linear(x.ptr, K, W.ptr, T, K, N, y.ptr, 1, st))
        else: nvr.check(l.nvr_lm_head(x.ptr, K, W.ptr, T, K, N, None if os.environ.get('NOSTORE') else y.ptr, pv.ptr, pi.ptr, C.byref(npart), st))
        l.nvr_event_record(b, st)
        evs.append((a, b))
    nvr.check(l.nvr_stream_synchronize(st))
    ts = []
    for a, b in evs:
        ms = C.c_float(); nvr.check(l.nvr_event_elapsed_ms(a, b, C.byref(ms))); ts.append(ms.value * 1e3)
    best = sorted(ts[2:])[len(ts[2:]) // 2]
    return best, y.to_numpy((T, 64), np.float32)[0, :4]
for t in sys.argv[1:] or ["old", "", "4,8,1", "4,8,2", "8,4,1", "8,4,2", "4,4,1", "4,4,2", "8,8,1", "8,8,2", "2,8,1", "2,8,2", "2,16,1", "4,16,1", "2,4,2"]:
    us, v = run(t); print(f"{t or 'default':14s} {us:8.1f} us  {N*K*2/us/1e3:7.1f} GB/s", v[:2], flush=True)
