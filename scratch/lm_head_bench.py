"""r06: LM head over mid / large batches: us per launch of nvr_lm_head (logits NULL: arg-max partials only) by row count.
T < 256: 128x128 tiles (gemm_tiled_lm_head); T >= 256: 256x256 tiles (gemm256_lm_head, r06)."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nvr_import
nvr = nvr_import.load(); l = nvr.lib()
K, N = 1024, 151936
rng = np.random.default_rng(0)
W = nvr.DeviceBuffer.from_numpy((rng.standard_normal((N, K), dtype=np.float32) * 0.05).astype(np.float16))
st = C.c_void_p(); nvr.check(l.nvr_stream_create(C.byref(st)))
e0, e1 = C.c_void_p(), C.c_void_p(); nvr.check(l.nvr_event_create(C.byref(e0))); nvr.check(l.nvr_event_create(C.byref(e1)))
for T in [int(a) for a in sys.argv[1:]] or [64, 128, 255, 256, 384, 512, 768, 1024]:
    x = nvr.DeviceBuffer.from_numpy(rng.standard_normal((T, K)).astype(np.float16))
    pv, pi = nvr.DeviceBuffer(2048 * T * 4), nvr.DeviceBuffer(2048 * T * 4)
    lg = nvr.DeviceBuffer(T * N * 4)
    for logits in (None, lg.ptr):
        npart = C.c_int32(0)
        for _ in range(3):
            nvr.check(l.nvr_lm_head(x.ptr, K, W.ptr, T, K, N, logits, pv.ptr, pi.ptr, C.byref(npart), st))
        nvr.check(l.nvr_stream_synchronize(st))
        reps = 20
        nvr.check(l.nvr_event_record(e0, st))
        for _ in range(reps):
            nvr.check(l.nvr_lm_head(x.ptr, K, W.ptr, T, K, N, logits, pv.ptr, pi.ptr, C.byref(npart), st))
        nvr.check(l.nvr_event_record(e1, st))
        nvr.check(l.nvr_stream_synchronize(st))
        ms = C.c_float(); nvr.check(l.nvr_event_elapsed_ms(e0, e1, C.byref(ms)))
        us = ms.value * 1e3 / reps
        print(f"T={T:5d} logits={'f32 stored' if logits else 'none      '} parts={npart.value:4d}  {us:8.1f} us  {2.0 * T * K * N / us / 1e9:7.1f} TFLOP/s", flush=True)
