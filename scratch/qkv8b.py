"""The qkv + RoPE + KV-store launch of a Qwen3-8B decode step (T=32, K=4096, H=32, KVH=8, D=128: 50 MB of weights) on the tiled weight copy, weights cycled
over 4 buffers (HBM-cold), us per launch; and the gate_up + SiLU launch (201 MB) beside it."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nvr_import
nvr = nvr_import.load(); l = nvr.lib(); nvr.check(l.nvr_device_set(0))
T, K, H, KVH, D, I = 32, 4096, 32, 8, 128, 12288
N = (H + 2 * KVH) * D
st = C.c_void_p(); l.nvr_stream_create(C.byref(st)); e0, e1 = C.c_void_p(), C.c_void_p(); l.nvr_event_create(C.byref(e0)); l.nvr_event_create(C.byref(e1))
rng = np.random.default_rng(0)
x = nvr.DeviceBuffer.from_numpy((rng.standard_normal((T, K)) * 0.5).astype(np.float16))
def timed(fn, n=16):
    best = 1e9
    for rnd in range(4):
        for i in range(8): fn(i)
        nvr.check(l.nvr_stream_synchronize(st)); l.nvr_event_record(e0, st)
        for i in range(n): fn(i)
        l.nvr_event_record(e1, st); nvr.check(l.nvr_stream_synchronize(st))
        ms = C.c_float(); l.nvr_event_elapsed_ms(e0, e1, C.byref(ms)); best = min(best, ms.value * 1e3 / n)
    return best
# qkv
Ws, Wts = [], []
for b in range(4):
    w = nvr.DeviceBuffer(N * K * 2); nvr.check(l.nvr_fill_weight(w.ptr, N, K, K, K, 0, 0, 5 + b, 1e-2, None))
    wt = nvr.DeviceBuffer(N * K * 2); nvr.check(l.nvr_retile_weight(w.ptr, wt.ptr, N, K, 1, H, KVH, D, None))
    Ws.append(w); Wts.append(wt)
cos, sin = nvr.DeviceBuffer(4096 * (D // 2) * 4), nvr.DeviceBuffer(4096 * (D // 2) * 4)
nvr.check(l.nvr_rope_table(4096, D, 1e6, cos.ptr, sin.ptr))
pos = nvr.DeviceBuffer.from_numpy((2048 + np.arange(T)).astype(np.int64)); slots = nvr.DeviceBuffer.from_numpy((np.arange(T) * 7).astype(np.int32))
qkv = nvr.DeviceBuffer(T * N * 2); kc = nvr.DeviceBuffer(256 * KVH * D * 2); vc = nvr.DeviceBuffer(256 * KVH * D * 2)
t = timed(lambda i: nvr.check(l.nvr_linear_qkv_rope_store_tiled(x.ptr, K, Ws[i % 4].ptr, Wts[i % 4].ptr, T, K, H, KVH, D, pos.ptr, slots.ptr, cos.ptr, sin.ptr, qkv.ptr, kc.ptr, vc.ptr, st)))
print(f"qkv + RoPE + store  {t:7.2f} us  {N * K * 2 / t / 1e6:5.2f} TB/s  crc {int(qkv.to_numpy((T, N), np.uint16).astype(np.uint64).sum())}", flush=True)
del Ws, Wts
Ws, Wts = [], []
for b in range(3):
    w = nvr.DeviceBuffer(2 * I * K * 2); nvr.check(l.nvr_fill_weight(w.ptr, 2 * I, K, K, K, 0, 0, 9 + b, 1e-2, None))
    wt = nvr.DeviceBuffer(2 * I * K * 2); nvr.check(l.nvr_retile_weight(w.ptr, wt.ptr, 2 * I, K, 0, H, KVH, D, None))
    Ws.append(w); Wts.append(wt)
out = nvr.DeviceBuffer(T * I * 2)
t = timed(lambda i: nvr.check(l.nvr_linear_silu_mul_tiled(x.ptr, K, Ws[i % 3].ptr, Wts[i % 3].ptr, T, K, I, out.ptr, st)), n=12)
print(f"gate_up + SiLU      {t:7.2f} us  {2 * I * K * 2 / t / 1e6:5.2f} TB/s  crc {int(out.to_numpy((T, I), np.uint16).astype(np.uint64).sum())}", flush=True)
os._exit(0)
