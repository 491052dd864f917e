"""us per launch of the four layer GEMMs (Qwen3-0.6B shapes) vs token count T, under the routing the environment selects
(NVR_GEMM256=0 / NVR_GEMM_TILED=0 force the smaller-tile kernels)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nvr_import
nvr = nvr_import.load(); L = nvr.lib()
Hd, H, KVH, D, I = 1024, 16, 8, 128, 3072
rng = np.random.default_rng(0)
def dev(a): return nvr.DeviceBuffer.from_numpy(a)
def h16(shape, s=0.05): return (rng.standard_normal(shape) * s).astype(np.float16).view(np.uint16)
W_qkv, W_o, W_gu, W_dn = dev(h16(((H + 2 * KVH) * D, Hd))), dev(h16((Hd, H * D))), dev(h16((2 * I, Hd))), dev(h16((Hd, I)))
Ts = [int(t) for t in os.environ.get("TS", "32,64,96,128,192,256,384,512,1024,2048,4096").split(",")]
TM = max(Ts)
x1, x2, x3 = dev(h16((TM, Hd), 1.0)), dev(h16((TM, H * D), 1.0)), dev(h16((TM, I), 1.0))
y = nvr.DeviceBuffer(TM * 4096 * 2)
NB = TM // 16 + 4
kc, vc = nvr.DeviceBuffer(NB * 16 * KVH * D * 2), nvr.DeviceBuffer(NB * 16 * KVH * D * 2)
pos = dev(np.arange(TM, dtype=np.int64) % 1024); slots = dev(np.arange(TM, dtype=np.int32))
half = D // 2
inv = 1.0 / (1e6 ** (np.arange(half, dtype=np.float64) * 2 / D))
ang = (np.arange(1024, dtype=np.float64)[:, None] * inv[None, :]).astype(np.float32)
cos_t, sin_t = dev(np.cos(ang).astype(np.float32)), dev(np.sin(ang).astype(np.float32))
def timeit(fn, reps=30):
    for _ in range(3): fn()
    nvr.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    nvr.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
print("route:", {k: os.environ.get(k) for k in ("NVR_GEMM256", "NVR_GEMM_TILED")}, flush=True)
for T in Ts:
    a = timeit(lambda: nvr.check(L.nvr_linear_qkv_rope_store(x1.ptr, Hd, W_qkv.ptr, T, Hd, H, KVH, D, pos.ptr, slots.ptr, cos_t.ptr, sin_t.ptr, y.ptr, kc.ptr, vc.ptr, None)))
    b = timeit(lambda: nvr.check(L.nvr_linear(x2.ptr, H * D, W_o.ptr, T, H * D, Hd, y.ptr, 0, None)))
    c = timeit(lambda: nvr.check(L.nvr_linear_silu_mul(x1.ptr, Hd, W_gu.ptr, T, Hd, I, y.ptr, None)))
    d = timeit(lambda: nvr.check(L.nvr_linear(x3.ptr, I, W_dn.ptr, T, I, Hd, y.ptr, 0, None)))
    print(f"T={T:5d}: qkv+rope {a:8.1f}  o {b:8.1f}  gate_up+silu {c:8.1f}  down {d:8.1f}  us", flush=True)
if not os.environ.get("NVR_NO_EXIT"): os._exit(0)
