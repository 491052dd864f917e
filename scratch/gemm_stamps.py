"""Diagnostic (libnvr_exp.so built with -DG256_STAMPS=1): where workgroup 0 of gemm256<plain> spends a tile (shader-clock stamps)."""
import ctypes as C, os, sys
os.environ.setdefault("NVR_LIBNVR", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nano-vllm-rs_amd", "libnvr_exp.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nvr_import
nvr = nvr_import.load(); l = nvr.lib(); nvr.check(l.nvr_device_set(0))
T, N = 32768, 4096
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
def buf(n):
    b = nvr.DeviceBuffer(n * 2); nvr.check(l.nvr_fill_weight(b.ptr, 1, n, n, n, 0, 0, 7, 0.01, None)); return b
x = buf(T * K); W = buf(N * K); y = nvr.DeviceBuffer(T * N * 2)
mode = sys.argv[2] if len(sys.argv) > 2 else "plain"
H, KVH, D = 16, 8, 128
if mode == "rope":
    pos = nvr.DeviceBuffer.from_numpy((np.arange(T) % 1024).astype(np.int64)); slots = nvr.DeviceBuffer.from_numpy(np.arange(T, dtype=np.int32))
    cos = nvr.DeviceBuffer(1024 * 64 * 4); sin = nvr.DeviceBuffer(1024 * 64 * 4); nvr.check(l.nvr_rope_table(D, 1024, 1e6, cos.ptr, sin.ptr))
    kc = nvr.DeviceBuffer(T * KVH * D * 2); vc = nvr.DeviceBuffer(T * KVH * D * 2)
if mode == "silu":
    W = buf(6144 * K)
if mode == "resid":
    N = 1024; W = buf(N * K)
for _ in range(3):
    if mode == "plain": nvr.check(l.nvr_linear(x.ptr, K, W.ptr, T, K, N, y.ptr, 0, None))
    elif mode == "rope": nvr.check(l.nvr_linear_qkv_rope_store(x.ptr, K, W.ptr, T, K, H, KVH, D, pos.ptr, slots.ptr, cos.ptr, sin.ptr, y.ptr, kc.ptr, vc.ptr, None))
    elif mode == "silu": nvr.check(l.nvr_linear_silu_mul(x.ptr, K, W.ptr, T, K, 3072, y.ptr, None))
    elif mode == "resid": nvr.check(l.nvr_linear_add_residual(x.ptr, K, W.ptr, T, K, N, y.ptr, None))
nvr.synchronize()
raw = C.CDLL(os.environ["NVR_LIBNVR"])
out = (C.c_uint64 * (2 * 64 * 8))()
assert raw.nvr_debug_g256_stamps(out) == 0
a = np.frombuffer(out, dtype=np.uint64).reshape(2, 64, 8).astype(np.int64)
names = ["tile start -> first R_END", "-> K-tile 0 done", "-> K loop done (+realign)", "-> staged hB0 (sync)", "-> hB0 stores issued (sync)", "-> hB1 stored (sync)"]
for grp in (0, 1):
    print(f"group {grp} (wave {grp * 4}), tiles 1..6, cycles of the shader clock:")
    for t in range(1, 7):
        st = a[grp, t]
        d = [st[i + 1] - st[i] for i in range(6)]
        nxt = a[grp, t + 1][0] - st[6]
        print(f"  tile {t}: " + "  ".join(f"{n.split('->')[-1].strip()[:22]}={v}" for n, v in zip(names, d)) + f"  | to next tile start={nxt}  total={a[grp, t + 1][0] - st[0]}")
k = (C.c_uint64 * 16)()
assert raw.nvr_debug_g256_kstamps(k) == 0
k = np.frombuffer(k, dtype=np.uint64).reshape(2, 8).astype(np.int64)
base = k.min()
for grp in (0, 1):
    print(f"group {grp}: barrier exits of K-tile 5 (R0 M0 R1 M1 R2 M2 R3 M3), cycles from the first:", (k[grp] - base).tolist(), "intervals", np.diff(k[grp]).tolist())
sys.stdout.flush(); os._exit(0)
