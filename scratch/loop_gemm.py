"""Loops one prefill kernel for ~6 s (power / clock probe): python scratch/loop_gemm.py silu|resid|rope|flash|rmsnorm"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nvr_import
nvr = nvr_import.load(); l = nvr.lib(); nvr.check(l.nvr_device_set(0))
which = sys.argv[1]; T = 32768
def buf(n):
    b = nvr.DeviceBuffer(n * 2); nvr.check(l.nvr_fill_weight(b.ptr, 1, n, n, n, 0, 0, 7, 0.02, None)); return b
H, KVH, D = 16, 8, 128
if which == "silu":
    x = buf(T * 1024); W = buf(6144 * 1024); y = nvr.DeviceBuffer(T * 3072 * 2)
    f = lambda: l.nvr_linear_silu_mul(x.ptr, 1024, W.ptr, T, 1024, 3072, y.ptr, None)
elif which == "resid":
    x = buf(T * 3072); W = buf(1024 * 3072); y = buf(T * 1024)
    f = lambda: l.nvr_linear_add_residual(x.ptr, 3072, W.ptr, T, 3072, 1024, y.ptr, None)
elif which == "flash":
    QKV = (H + 2 * KVH) * D
    y = buf(T * QKV); cu = nvr.DeviceBuffer.from_numpy((np.arange(33) * 1024).astype(np.int32))
    meta = nvr.AttnMetaC(); meta.is_prefill = 1; meta.cu_seqlens_q = cu.ptr; meta.cu_seqlens_k = cu.ptr; meta.max_seqlen_q = 1024; meta.max_seqlen_k = 1024; meta.batch = 32
    out = nvr.DeviceBuffer(T * H * D * 2)
    f = lambda: l.nvr_attn_prefill_varlen(y.ptr, y.ptr + H * D * 2, y.ptr + (H + KVH) * D * 2, QKV, C.byref(meta), T, H, KVH, D, float(1 / np.sqrt(D)), out.ptr, None)
else:
    x = buf(T * 1024); w = buf(1024); y = nvr.DeviceBuffer(T * 1024 * 2)
    f = lambda: l.nvr_rmsnorm(x.ptr, w.ptr, 1e-6, T, 1024, y.ptr, None)
t0 = time.time()
while time.time() - t0 < 6:
    for _ in range(50): nvr.check(f())
    nvr.synchronize()
