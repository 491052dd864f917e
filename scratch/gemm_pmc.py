"""Three launches of the prefill GEMM (T=32768, K=3072, N=1024 and K=1024, N=4096) for a PMC pass."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nvr_import
nvr = nvr_import.load(); l = nvr.lib(); nvr.check(l.nvr_device_set(0))
T = 32768
def buf(n):
    b = nvr.DeviceBuffer(n * 2); nvr.check(l.nvr_fill_weight(b.ptr, 1, n, n, n, 0, 0, 7, 0.01, None)); return b
for K, N in ((3072, 1024), (1024, 4096)):
    x, W, y = buf(T * K), buf(N * K), nvr.DeviceBuffer(T * N * 2)
    for _ in range(3):
        nvr.check(l.nvr_linear(x.ptr, K, W.ptr, T, K, N, y.ptr, 0, None))
    nvr.synchronize()
print("done")
