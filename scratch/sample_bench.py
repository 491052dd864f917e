"""nvr_sample alone: us per launch for B rows of V f32 logits under different filters (wall clock over back-to-back launches)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nvr_import
nvr = nvr_import.load()
L = nvr.lib()
B, V = int(os.environ.get("B", "32")), int(os.environ.get("V", "151936"))
rng = np.random.default_rng(0)
scale = float(os.environ.get("SCALE", "0.05"))           # random-init logits are nearly flat; SCALE=3 gives a peaked row
x = (rng.standard_normal((B, V)) * scale).astype(np.float32)
d_x = nvr.DeviceBuffer.from_numpy(x)
ws = nvr.DeviceBuffer(L.nvr_sample_workspace_bytes(B, V)); d_out = nvr.DeviceBuffer(B * 8)
keys = nvr.DeviceBuffer.from_numpy(np.arange(B, dtype=np.uint64) * 7919 + 1)
def run(name, temp, k, p, reps=50):
    t = nvr.DeviceBuffer.from_numpy(np.full(B, temp, np.float32)); kk = nvr.DeviceBuffer.from_numpy(np.full(B, k, np.int64))
    pp = nvr.DeviceBuffer.from_numpy(np.full(B, p, np.float32))
    for _ in range(3): nvr.check(L.nvr_sample(d_x.ptr, B, V, t.ptr, kk.ptr, pp.ptr, keys.ptr, d_out.ptr, ws.ptr, None))
    nvr.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): nvr.check(L.nvr_sample(d_x.ptr, B, V, t.ptr, kk.ptr, pp.ptr, keys.ptr, d_out.ptr, ws.ptr, None))
    nvr.synchronize(); dt = (time.perf_counter() - t0) / reps
    print(f"B={B} V={V} {name:26s}: {dt * 1e6:8.1f} us/launch  tokens {d_out.to_numpy((B,), np.int64)[:4].tolist()}", flush=True)
run("greedy", 0.0, 0, -1.0)
run("temperature 0.8", 0.8, 0, -1.0)
run("top_k 50", 0.8, 50, -1.0)
run("top_p 0.9", 0.8, 0, 0.9)
run("top_k 50 + top_p 0.9", 0.8, 50, 0.9)
if not os.environ.get("NVR_NO_EXIT"): os._exit(0)
