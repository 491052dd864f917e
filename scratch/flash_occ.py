"""Diagnostic (libnvr_occ.so): start / end wall-clock stamps and hardware ids of EVERY workgroup of one flash prefill launch ->
how many workgroups a CU really runs at a time."""
import ctypes as C, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("NVR_LIBNVR", os.path.join(ROOT, "nano-vllm-rs_amd", "libnvr_occ.so"))
sys.path.insert(0, ROOT)
import numpy as np, nvr_import
nvr = nvr_import.load(); l = nvr.lib(); nvr.check(l.nvr_device_set(0))
L, H, KVH, D, T = 1024, 16, 8, 128, 32768
B = T // L; QKV = (H + 2 * KVH) * D
y = nvr.DeviceBuffer(T * QKV * 2); nvr.check(l.nvr_fill_weight(y.ptr, 1, T * QKV, T * QKV, T * QKV, 0, 0, 7, 0.02, None))
cu = nvr.DeviceBuffer.from_numpy((np.arange(B + 1) * L).astype(np.int32))
meta = nvr.AttnMetaC(); meta.is_prefill = 1; meta.cu_seqlens_q = cu.ptr; meta.cu_seqlens_k = cu.ptr; meta.max_seqlen_q = L; meta.max_seqlen_k = L; meta.batch = B
out = nvr.DeviceBuffer(T * H * D * 2)
for _ in range(3):
    nvr.check(l.nvr_attn_prefill_varlen(y.ptr, y.ptr + H * D * 2, y.ptr + (H + KVH) * D * 2, QKV, C.byref(meta), T, H, KVH, D, float(1 / np.sqrt(D)), out.ptr, None))
nvr.synchronize()
raw = C.CDLL(os.environ["NVR_LIBNVR"])
buf = (C.c_uint64 * (8192 * 4))(); assert raw.nvr_debug_flash_wg(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 4).astype(np.int64)
a = a[a[:, 0] > 0]
t0 = a[:, 0].min(); print(f"{len(a)} workgroups, launch span {(a[:, 1].max() - t0) / 100:.1f} us")
hw = a[:, 2]; xcc = a[:, 3] & 0xf
cu_id = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key = xcc * 1000 + se * 100 + sh * 20 + cu_id
cus = collections.defaultdict(list)
for k, s, e in zip(key, a[:, 0], a[:, 1]): cus[int(k)].append((int(s - t0), int(e - t0)))
print(f"{len(cus)} distinct (xcc, se, sh, cu) ids")
conc = []; busy = []
for k, iv in cus.items():
    ev = sorted([(s, 1) for s, e in iv] + [(e, -1) for s, e in iv])
    cur = 0; last = 0; area = 0; mx = 0; b = 0
    for t, d in ev:
        area += cur * (t - last); b += (t - last) if cur > 0 else 0; last = t; cur += d; mx = max(mx, cur)
    conc.append((area / max(b, 1), mx, len(iv), b / 100))
c = np.asarray(conc)
print(f"per CU: workgroups {c[:,2].mean():.1f} (min {c[:,2].min():.0f}, max {c[:,2].max():.0f}); mean concurrency while busy {c[:,0].mean():.2f}; max concurrency {c[:,1].max():.0f} (median {np.median(c[:,1]):.0f}); busy {c[:,3].mean():.1f} us (min {c[:,3].min():.1f}, max {c[:,3].max():.1f})")
d = (a[:, 1] - a[:, 0]) / 100.0
print(f"workgroup lifetime: mean {d.mean():.1f} us, min {d.min():.1f}, max {d.max():.1f}")
