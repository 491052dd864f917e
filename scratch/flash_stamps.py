"""Diagnostic (libnvr_fst.so = tools/build_variant.sh fst flash_prefill.hip "-DNVR_FLASH_STAMPS=1"): where wave 0 of two workgroups of the flash
prefill kernel spends a 64-key step (shader-clock stamps): workgroup 0 (first dispatched: its CU is alone at first) and workgroup 1031."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("NVR_LIBNVR", os.path.join(ROOT, "nano-vllm-rs_amd", "libnvr_fst.so"))
sys.path.insert(0, ROOT)
import numpy as np, nvr_import
nvr = nvr_import.load(); l = nvr.lib(); nvr.check(l.nvr_device_set(0))
L, H, KVH, D, T = 1024, 16, 8, 128, 32768
B = T // L
QKV = (H + 2 * KVH) * D
y = nvr.DeviceBuffer(T * QKV * 2); nvr.check(l.nvr_fill_weight(y.ptr, 1, T * QKV, T * QKV, T * QKV, 0, 0, 7, 0.02, None))
cu = nvr.DeviceBuffer.from_numpy((np.arange(B + 1) * L).astype(np.int32))
meta = nvr.AttnMetaC(); meta.is_prefill = 1; meta.cu_seqlens_q = cu.ptr; meta.cu_seqlens_k = cu.ptr; meta.max_seqlen_q = L; meta.max_seqlen_k = L; meta.batch = B
out = nvr.DeviceBuffer(T * H * D * 2)
for _ in range(3):
    nvr.check(l.nvr_attn_prefill_varlen(y.ptr, y.ptr + H * D * 2, y.ptr + (H + KVH) * D * 2, QKV, C.byref(meta), T, H, KVH, D, float(1 / np.sqrt(D)), out.ptr, None))
nvr.synchronize()
raw = C.CDLL(os.environ["NVR_LIBNVR"])
buf = (C.c_uint64 * (2 * 64 * 8))()
assert raw.nvr_debug_flash_stamps(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(2, 64, 8).astype(np.int64)
names = ["stage issue", "QK (reads + 32 MFMA)", "mask + max + exp", "rescale test", "PV (reads + 36 MFMA)", "vmcnt wait", "barrier"]
for w, blk in enumerate((120, 2168)):
    steps = [i for i in range(64) if a[w, i, 0] > 0]
    print(f"workgroup {blk}: {len(steps)} steps stamped")
    for i in steps[:16]:
        st = a[w, i]
        d = [st[1] - st[0], st[2] - st[1], st[3] - st[2], 0, st[4] - st[3], st[5] - st[4], st[6] - st[5]]
        nxt = a[w, i + 1, 0] - st[6] if i + 1 in steps else 0
        print(f"  step {i:2d}: " + "  ".join(f"{n}={v}" for n, v in zip(names, d) if n != "rescale test") + f"  | total {st[6] - st[0]}  to next {nxt}")
