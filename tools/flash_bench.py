"""flash prefill alone: B sequences x L tokens of Qwen3-0.6B geometry (H 16, KVH 8, D 128), HIP-event timing (median) — for A/B of
library builds (NVR_LIBNVR=...) and counter passes.  python tools/flash_bench.py [L] [H] [KVH] [reps]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nvr_import
nvr = nvr_import.load(); l = nvr.lib(); nvr.check(l.nvr_device_set(0))
L = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
H = int(sys.argv[2]) if len(sys.argv) > 2 else 16
KVH = int(sys.argv[3]) if len(sys.argv) > 3 else 8
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 12
D, T = 128, 32768
B = T // L
st = C.c_void_p(); l.nvr_stream_create(C.byref(st))
QKV = (H + 2 * KVH) * D
y = nvr.DeviceBuffer(T * QKV * 2); nvr.check(l.nvr_fill_weight(y.ptr, 1, T * QKV, T * QKV, T * QKV, 0, 0, 7, 0.02, None))
cu = nvr.DeviceBuffer.from_numpy((np.arange(B + 1) * L).astype(np.int32))
meta = nvr.AttnMetaC(); meta.is_prefill = 1; meta.cu_seqlens_q = cu.ptr; meta.cu_seqlens_k = cu.ptr; meta.max_seqlen_q = L; meta.max_seqlen_k = L; meta.batch = B
out = nvr.DeviceBuffer(T * H * D * 2)
qp = y.ptr; kp = y.ptr + H * D * 2; vp = y.ptr + (H + KVH) * D * 2
evs = []
for rep in range(reps):
    a, b = C.c_void_p(), C.c_void_p(); l.nvr_event_create(C.byref(a)); l.nvr_event_create(C.byref(b))
    l.nvr_event_record(a, st)
    nvr.check(l.nvr_attn_prefill_varlen(qp, kp, vp, QKV, C.byref(meta), T, H, KVH, D, float(1 / np.sqrt(D)), out.ptr, st))
    l.nvr_event_record(b, st); evs.append((a, b))
nvr.check(l.nvr_stream_synchronize(st)); ts = []
for a, b in evs:
    ms = C.c_float(); nvr.check(l.nvr_event_elapsed_ms(a, b, C.byref(ms))); ts.append(ms.value * 1e3)
ts = sorted(ts[2:]); us = ts[len(ts) // 2]
fl = 4 * H * D * B * (L * (L + 1) // 2)
print(f"flash {B} x {L} H={H} KVH={KVH}: {us:8.1f} us (min {ts[0]:.1f})  {fl/us/1e6:7.1f} TF/s causal  lib={os.environ.get('NVR_LIBNVR','default')}")
