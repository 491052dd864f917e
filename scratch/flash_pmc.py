"""A few launches of the flash prefill kernel at the bench shape (32 x 1024 tokens, H=16, KVH=8, D=128) for PMC passes
and for timing (the stateless entry point builds its tile list per call: only kernel time is meaningful)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nvr_import
nvr = nvr_import.load(); l = nvr.lib(); nvr.check(l.nvr_device_set(0))
B, L, H, KVH, D = 32, int(os.environ.get("SEQ", "1024")), 16, 8, 128
T, ld = B * L, (H + 2 * KVH) * D
qkv = nvr.DeviceBuffer(T * ld * 2)
nvr.check(l.nvr_fill_weight(qkv.ptr, T, ld, ld, ld, 0, 0, l.nvr_weight_key(3, 9), l.nvr_weight_scale(1.0), None))
cu = nvr.DeviceBuffer.from_numpy((np.arange(B + 1) * L).astype(np.int32))
out = nvr.DeviceBuffer(T * H * D * 2)
meta = nvr.AttnMetaC(); meta.is_prefill, meta.cu_seqlens_q, meta.batch, meta.max_context_len = 1, cu.ptr, B, L
scale = float(1 / np.sqrt(np.float32(D)))
st = C.c_void_p(); l.nvr_stream_create(C.byref(st)); e0, e1 = C.c_void_p(), C.c_void_p(); l.nvr_event_create(C.byref(e0)); l.nvr_event_create(C.byref(e1))
for rep in range(int(os.environ.get("REPS", "4"))):
    l.nvr_event_record(e0, st)
    nvr.check(l.nvr_attn_prefill_varlen(qkv.ptr, qkv.ptr + H * D * 2, qkv.ptr + (H + KVH) * D * 2, ld, C.byref(meta), T, H, KVH, D, scale, out.ptr, st))
    l.nvr_event_record(e1, st); nvr.check(l.nvr_stream_synchronize(st))
    ms = C.c_float(); l.nvr_event_elapsed_ms(e0, e1, C.byref(ms))
    print(f"launch {rep}: {ms.value * 1e3:8.1f} us incl. host tile build", flush=True)
