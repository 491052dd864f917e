"""Decode attention (Qwen3-0.6B heads: H=16, KVH=8, D=128) over batch x context shapes; NVR_ATTN_WAVES (temporary switch) = waves per workgroup."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nvr_import
nvr = nvr_import.load(); l = nvr.lib(); nvr.check(l.nvr_device_set(0))
H, KVH, D, bs, L = 16, 8, 128, 256, 8
stream = C.c_void_p(); nvr.check(l.nvr_stream_create(C.byref(stream)))
e0, e1 = C.c_void_p(), C.c_void_p(); l.nvr_event_create(C.byref(e0)); l.nvr_event_create(C.byref(e1))
scale = float(1 / np.sqrt(np.float32(D)))
for B, ctx_len in [(512, 100), (512, 300), (256, 200), (256, 1024), (128, 1024), (64, 1024), (1024, 100)]:
    nblk = (ctx_len + bs - 1) // bs; NB = B * nblk; layer_elems = NB * bs * KVH * D
    pool = nvr.DeviceBuffer(L * 2 * layer_elems * 2)
    nvr.check(l.nvr_fill_weight(pool.ptr, L * 2 * NB * bs, KVH * D, KVH * D, KVH * D, 0, 0, l.nvr_weight_key(3, 77), l.nvr_weight_scale(1.0), None))
    rng = np.random.default_rng(0)
    bt = np.concatenate([rng.permutation(NB).astype(np.int32).reshape(B, nblk), -np.ones((B, 1), np.int32)], 1)
    q = nvr.DeviceBuffer.from_numpy(rng.standard_normal((B, H * D)).astype(np.float16))
    ctxs = np.maximum(1, ctx_len - (np.arange(B) * 7) % max(1, ctx_len // 3)).astype(np.int32)
    d_ctx, d_bt = nvr.DeviceBuffer.from_numpy(ctxs), nvr.DeviceBuffer.from_numpy(bt)
    out = nvr.DeviceBuffer(B * H * D * 2)
    bucket = (ctx_len + 255) // 256 * 256
    ws = nvr.DeviceBuffer(l.nvr_paged_attn_workspace_bytes(B, H, D, bucket))
    meta = nvr.AttnMetaC(); meta.context_lens, meta.block_tables, meta.max_blocks, meta.batch, meta.max_context_len = d_ctx.ptr, d_bt.ptr, nblk + 1, B, bucket
    def sweep():
        for i in range(L):
            kc = pool.ptr + (2 * i) * layer_elems * 2; vc = kc + layer_elems * 2
            nvr.check(l.nvr_paged_attn_decode(q.ptr, H * D, kc, vc, C.byref(meta), H, KVH, D, bs, scale, out.ptr, ws.ptr, stream))
    base = None; line = f"B={B:5d} ctx<={ctx_len:5d}:"
    for v in [""]:
        if v: os.environ["NVR_ATTN_WAVES"] = v
        else: os.environ.pop("NVR_ATTN_WAVES", None)
        best = 1e9
        for rnd in range(3):
            sweep(); nvr.check(l.nvr_stream_synchronize(stream))
            l.nvr_event_record(e0, stream)
            for _ in range(4): sweep()
            l.nvr_event_record(e1, stream)
            ms = C.c_float(); nvr.check(l.nvr_event_elapsed_ms(e0, e1, C.byref(ms))); best = min(best, ms.value * 1e3 / (4 * L))
        res = out.to_numpy((B, H * D), np.float16).astype(np.float32)
        if base is None: base = res
        alg = int(ctxs.sum()) * 2 * KVH * D * 2
        line += f"  {v or 'now':>3s}: {best:7.2f} us {alg / best / 1e6:5.2f} TB/s (d {np.abs(res - base).max():.0e})"
    print(line, flush=True)
    del pool
os._exit(0)
