"""Does a preceding plain-load sweep over a layer's K/V (pulling them into the memory-side cache) speed up the decode attention
kernel's non-temporal stream?  Run under rocprofv3 --kernel-trace --stats: compare attn_rows_kernel's average duration in the two
phases (the reader is nvr_argmax over the pools viewed as f32 rows)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nvr_import
nvr = nvr_import.load(); l = nvr.lib(); nvr.check(l.nvr_device_set(0))
B, H, KVH, D, BLOCK, CTX, L = 32, 16, 8, 128, 256, 1040, 28
rng = np.random.default_rng(0)
nb_seq = CTX // BLOCK + 1
NB = B * nb_seq
pool_bytes = NB * BLOCK * KVH * D * 2
pools = [(nvr.DeviceBuffer(pool_bytes), nvr.DeviceBuffer(pool_bytes)) for _ in range(L)]
for a, b in pools: a.zero(); b.zero()
bt = -np.ones((B, nb_seq + 1), np.int32)
for i in range(B): bt[i, :nb_seq] = np.arange(nb_seq) + i * nb_seq
d_q = nvr.DeviceBuffer.from_numpy(rng.standard_normal((B, H * D)).astype(np.float16))
d_ctx, d_bt = nvr.DeviceBuffer.from_numpy(np.full(B, CTX, np.int32)), nvr.DeviceBuffer.from_numpy(bt)
d_out = nvr.DeviceBuffer(B * H * D * 2)
bucket = (CTX + 255) // 256 * 256
ws = nvr.DeviceBuffer(l.nvr_paged_attn_workspace_bytes(B, H, D, bucket))
meta = nvr.AttnMetaC()
meta.context_lens, meta.block_tables, meta.max_blocks, meta.batch, meta.max_context_len = d_ctx.ptr, d_bt.ptr, nb_seq + 1, B, bucket
scale = float(1 / np.sqrt(np.float32(D)))
rows = 256; V = pool_bytes // 4 // rows
d_idx = nvr.DeviceBuffer(rows * 8)
mode = sys.argv[1]
for rep in range(6):
    for kc, vc in pools:
        if mode == "prefetch":
            nvr.check(l.nvr_argmax(kc.ptr, rows, V, d_idx.ptr, None)); nvr.check(l.nvr_argmax(vc.ptr, rows, V, d_idx.ptr, None))
        nvr.check(l.nvr_paged_attn_decode(d_q.ptr, H * D, kc.ptr, vc.ptr, C.byref(meta), H, KVH, D, BLOCK, scale, d_out.ptr, ws.ptr, None))
nvr.synchronize()
