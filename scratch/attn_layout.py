"""Does a head-major KV block layout ([NB, KVH, bs, D]) stream faster than the token-major one ([NB, bs, KVH, D])?
Emulated with the existing kernel: B=256 'sequences' x KVH=1 x H=2 reads the same bytes with the same workgroup
count as B=32 x KVH=8 x H=16, but each workgroup's rows are contiguous (256 B apart instead of 2 KB apart)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nvr_import
nvr = nvr_import.load(); l = nvr.lib(); nvr.check(l.nvr_device_set(0))
D, bs, L = 128, 256, 28
ctx_len = int(os.environ.get("CTX", "1044"))
stream = C.c_void_p(); nvr.check(l.nvr_stream_create(C.byref(stream)))
e0, e1 = C.c_void_p(), C.c_void_p(); l.nvr_event_create(C.byref(e0)); l.nvr_event_create(C.byref(e1))
scale = float(1 / np.sqrt(np.float32(D)))
def case(B, H, KVH, perm, bs=bs):
    nblk = (ctx_len + bs - 1) // bs
    NB = B * nblk
    layer_elems = NB * bs * KVH * D
    pool = nvr.DeviceBuffer(L * 2 * layer_elems * 2)
    nvr.check(l.nvr_fill_weight(pool.ptr, L * 2 * NB * bs, KVH * D, KVH * D, KVH * D, 0, 0, l.nvr_weight_key(3, 77), l.nvr_weight_scale(1.0), None))
    rng = np.random.default_rng(0)
    bt = (rng.permutation(NB) if perm else np.arange(NB)).astype(np.int32).reshape(B, nblk)
    bt = np.concatenate([bt, -np.ones((B, 1), np.int32)], 1)
    q = nvr.DeviceBuffer.from_numpy(rng.standard_normal((B, H * D)).astype(np.float16))
    d_ctx, d_bt = nvr.DeviceBuffer.from_numpy(np.full(B, ctx_len, np.int32)), nvr.DeviceBuffer.from_numpy(bt)
    out = nvr.DeviceBuffer(B * H * D * 2)
    bucket = (ctx_len + 255) // 256 * 256
    ws = nvr.DeviceBuffer(l.nvr_paged_attn_workspace_bytes(B, H, D, bucket))
    meta = nvr.AttnMetaC(); meta.context_lens, meta.block_tables, meta.max_blocks, meta.batch, meta.max_context_len = d_ctx.ptr, d_bt.ptr, nblk + 1, B, bucket
    def sweep():
        for i in range(L):
            kc = pool.ptr + (2 * i) * layer_elems * 2; vc = kc + layer_elems * 2
            nvr.check(l.nvr_paged_attn_decode(q.ptr, H * D, kc, vc, C.byref(meta), H, KVH, D, bs, scale, out.ptr, ws.ptr, stream))
    best = 1e9
    for rnd in range(4):
        sweep(); nvr.check(l.nvr_stream_synchronize(stream))
        l.nvr_event_record(e0, stream)
        for _ in range(6): sweep()
        l.nvr_event_record(e1, stream)
        ms = C.c_float(); nvr.check(l.nvr_event_elapsed_ms(e0, e1, C.byref(ms)))
        best = min(best, ms.value * 1e3 / (6 * L))
    alg = B * ctx_len * 2 * KVH * D * 2
    print(f"B={B:4d} H={H:2d} KVH={KVH} bs={bs:3d} perm={int(perm)}  {best:7.2f} us  {alg / best / 1e3:7.1f} GB/s", flush=True)
    del pool
for perm in (1, 0):
    case(32, 16, 8, perm)
    case(256, 2, 1, perm)
case(256, 2, 1, 1, bs=16)
case(32, 16, 8, 1, bs=16)
