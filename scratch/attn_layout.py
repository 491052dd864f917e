"""Does a head-major KV layout ([block][head][token][D]: one kv head's rows of a block contiguous) stream faster than the
reference's token-major [block][token][head][D]?  Emulated with the product kernel: 32 sequences x 8 kv heads (token-major,
256-B row pieces at 2 KiB stride) against 256 "sequences" x 1 kv head (every block 64 KiB contiguous) — the same 256 workgroups,
the same bytes, cold pools cycled like the 28 layers of a step."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nvr_import
nvr = nvr_import.load(); l = nvr.lib(); nvr.check(l.nvr_device_set(0))
D, BLOCK, CTX, L = 128, 256, 1040, 28
rng = np.random.default_rng(0)
st = C.c_void_p(); nvr.check(l.nvr_stream_create(C.byref(st)))
e0, e1 = C.c_void_p(), C.c_void_p(); l.nvr_event_create(C.byref(e0)); l.nvr_event_create(C.byref(e1))
def run(B, H, KVH, order):
    nb_seq = CTX // BLOCK + 1
    NB = B * nb_seq
    pools = [(nvr.DeviceBuffer(NB * BLOCK * KVH * D * 2), nvr.DeviceBuffer(NB * BLOCK * KVH * D * 2)) for _ in range(L)]
    for a, b in pools: a.zero(); b.zero()
    bt = -np.ones((B, nb_seq + 1), np.int32)
    ids = np.arange(NB) if order == "seq" else rng.permutation(NB)
    for i in range(B): bt[i, :nb_seq] = ids[i * nb_seq:(i + 1) * nb_seq]
    ctx = np.full(B, CTX, np.int32)
    d_q = nvr.DeviceBuffer.from_numpy(rng.standard_normal((B, H * D)).astype(np.float16))
    d_ctx, d_bt = nvr.DeviceBuffer.from_numpy(ctx), nvr.DeviceBuffer.from_numpy(bt)
    d_out = nvr.DeviceBuffer(B * H * D * 2)
    bucket = (CTX + 255) // 256 * 256
    ws = nvr.DeviceBuffer(l.nvr_paged_attn_workspace_bytes(B, H, D, bucket))
    meta = nvr.AttnMetaC()
    meta.context_lens, meta.block_tables, meta.max_blocks, meta.batch, meta.max_context_len = d_ctx.ptr, d_bt.ptr, nb_seq + 1, B, bucket
    scale = float(1 / np.sqrt(np.float32(D)))
    def sweep():
        for kc, vc in pools:
            nvr.check(l.nvr_paged_attn_decode(d_q.ptr, H * D, kc.ptr, vc.ptr, C.byref(meta), H, KVH, D, BLOCK, scale, d_out.ptr, ws.ptr, st))
    sweep(); nvr.check(l.nvr_stream_synchronize(st))
    best = 1e9
    for _ in range(3):
        l.nvr_event_record(e0, st)
        for _ in range(10): sweep()
        l.nvr_event_record(e1, st)
        ms = C.c_float(); nvr.check(l.nvr_event_elapsed_ms(e0, e1, C.byref(ms)))
        best = min(best, ms.value * 1e3 / (10 * L))
    by = B * CTX * KVH * D * 2 * 2
    print(f"B={B:4d} H={H:2d} KVH={KVH} blocks {order:6s}: {best:6.2f} us per launch  {by / best / 1e6:6.2f} TB/s", flush=True)
run(32, 16, 8, "seq"); run(32, 16, 8, "random"); run(256, 2, 1, "seq"); run(256, 2, 1, "random")
os._exit(0)
