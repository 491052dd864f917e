"""Is decode attention faster when its K/V already sit in the memory-side cache (MALL, 256 MB)?  The product kernel on ONE layer pool
(144 MB, re-read every launch: cache-resident if reads allocate) against the 28 pools of a step cycled (HBM-cold)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nvr_import
nvr = nvr_import.load(); l = nvr.lib(); nvr.check(l.nvr_device_set(0))
B, H, KVH, D, BLOCK, CTX = 32, 16, 8, 128, 256, 1040
rng = np.random.default_rng(0)
st = C.c_void_p(); nvr.check(l.nvr_stream_create(C.byref(st)))
e0, e1 = C.c_void_p(), C.c_void_p(); l.nvr_event_create(C.byref(e0)); l.nvr_event_create(C.byref(e1))
nb_seq = CTX // BLOCK + 1
NB = B * nb_seq
def run(L, ctx_len):
    pools = [(nvr.DeviceBuffer(NB * BLOCK * KVH * D * 2), nvr.DeviceBuffer(NB * BLOCK * KVH * D * 2)) for _ in range(L)]
    for a, b in pools: a.zero(); b.zero()
    bt = -np.ones((B, nb_seq + 1), np.int32)
    for i in range(B): bt[i, :nb_seq] = np.arange(nb_seq) + i * nb_seq
    ctx = np.full(B, ctx_len, np.int32)
    d_q = nvr.DeviceBuffer.from_numpy(rng.standard_normal((B, H * D)).astype(np.float16))
    d_ctx, d_bt = nvr.DeviceBuffer.from_numpy(ctx), nvr.DeviceBuffer.from_numpy(bt)
    d_out = nvr.DeviceBuffer(B * H * D * 2)
    bucket = (ctx_len + 255) // 256 * 256
    ws = nvr.DeviceBuffer(l.nvr_paged_attn_workspace_bytes(B, H, D, bucket))
    meta = nvr.AttnMetaC()
    meta.context_lens, meta.block_tables, meta.max_blocks, meta.batch, meta.max_context_len = d_ctx.ptr, d_bt.ptr, nb_seq + 1, B, bucket
    scale = float(1 / np.sqrt(np.float32(D)))
    ge = C.c_void_p()
    nvr.check(l.nvr_graph_capture_begin(st))
    for rep in range(28 // L):
        for kc, vc in pools:
            nvr.check(l.nvr_paged_attn_decode(d_q.ptr, H * D, kc.ptr, vc.ptr, C.byref(meta), H, KVH, D, BLOCK, scale, d_out.ptr, ws.ptr, st))
    nvr.check(l.nvr_graph_capture_end(st, C.byref(ge)))
    for _ in range(3): nvr.check(l.nvr_graph_launch(ge, st))
    nvr.check(l.nvr_stream_synchronize(st))
    best = 1e9
    for _ in range(3):
        l.nvr_event_record(e0, st)
        for _ in range(10): nvr.check(l.nvr_graph_launch(ge, st))
        l.nvr_event_record(e1, st)
        ms = C.c_float(); nvr.check(l.nvr_event_elapsed_ms(e0, e1, C.byref(ms)))
        best = min(best, ms.value * 1e3 / (10 * 28))
    by = B * ctx_len * KVH * D * 2 * 2
    print(f"{L:2d} pool(s) x {by / 1e6:6.1f} MB, ctx {ctx_len}: {best:6.2f} us per launch (in-graph, incl. boundary)  {by / best / 1e6:6.2f} TB/s", flush=True)
    nvr.check(l.nvr_graph_destroy(ge))
run(28, 1040); run(1, 1040); run(2, 1040); run(1, 512); run(28, 512); run(1, 256); run(28, 256)
os._exit(0)
