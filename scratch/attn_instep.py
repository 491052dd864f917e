"""Does the paged attention kernel slow down when interleaved with small GEMMs (as in the real step)?"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nvr_import
nvr = nvr_import.load(); l = nvr.lib(); nvr.check(l.nvr_device_set(0))
B, H, KVH, D, bs, L = 32, 16, 8, 128, 256, 28
ctx_len = 1040; nblk = 5; NB = B * nblk; layer_elems = NB * bs * KVH * D
pool = nvr.DeviceBuffer(L * 2 * layer_elems * 2)
nvr.check(l.nvr_fill_weight(pool.ptr, L * 2 * NB * bs, KVH * D, KVH * D, KVH * D, 0, 0, 5, 0.01, None))
ctx = np.full(B, ctx_len, np.int32); rng = np.random.default_rng(0)
bt = np.concatenate([np.arange(NB, dtype=np.int32).reshape(B, nblk), -np.ones((B, 1), np.int32)], 1)
q = nvr.DeviceBuffer.from_numpy(rng.standard_normal((B, H * D)).astype(np.float16))
d_ctx, d_bt = nvr.DeviceBuffer.from_numpy(ctx), nvr.DeviceBuffer.from_numpy(bt)
out = nvr.DeviceBuffer(B * H * D * 2); ws = nvr.DeviceBuffer(l.nvr_paged_attn_workspace_bytes(B, H, D, 1280))
meta = nvr.AttnMetaC(); meta.context_lens, meta.block_tables, meta.max_blocks, meta.batch, meta.max_context_len = d_ctx.ptr, d_bt.ptr, nblk + 1, B, 1280
Ws = [nvr.DeviceBuffer(4096 * 1024 * 2) for _ in range(L)]
x = nvr.DeviceBuffer.from_numpy(rng.standard_normal((B, 1024)).astype(np.float16)); y = nvr.DeviceBuffer(B * 4096 * 2)
st = C.c_void_p(); l.nvr_stream_create(C.byref(st)); e0, e1 = C.c_void_p(), C.c_void_p(); l.nvr_event_create(C.byref(e0)); l.nvr_event_create(C.byref(e1))
scale = float(1 / np.sqrt(np.float32(D)))
def attn(i):
    kc = pool.ptr + (2 * i) * layer_elems * 2
    nvr.check(l.nvr_paged_attn_decode(q.ptr, H * D, kc, kc + layer_elems * 2, C.byref(meta), H, KVH, D, bs, scale, out.ptr, ws.ptr, st))
def gemm(i): nvr.check(l.nvr_linear(x.ptr, 1024, Ws[i].ptr, B, 1024, 4096, y.ptr, 0, st))
def run(fn, reps=6):
    fn(); nvr.check(l.nvr_stream_synchronize(st)); l.nvr_event_record(e0, st)
    for _ in range(reps): fn()
    l.nvr_event_record(e1, st); ms = C.c_float(); nvr.check(l.nvr_event_elapsed_ms(e0, e1, C.byref(ms))); return ms.value * 1e3 / (reps * L)
a = run(lambda: [attn(i) for i in range(L)])
g = run(lambda: [gemm(i) for i in range(L)])
ag = run(lambda: [(attn(i), gemm(i)) for i in range(L)])
agg = run(lambda: [(attn(i), gemm(i), gemm(i), gemm(i), gemm(i)) for i in range(L)])
print(f"attn only {a:.2f} us | gemm only {g:.2f} us | attn+gemm {ag:.2f} (sum {a+g:.2f}) | attn+4gemm {agg:.2f} (sum {a+4*g:.2f})")
