"""Where does a decode-attention launch spend its time?  Needs libnvr.so built with -DNVR_ATTN_EXPERIMENTS;
NVR_ATTN_STAMPS=1 makes workgroup thread 0 write wall_clock64() (100 MHz) at: entry, ctx known, first round done,
loop + remainder done, output stored."""
import ctypes as C, os, sys
os.environ["NVR_ATTN_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nvr_import
nvr = nvr_import.load(); l = nvr.lib(); nvr.check(l.nvr_device_set(0))
B, H, KVH, D, bs, L = 32, 16, 8, 128, 256, 28
ctx_len = int(os.environ.get("CTX", "1044"))
nblk = (ctx_len + bs - 1) // bs; NB = B * nblk; layer_elems = NB * bs * KVH * D
pool = nvr.DeviceBuffer(L * 2 * layer_elems * 2)
nvr.check(l.nvr_fill_weight(pool.ptr, L * 2 * NB * bs, KVH * D, KVH * D, KVH * D, 0, 0, l.nvr_weight_key(3, 77), l.nvr_weight_scale(1.0), None))
rng = np.random.default_rng(0)
bt = np.concatenate([rng.permutation(NB).astype(np.int32).reshape(B, nblk), -np.ones((B, 1), np.int32)], 1)
q = nvr.DeviceBuffer.from_numpy(rng.standard_normal((B, H * D)).astype(np.float16))
d_ctx, d_bt = nvr.DeviceBuffer.from_numpy(np.full(B, ctx_len, np.int32)), nvr.DeviceBuffer.from_numpy(bt)
out = nvr.DeviceBuffer(B * H * D * 2)
bucket = (ctx_len + 255) // 256 * 256
wsb = max(l.nvr_paged_attn_workspace_bytes(B, H, D, bucket), B * KVH * 24 * 8)
wss = [nvr.DeviceBuffer(wsb) for _ in range(L)]
meta = nvr.AttnMetaC(); meta.context_lens, meta.block_tables, meta.max_blocks, meta.batch, meta.max_context_len = d_ctx.ptr, d_bt.ptr, nblk + 1, B, bucket
stream = C.c_void_p(); nvr.check(l.nvr_stream_create(C.byref(stream)))
scale = float(1 / np.sqrt(np.float32(D)))
if len(sys.argv) > 1: os.environ["NVR_ATTN_TUNE"] = sys.argv[1]
def sweep():
    for i in range(L):
        kc = pool.ptr + (2 * i) * layer_elems * 2; vc = kc + layer_elems * 2
        nvr.check(l.nvr_paged_attn_decode(q.ptr, H * D, kc, vc, C.byref(meta), H, KVH, D, bs, scale, out.ptr, wss[i].ptr, stream))
for _ in range(3): sweep()
nvr.check(l.nvr_stream_synchronize(stream))
nwg = B * KVH
st = np.stack([w.to_numpy((wsb // 8,), np.uint64)[:nwg * 24].reshape(nwg, 24).astype(np.int64) for w in wss[4:]])  # [layers, wg, 5]
t0 = st[:, :, 0].min(axis=1, keepdims=True)
rel = (st - t0[:, :, None]) * 10.0 / 1e3       # us since first workgroup entry
names = ["entry", "ctx known", "first round done", "loop+remainder done (wave 0)", "stored", "after barrier"]
for i, n in enumerate(names):
    print(f"{n:22s} mean {rel[:, :, i].mean():6.2f} us   min {rel[:, :, i].min(axis=1).mean():6.2f}   max {rel[:, :, i].max(axis=1).mean():6.2f}")
W = int((os.environ.get("NVR_ATTN_TUNE") or "2,8").split(",")[1])
wv = rel[:, :, 8:8 + W]
print(f"per-wave loop done: first wave {wv.min(axis=2).mean():6.2f}  last wave {wv.max(axis=2).mean():6.2f}  (last - first) mean {(wv.max(axis=2) - wv.min(axis=2)).mean():5.2f} us; latest anywhere {wv.max(axis=(1,2)).mean():6.2f}")
#print("launch-to-launch (entry of first wg, consecutive layers):", np.diff(t0[:, 0, 0]).mean() * 10 / 1e3, "us")
