"""Upper bound of "early-launched consumer GEMMs": the decode chain of bench.time_decode_chain (six launches per layer, no attention) as one
captured graph, (a) in stream order and (b) with the two GEMMs that consume a norm's output (gate_up + SiLU behind the post-attention norm, the
next layer's qkv behind the final norm) forked onto a second stream so that they run CONCURRENTLY with that norm — without the data dependency
(their results are garbage: this only asks what the launch floor would give back if a device-side flag replaced the graph edge).
    python3 scratch/early_launch_probe.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nvr_import
nvr = nvr_import.load(); l = nvr.lib(); nvr.check(l.nvr_device_set(0))
mc = nvr.ModelConfig("qwen3-0.6b"); c = mc.c
T, Hd, H, KVH, D, I, L = 32, c.hidden_size, c.num_attention_heads, c.num_key_value_heads, mc.head_dim(), c.intermediate_size, c.num_hidden_layers
QKV = (H + 2 * KVH) * D
keep = []
def buf(n): b = nvr.DeviceBuffer(n); keep.append(b); return b
def arr(a): b = nvr.DeviceBuffer.from_numpy(np.ascontiguousarray(a)); keep.append(b); return b
def weights(rows, cols, mode):
    out = []
    for i in range(L):
        w, t = buf(rows * cols * 2), buf(rows * cols * 2)
        nvr.check(l.nvr_fill_weight(w.ptr, rows, cols, cols, cols, 0, 0, 5 + i, 1e-6, None))
        nvr.check(l.nvr_retile_weight(w.ptr, t.ptr, rows, cols, mode, H, KVH, D, None))
        out.append((w.ptr, t.ptr))
    return out
Wqkv, Wo, Wgu, Wd = weights(QKV, Hd, 1), weights(Hd, H * D, 0), weights(2 * I, Hd, 0), weights(Hd, I, 0)
rng = np.random.default_rng(0)
h = arr(rng.standard_normal((T, Hd)).astype(np.float16)); n = buf(T * Hd * 2); g = arr(np.ones(Hd, np.float16))
qkv, attn, act = buf(T * QKV * 2), arr(rng.standard_normal((T, H * D)).astype(np.float16) * 0.1), buf(T * I * 2)
slabs = buf(4 * T * Hd * 4)
pos = arr(np.arange(T, dtype=np.int64) + 1000); slots = arr(np.arange(T, dtype=np.int32))
cos = arr(np.ones((2048, D // 2), np.float32)); sin = arr(np.zeros((2048, D // 2), np.float32))
kc, vc = buf(64 * KVH * D * 2), buf(64 * KVH * D * 2)
So, Sd = l.nvr_decode_splitk_slices(T, H * D, Hd), l.nvr_decode_splitk_slices(T, I, Hd)
s1, s2 = C.c_void_p(), C.c_void_p(); nvr.check(l.nvr_stream_create(C.byref(s1))); nvr.check(l.nvr_stream_create(C.byref(s2)))
evs = []
def ev():
    e = C.c_void_p(); nvr.check(l.nvr_event_create(C.byref(e))); evs.append(e); return e
def qkv_k(i, st): nvr.check(l.nvr_linear_qkv_rope_store_tiled(n.ptr, Hd, Wqkv[i][0], Wqkv[i][1], T, Hd, H, KVH, D, pos.ptr, slots.ptr, cos.ptr, sin.ptr, qkv.ptr, kc.ptr, vc.ptr, st))
def o_k(i, st): nvr.check(l.nvr_linear_splitk_tiled(attn.ptr, H * D, Wo[i][0], Wo[i][1], T, H * D, Hd, So, slabs.ptr, st))
def norm_k(S, st): nvr.check(l.nvr_add_rmsnorm_slabs(h.ptr, slabs.ptr, S, g.ptr, 1e-6, T, Hd, n.ptr, st))
def gu_k(i, st): nvr.check(l.nvr_linear_silu_mul_tiled(n.ptr, Hd, Wgu[i][0], Wgu[i][1], T, Hd, I, act.ptr, st))
def dn_k(i, st): nvr.check(l.nvr_linear_splitk_tiled(act.ptr, I, Wd[i][0], Wd[i][1], T, I, Hd, Sd, slabs.ptr, st))
def fork(a, b):                      # b starts where a is now
    e = ev(); nvr.check(l.nvr_event_record(e, a)); nvr.check(l.nvr_stream_wait_event(b, e))
def capture(forked):
    ge = C.c_void_p()
    nvr.check(l.nvr_graph_capture_begin(s1))
    qkv_k(0, s1)
    for i in range(L):
        o_k(i, s1)
        if forked:
            fork(s1, s2); gu_k(i, s2); norm_k(So, s1); fork(s2, s1)
        else:
            norm_k(So, s1); gu_k(i, s1)
        dn_k(i, s1)
        if forked and i + 1 < L:
            fork(s1, s2); qkv_k(i + 1, s2); norm_k(Sd, s1); fork(s2, s1)
        else:
            norm_k(Sd, s1)
            if i + 1 < L: qkv_k(i + 1, s1)
    nvr.check(l.nvr_graph_capture_end(s1, C.byref(ge)))
    return ge
def time_graph(ge, reps=20):
    e0, e1 = ev(), ev()
    for _ in range(3): nvr.check(l.nvr_graph_launch(ge, s1))
    nvr.check(l.nvr_stream_synchronize(s1)); nvr.check(l.nvr_event_record(e0, s1))
    for _ in range(reps): nvr.check(l.nvr_graph_launch(ge, s1))
    nvr.check(l.nvr_event_record(e1, s1)); ms = C.c_float(); nvr.check(l.nvr_event_elapsed_ms(e0, e1, C.byref(ms)))
    return ms.value * 1e3 / reps / L
nvr.synchronize()
ga, gb = capture(False), capture(True)
for r in range(3):
    print(f"chain in stream order: {time_graph(ga):.2f} us per layer    consumers of a norm forked beside it (no data dependency): {time_graph(gb):.2f} us per layer", flush=True)
os._exit(0)
