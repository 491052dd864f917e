"""Whole synthetic decode layer stack of Qwen3-0.6B (GEMM / norm chain + paged attention over a cold KV pool) as ONE captured
hipGraph, (a) the batch of B sequences on one stream — the engine's step — and (b) split into two micro-batches of B/2 on two
forked streams (the latency-bound chain of one half running next to the bandwidth-bound attention of the other; every weight
byte is then read twice per step).  Microseconds per layer.   python tools/overlap_bench.py [B] [ctx]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nvr_import

nvr = nvr_import.load(); l = nvr.lib(); nvr.check(l.nvr_device_set(0))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
CTX = int(sys.argv[2]) if len(sys.argv) > 2 else 1040
Hd, H, KVH, D, I, L, BLOCK = 1024, 16, 8, 128, 3072, 28, 256
QKV = (H + 2 * KVH) * D
rng = np.random.default_rng(0)
keep = []


def buf(nbytes):
    b = nvr.DeviceBuffer(nbytes); keep.append(b); return b


def arr(a):
    b = nvr.DeviceBuffer.from_numpy(np.ascontiguousarray(a)); keep.append(b); return b


def weights(rows, cols, mode):
    out = []
    for i in range(L):
        w, t = buf(rows * cols * 2), buf(rows * cols * 2)
        nvr.check(l.nvr_fill_weight(w.ptr, rows, cols, cols, cols, 0, 0, 5 + i, 1e-6, None))
        nvr.check(l.nvr_retile_weight(w.ptr, t.ptr, rows, cols, mode, H, KVH, D, None))
        out.append((w, t))
    return out


Wqkv, Wo, Wgu, Wd = weights(QKV, Hd, 1), weights(Hd, H * D, 0), weights(2 * I, Hd, 0), weights(Hd, I, 0)
g = arr(np.ones(Hd, np.float16))
blocks_per_seq = (CTX + BLOCK) // BLOCK + 1
NB = B * blocks_per_seq
caches = [(buf(NB * BLOCK * KVH * D * 2), buf(NB * BLOCK * KVH * D * 2)) for _ in range(L)]
for kc, vc in caches:
    kc.zero(); vc.zero()
cos = arr(np.ones((4096, D // 2), np.float32)); sin = arr(np.zeros((4096, D // 2), np.float32))
scale = float(1.0 / np.sqrt(np.float32(D)))
bucket = (CTX + 255) // 256 * 256


class Half:
    """Buffers and metadata of one (micro-)batch of n sequences starting at sequence s0."""

    def __init__(self, s0, n):
        self.n = n
        self.h = arr(rng.standard_normal((n, Hd)).astype(np.float16)); self.nrm = buf(n * Hd * 2)
        self.qkv, self.attn, self.act = buf(n * QKV * 2), buf(n * H * D * 2), buf(n * I * 2)
        self.slabs = buf(4 * n * Hd * 4)
        ctx = np.full(n, CTX, np.int32)
        bt = -np.ones((n, blocks_per_seq), np.int32)
        for i in range(n):
            bt[i] = np.arange(blocks_per_seq) + (s0 + i) * blocks_per_seq
        self.pos = arr((ctx - 1).astype(np.int64))
        self.slots = arr(np.asarray([bt[i, (CTX - 1) // BLOCK] * BLOCK + (CTX - 1) % BLOCK for i in range(n)], np.int32))
        self.ctx, self.bt = arr(ctx), arr(bt)
        self.ws = buf(l.nvr_paged_attn_workspace_bytes(n, H, D, bucket))
        self.meta = nvr.AttnMetaC()
        self.meta.context_lens, self.meta.block_tables, self.meta.max_blocks = self.ctx.ptr, self.bt.ptr, blocks_per_seq
        self.meta.batch, self.meta.max_context_len = n, bucket
        self.So, self.Sd = l.nvr_decode_splitk_slices(n, H * D, Hd), l.nvr_decode_splitk_slices(n, I, Hd)

    def layer(self, i, st, attention=True, chain=True):
        T = self.n
        kc, vc = caches[i]
        if chain:
            nvr.check(l.nvr_linear_qkv_rope_store_tiled(self.nrm.ptr, Hd, Wqkv[i][0].ptr, Wqkv[i][1].ptr, T, Hd, H, KVH, D, self.pos.ptr, self.slots.ptr,
                                                        cos.ptr, sin.ptr, self.qkv.ptr, kc.ptr, vc.ptr, st))
        if attention:
            nvr.check(l.nvr_paged_attn_decode(self.qkv.ptr, QKV, kc.ptr, vc.ptr, C.byref(self.meta), H, KVH, D, BLOCK, scale, self.attn.ptr,
                                              self.ws.ptr, st))
        if chain:
            nvr.check(l.nvr_linear_splitk_tiled(self.attn.ptr, H * D, Wo[i][0].ptr, Wo[i][1].ptr, T, H * D, Hd, self.So, self.slabs.ptr, st))
            nvr.check(l.nvr_add_rmsnorm_slabs(self.h.ptr, self.slabs.ptr, self.So, g.ptr, 1e-6, T, Hd, self.nrm.ptr, st))
            nvr.check(l.nvr_linear_silu_mul_tiled(self.nrm.ptr, Hd, Wgu[i][0].ptr, Wgu[i][1].ptr, T, Hd, I, self.act.ptr, st))
            nvr.check(l.nvr_linear_splitk_tiled(self.act.ptr, I, Wd[i][0].ptr, Wd[i][1].ptr, T, I, Hd, self.Sd, self.slabs.ptr, st))
            nvr.check(l.nvr_add_rmsnorm_slabs(self.h.ptr, self.slabs.ptr, self.Sd, g.ptr, 1e-6, T, Hd, self.nrm.ptr, st))


st = C.c_void_p(); nvr.check(l.nvr_stream_create(C.byref(st)))
st2 = C.c_void_p(); nvr.check(l.nvr_stream_create(C.byref(st2)))
e0, e1, ef, ej = (C.c_void_p() for _ in range(4))
for e in (e0, e1, ef, ej):
    nvr.check(l.nvr_event_create(C.byref(e)))


def measure(capture, reps=20):
    ge = C.c_void_p()
    nvr.check(l.nvr_graph_capture_begin(st))
    capture()
    nvr.check(l.nvr_graph_capture_end(st, C.byref(ge)))
    for _ in range(3):
        nvr.check(l.nvr_graph_launch(ge, st))
    nvr.check(l.nvr_stream_synchronize(st))
    best = 1e9
    for _ in range(3):
        l.nvr_event_record(e0, st)
        for _ in range(reps):
            nvr.check(l.nvr_graph_launch(ge, st))
        l.nvr_event_record(e1, st)
        ms = C.c_float(); nvr.check(l.nvr_event_elapsed_ms(e0, e1, C.byref(ms)))
        best = min(best, ms.value * 1e3 / reps / L)
    nvr.check(l.nvr_graph_destroy(ge))
    return best


full, ha, hb = Half(0, B), Half(0, B // 2), Half(B // 2, B - B // 2)


def one_stream(**kw):
    for i in range(L):
        full.layer(i, st, **kw)


def halves_serial():
    for i in range(L):
        ha.layer(i, st); hb.layer(i, st)


def two_streams():
    nvr.check(l.nvr_event_record(ef, st)); nvr.check(l.nvr_stream_wait_event(st2, ef))       # fork
    for i in range(L):
        ha.layer(i, st)
        hb.layer(i, st2)
    nvr.check(l.nvr_event_record(ej, st2)); nvr.check(l.nvr_stream_wait_event(st, ej))       # join


print(f"B={B} ctx={CTX}  us per layer (28 layers per graph, own weights and KV pool per layer, best of 3 x 20 replays)")
print(f"{measure(one_stream):8.2f}  one stream, batch {B} (the engine's step)", flush=True)
print(f"{measure(lambda: one_stream(attention=False)):8.2f}    its chain alone", flush=True)
print(f"{measure(lambda: one_stream(chain=False)):8.2f}    its attention alone", flush=True)
print(f"{measure(halves_serial):8.2f}  two micro-batches of {B // 2}, one stream", flush=True)
print(f"{measure(two_streams):8.2f}  two micro-batches of {B // 2}, two forked streams", flush=True)
