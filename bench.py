#!/usr/bin/env python3
"""bench.py — decode throughput of the paged-attention hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): Qwen3-0.6B shape, fp16, synthetic random-init weights, 32
sequences with 1024-token synthetic prompts, greedy decode, block size 256, hipGraph decode steps.
A "step" is one decode pass of the engine over the whole batch (schedule -> execute_model ->
sample_tokens -> postprocess through the C ABI), i.e. 32 generated tokens.  The prefill of the
32x1024 prompt tokens happens before the timed region; K steps are timed between barriers and
device synchronisation; the maximum over ranks is reported.  N>1: `value` is the north star's TENSOR-PARALLEL
configuration — one engine over the N GPUs (heads / MLP columns / vocabulary sharded, the same 32 sequences: "scaling":
"strong"), its exchanges done by the one-shot peer-to-peer kernels over xGMI (kernels/comm_p2p.hip; RCCL for messages
that do not fit an arena slot, and as the fallback when the peer mapping cannot be built).  Decode sequences are also
independent units, so the no-exchange deployment (N replicas, one engine with its own 32 sequences per GPU, "weak") is
measured first on the same ranks and attached as "replicas".  The tensor-parallel phase runs in a child process per
rank: the multi-GPU path cannot be exercised on the 1-GPU development boxes, and a crash or hang in it must not cost the
measurement already taken — if it fails twice (with and without the peer-to-peer kernels), the replicas number is
reported as `value`, labelled as such, with the error attached.

The JSON line also carries
  roofline     — the dominant kernel (paged decode attention): algorithmic K/V bytes per launch
                 divided by its average launch duration, measured here with HIP events on the
                 kernel's own stream over back-to-back launches on the live KV pool (all layers
                 cycled, so every launch streams from HBM like in the real step);
  cpu_baseline — the CPU oracle (a port: the reference's Rust path cannot be built)
                 timed on this box's host cores on BASELINE.json configs[0] (bs=1, prompt 128, f32).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
import zlib
import array

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import nvr_import  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)

BATCH, PROMPT_LEN, BLOCK = 32, 1024, 256
TP_ASYNC = True           # launch-ahead also runs on tensor-parallel ranks (device-side cross-rank arg-max merge, r03)


def model_bytes_per_step(mc, ctx_mean: float) -> dict:
    """Algorithmic HBM bytes of one decode step, fp16."""
    Hd, L, H, KVH, D, I, V = (mc.hidden_size, mc.num_hidden_layers, mc.num_attention_heads, mc.num_key_value_heads,
                              mc.head_dim or mc.hidden_size // mc.num_attention_heads, mc.intermediate_size, mc.vocab_size)
    per_layer = 2 * ((H + 2 * KVH) * D * Hd + Hd * H * D + 2 * I * Hd + Hd * I + 2 * Hd)
    weights = per_layer * L + 2 * Hd + 2 * V * Hd
    kv_per_token = L * 2 * KVH * D * 2
    return dict(weights=weights, kv_read=BATCH * ctx_mean * kv_per_token, kv_write=BATCH * kv_per_token,
                kv_per_token_layer=2 * KVH * D * 2)


def time_attention_kernel(nvr, eng, mc, reps: int) -> dict:
    """Average duration of one decode paged-attention launch on the live KV pool (HIP events)."""
    l = nvr.lib()
    seqs = eng.last_batch()
    B = len(seqs)
    H, KVH = mc.c.num_attention_heads, mc.c.num_key_value_heads
    D, L = mc.head_dim(), mc.c.num_hidden_layers
    tp = eng.config.c.tensor_parallel_size
    H, KVH = H // tp, KVH // tp
    ctx = np.asarray([len(s) for s in seqs], np.int32)
    max_blocks = max(s.num_blocks() for s in seqs) + 1
    bt = -np.ones((B, max_blocks), np.int32)
    for i, s in enumerate(seqs):
        t = s.block_table
        bt[i, :len(t)] = t
    rng = np.random.default_rng(0)
    d_q = nvr.DeviceBuffer.from_numpy(rng.standard_normal((B, H * D)).astype(np.float16))
    d_ctx, d_bt = nvr.DeviceBuffer.from_numpy(ctx), nvr.DeviceBuffer.from_numpy(bt)
    d_out = nvr.DeviceBuffer(B * H * D * 2)
    bucket = (int(ctx.max()) + 255) // 256 * 256                 # same partitioning as the engine's graph
    ws = nvr.DeviceBuffer(l.nvr_paged_attn_workspace_bytes(B, H, D, bucket))
    meta = nvr.AttnMetaC()
    meta.context_lens, meta.block_tables, meta.max_blocks, meta.batch, meta.max_context_len = d_ctx.ptr, d_bt.ptr, max_blocks, B, bucket
    stream = C.c_void_p(); nvr.check(l.nvr_stream_create(C.byref(stream)))
    e0, e1 = C.c_void_p(), C.c_void_p()
    nvr.check(l.nvr_event_create(C.byref(e0))); nvr.check(l.nvr_event_create(C.byref(e1)))
    caches = [eng.model_runner.kv_cache(i) for i in range(L)]
    scale = float(1.0 / np.sqrt(np.float32(D)))

    def sweep():
        for kc, vc in caches:
            nvr.check(l.nvr_paged_attn_decode(d_q.ptr, H * D, kc, vc, C.byref(meta), H, KVH, D, BLOCK, scale, d_out.ptr, ws.ptr, stream))
    sweep()
    nvr.check(l.nvr_stream_synchronize(stream))
    # the step replays a captured graph, so the launches are timed the same way: reps sweeps over the 28 layer pools as ONE graph (events on its
    # stream around the replay) — eager back-to-back launches add ~1 us of host launch gap each, which is not in the step
    graph = C.c_void_p()
    captured = False
    if reps >= 1 and hasattr(l, "nvr_graph_capture_begin") and l.nvr_graph_capture_begin(stream) == 0:
        for _ in range(reps):
            sweep()
        captured = l.nvr_graph_capture_end(stream, C.byref(graph)) == 0 and bool(graph)
    if captured:
        nvr.check(l.nvr_graph_launch(graph, stream)); nvr.check(l.nvr_stream_synchronize(stream))       # warm replay
        nvr.check(l.nvr_event_record(e0, stream))
        nvr.check(l.nvr_graph_launch(graph, stream))
        nvr.check(l.nvr_event_record(e1, stream))
    else:
        nvr.check(l.nvr_event_record(e0, stream))
        for _ in range(reps):
            sweep()
        nvr.check(l.nvr_event_record(e1, stream))
    ms = C.c_float()
    nvr.check(l.nvr_event_elapsed_ms(e0, e1, C.byref(ms)))
    if captured: l.nvr_graph_destroy(graph)
    launches = reps * L
    us = ms.value * 1e3 / launches
    alg_bytes = float(ctx.sum()) * 2 * KVH * D * 2 + 2 * B * H * D * 2          # K+V rows read + q in + out
    l.nvr_event_destroy(e0); l.nvr_event_destroy(e1); l.nvr_stream_destroy(stream)
    return dict(us_per_launch=us, launches=launches, alg_bytes=alg_bytes, ctx_sum=int(ctx.sum()), as_graph=captured)


def time_decode_chain(nvr, mc, reps: int = 20) -> dict:
    """The decode step's GEMM / norm chain without attention (per layer: qkv+RoPE+store, o_proj split-k, add+RMSNorm,
    gate_up+SiLU, down split-k, add+RMSNorm — the six launches of the default chain) as one captured hipGraph of L layers with
    their own weights (HBM-cold every replay), replayed back to back on its own stream: microseconds per layer."""
    l = nvr.lib()
    c = mc.c
    T, Hd, H, KVH, D, I, L = BATCH, c.hidden_size, c.num_attention_heads, c.num_key_value_heads, mc.head_dim(), c.intermediate_size, c.num_hidden_layers
    QKV = (H + 2 * KVH) * D
    st = C.c_void_p(); nvr.check(l.nvr_stream_create(C.byref(st)))
    e0, e1 = C.c_void_p(), C.c_void_p(); nvr.check(l.nvr_event_create(C.byref(e0))); nvr.check(l.nvr_event_create(C.byref(e1)))
    keep = []

    def buf(nbytes):
        b = nvr.DeviceBuffer(nbytes); keep.append(b); return b

    def arr(a):
        b = nvr.DeviceBuffer.from_numpy(np.ascontiguousarray(a)); keep.append(b); return b

    def weights(rows, cols):
        ws = [buf(rows * cols * 2) for _ in range(L)]
        for i, w in enumerate(ws):
            nvr.check(l.nvr_fill_weight(w.ptr, rows, cols, cols, cols, 0, 0, 5 + i, 1e-6, None))
        return ws
    Wqkv, Wo, Wgu, Wd = weights(QKV, Hd), weights(Hd, H * D), weights(2 * I, Hd), weights(Hd, I)

    def tiled(ws, rows, cols, mode):                      # the runner's tiled copies: what the decode kernels read
        if os.environ.get("NVR_TILED_WEIGHTS", "1") == "0":
            return [C.c_void_p(None)] * len(ws)
        ts = [buf(rows * cols * 2) for _ in ws]
        for w, t in zip(ws, ts):
            nvr.check(l.nvr_retile_weight(w.ptr, t.ptr, rows, cols, mode, H, KVH, D, None))
        return [t.ptr for t in ts]
    Tqkv, To, Tgu, Td = tiled(Wqkv, QKV, Hd, 1), tiled(Wo, Hd, H * D, 0), tiled(Wgu, 2 * I, Hd, 0), tiled(Wd, Hd, I, 0)
    rng = np.random.default_rng(0)
    h = arr(rng.standard_normal((T, Hd)).astype(np.float16)); n = buf(T * Hd * 2); g = arr(np.ones(Hd, np.float16))
    qkv, attn, act = buf(T * QKV * 2), arr(rng.standard_normal((T, H * D)).astype(np.float16) * 0.1), buf(T * I * 2)
    slabs = buf(4 * T * Hd * 4)
    pos = arr(np.arange(T, dtype=np.int64) + 1000); slots = arr(np.arange(T, dtype=np.int32))
    cos = arr(np.ones((2048, D // 2), np.float32)); sin = arr(np.zeros((2048, D // 2), np.float32))
    kc, vc = buf(64 * KVH * D * 2), buf(64 * KVH * D * 2)
    So, Sd = l.nvr_decode_splitk_slices(T, H * D, Hd), l.nvr_decode_splitk_slices(T, I, Hd)
    nvr.synchronize()
    ge = C.c_void_p()
    nvr.check(l.nvr_graph_capture_begin(st))
    for i in range(L):
        nvr.check(l.nvr_linear_qkv_rope_store_tiled(n.ptr, Hd, Wqkv[i].ptr, Tqkv[i], T, Hd, H, KVH, D, pos.ptr, slots.ptr, cos.ptr, sin.ptr,
                                                    qkv.ptr, kc.ptr, vc.ptr, st))
        nvr.check(l.nvr_linear_splitk_tiled(attn.ptr, H * D, Wo[i].ptr, To[i], T, H * D, Hd, So, slabs.ptr, st))
        nvr.check(l.nvr_add_rmsnorm_slabs(h.ptr, slabs.ptr, So, g.ptr, 1e-6, T, Hd, n.ptr, st))
        nvr.check(l.nvr_linear_silu_mul_tiled(n.ptr, Hd, Wgu[i].ptr, Tgu[i], T, Hd, I, act.ptr, st))
        nvr.check(l.nvr_linear_splitk_tiled(act.ptr, I, Wd[i].ptr, Td[i], T, I, Hd, Sd, slabs.ptr, st))
        nvr.check(l.nvr_add_rmsnorm_slabs(h.ptr, slabs.ptr, Sd, g.ptr, 1e-6, T, Hd, n.ptr, st))
    nvr.check(l.nvr_graph_capture_end(st, C.byref(ge)))
    for _ in range(3):
        nvr.check(l.nvr_graph_launch(ge, st))
    nvr.check(l.nvr_stream_synchronize(st))
    nvr.check(l.nvr_event_record(e0, st))
    for _ in range(reps):
        nvr.check(l.nvr_graph_launch(ge, st))
    nvr.check(l.nvr_event_record(e1, st))
    ms = C.c_float(); nvr.check(l.nvr_event_elapsed_ms(e0, e1, C.byref(ms)))
    us_layer = ms.value * 1e3 / reps / L
    nvr.check(l.nvr_graph_destroy(ge)); l.nvr_event_destroy(e0); l.nvr_event_destroy(e1); l.nvr_stream_destroy(st)
    alg = 2.0 * (QKV * Hd + Hd * H * D + 2 * I * Hd + Hd * I)                  # weight bytes of one layer, read once
    return dict(us_per_layer=us_layer, alg_bytes_per_layer=alg, layers=L, replays=reps)


def shared_prefix_decode(nvr, mc, nseq: int = 512, steps: int = 24) -> dict:
    """BASELINE.json configs[4] as a side measurement of the N = 1 line: nseq sequences = the same 512-token system prompt + 64 own
    tokens each (BlockManager::allocate shares the two full prefix blocks, block_manager.rs:181-197), greedy decode through the engine
    with hipGraph steps; decode batches whose block tables all start with the same blocks attend to them in one MFMA pass."""
    eng = nvr.LLMEngine(nvr.Config(max_num_seqs=nseq, max_num_batched_tokens=65536, max_model_len=1024, kvcache_block_size=BLOCK,
                                   num_kvcache_blocks=nseq * 2 + 16, async_decode=1), mc)
    shared = nvr.synthetic_tokens(512, 2, 0, mc.c.vocab_size).tolist()
    for i in range(nseq):
        eng.add_request(shared + nvr.synthetic_tokens(64, 1, i, mc.c.vocab_size).tolist(),
                        nvr.SamplingParams(temperature=0.0, max_tokens=steps + 12, ignore_eos=True))
    nvr.synchronize(); t0 = time.perf_counter(); npre = 0
    while True:
        rec = eng.step()
        if not rec["is_prefill"]:
            break
        npre += 1
    nvr.synchronize(); t_pre = time.perf_counter() - t0
    for _ in range(4):
        eng.step()
    h0 = eng.host_times()
    nvr.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        eng.step()
    nvr.synchronize(); dt = (time.perf_counter() - t0) / steps
    h1 = eng.host_times()
    st = eng.scheduler.block_manager.get_stats()
    out = dict(workload=f"{nseq} sequences x (512 shared + 64 own prompt tokens), greedy decode (BASELINE.json configs[4])",
               ms_per_step=round(dt * 1e3, 4), tokens_per_s=round(nseq / dt, 1), steps=steps,
               shared_prefix_tokens=eng.model_runner.last_shared_prefix_len(), prefill_steps=npre,
               prefill_seconds=round(t_pre, 4), prompt_tokens=nseq * 576,
               host_us_per_step={k: round((h1[k + "_us"] - h0[k + "_us"]) / max(1, h1["steps"] - h0["steps"]), 2) for k in ("schedule", "postprocess")})
    if st:
        out["kv_blocks_used"] = int(st.get("used_blocks", 0))
    del eng
    return out


MODELS = {"qwen3-0.6b": dict(batch=32, prompt_len=1024, label="Qwen3-0.6B", baseline_config=1, shape=dict(h=16, kvh=8, d=128, layers=28)),
          "qwen3-8b": dict(batch=32, prompt_len=2048, label="Qwen3-8B", baseline_config=3, shape=dict(h=32, kvh=8, d=128, layers=36))}


def prefill_flops(c, lens) -> float:
    """Algorithmic FLOPs of prefilling sequences of the given lengths: GEMMs per token, causal attention, one LM-head row
    per sequence."""
    Dh = c.head_dim or c.hidden_size // c.num_attention_heads
    gemm_tok = 2 * c.num_hidden_layers * ((c.num_attention_heads + 2 * c.num_key_value_heads) * Dh * c.hidden_size
                                          + c.hidden_size * c.num_attention_heads * Dh + 3 * c.intermediate_size * c.hidden_size)
    attn = sum(4 * c.num_attention_heads * Dh * (n * (n + 1) // 2) for n in lens) * c.num_hidden_layers
    return float(sum(lens)) * gemm_tok + attn + len(lens) * 2 * c.vocab_size * c.hidden_size


def side_decode(nvr, preset: str, steps: int = 16, warmup: int = 4, tp_size: int = 1, tp_rank: int = 0, device: int = 0,
                attach=None, barrier=None, reduce_max=None, batch: int = None, prompt_len: int = None, dtype: str = "float16",
                async_decode=None) -> dict:
    """One more BASELINE workload measured next to the headline (same engine path: prefill untimed, W warm-up steps, K timed decode
    steps): used for BASELINE.json configs[3] (Qwen3-8B, bs 32 x 2048; src/models/qwen3.rs:70-125 with the 8B numbers) on one GPU and,
    in a tensor-parallel child, over the N GPUs (attach = communicator set-up of the engine's runner)."""
    w = MODELS[preset]
    B, P = batch or w["batch"], prompt_len or w["prompt_len"]
    mc = nvr.ModelConfig(preset)
    total_new = warmup + steps + 1
    t0 = time.perf_counter()
    eng = nvr.LLMEngine(nvr.Config(max_num_seqs=B, max_num_batched_tokens=32768, max_model_len=P + total_new + 16, kvcache_block_size=BLOCK,
                                   num_kvcache_blocks=B * ((P + total_new + 16) // BLOCK + 2), tensor_parallel_size=tp_size,
                                   tensor_parallel_rank=tp_rank, device_ordinal=device, dtype=dtype, **({} if async_decode is None else {"async_decode": async_decode})), mc)
    if attach is not None:
        ok, desc = attach(eng)
        if not ok:
            return {"error": desc}
    nvr.synchronize(); t_init = time.perf_counter() - t0
    exchange_modes = None
    if tp_size > 1:
        # row g on real links: the same prefill (B x P tokens, one token sampled per sequence: the sequences finish and free their blocks)
        # under the three forms of the tensor-parallel prefill exchange — serial on one stream, token chunks of one GEMM on a second
        # stream (default), two micro-batches under each other's compute (nvr_runner_set_tp_prefill_overlap 0 / 1 / 2; same bits)
        exchange_modes = {}
        try:
          for mode, name in ((1, "warm_up"), (0, "serial"), (1, "chunks"), (2, "two_microbatches")):
            eng.model_runner.set_tp_prefill_overlap(mode)
            for i in range(B):
                eng.add_request(nvr.synthetic_tokens(P, 1, 1000 * (mode + 1) + i, mc.c.vocab_size).tolist(),
                                nvr.SamplingParams(temperature=0.0, max_tokens=1, ignore_eos=True))
            if barrier: barrier()
            nvr.synchronize(); t1 = time.perf_counter(); forms = []
            while not eng.is_finished():
                rec = eng.step(); forms.append(eng.model_runner.last_overlap_chunks() if rec["is_prefill"] else -1)
            nvr.synchronize(); dt = time.perf_counter() - t1
            if reduce_max: dt = reduce_max(dt)
            eng.take_finished()
            if name != "warm_up":
                exchange_modes[name] = {"seconds": round(dt, 4), "prefill_steps": len(forms), "chunks_or_microbatches_per_step": forms}
        except Exception as ex:                                              # noqa: BLE001 (every rank fails alike: the schedule is deterministic)
            exchange_modes["error"] = str(ex)[:200]
            while not eng.is_finished():
                try: eng.step()
                except Exception: break                                      # noqa: BLE001
            eng.take_finished()
        eng.model_runner.set_tp_prefill_overlap(1)
    for i in range(B):
        eng.add_request(nvr.synthetic_tokens(P, 1, i, mc.c.vocab_size).tolist(),
                        nvr.SamplingParams(temperature=0.0, max_tokens=total_new + 8, ignore_eos=True))
    nvr.synchronize(); t0 = time.perf_counter(); npre = 0
    while True:
        rec = eng.step()
        if not rec["is_prefill"]:
            break
        npre += 1
    nvr.synchronize(); t_pre = time.perf_counter() - t0       # (includes the first decode step and its graph capture)
    crc = zlib.crc32(array.array("q", rec["tokens"]).tobytes(), 0)          # crc32 over the tokens of every decode step from the first
    for _ in range(warmup - 1):
        crc = zlib.crc32(array.array("q", eng.step()["tokens"]).tobytes(), crc)
    if barrier: barrier()
    nvr.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        crc = zlib.crc32(array.array("q", eng.step()["tokens"]).tobytes(), crc)
    nvr.synchronize(); el = time.perf_counter() - t0
    if barrier: barrier()
    if reduce_max: el = reduce_max(el)
    c = mc.c
    ctx_mean = P + 1 + warmup + (steps - 1) / 2.0
    Dh = c.head_dim or c.hidden_size // c.num_attention_heads
    per_layer = 2 * ((c.num_attention_heads + 2 * c.num_key_value_heads) * Dh * c.hidden_size + c.hidden_size * c.num_attention_heads * Dh
                     + 3 * c.intermediate_size * c.hidden_size + 2 * c.hidden_size)
    weights = per_layer * c.num_hidden_layers + 2 * c.hidden_size + 2 * c.vocab_size * c.hidden_size
    kv_tok = c.num_hidden_layers * 2 * c.num_key_value_heads * Dh * 2
    step_bytes = weights + B * ctx_mean * kv_tok + B * kv_tok
    ms = el * 1e3 / steps
    which = f"(BASELINE.json configs[{w['baseline_config']}])" if (batch is None and prompt_len is None) else "(batch-size sweep of the headline workload)"
    if dtype != "float16":
        which = f"(the headline workload on the {dtype} build of the kernels)"
    out = dict(workload=f"{w['label']} {'fp16' if dtype == 'float16' else 'bf16'} random-init, bs={B} x {P}-token prompts, greedy paged-attention decode, hipGraph steps "
                        f"{which}, {'one GPU' if tp_size == 1 else f'tensor parallel over {tp_size} GPUs'}",
               parallelism=f"tp{tp_size}", ms_per_step=round(ms, 4), tokens_per_s=round(B * steps / el, 1), steps=steps, warmup=warmup,
               step_algorithmic_bytes=int(step_bytes), step_hbm_frac_per_gpu=round(step_bytes / tp_size / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
               prefill_steps=npre, prefill_plus_first_decode_seconds=round(t_pre, 3),
               prefill_tflop_per_s_lower_bound=round(prefill_flops(c, [P] * B) / t_pre / 1e12, 1), init_seconds=round(t_init, 1),
               async_decode="off (nvr_config.async_decode = 0)" if async_decode == 0 else "on (the nvr_config default)",
               decode_token_crc=f"{crc:08x}", decode_steps_in_crc=warmup + steps)
    if exchange_modes is not None:
        out["prefill_exchange_forms"] = exchange_modes
    del eng
    return out


def ragged_decode(nvr, mc, steps: int = 16, warmup: int = 4, lo: int = 256, hi: int = 8192) -> dict:
    """A decode batch as serving sees it: the headline's 32 sequences, but with contexts spread geometrically between `lo` and `hi` keys (every BASELINE config is
    uniform).  The runner passes the attention launch its balance hint (nvr_runner_last_decode_ragged) and the launch cuts ALL pairs' keys into equal shares
    (attn_share_kernel, r06) instead of sizing every pair's workgroups by the longest context.  Same engine path as the headline; the step's roofline fraction is
    over its own algorithmic bytes (weights + the K/V actually visible)."""
    B, V, c = BATCH, mc.c.vocab_size, mc.c
    lens = [int(lo * (hi / lo) ** (i / (B - 1))) for i in range(B)]
    total_new = warmup + steps + 4
    nvr.lib().nvr_seq_reset_id_counter()
    eng = nvr.LLMEngine(nvr.Config(max_num_seqs=B, max_num_batched_tokens=32768, max_model_len=hi + total_new + 16, kvcache_block_size=BLOCK,
                                   num_kvcache_blocks=sum((n + total_new + 16) // BLOCK + 2 for n in lens)), mc)
    for i, n in enumerate(lens):
        eng.add_request(nvr.synthetic_tokens(n, 1, i, V).tolist(), nvr.SamplingParams(temperature=0.0, max_tokens=total_new, ignore_eos=True))
    while True:
        rec = eng.step()
        if not rec["is_prefill"]:
            break
    for _ in range(warmup - 1):
        eng.step()
    nvr.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        eng.step()
    nvr.synchronize(); ms = (time.perf_counter() - t0) * 1e3 / steps
    Dh = c.head_dim or c.hidden_size // c.num_attention_heads
    per_layer = 2 * ((c.num_attention_heads + 2 * c.num_key_value_heads) * Dh * c.hidden_size + c.hidden_size * c.num_attention_heads * Dh
                     + 3 * c.intermediate_size * c.hidden_size + 2 * c.hidden_size)
    weights = per_layer * c.num_hidden_layers + 2 * c.hidden_size + 2 * c.vocab_size * c.hidden_size
    kv_tok = c.num_hidden_layers * 2 * c.num_key_value_heads * Dh * 2
    ctx_sum = sum(lens) + B * (1 + warmup + (steps - 1) / 2.0)
    step_bytes = weights + ctx_sum * kv_tok + B * kv_tok
    return {"workload": f"{MODELS['qwen3-0.6b']['label']} fp16, bs={B}, contexts spread geometrically over {lo}..{hi} keys (sum {sum(lens)}: the K/V bytes of a uniform "
                        f"batch of {sum(lens) // B}-key contexts), greedy decode, hipGraph steps",
            "ms_per_step": round(ms, 4), "tokens_per_s": round(B / ms * 1e3, 1), "steps": steps,
            "step_algorithmic_bytes": int(step_bytes), "step_hbm_frac": round(step_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "attention_launch": "work-balanced (attn_share_kernel: equal shares of all pairs' keys)" if eng.model_runner.last_decode_ragged()
                                else "one workgroup (set) per (sequence, kv head) pair",
            "steps_launched_ahead": eng.ahead_launched(),
            "note": "per-pair launches (NVR_ATTN_SHARE=0) take 4.09 ms for this step: profiles/r06_priced_levers.txt 14."}


def stochastic_decode(nvr, mc, steps: int = 16, warmup: int = 4, check: bool = True) -> dict:
    """The non-greedy branch of the path (Sampler::forward top-k / top-p / Gumbel-max, src/layers/sampler.rs:71-218; named in north_star) on the
    headline workload: BASELINE.json configs[1] with temperature 0.8, top_k 50, top_p 0.9 on every sequence.  A stochastic step has no launch-ahead
    (its tokens are needed on the host before the next step can be built) and no fused arg-max head: the LM head writes the 32 x 151 936 f32 logits
    (19.4 MB) and the sampler reads them back.  Reports ms per step next to the greedy step, the sampler launch timed alone (HIP events on its stream,
    20 back-to-back launches on one step's logits) with its algorithmic bytes, and — check — one step's sampled ids against the CPU oracle's kept
    sets (temperature scaling, top-k, top-p restated on the CPU from the same logits; the checker, outside every timed region)."""
    B, P, V = BATCH, PROMPT_LEN, mc.c.vocab_size
    sp = dict(temperature=0.8, top_k=50, top_p=0.9, ignore_eos=True)
    total_new = warmup + steps + 4
    nvr.lib().nvr_seq_reset_id_counter()
    eng = nvr.LLMEngine(nvr.Config(max_num_seqs=B, max_num_batched_tokens=32768, max_model_len=P + total_new + 16, kvcache_block_size=BLOCK,
                                   num_kvcache_blocks=B * ((P + total_new + 16) // BLOCK + 2), sample_seed=1234), mc)
    for i in range(B):
        eng.add_request(nvr.synthetic_tokens(P, 1, i, V).tolist(), nvr.SamplingParams(max_tokens=total_new, **sp))
    while True:
        rec = eng.step()
        if not rec["is_prefill"]:
            break
    for _ in range(warmup - 1):
        eng.step()
    nvr.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        rec = eng.step()
    nvr.synchronize(); ms = (time.perf_counter() - t0) * 1e3 / steps
    out = {"workload": f"{MODELS['qwen3-0.6b']['label']} fp16, bs={B} x {P}-token prompts, temperature 0.8 / top_k 50 / top_p 0.9 on every sequence (BASELINE.json configs[1], "
                       "the sampler's stochastic branch)", "ms_per_step": round(ms, 4), "tokens_per_s": round(B / ms * 1e3, 1), "steps": steps,
           "steps_launched_ahead": eng.ahead_launched(),
           "note": "no launch-ahead and no fused arg-max head on a stochastic step: f32 logits written (19.4 MB) and read back by sample_rows_kernel; "
                   "the host waits for the tokens before it schedules the next step"}
    # the last step's logits and tokens: the sampler launch timed alone on them, and the kept-set check
    logits = eng.model_runner.logits(B).copy()
    toks = list(rec["tokens"])
    seqs = eng.last_batch()
    keys = np.asarray([nvr.lib().nvr_sample_key(1234, s.seq_id, s.num_completion_tokens() - 1) for s in seqs], np.uint64)
    l = nvr.lib()
    d_lg = nvr.DeviceBuffer.from_numpy(logits)
    d_t = nvr.DeviceBuffer.from_numpy(np.full(B, 0.8, np.float32)); d_k = nvr.DeviceBuffer.from_numpy(np.full(B, 50, np.int64))
    d_p = nvr.DeviceBuffer.from_numpy(np.full(B, 0.9, np.float32)); d_keys = nvr.DeviceBuffer.from_numpy(keys)
    d_out = nvr.DeviceBuffer(B * 8); ws = nvr.DeviceBuffer(l.nvr_sample_workspace_bytes(B, V))
    st = C.c_void_p(); nvr.check(l.nvr_stream_create(C.byref(st)))
    e0, e1 = C.c_void_p(), C.c_void_p(); nvr.check(l.nvr_event_create(C.byref(e0))); nvr.check(l.nvr_event_create(C.byref(e1)))
    launch = lambda: nvr.check(l.nvr_sample(d_lg.ptr, B, V, d_t.ptr, d_k.ptr, d_p.ptr, d_keys.ptr, d_out.ptr, ws.ptr, st))   # noqa: E731
    for _ in range(3): launch()
    nvr.check(l.nvr_stream_synchronize(st))
    reps = 20
    nvr.check(l.nvr_event_record(e0, st))
    for _ in range(reps): launch()
    nvr.check(l.nvr_event_record(e1, st)); nvr.check(l.nvr_stream_synchronize(st))
    msv = C.c_float(); nvr.check(l.nvr_event_elapsed_ms(e0, e1, C.byref(msv)))
    us = msv.value * 1e3 / reps
    alg = B * V * 4
    out["sampler_launch"] = {"kernel": "sample_rows_kernel (temperature, radix-select top-k, top-p, Gumbel-max; one launch)", "us_per_launch": round(us, 2),
                             "algorithmic_bytes": alg, "achieved_gbs": round(alg / (us * 1e-6) / 1e9, 1), "hbm_frac": round(alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                             "note": "eager back-to-back launches incl. the filtered-row stores of the stateless entry point (the engine's launch skips them); "
                                     "19.4 MB of logits resident in the 256 MB Infinity Cache between launches"}
    again = d_out.to_numpy((B,), np.int64).tolist()
    out["sampler_launch"]["replayed_ids_equal_the_steps"] = bool(again == toks)
    l.nvr_event_destroy(e0); l.nvr_event_destroy(e1); l.nvr_stream_destroy(st)
    if check:
        import oracle                                                        # the checker (CPU restatement), never timed
        inside, exact = 0, 0
        for b in range(B):
            a = logits[b] / np.float32(0.8)
            kept = np.isfinite(oracle.top_p(oracle.top_k(a, 50), 0.9))
            inside += int(kept[toks[b]])
            exact += int(oracle.sample(logits[b], 0.8, 50, 0.9, int(keys[b])) == toks[b])
        out["oracle_check"] = {"rows": B, "sampled_id_inside_the_oracles_kept_set": inside, "sampled_id_equals_the_oracles": exact,
                               "note": "one step: ids against oracle.top_k / top_p kept sets and oracle.sample (same counter RNG key) on the step's own logits"}
    del eng
    return out


def float32_path(nvr, steps: int = 64) -> dict:
    """BASELINE.json configs[0] (Qwen3-0.6B, bs = 1, 128-token prompt, greedy decode: the reference's own runnable configuration, f32 on its
    CPU path) on the product's Config.dtype = "float32" path (kernels/f32_path.hip: reference precision, plain FMA kernels, eager) — the GPU
    twin of cpu_baseline's workload."""
    mc = nvr.ModelConfig("qwen3-0.6b")
    eng = nvr.LLMEngine(nvr.Config(max_num_seqs=1, max_num_batched_tokens=256, max_model_len=256, kvcache_block_size=BLOCK, num_kvcache_blocks=2,
                                   dtype="float32"), mc)
    # (a warm engine, as for the headline: the same-shape prefill once on other tokens, so that the measured one does not pay code-object loading)
    eng.add_request(nvr.synthetic_tokens(128, 3, 0, mc.c.vocab_size).tolist(), nvr.SamplingParams(temperature=0.0, max_tokens=1, ignore_eos=True))
    while not eng.is_finished(): eng.step()
    eng.take_finished()
    eng.add_request(nvr.synthetic_tokens(128, 1, 0, mc.c.vocab_size).tolist(), nvr.SamplingParams(temperature=0.0, max_tokens=steps + 5, ignore_eos=True))
    nvr.synchronize(); t0 = time.perf_counter()
    eng.step()
    nvr.synchronize(); t_pre = time.perf_counter() - t0
    for _ in range(3): eng.step()
    nvr.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): eng.step()
    nvr.synchronize(); el = time.perf_counter() - t0
    del eng
    return {"workload": "Qwen3-0.6B f32 random-init (unrounded weights), bs=1, 128-token prompt, greedy decode (BASELINE.json configs[0]) on the float32 path",
            "decode_tokens_per_s": round(steps / el, 1), "ms_per_step": round(el * 1e3 / steps, 3), "prefill_tokens_per_s": round(128 / t_pre, 1),
            "parity": "tests/test_baseline_parity.py: 64 greedy ids == the f32 CPU-path oracle's, max |dlogit| 7e-6",
            "note": "the reference-precision path: a parity vehicle (every greedy id equal to the f32 CPU-path oracle's on configs[0] / [1] / [2] / [4]), not a tuned "
                    "product path — decode-sized steps on FMA kernels (five launches per layer), prefill GEMMs on the f32 matrix cores since r06"}


def prefill_sweep(nvr, lens=(128, 256, 512, 1024, 2048, 4096), nseq: int = 256) -> dict:
    """BASELINE.json configs[2]: Qwen3-0.6B fp16, 256 sequences at L in {128 .. 4096} through the engine under the reference's 32 768-token
    prefill budget (config.rs:58; Scheduler::try_schedule_prefill, scheduler.rs:119-168, batches whole sequences), and again with the
    chunked-prefill extension (A-23: the budget is filled to the last token).  Wall time of the prefill steps only (a step ends with the
    D2H of its tokens, so the device is idle when it returns); model FLOPs per SURVEY §8d against 2.5 PFLOP/s dense fp16."""
    mc = nvr.ModelConfig("qwen3-0.6b")
    rows = []
    for L in lens:
        rec = {"len": L, "sequences": nseq, "tokens": nseq * L}
        for chunked in (0, 1):
            nblk = (L + 2 + BLOCK - 1) // BLOCK + 1
            nvr.lib().nvr_seq_reset_id_counter()
            eng = nvr.LLMEngine(nvr.Config(max_num_seqs=nseq, max_num_batched_tokens=32768, max_model_len=L + 16, kvcache_block_size=BLOCK,
                                           num_kvcache_blocks=nseq * nblk + 8, enable_chunked_prefill=chunked), mc)
            for i in range(nseq):
                eng.add_request(nvr.synthetic_tokens(L, 1, i, mc.c.vocab_size).tolist(), nvr.SamplingParams(temperature=0.0, max_tokens=2, ignore_eos=True))
            nvr.synchronize(); steps = 0; dt = 0.0
            while True:
                t0 = time.perf_counter(); r = eng.step(); t1 = time.perf_counter()
                if not r["is_prefill"]:
                    break
                steps += 1; dt += t1 - t0
            fl = prefill_flops(mc.c, [L] * nseq)
            rec["chunked" if chunked else "whole_sequences"] = dict(prefill_steps=steps, ms=round(dt * 1e3, 2), k_tokens_per_s=round(nseq * L / dt / 1e3, 1),
                                                                     tflop_per_s=round(fl / dt / 1e12, 1), mfma_frac_of_2500=round(fl / dt / 2.5e15, 4))
            del eng
        rows.append(rec)
    return {"workload": "Qwen3-0.6B fp16, 256 sequences x L prompt tokens, 32 768-token prefill budget (BASELINE.json configs[2])", "rows": rows}



def prefill_recycled(nvr, preset: str = "qwen3-0.6b") -> dict:
    """The headline's prefill under the LESS favourable K/V layout: the pool holds the batch exactly once, so the measured batch takes the
    blocks the warm-up batch returned (a recycled free list: not consecutive) and the flash kernel walks the block tables instead of
    reading contiguous cache rows.  Same engine step, wall clock."""
    w = MODELS[preset]
    B, P = w["batch"], w["prompt_len"]
    mc = nvr.ModelConfig(preset)
    nvr.lib().nvr_seq_reset_id_counter()
    eng = nvr.LLMEngine(nvr.Config(max_num_seqs=B, max_num_batched_tokens=min(B * P, 32768), max_model_len=P + 32, kvcache_block_size=BLOCK,
                                   num_kvcache_blocks=B * ((P + 32) // BLOCK + 1)), mc)
    out = {}
    for seed, tag in ((3, "warm-up"), (1, "measured")):
        for i in range(B):
            eng.add_request(nvr.synthetic_tokens(P, seed, i, mc.c.vocab_size).tolist(), nvr.SamplingParams(temperature=0.0, max_tokens=1, ignore_eos=True))
        nvr.synchronize(); t0 = time.perf_counter(); n = 0
        while not eng.is_finished():
            eng.step(); n += 1
        nvr.synchronize(); dt = time.perf_counter() - t0
        src = eng.model_runner.last_prefill_kv_source()
        eng.take_finished()
        if tag == "measured":
            fl = prefill_flops(mc.c, [P] * B)
            out = {"tokens": B * P, "seconds": round(dt, 4), "prefill_steps": n, "tflop_per_s": round(fl / dt / 1e12, 1),
                   "mfma_frac_of_2500": round(fl / dt / 2.5e15, 4),
                   "kv_source": {0: "the step's qkv buffer", 1: "the cache rows, contiguous (consecutive blocks)", 2: "the caches through the block tables"}.get(src, "n/a"),
                   "note": "pool = the batch exactly once: the measured batch sits in the blocks the warm-up batch returned (recycled free list)"}
    del eng
    return out


def _pmc_child(model: str, counters, steps: int, warm: int, timeout_s: float = 240.0):
    """One counter pass of this workload in a CHILD process: `rocprofv3 --kernel-trace --pmc <counters> -- python3 bench.py ...` with nothing but
    --kernel-trace beside --pmc (MI355X_MICROARCH.md: counters in their own run), eager synchronous decode steps (every launch its own
    dispatch), started BEFORE this process touches the GPU.  Returns the rows of the counter table (dicts) or a string saying why not."""
    import csv, glob, shutil, subprocess, tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return "rocprofv3 not found"
    if any(k.startswith("ROCPROF") or k.startswith("ROCP_") for k in os.environ):
        return "this process already runs under a profiler"
    out = tempfile.mkdtemp(prefix="nvr_pmc_", dir="/tmp")
    cmd = [exe, "--kernel-trace", "--pmc", *counters, "--output-format", "csv", "-d", out, "-o", "pmc", "--",
           sys.executable, os.path.abspath(__file__), "--steps", str(steps), "--warmup", str(warm), "--model", model, "--no-cpu-baseline", "--no-chain",
           "--no-shared-prefix", "--no-configs3", "--no-prefill-sweep", "--no-batch-sweep", "--no-default-engine", "--sync-decode", "--eager",
           "--no-live-pmc", "--attn-reps", "1"]
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout_s)
        files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode != 0 or not files:
            return f"counter pass failed (rc {r.returncode})"
        return list(csv.DictReader(open(files[0])))
    except subprocess.TimeoutExpired:
        return "counter pass timed out"
    except Exception as ex:                                                      # noqa: BLE001
        return f"counter pass: {str(ex)[:120]}"
    finally:
        shutil.rmtree(out, ignore_errors=True)


def live_pmc_attention(model: str):
    """HBM traffic of the dominant kernel counted IN THIS RUN (FETCH_SIZE pass).  gfx950: FETCH_SIZE counts KiB at 64 B per 128-byte
    request of a 16-B-per-lane streaming read -> bytes = FETCH_SIZE * 1024 * 2.  Returns a dict or a string saying why there is no live
    number (the committed pass under profiles/ is quoted then)."""
    steps, warm = 6, 2
    rows = _pmc_child(model, ["FETCH_SIZE"], steps, warm)
    return rows if isinstance(rows, str) else attention_traffic_from_rows(rows, model, steps, warm)


def attention_traffic_from_rows(rows, model: str, steps: int, warm: int):
    """The FETCH_SIZE rows of a counter pass (dicts of rocprofv3's counter_collection.csv) -> HBM bytes over algorithmic bytes of the
    decode attention dispatches (pure function: tests/test_bench_counters.py feeds it synthetic tables)."""
    import statistics
    vals = [float(r["Counter_Value"]) for r in rows if r.get("Counter_Name") == "FETCH_SIZE" and "attn_rows_kernel" in r.get("Kernel_Name", "")]
    m = MODELS[model]
    mc = m["shape"]
    # dispatches of the pass: n = warm + steps engine decode steps at contexts P+1 .. P+n (one launch per layer each), then the 2 or 3 x layers
    # launches of the live kernel timing at the final context P+n+1: mean FETCH_SIZE over ALL of them against their mean algorithmic bytes
    n, P, L = warm + steps, m["prompt_len"], mc["layers"]
    x = len(vals) // L - n                      # sweeps of the live kernel timing: a warm-up sweep + the timed one (eager), or + a warm graph replay too
    if len(vals) % L or x not in (2, 3):
        return f"counter pass saw {len(vals)} attention dispatches, expected {L * (n + 2)} or {L * (n + 3)}"
    kib = statistics.fmean(vals)
    ctx = (n * P + n * (n + 1) / 2.0 + x * (P + n + 1)) / (n + x)
    alg = m["batch"] * ctx * mc["kvh"] * mc["d"] * 2 * 2 + 2 * m["batch"] * mc["h"] * mc["d"] * 2
    return {"fetch_size_kib_mean": round(kib, 1), "dispatches": len(vals), "hbm_bytes_per_launch": int(kib * 1024 * 2),
            "algorithmic_bytes_per_launch": int(alg), "traffic_over_algorithmic": round(kib * 1024 * 2 / alg, 4), "mean_context": round(ctx, 2)}


def live_pmc_prefill(model: str):
    """Matrix-pipe busy share of the measured prefill step counted IN THIS RUN: SQ_VALU_MFMA_BUSY_CYCLES (16 cycles per 16x16x32 MFMA, summed
    over the SIMDs) against GRBM_GUI_ACTIVE (summed over the 8 XCDs: cycles = GRBM / 8) per dispatch: busy = BUSY / (1024 SIMDs x cycles).
    The step = the dispatches from the last prefill embedding launch to the first decode step.  Returns a dict or a string."""
    rows = _pmc_child(model, ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"], 1, 0)
    return rows if isinstance(rows, str) else prefill_busy_from_rows(rows)


def prefill_busy_from_rows(rows):
    """Counter rows of a pass over the bench workload -> matrix-pipe busy share of its LAST prefill step (pure function, see above)."""
    import collections
    disp = collections.OrderedDict()
    for r in sorted(rows, key=lambda r: int(r["Dispatch_Id"])):
        d = disp.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"]})
        d[r["Counter_Name"]] = float(r["Counter_Value"])
    ds = list(disp.values())
    emb = [i for i, d in enumerate(ds) if "embedding_kernel" in d["name"]]
    if not emb:
        return "no prefill step in the counter pass"
    step = []
    for d in ds[emb[-1]:]:
        if "embed_rmsnorm" in d["name"]:
            break
        step.append(d)
    agg = collections.OrderedDict()
    for d in step:
        nm = d["name"].split("(")[0]
        for tag in ("gemm256_kernelILi2", "gemm256_kernelILi3", "gemm256_kernelILi1", "flash_prefill_kernel", "rmsnorm_kernel"):
            if tag in nm:
                a = agg.setdefault(tag, [0.0, 0.0, 0]); a[0] += d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0); a[1] += d.get("GRBM_GUI_ACTIVE", 0.0) / 8; a[2] += 1
    busy = sum(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for d in step); cyc = sum(d.get("GRBM_GUI_ACTIVE", 0.0) / 8 for d in step)
    if cyc <= 0:
        return "no cycles counted"
    names = {"gemm256_kernelILi2": "gate_up+SiLU", "gemm256_kernelILi3": "qkv+RoPE+store", "gemm256_kernelILi1": "o / down + residual",
             "flash_prefill_kernel": "flash_prefill", "rmsnorm_kernel": "rmsnorm"}
    return {"step_weighted": round(busy / (1024 * cyc), 4), "dispatches": len(step),
            "per_kernel": {names[k]: round(a[0] / (1024 * a[1]), 4) for k, a in agg.items() if a[1] > 0}}


def _pmc_prefill_busy():
    """Counter-based MFMA utilisation of the prefill step (separate rocprofv3 --pmc pass, committed under profiles/)."""
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_mfma_prefill_latest.json")) as f:
            return json.load(f).get("prefill_step_weighted")
    except Exception:                                        # noqa: BLE001
        return None


def cpu_baseline(decode_steps: int = 12) -> dict:
    """Oracle engine on BASELINE.json configs[0]: Qwen3-0.6B f32, bs=1, prompt 128, greedy (a port)."""
    import oracle
    from oracle import engine_oracle as eo, model_oracle as mo
    # threads actually used: the CPUs this process may run on (affinity and cgroup quota), capped at 32 —
    # the GPU box reports 256 logical CPUs, but an OpenMP team wider than the usable cores only spins
    cores = oracle.usable_cores()
    oracle.lib().nvo_set_num_threads(cores)
    mcfg = mo.qwen3_0_6b()
    eo.reset_sequence_counter()
    eng = mo.OracleEngine(mcfg, eo.Config(kvcache_block_size=256, num_kvcache_blocks=2, max_num_seqs=1,
                                          max_num_batched_tokens=256, max_model_len=256), fp16=False, max_pos=256)
    eng.add_request(oracle.fill_tokens(128, 1, 0, mcfg.vocab_size).tolist(),
                    eo.SamplingParams(temperature=0.0, max_tokens=decode_steps + 1, ignore_eos=True))
    t0 = time.perf_counter(); eng.step(); t_prefill = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(decode_steps):
        eng.step()
    t_dec = time.perf_counter() - t0
    return dict(value=round(decode_steps / t_dec, 3), unit="tokens/s", cores=int(cores), kind="port",
                sample=f"oracle (CPU restatement; reference Rust path unbuildable) Qwen3-0.6B f32 bs=1: prefill 128 tokens "
                       f"({128 / t_prefill:.1f} tok/s), then {decode_steps} greedy decode steps",
                prefill_tokens_per_s=round(128 / t_prefill, 2))


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--attn-reps", type=int, default=8)
    ap.add_argument("--no-chain", action="store_true", help="skip the GEMM / norm chain timing (roofline_chain)")
    ap.add_argument("--no-shared-prefix", action="store_true", help="skip the configs[4] side measurement (shared_prefix)")
    ap.add_argument("--model", choices=sorted(MODELS), default="qwen3-0.6b",
                    help="headline workload: qwen3-0.6b = BASELINE.json configs[1] (bs 32 x 1024, the metric's config); qwen3-8b = configs[3] "
                         "(bs 32 x 2048; --gpus 1: the whole model on one GPU, --gpus 8: tensor parallel over xGMI)")
    ap.add_argument("--no-configs3", action="store_true", help="skip the Qwen3-8B side measurement (configs3 block of the default line)")
    ap.add_argument("--no-prefill-sweep", action="store_true", help="skip the configs[2] prefill sweep (prefill_sweep block)")
    ap.add_argument("--no-batch-sweep", action="store_true", help="skip the bs 64 / 128 decode side measurements (batch_sweep block)")
    ap.add_argument("--no-default-engine", action="store_true",
                    help="skip the default_engine block (a fresh default engine, async_decode = 0 decode steps, the prefill on a recycled free list)")
    ap.add_argument("--sync-decode", action="store_true",
                    help="nvr_config.async_decode = 0: wait for every step's tokens on the host before the next step is scheduled "
                         "(default: the next greedy decode step is launched ahead; same batches, tokens and statistics)")
    ap.add_argument("--eager", action="store_true", help="enforce_eager: launch decode kernels one by one instead of replaying a hipGraph")
    ap.add_argument("--parallel", choices=["both", "tp", "replicas"], default=os.environ.get("NVR_BENCH_PARALLEL", "both"),
                    help="--gpus N > 1: 'both' (default) = N replicas measured first (attached as \"replicas\"), then the tensor-parallel "
                         "engine over the N GPUs in child processes: its tokens/s is the value ('strong' scaling); 'tp' = only the "
                         "tensor-parallel run, in this process; 'replicas' = only the N independent engines ('weak' scaling)")
    ap.add_argument("--materialize-logits", action="store_true",
                    help="write the f32 logits of every step to HBM (default: a greedy batch takes its tokens from the arg-max "
                         "partials of the LM-head epilogue and the logits are written only when someone asks for them)")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="do not run the child rocprofv3 --pmc FETCH_SIZE pass that counts roofline.traffic in this run (N = 1; "
                         "NVR_BENCH_LIVE_PMC=0 does the same): the committed pass under profiles/ is quoted instead")
    args = ap.parse_args()
    global BATCH, PROMPT_LEN
    BATCH, PROMPT_LEN = MODELS[args.model]["batch"], MODELS[args.model]["prompt_len"]
    # Control-plane dry runs of 4 / 8 rank PROCESSES on ONE GPU only (tools/tp_dryrun.sh, NRANKS >= 4): every cross-process rendezvous of a one-shot collective
    # then costs a scheduling quantum of the shared device (~30 ms), and the 3 584 slot-sized launches of a 32 x 1024 prefill without RCCL (which refuses two ranks
    # on one device) outlast the 30 s token wait.  Short prompts keep the dry run a test of the flow — rendezvous, self-tests, protocol chain, token crc, the
    # configs[3] child — and the line says so (config.workload, config.dry_run).  Ignored unless NVR_BENCH_SHARED_GPU is set: never in a measurement.
    dry_prompt = int(os.environ.get("NVR_BENCH_DRYRUN_PROMPT_LEN", "0")) if os.environ.get("NVR_BENCH_SHARED_GPU") else 0
    if dry_prompt > 0:
        PROMPT_LEN = dry_prompt
    if args.materialize_logits:
        os.environ["NVR_LAZY_LOGITS"] = "0"

    live_pmc = live_prefill = None
    if args.gpus == 1 and not args.no_live_pmc and os.environ.get("NVR_BENCH_LIVE_PMC", "1") != "0" and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        live_pmc = live_pmc_attention(args.model)                                # child processes, before this one initialises the GPU
        live_prefill = live_pmc_prefill(args.model)
    nvr = nvr_import.load()                # loads libnvr.so (and the ROCm HIP runtime)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("NVR_BENCH_SHARED_GPU"):          # control-plane dry run of the multi-rank path on a 1-GPU box
        local_rank = 0
    dist = None
    watchdog = None
    fallback_state: dict = {}
    arm = lambda phase, secs=None: None                                  # noqa: E731 (N == 1: no watchdog)
    if args.gpus > 1:
        # the multi-rank path cannot be exercised on the 1-GPU development boxes: never hang the driver — if a rank is still
        # stuck (a collective that never completes, a rendezvous that never forms) after 7 minutes (a tensor-parallel child: 110 s), every rank exits
        import threading
        wd_secs = 150.0 if os.environ.get("NVR_BENCH_CHILD") == "1" else 1500.0   # (parent: its phases include the children's limits, 420 + 300 + 300 s)

        def _bail():
            print(f"[bench] rank {rank}: multi-GPU run made no progress in its current phase ({fallback_state.get('phase', '?')}), giving up", file=sys.stderr, flush=True)
            line, rep = fallback_state.get("line"), fallback_state.get("replicas")
            if rank == 0 and line is not None:
                print(json.dumps(line), flush=True)                     # a complete measurement exists (e.g. the side block hung): report it
            elif rank == 0 and rep is not None:
                # the tensor-parallel phase hung after the replicas phase finished: report what was measured
                print(json.dumps({"metric": "decode tokens/s + %HBM-roofline, Qwen3-0.6B bs=32 seq=1024, 1/2/4/8 GPU", "value": rep["value"],
                                  "unit": "tokens/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
                                  "ms_per_step": rep["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                                  "dtype": "f16", "data": "synthetic",
                                  "config": {"workload": "Qwen3-0.6B fp16 random-init, bs=32 x 1024-token prompts per GPU, greedy paged-attention decode",
                                             "parallelism": rep["parallelism"] + " (the tensor-parallel phase did not complete in time)"},
                                  "roofline": None, "replicas": rep}), flush=True)
            os._exit(0 if (line is not None or rep is not None) else 4)   # a measured line went out: let the launcher finish normally

        def arm(phase: str, secs: float = None):
            """(re)start the no-progress timer at a phase boundary"""
            nonlocal watchdog
            if watchdog is not None:
                watchdog.cancel()
            fallback_state["phase"] = phase
            watchdog = threading.Timer(secs or wd_secs, _bail)
            watchdog.daemon = True
            watchdog.start()
        arm("rendezvous")
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} needs WORLD_SIZE={args.gpus} (launch with torch.distributed.run)")
        # control plane only (unique-id broadcast, handle exchange, agreements, barrier, max): a small TCP rendezvous on MASTER_ADDR /
        # MASTER_PORT (nano-vllm-rs_amd/ctrl.py) — a rank process holds ONE ROCm stack (libnvr.so's) and exits normally
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist = nvr_import.load_ctrl().SocketGroup(rank=rank, world=world, timeout=140.0 if os.environ.get("NVR_BENCH_CHILD") == "1" else 1400.0)
        if os.environ.get("NVR_BENCH_INJECT") == "child_hang" and os.environ.get("NVR_BENCH_CHILD") == "1" and os.environ.get("NVR_BENCH_ATTEMPT", "0") == "0":
            time.sleep(100000)                                               # (fallback dry run: the first child never gets anywhere; its watchdog ends it)

    total_new = args.warmup + args.steps + 1
    mc = nvr.ModelConfig(args.model)
    nvr.check(nvr.lib().nvr_device_set(local_rank))

    def make_engine(tp_size: int, tp_rank: int):
        cfg = nvr.Config(max_num_seqs=BATCH, max_num_batched_tokens=min(BATCH * PROMPT_LEN, 32768), max_model_len=PROMPT_LEN + total_new + 16,
                         # the pool holds the batch twice: the measured batch then takes never-used, consecutive blocks after the warm-up
                         # prefill's have gone to the back of the free list (its attention reads K/V as contiguous cache rows; on a
                         # recycled free list it walks the block tables: +1.1 ms per 32 x 1024 prefill, prefill.kv_source says which)
                         kvcache_block_size=BLOCK, num_kvcache_blocks=2 * BATCH * ((PROMPT_LEN + total_new + 16) // BLOCK + 2),
                         tensor_parallel_size=tp_size, tensor_parallel_rank=tp_rank,
                         device_ordinal=local_rank, enforce_eager=args.eager, **({"async_decode": 0} if args.sync_decode else {}))   # (default: launch-ahead on, nvr_config_default)
        e = nvr.LLMEngine(cfg, mc)
        if tp_size > 1 and os.environ.get("NVR_BENCH_TP_PREFILL_OVERLAP"):   # (0 = serial prefill exchange on the compute stream: see tools/tp_dryrun.sh, NRANKS >= 4)
            e.model_runner.set_tp_prefill_overlap(int(os.environ["NVR_BENCH_TP_PREFILL_OVERLAP"]))
        return e

    def barrier():
        nvr.synchronize()
        if dist is not None:
            dist.barrier()

    kv_source = [-1]

    token_crc = [0]                           # crc32 over the tokens sampled in the timed steps (tensor-parallel ranks must agree on it)
    stream_crc = [0]                          # ... and over every decode step from the first (warm-up steps included): comparable between engines
    host_us = [None]                          # host us per timed step inside Scheduler::schedule / ::postprocess

    def run_decode(eng):
        """prefill (untimed), W warm-up steps, K timed steps between barriers; max over ranks"""
        # a warm engine, as in serving: the same-shape prefill run once first on other tokens (seed 3; one new token per sequence, so
        # the batch finishes with its prefill and its blocks return to the pool), so that the measured prefill step does not pay
        # first-use costs — code-object loading of the kernels this shape routes to and first-touch of the workspaces: 2.5 ms of a
        # first 32 x 1024 prefill against 35.2-35.3 ms for every later one (scratch/prefill_wall.py)
        for i in range(BATCH):
            eng.add_request(nvr.synthetic_tokens(PROMPT_LEN, 3, i, mc.c.vocab_size).tolist(), nvr.SamplingParams(temperature=0.0, max_tokens=1, ignore_eos=True))
        while not eng.is_finished():
            eng.step()
        eng.take_finished()
        nvr.synchronize()
        for i in range(BATCH):               # synthetic prompts, SURVEY §8d: seed 1, one stream per sequence
            eng.add_request(nvr.synthetic_tokens(PROMPT_LEN, 1, i, mc.c.vocab_size).tolist(),
                            nvr.SamplingParams(temperature=0.0, max_tokens=total_new + 8, ignore_eos=True))
        t0 = time.perf_counter()
        pre_seqs = 0
        while pre_seqs < BATCH:              # prefill of the prompt tokens (untimed): one step per 32 768-token budget (config.rs:58)
            info = eng.step()
            assert info["is_prefill"], info
            pre_seqs += info["num_seqs"]
        nvr.synchronize()
        t_pre = time.perf_counter() - t0
        kv_source[0] = eng.model_runner.last_prefill_kv_source()
        scrc = 0
        for _ in range(args.warmup):
            info = eng.step()
            assert not info["is_prefill"] and info["num_seqs"] == BATCH
            scrc = zlib.crc32(array.array("q", info["tokens"]).tobytes(), scrc)
        barrier()
        h0 = eng.host_times()
        crc = 0
        t0 = time.perf_counter()
        for _ in range(args.steps):
            toks = array.array("q", eng.step()["tokens"]).tobytes()                     # (~1 us of host time per step, next to a 1.4 ms step)
            crc = zlib.crc32(toks, crc); scrc = zlib.crc32(toks, scrc)
        nvr.synchronize()
        el = time.perf_counter() - t0
        token_crc[0] = crc; stream_crc[0] = scrc
        h1 = eng.host_times()
        host_us[0] = {k: round((h1[k + "_us"] - h0[k + "_us"]) / max(1, h1["steps"] - h0["steps"]), 2) for k in ("schedule", "postprocess")}
        barrier()
        if dist is not None:
            el = dist.max(el)
        return el, t_pre

    parallelism, scaling, jobs = "tp1", "strong", 1
    replicas = None                           # N > 1: the no-exchange measurement (N independent engines), always taken first
    tensor_parallel = None
    collective = None
    backend_info: dict = {}                   # which collective backend the tensor-parallel engine ended up on (init_tensor_parallel)
    child = os.environ.get("NVR_BENCH_CHILD") == "1"

    def init_tensor_parallel(eng):
        """Communicators of a tensor-parallel engine: RCCL (large messages, fallback) and the one-shot peer-to-peer arenas
        (hipIpc handles gathered over the TCP control plane).  Every decision is agreed by all ranks (MIN over ranks).
        Returns (ok, description)."""
        all_ok = dist.all_ok
        uid = bytes(128)
        if rank == 0:
            try:
                uid = bytes(nvr.comm_unique_id())
            except Exception:                                                    # noqa: BLE001
                pass
        uid = dist.broadcast(uid, 0)
        rccl_ok, why = True, ""
        try:
            if os.environ.get("NVR_BENCH_RCCL", "1") == "0":
                raise RuntimeError("disabled by NVR_BENCH_RCCL=0")
            eng.model_runner.init_comm(uid)                                      # RCCL communicator + collective self-test
        except Exception as ex:                                                  # noqa: BLE001
            rccl_ok, why = False, str(ex)
        mine = rccl_ok
        rccl_ok = all_ok(rccl_ok)
        if mine and not rccl_ok:
            eng.model_runner.comm_drop_rccl()        # a peer has no communicator: nobody uses RCCL (or the ranks would pick different backends)
        # every rank goes through the SAME sequence of control-plane collectives below, whatever fails locally
        p2p_ok, why2 = os.environ.get("NVR_BENCH_P2P", "1") != "0", "disabled by NVR_BENCH_P2P=0"
        try:
            handle = eng.model_runner.p2p_export() if p2p_ok else b"\0" * 64
        except Exception as ex:                                                  # noqa: BLE001
            handle, p2p_ok, why2 = b"\0" * 64, False, str(ex)
        gathered = dist.all_gather((handle, local_rank))
        p2p_ok = all_ok(p2p_ok)
        if p2p_ok:
            try:
                eng.model_runner.p2p_attach([g[0] for g in gathered], [g[1] for g in gathered])
            except Exception as ex:                                              # noqa: BLE001
                p2p_ok, why2 = False, str(ex)
        p2p_ok = all_ok(p2p_ok)
        dist.barrier()                                                           # every rank has mapped every arena (or nobody uses them)
        # the chain fence-free -> fenced -> RCCL: the communicator self-test (all-reduce + all-gather through the arenas, then the
        # largest decode message back to back on both slot parities) decides, every rank alike; a protocol that fails it is never timed
        protocol, tried = None, []
        if p2p_ok:
            for proto in (("fenced",) if eng.model_runner.p2p_fenced() else ("fence_free", "fenced")):
                if proto == "fenced" and not eng.model_runner.p2p_fenced():
                    try:
                        eng.model_runner.p2p_reset()                             # (a timed-out self-test leaves the ranks' epochs apart)
                    except Exception:                                            # noqa: BLE001
                        pass
                    dist.barrier()
                    eng.model_runner.p2p_set_fenced(True)
                    dist.barrier()
                ok1, err1 = True, ""
                try:
                    eng.model_runner.comm_selftest()
                except Exception as ex:                                          # noqa: BLE001
                    ok1, err1 = False, str(ex)
                ok1 = all_ok(ok1)
                tried.append({"protocol": proto, "selftest": "passed" if ok1 else ("failed: " + (err1 or "on a peer"))[:200]})
                if ok1:
                    protocol = proto
                    break
                why2 = err1 or why2
            p2p_ok = protocol is not None
        p2p_ok = all_ok(p2p_ok)
        if not p2p_ok:
            eng.model_runner.p2p_disable()
        backend_info.update({"collective_backend": ("p2p_" + protocol) if p2p_ok else ("rccl" if rccl_ok else "none"),
                             "rccl_nranks": args.gpus if rccl_ok else 0, "p2p_protocols_tried": tried})
        if not rccl_ok and not p2p_ok:
            return False, f"no collective backend: RCCL: {why or 'a peer failed'}; peer-to-peer: {why2 or 'a peer failed'}"
        desc = (f"one-shot peer-to-peer kernels over xGMI, {protocol.replace('_', '-')} protocol (all-reduce + residual + RMSNorm in one launch, captured in the decode graph)"
                + ("; RCCL for messages larger than an arena slot" if rccl_ok else "")) if p2p_ok else "RCCL all-reduce (peer-to-peer arenas unavailable: " + (why2 or "a peer failed") + ")"
        return True, desc

    if args.gpus > 1 and (child or args.parallel == "tp"):
        # the tensor-parallel engine over all ranks (as the child of a "both" parent, or directly with --parallel tp);
        # a failure to build the communicators is reported as a JSON error line
        eng = make_engine(args.gpus, rank)
        ok, collective = init_tensor_parallel(eng)
        if not ok:
            if rank == 0:
                print(json.dumps({"error": "tensor-parallel communicators could not be built on this node: " + collective}), flush=True)
            dist.barrier(); dist.close()
            sys.stdout.flush(); sys.exit(0 if child else 3)
        parallelism, scaling, jobs = f"tp{args.gpus}", "strong", 1
        arm("tensor-parallel decode")
        elapsed, t_prefill = run_decode(eng)
        # every rank adds the same partial sums in the same order and merges the same (max, arg-max) records: the ranks' token streams are identical
        # or the exchange is broken on this node (stale peer data) — then this attempt reports an error and the parent tries the next backend
        crcs = dist.all_gather(token_crc[0])
        if os.environ.get("NVR_BENCH_INJECT") == "crc_mismatch" and os.environ.get("NVR_BENCH_ATTEMPT", "0") == "0":
            crcs = [c ^ (1 if i == 1 else 0) for i, c in enumerate(crcs)]     # (fallback dry run: the first attempt's ranks 'disagree')
        if len(set(crcs)) != 1:
            if rank == 0:
                print(json.dumps({"error": f"tensor-parallel ranks disagree on the sampled tokens ({collective}): token crc per rank {crcs}"}), flush=True)
            dist.barrier(); dist.close()
            sys.stdout.flush(); sys.exit(0 if child else 3)
        collective += f"; the {args.gpus} ranks' token streams of the timed steps are identical (crc {crcs[0]:08x})"
    elif args.gpus > 1:
        eng = make_engine(1, 0)
        r_el, r_pre = run_decode(eng)
        replicas = {"value": round(args.gpus * BATCH * args.steps / r_el, 2), "unit": "tokens/s", "ms_per_step": round(r_el * 1e3 / args.steps, 4),
                    "scaling": "weak", "parallelism": f"replicas{args.gpus}",
                    "note": f"{args.gpus} independent engines (one full model and its own 32 sequences per GPU), no exchange between ranks"}
        fallback_state["replicas"] = replicas
        arm("tensor-parallel children", 1150.0)                              # (their limits: 420 + 300 + 300 s)
        parallelism, scaling, jobs = f"replicas{args.gpus}", "weak", args.gpus
        elapsed, t_prefill = r_el, r_pre
        if args.parallel == "both":
            # the north star's tensor-parallel configuration, measured on the same ranks right after in a CHILD process per rank
            # (its own rendezvous on the next port); first with the peer-to-peer kernels, then — if that attempt died — RCCL only
            import subprocess
            tp_errors = []
            # attempt 0 walks the chain fence-free -> fenced -> RCCL by itself (self-tests, every rank alike); the later attempts are for a child
            # that DIED, HUNG or whose ranks disagreed on the tokens: start fenced, then RCCL alone.  Time limits: 8-rank RCCL init + Qwen3-0.6B,
            # then Qwen3-8B with its own communicator and the three prefill exchange forms; 420 + 300 + 300 s fit the driver's 1800 s with the
            # replicas phase in front
            attempts = (({}, 420), ({"NVR_P2P_FENCED": "1"}, 300), ({"NVR_BENCH_P2P": "0"}, 300))
            for attempt, (extra_env, limit) in enumerate(attempts):
                env = dict(os.environ)
                env["MASTER_PORT"] = str(int(env.get("MASTER_PORT", "29500")) + 1 + attempt)
                env["NVR_BENCH_CHILD"] = "1"; env["NVR_BENCH_ATTEMPT"] = str(attempt)
                env.update(extra_env)
                for k in [k for k in env if k.startswith("TORCHELASTIC_")]:   # (the children are plain processes, not the launcher agent's workers)
                    del env[k]
                cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup),
                       "--parallel", "tp", "--no-cpu-baseline", "--attn-reps", "1"] + (["--eager"] if args.eager else [])
                child_out, child_err, child_rc = "", "", None
                txt = lambda b: b.decode("utf-8", "replace") if isinstance(b, (bytes, bytearray)) else (b or "")   # noqa: E731
                try:
                    cp = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=limit)
                    child_out, child_err, child_rc = cp.stdout, cp.stderr, cp.returncode
                except subprocess.TimeoutExpired as te:
                    # what the child printed before the limit counts: it prints its tensor-parallel line as soon as that is measured
                    child_out = txt(te.stdout)
                    child_err = f"timed out after {limit} s: " + txt(te.stderr)[-300:]
                done = 0
                if rank == 0:
                    line = next((l for l in reversed(child_out.splitlines()) if l.startswith("{")), None)
                    try:
                        cj = json.loads(line) if line else None
                    except Exception:                                        # noqa: BLE001
                        cj = None
                    if cj and cj.get("config", {}).get("parallelism") == f"tp{args.gpus}":
                        tensor_parallel = cj
                        if tp_errors:
                            tensor_parallel["config"]["earlier_attempts"] = list(tp_errors)
                        done = 1
                    else:
                        tp_errors.append(f"attempt {attempt} ({extra_env or 'default chain'}): " + ((cj or {}).get("error") or f"child exit code {child_rc}: {child_err.strip()[-400:]}"))
                done = dist.broadcast(done, 0)
                barrier()
                if done:
                    break
            if rank == 0 and tensor_parallel is None:
                tensor_parallel = {"error": " | ".join(tp_errors)}
    else:
        eng = make_engine(1, 0)
        elapsed, t_prefill = run_decode(eng)

    # prefill of the 32 x 1024 prompt tokens (one engine step, wall clock incl. host input preparation): MFMA-bound,
    # 880.8 MFLOP/token of GEMM + 114 688*l flop/token of causal attention (+ LM head per sequence)
    prefill_flop = prefill_flops(mc.c, [PROMPT_LEN] * BATCH)
    ms_per_step = elapsed * 1e3 / args.steps
    tokens_per_s = jobs * BATCH * args.steps / elapsed
    ctx_mean = PROMPT_LEN + 1 + args.warmup + (args.steps - 1) / 2.0   # keys visible per sequence, averaged over timed steps
    mb = model_bytes_per_step(mc.c, ctx_mean)
    step_bytes = mb["weights"] + mb["kv_read"] + mb["kv_write"]
    step_gbs = step_bytes / (args.gpus / jobs) / (ms_per_step * 1e-3) / 1e9   # per-GPU share of the algorithmic bytes

    attn = time_attention_kernel(nvr, eng, mc, args.attn_reps)
    chain = None
    if args.gpus == 1 and rank == 0 and not args.no_chain and args.model == "qwen3-0.6b":
        try:
            chain = time_decode_chain(nvr, mc)
        except Exception as ex:                                              # noqa: BLE001
            print(f"[bench] decode chain timing failed: {ex}", file=sys.stderr, flush=True)
    shared_prefix = None
    if args.gpus == 1 and rank == 0 and not args.no_shared_prefix and args.model == "qwen3-0.6b":
        try:
            shared_prefix = shared_prefix_decode(nvr, mc)
        except Exception as ex:                                              # noqa: BLE001
            shared_prefix = {"error": str(ex)[:300]}
    configs3 = None
    if args.gpus == 1 and rank == 0 and not args.no_configs3 and args.model == "qwen3-0.6b":
        try:
            configs3 = side_decode(nvr, "qwen3-8b")
        except Exception as ex:                                              # noqa: BLE001
            configs3 = {"error": str(ex)[:300]}
    configs3_tp = args.gpus > 1 and parallelism.startswith("tp") and not args.no_configs3 and args.model == "qwen3-0.6b"
    batch_sweep = None
    if args.gpus == 1 and rank == 0 and not args.no_batch_sweep and args.model == "qwen3-0.6b":
        # the same engine path at larger batches: the step's share of the HBM roofline grows with the K/V bytes per launch
        batch_sweep = []
        for preset, bsz, plen in (("qwen3-0.6b", 64, 1024), ("qwen3-0.6b", 128, 1024), ("qwen3-0.6b", 256, 1024), ("qwen3-0.6b", 512, 1024),
                                  ("qwen3-0.6b", 1024, 1024), ("qwen3-8b", 64, 2048), ("qwen3-8b", 128, 2048), ("qwen3-8b", 256, 2048)):
            try:
                r = side_decode(nvr, preset, batch=bsz, prompt_len=plen, steps=8, warmup=3)
                row = {k: r[k] for k in ("workload", "ms_per_step", "tokens_per_s", "step_algorithmic_bytes", "step_hbm_frac_per_gpu")}
                row["model"], row["batch"] = preset, bsz
                batch_sweep.append(row)
            except Exception as ex:                                          # noqa: BLE001
                batch_sweep.append({"model": preset, "batch": bsz, "error": str(ex)[:200]})
    ragged = None
    if args.gpus == 1 and rank == 0 and not args.no_batch_sweep and args.model == "qwen3-0.6b":
        try:
            ragged = ragged_decode(nvr, mc)
        except Exception as ex:                                              # noqa: BLE001
            ragged = {"error": str(ex)[:300]}
    stochastic = None
    if args.gpus == 1 and rank == 0 and not args.no_batch_sweep and args.model == "qwen3-0.6b":
        try:
            stochastic = stochastic_decode(nvr, mc, check=not args.no_cpu_baseline)
            stochastic["greedy_ms_per_step"] = round(ms_per_step, 4)
        except Exception as ex:                                              # noqa: BLE001
            stochastic = {"error": str(ex)[:300]}
    bf16_block = None
    if args.gpus == 1 and rank == 0 and not args.no_batch_sweep and args.model == "qwen3-0.6b":
        # Config.dtype = "bfloat16" (config.rs:51,113-116): the same workload on the bf16 build of every kernel (same bytes, same MFMA rate)
        try:
            r = side_decode(nvr, "qwen3-0.6b", batch=BATCH, prompt_len=PROMPT_LEN, dtype="bfloat16")
            bf16_block = {k: r[k] for k in ("workload", "ms_per_step", "tokens_per_s", "step_hbm_frac_per_gpu", "prefill_plus_first_decode_seconds")}
        except Exception as ex:                                              # noqa: BLE001
            bf16_block = {"error": str(ex)[:200]}
    default_engine = None
    if args.gpus == 1 and rank == 0 and not args.no_default_engine and args.model == "qwen3-0.6b":
        # the headline engine IS the engine nvr_config_default builds (r05: async_decode defaults to 1); measured again here on a fresh engine of
        # its own, next to the opt-out (async_decode = 0: the host waits for every step's tokens before it schedules the next step) and the
        # prefill on a recycled free list (block tables, not contiguous rows)
        default_engine = {"note": "nvr_config_default() + the workload's sizes, nothing else set: the same configuration as the headline line"}
        try:
            # (two interleaved pairs, the faster of each kind: the side blocks of this line run minutes apart on a part whose clocks move with its
            #  temperature — the bf16 block, the same step, differs from the headline by 1-2 % either way)
            runs = {None: [], 0: []}
            for _ in range(2):
                for mode in (None, 0):
                    runs[mode].append(side_decode(nvr, "qwen3-0.6b", steps=max(16, min(args.steps, 64)), warmup=args.warmup, batch=BATCH, prompt_len=PROMPT_LEN,
                                                  async_decode=mode))
            r = min(runs[None], key=lambda x: x["ms_per_step"])
            default_engine["ms_per_step"] = r["ms_per_step"]; default_engine["tokens_per_s"] = r["tokens_per_s"]
            default_engine["ms_per_step_runs"] = [x["ms_per_step"] for x in runs[None]]
            r = min(runs[0], key=lambda x: x["ms_per_step"])
            default_engine["sync_decode_ms_per_step"] = r["ms_per_step"]
            default_engine["sync_decode_tokens_per_s"] = r["tokens_per_s"]
            default_engine["sync_decode_step_hbm_frac"] = r["step_hbm_frac_per_gpu"]
            default_engine["sync_decode_ms_per_step_runs"] = [x["ms_per_step"] for x in runs[0]]
            # the token stream of the timed (launch-ahead) engine against the synchronous engine's — the one the parity suite checks against the oracle
            # step by step (tests/test_baseline_parity.py: configs[1] vs the oracle, default engine == synchronous engine bit for bit)
            sync_crc = {x["decode_token_crc"] for x in runs[0]}; ahead_crc = {x["decode_token_crc"] for x in runs[None]}
            same_window = runs[0][0]["decode_steps_in_crc"] == args.warmup + args.steps
            default_engine["token_stream"] = {
                "timed_engine_crc_all_decode_steps": f"{stream_crc[0]:08x}", "timed_engine_crc_timed_steps": f"{token_crc[0]:08x}",
                "fresh_default_engine_crc": sorted(ahead_crc), "synchronous_engine_crc": sorted(sync_crc), "decode_steps_in_crc": runs[0][0]["decode_steps_in_crc"],
                "equal": bool(len(sync_crc) == 1 and ahead_crc == sync_crc and (not same_window or f"{stream_crc[0]:08x}" in sync_crc)),
                "note": "crc32 over the sampled token ids of every decode step from the first; 'equal' = launch-ahead engines (the timed one when its window is the "
                        "same) and synchronous engines produced the same stream"}
        except Exception as ex:                                              # noqa: BLE001
            default_engine["sync_decode_error"] = str(ex)[:200]
        try:
            default_engine["prefill_recycled_free_list"] = prefill_recycled(nvr)
        except Exception as ex:                                              # noqa: BLE001
            default_engine["prefill_recycled_free_list"] = {"error": str(ex)[:200]}
    f32_block = None
    if args.gpus == 1 and rank == 0 and not args.no_default_engine and args.model == "qwen3-0.6b":
        try:
            f32_block = float32_path(nvr)
        except Exception as ex:                                              # noqa: BLE001
            f32_block = {"error": str(ex)[:200]}
    sweep = None
    if args.gpus == 1 and rank == 0 and not args.no_prefill_sweep and args.model == "qwen3-0.6b":
        try:
            sweep = prefill_sweep(nvr)
        except Exception as ex:                                              # noqa: BLE001
            sweep = {"error": str(ex)[:300]}
    achieved = attn["alg_bytes"] / (attn["us_per_launch"] * 1e-6) / 1e9
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_attn_latest.json")
    traffic_live = isinstance(live_pmc, dict)
    if traffic_live:
        traffic = int(round(live_pmc["traffic_over_algorithmic"] * attn["alg_bytes"]))
    elif os.path.exists(pmc_path):
        try:
            # PMC pass (profiles/): HBM bytes / algorithmic bytes of this kernel, scaled to this run's launch
            traffic = int(round(json.load(open(pmc_path))["traffic_over_algorithmic"] * attn["alg_bytes"]))
        except Exception:
            traffic = None

    line = None
    if rank == 0:
        out = {
            "metric": "decode tokens/s + %HBM-roofline, Qwen3-0.6B bs=32 seq=1024, 1/2/4/8 GPU",
            "value": round(tokens_per_s, 2), "unit": "tokens/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "host_us_per_step": (dict(host_us[0], note="host time inside Scheduler::schedule / ::postprocess (block manager included) per timed decode step, "
                                                       "rank 0 (SURVEY section 8d); with launch-ahead it runs while the GPU executes the previous step") if host_us[0] else None),
            "config": {"workload": f"{MODELS[args.model]['label']} fp16 random-init, bs={BATCH} x {PROMPT_LEN}-token prompts, greedy paged-attention decode, "
                                   f"block_size=256, hipGraph decode steps (BASELINE.json configs[{MODELS[args.model]['baseline_config']}])",
                       "batch": BATCH, "prompt_len": PROMPT_LEN, "mean_context": ctx_mean,
                       **({"dry_run": f"control-plane dry run: {world} rank processes on ONE GPU, prompts shortened to {dry_prompt} tokens (NVR_BENCH_DRYRUN_PROMPT_LEN); not a measurement"} if dry_prompt > 0 else {}),
                       "parallelism": parallelism, "hipgraph": not args.eager,
                       "async_decode": ("nvr_config.async_decode (default 1 since r05): "
                                        + ("on" if (not args.sync_decode and (args.gpus == 1 or parallelism.startswith("replicas") or TP_ASYNC)) else "off")),
                       "logits": "materialised every step" if args.materialize_logits else "greedy arg-max fused into the LM head; f32 logits on demand"},
            "prefill": {"tokens": BATCH * PROMPT_LEN, "seconds": round(t_prefill, 4), "tokens_per_s": round(BATCH * PROMPT_LEN / t_prefill, 1),
                        "tflop_per_s": round(prefill_flop / t_prefill / 1e12, 1), "mfma_frac_of_2500": round(prefill_flop / t_prefill / 2.5e15, 4),
                        "mfma_busy_frac_pmc": _pmc_prefill_busy(),
                        "kv_source": {0: "the step's qkv buffer", 1: "the cache rows, contiguous (consecutive blocks)", 2: "the caches through the block tables"}.get(kv_source[0], "n/a"),
                        "note": "one untimed engine prefill step on a warm engine (wall clock, includes host input preparation and upload; a same-shape warm-up prefill on other tokens "
                                "request ran before it); "
                                "mfma_busy_frac_pmc = matrix-pipe busy cycles / available cycles at the clock the chip held (counter pass: see "
                                "mfma_busy_frac_pmc_source)"},
            "step_hbm_frac": round(step_gbs / HBM_PEAK_GBS, 4),
            "step_algorithmic_bytes": int(step_bytes),
            "roofline": {"kernel": "attn_rows_kernel (paged decode attention, K9)", "bound": "hbm", "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "us_per_launch": round(attn["us_per_launch"], 2), "launches_timed": attn["launches"],
                         "timed_as": "one captured graph of the launches, replayed (like the step)" if attn.get("as_graph") else "eager back-to-back launches",
                         "algorithmic_bytes_per_launch": int(attn["alg_bytes"])},
        }
        if traffic_live:
            out["roofline"]["traffic_source"] = ("counted in this run: a child `rocprofv3 --kernel-trace --pmc FETCH_SIZE` pass of eager decode steps "
                                                 "(FETCH_SIZE KiB x 1024 x 2 on gfx950), HBM bytes / algorithmic bytes over its attention dispatches, "
                                                 "scaled to the algorithmic bytes of the timed launches")
            out["roofline"]["traffic_pass"] = live_pmc
        elif traffic is not None:
            out["roofline"]["traffic_source"] = ("profiles/pmc_attn_latest.json (separate rocprofv3 --pmc FETCH_SIZE pass of this kernel: HBM bytes / "
                                                 "algorithmic bytes), scaled by this run's algorithmic bytes — not counted in this run"
                                                 + (f" ({live_pmc})" if isinstance(live_pmc, str) else ""))
        if isinstance(live_prefill, dict):
            out["prefill"]["mfma_busy_frac_pmc"] = live_prefill["step_weighted"]
            out["prefill"]["mfma_busy_frac_pmc_per_kernel"] = live_prefill["per_kernel"]
            out["prefill"]["mfma_busy_frac_pmc_source"] = ("counted in this run: a child `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` pass of "
                                                           f"the same prefill step ({live_prefill['dispatches']} dispatches; profiled clocks run ~3 % below un-profiled ones)")
        else:
            out["prefill"]["mfma_busy_frac_pmc_source"] = ("profiles/pmc_mfma_prefill_latest.json (separate rocprofv3 --pmc pass, not counted in this run"
                                                           + (f": {live_prefill}" if isinstance(live_prefill, str) else "") + ")")
        if collective is not None:
            out["config"]["collectives"] = collective
            out["config"].update(backend_info)
            out["config"]["tp_attempt"] = int(os.environ.get("NVR_BENCH_ATTEMPT", "0"))
        if chain is not None:
            gbs = chain["alg_bytes_per_layer"] / (chain["us_per_layer"] * 1e-6) / 1e9
            out["roofline_chain"] = {"kernel": "decode GEMM / norm chain, 6 launches per layer (linear_skinny_kernel x4, add_rmsnorm_slabs_kernel x2), "
                                               "no attention", "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None, "us_per_layer": round(chain["us_per_layer"], 2),
                                     "algorithmic_bytes_per_layer": int(chain["alg_bytes_per_layer"]),
                                     "note": "one hipGraph of 28 layers with their own weights (HBM-cold), replayed back to back (HIP events on its stream)"}
        if shared_prefix is not None:
            out["shared_prefix"] = shared_prefix
        if configs3 is not None:
            out["configs3"] = configs3
        if batch_sweep is not None:
            out["batch_sweep"] = batch_sweep
            # the north star's 0.70 of the decode-step HBM roofline: the smallest measured batch that reaches it, per model (the headline's own
            # fraction is step_hbm_frac, the 8B headline configs3.step_hbm_frac_per_gpu)
            reach = {}
            pts = {"qwen3-0.6b": [(BATCH, out["step_hbm_frac"])], "qwen3-8b": ([(32, configs3["step_hbm_frac_per_gpu"])] if configs3 and "step_hbm_frac_per_gpu" in configs3 else [])}
            for row in batch_sweep:
                if "error" not in row: pts[row["model"]].append((row["batch"], row["step_hbm_frac_per_gpu"]))
            for mdl, pp in pts.items():
                ok = sorted(b for b, f in pp if f >= 0.70)
                reach[mdl] = {"smallest_batch_with_step_hbm_frac_ge_0.70": ok[0] if ok else None, "measured": {str(b): f for b, f in sorted(pp)}}
            out["batch_for_0.70_of_step_roofline"] = reach
        if ragged is not None:
            out["ragged_batch"] = ragged
        if stochastic is not None:
            out["stochastic"] = stochastic
        if bf16_block is not None:
            out["bf16"] = bf16_block
        if default_engine is not None:
            out["default_engine"] = default_engine
        if f32_block is not None:
            out["float32_path"] = f32_block
        if sweep is not None:
            out["prefill_sweep"] = sweep
        if args.gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        if replicas is not None and tensor_parallel is not None and "error" not in tensor_parallel:
            # N > 1: the tensor-parallel engine's line is the result; the replicas measured on the same ranks ride along
            tp_line = dict(tensor_parallel)
            tp_line["replicas"] = replicas
            tp_line["speedup_vs_one_replica"] = round(tp_line["value"] / (replicas["value"] / args.gpus), 3)
            line = tp_line
        else:
            if replicas is not None:
                out["replicas"] = replicas
            if tensor_parallel is not None:                                 # both tensor-parallel attempts failed: say so in the line
                out["tensor_parallel"] = tensor_parallel
                out["config"]["parallelism"] = parallelism + " (the tensor-parallel phase failed: see tensor_parallel.error)"
            line = out
    del eng
    if configs3_tp:
        # BASELINE.json configs[3] on the same ranks: Qwen3-8B, bs 32 x 2048, tensor parallel over the N GPUs (every rank takes part;
        # a hang here costs nothing already measured: the watchdog prints the line as it stands)
        fallback_state["line"] = line
        if rank == 0 and line is not None and child:
            # partial result first (children only: their stdout is read by the parent, which prints ONE line): a parent that has to cut this child
            # short still has the tensor-parallel headline
            print(json.dumps(dict(line, partial="the configs[3] tensor-parallel side block was still running")), flush=True)
        arm("configs[3] tensor-parallel side block", 240.0)

        def rmax(v):
            return dist.max(v)
        try:
            c3 = side_decode(nvr, "qwen3-8b", tp_size=args.gpus, tp_rank=rank, device=local_rank, attach=init_tensor_parallel, barrier=barrier, reduce_max=rmax,
                             prompt_len=dry_prompt or None)
        except Exception as ex:                                              # noqa: BLE001
            c3 = {"error": str(ex)[:300]}
        if line is not None:
            line["configs3"] = c3
    if rank == 0 and line is not None:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.close()
        if watchdog is not None:
            watchdog.cancel()


if __name__ == "__main__":
    main()
