#!/usr/bin/env python3
"""bench.py — decode throughput of the paged-attention hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): Qwen3-0.6B shape, fp16, synthetic random-init weights, 32
sequences with 1024-token synthetic prompts, greedy decode, block size 256, hipGraph decode steps.
A "step" is one decode pass of the engine over the whole batch (schedule -> execute_model ->
sample_tokens -> postprocess through the C ABI), i.e. 32 generated tokens.  The prefill of the
32x1024 prompt tokens happens before the timed region; K steps are timed between barriers and
device synchronisation; the maximum over ranks is reported.  N>1 shards heads / MLP columns / vocab
across N ranks.  Sequences are independent units, so N>1 first measures N replicas (one engine and
its own 32 sequences per GPU, no exchange: "scaling": "weak") — that is `value` — and then, on the
same ranks, the north star's tensor-parallel configuration (heads / MLP columns / vocabulary sharded,
RCCL all-reduce: the same 32 sequences, "strong"), attached as "tensor_parallel".  `--parallel tp`
makes the tensor-parallel run the value instead; if its communicator cannot be built or its phase
does not finish, the replicas measurement is what is reported.

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

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import nvr_import  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)

BATCH, PROMPT_LEN, BLOCK = 32, 1024, 256


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
    nvr.check(l.nvr_event_record(e0, stream))
    for _ in range(reps):
        sweep()
    nvr.check(l.nvr_event_record(e1, stream))
    ms = C.c_float()
    nvr.check(l.nvr_event_elapsed_ms(e0, e1, C.byref(ms)))
    launches = reps * L
    us = ms.value * 1e3 / launches
    alg_bytes = float(ctx.sum()) * 2 * KVH * D * 2 + 2 * B * H * D * 2          # K+V rows read + q in + out
    l.nvr_event_destroy(e0); l.nvr_event_destroy(e1); l.nvr_stream_destroy(stream)
    return dict(us_per_launch=us, launches=launches, alg_bytes=alg_bytes, ctx_sum=int(ctx.sum()))


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
    ap.add_argument("--eager", action="store_true", help="enforce_eager: launch decode kernels one by one instead of replaying a hipGraph")
    ap.add_argument("--parallel", choices=["both", "tp", "replicas"], default=os.environ.get("NVR_BENCH_PARALLEL", "both"),
                    help="--gpus N > 1: 'replicas' = N independent engines, 32 sequences each, no exchange between ranks (sequences are "
                         "independent units: weak scaling); 'tp' = one tensor-parallel engine over N GPUs (RCCL all-reduce, strong scaling: "
                         "the same 32 sequences) as the reported value; 'both' (default) = value from the replicas, the tensor-parallel run "
                         "measured right after and attached as \"tensor_parallel\"")
    ap.add_argument("--materialize-logits", action="store_true",
                    help="write the f32 logits of every step to HBM (default: a greedy batch takes its tokens from the arg-max "
                         "partials of the LM-head epilogue and the logits are written only when someone asks for them)")
    args = ap.parse_args()
    if args.materialize_logits:
        os.environ["NVR_LAZY_LOGITS"] = "0"

    nvr = nvr_import.load()                # loads libnvr.so (and the ROCm HIP runtime) before torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("NVR_BENCH_SHARED_GPU"):          # control-plane dry run of the multi-rank path on a 1-GPU box
        local_rank = 0
    dist = None
    watchdog = None
    fallback_state: dict = {}
    if args.gpus > 1:
        # the multi-rank path cannot be exercised on the 1-GPU development boxes: never hang the driver — if a rank is still
        # stuck (a collective that never completes, a rendezvous that never forms) after 10 minutes, every rank exits
        import threading
        wd_secs = 240.0 if os.environ.get("NVR_BENCH_CHILD") == "1" else 600.0

        def _bail():
            print(f"[bench] rank {rank}: multi-GPU run made no progress for {int(wd_secs)} s, giving up", file=sys.stderr, flush=True)
            rep = fallback_state.get("replicas")
            if rank == 0 and rep is not None:
                # the tensor-parallel phase hung after the replicas phase finished: report what was measured
                print(json.dumps({"metric": "decode tokens/s + %HBM-roofline, Qwen3-0.6B bs=32 seq=1024, 1/2/4/8 GPU", "value": rep["value"],
                                  "unit": "tokens/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
                                  "ms_per_step": rep["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                                  "dtype": "f16", "data": "synthetic",
                                  "config": {"workload": "Qwen3-0.6B fp16 random-init, bs=32 x 1024-token prompts per GPU, greedy paged-attention decode",
                                             "parallelism": rep["parallelism"] + " (the tensor-parallel phase did not complete within 600 s)"},
                                  "roofline": None, "replicas": rep}), flush=True)
            os._exit(0 if rep is not None else 4)      # a measured line went out: let the launcher finish normally
        watchdog = threading.Timer(wd_secs, _bail)
        watchdog.daemon = True
        watchdog.start()
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} needs WORLD_SIZE={args.gpus} (launch with torch.distributed.run)")
        nvr.preload_rccl()                 # ROCm's librccl before torch's bundled copy can claim the soname
        import torch.distributed as dist   # control plane only (gloo): unique-id broadcast, barrier, max
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    total_new = args.warmup + args.steps + 1
    mc = nvr.ModelConfig("qwen3-0.6b")
    nvr.check(nvr.lib().nvr_device_set(local_rank))

    def make_engine(tp_size: int, tp_rank: int):
        cfg = nvr.Config(max_num_seqs=BATCH, max_num_batched_tokens=BATCH * PROMPT_LEN, max_model_len=PROMPT_LEN + total_new + 16,
                         kvcache_block_size=BLOCK, num_kvcache_blocks=BATCH * ((PROMPT_LEN + total_new + 16) // BLOCK + 2),
                         tensor_parallel_size=tp_size, tensor_parallel_rank=tp_rank,
                         device_ordinal=local_rank, enforce_eager=args.eager)
        return nvr.LLMEngine(cfg, mc)

    def barrier():
        nvr.synchronize()
        if dist is not None:
            dist.barrier()

    def run_decode(eng):
        """prefill (untimed), W warm-up steps, K timed steps between barriers; max over ranks"""
        for i in range(BATCH):               # synthetic prompts, SURVEY §8d: seed 1, one stream per sequence
            eng.add_request(nvr.synthetic_tokens(PROMPT_LEN, 1, i, mc.c.vocab_size).tolist(),
                            nvr.SamplingParams(temperature=0.0, max_tokens=total_new + 8, ignore_eos=True))
        t0 = time.perf_counter()
        info = eng.step()                    # prefill of 32 x 1024 tokens (untimed)
        nvr.synchronize()
        t_pre = time.perf_counter() - t0
        assert info["is_prefill"] and info["num_seqs"] == BATCH, info
        for _ in range(args.warmup):
            info = eng.step()
            assert not info["is_prefill"] and info["num_seqs"] == BATCH
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            eng.step()
        nvr.synchronize()
        el = time.perf_counter() - t0
        barrier()
        if dist is not None:
            import torch
            t = torch.tensor([el], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, t_pre

    parallelism, scaling, jobs = "tp1", "strong", 1
    replicas = None                           # N > 1: the no-exchange measurement (N independent engines), always taken first
    tensor_parallel = None
    child = os.environ.get("NVR_BENCH_CHILD") == "1"
    if args.gpus > 1 and child:
        # the tensor-parallel phase of a parent bench.py (see "both" below): no replicas leg; a failure is reported as a JSON error line
        import torch
        eng = make_engine(args.gpus, rank)
        uid = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            uid = torch.frombuffer(bytearray(nvr.comm_unique_id()), dtype=torch.uint8).clone()
        dist.broadcast(uid, 0)
        ok, why = 1, ""
        try:
            eng.model_runner.init_comm(bytes(uid.numpy().tobytes()))       # RCCL communicator + collective self-test
        except Exception as ex:                                              # noqa: BLE001
            ok, why = 0, str(ex)
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            if rank == 0:
                print(json.dumps({"error": "tensor-parallel communicator could not be built on this node" + (f": {why}" if why else "")}), flush=True)
            dist.barrier(); dist.destroy_process_group()
            sys.stdout.flush(); os._exit(0)
        parallelism, scaling, jobs = f"tp{args.gpus}", "strong", 1
        elapsed, t_prefill = run_decode(eng)
    elif args.gpus > 1:
        eng = make_engine(1, 0)
        r_el, r_pre = run_decode(eng)
        replicas = {"value": round(args.gpus * BATCH * args.steps / r_el, 2), "unit": "tokens/s", "ms_per_step": round(r_el * 1e3 / args.steps, 4),
                    "scaling": "weak", "parallelism": f"replicas{args.gpus}",
                    "note": f"{args.gpus} independent engines (one full model and its own 32 sequences per GPU), no exchange between ranks"}
        fallback_state["replicas"] = replicas
        if args.parallel == "replicas":
            parallelism, scaling, jobs = f"replicas{args.gpus}", "weak", args.gpus
            elapsed, t_prefill = r_el, r_pre
        elif args.parallel == "both":
            # value = the replicas (what a deployment picks for a model this small); the north star's tensor-parallel configuration
            # is measured on the same ranks right after, in a CHILD process per rank (its own gloo group on the next port): the
            # multi-GPU RCCL path could not be exercised on the 1-GPU development boxes, and a crash or hang in it must not cost
            # the measurement already taken
            import subprocess
            parallelism, scaling, jobs = f"replicas{args.gpus}", "weak", args.gpus
            elapsed, t_prefill = r_el, r_pre
            env = dict(os.environ)
            env["MASTER_PORT"] = str(int(env.get("MASTER_PORT", "29500")) + 1)
            env["NVR_BENCH_CHILD"] = "1"
            for k in [k for k in env if k.startswith("TORCHELASTIC_")]:       # the children rendezvous on their own store (rank 0 hosts it),
                del env[k]                                                    # not on the launcher agent's
            cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup),
                   "--parallel", "tp", "--no-cpu-baseline", "--attn-reps", "1"] + (["--eager"] if args.eager else [])
            child_out, child_err, child_rc = "", "", None
            try:
                cp = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
                child_out, child_err, child_rc = cp.stdout, cp.stderr, cp.returncode
            except subprocess.TimeoutExpired as te:
                child_err = "timed out after 300 s: " + str((te.stderr or b"")[-300:])
            if rank == 0:
                line = next((l for l in reversed(child_out.splitlines()) if l.startswith("{")), None)
                try:
                    cj = json.loads(line) if line else None
                except Exception:                                            # noqa: BLE001
                    cj = None
                if cj and cj.get("config", {}).get("parallelism") == f"tp{args.gpus}":
                    tensor_parallel = {"value": cj["value"], "unit": "tokens/s", "ms_per_step": cj["ms_per_step"], "scaling": "strong",
                                       "parallelism": f"tp{args.gpus}", "speedup_vs_one_gpu": round(cj["value"] / (replicas["value"] / args.gpus), 3),
                                       "roofline": cj.get("roofline"),
                                       "note": "one engine over all ranks: heads / MLP columns / vocabulary sharded, RCCL all-reduce after o_proj and "
                                               "down_proj (57 collectives per step), the same 32 sequences as at N=1"}
                else:
                    why = (cj or {}).get("error") or f"child exit code {child_rc}: {child_err.strip()[-400:]}"
                    tensor_parallel = {"error": why}
            barrier()
        else:
            import torch
            eng_rep = eng                     # kept alive (5.6 GB): its KV pool backs the attention timing if the TP phase fails
            eng = make_engine(args.gpus, rank)
            uid = torch.zeros(128, dtype=torch.uint8)
            if rank == 0:
                uid = torch.frombuffer(bytearray(nvr.comm_unique_id()), dtype=torch.uint8).clone()
            dist.broadcast(uid, 0)
            ok, why = 1, ""
            try:
                eng.model_runner.init_comm(bytes(uid.numpy().tobytes()))   # RCCL communicator + collective self-test
            except Exception as ex:                                          # noqa: BLE001
                ok, why = 0, str(ex)
            flag = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                # The tensor-parallel communicator could not be built on this node: report it; the replicas measurement stands.
                if why:
                    print(f"[bench] rank {rank}: tensor-parallel init failed: {why}", file=sys.stderr, flush=True)
                del eng
                eng = eng_rep
                parallelism, scaling, jobs = f"replicas{args.gpus} (tensor-parallel init failed)", "weak", args.gpus
                elapsed, t_prefill = r_el, r_pre
            else:
                del eng_rep
                parallelism, scaling, jobs = f"tp{args.gpus}", "strong", 1
                elapsed, t_prefill = run_decode(eng)
    else:
        eng = make_engine(1, 0)
        elapsed, t_prefill = run_decode(eng)

    # prefill of the 32 x 1024 prompt tokens (one engine step, wall clock incl. host input preparation): MFMA-bound,
    # 880.8 MFLOP/token of GEMM + 114 688*l flop/token of causal attention (+ LM head per sequence)
    c = mc.c
    Dh = c.head_dim or c.hidden_size // c.num_attention_heads
    gemm_flop_tok = 2 * c.num_hidden_layers * ((c.num_attention_heads + 2 * c.num_key_value_heads) * Dh * c.hidden_size
                                               + c.hidden_size * c.num_attention_heads * Dh + 3 * c.intermediate_size * c.hidden_size)
    attn_flop = BATCH * sum(4 * c.num_attention_heads * Dh * (l + 1) // 2 * 2 for l in range(PROMPT_LEN)) * c.num_hidden_layers // 2
    prefill_flop = BATCH * PROMPT_LEN * gemm_flop_tok + attn_flop + BATCH * 2 * c.vocab_size * c.hidden_size
    ms_per_step = elapsed * 1e3 / args.steps
    tokens_per_s = jobs * BATCH * args.steps / elapsed
    ctx_mean = PROMPT_LEN + 1 + args.warmup + (args.steps - 1) / 2.0   # keys visible per sequence, averaged over timed steps
    mb = model_bytes_per_step(mc.c, ctx_mean)
    step_bytes = mb["weights"] + mb["kv_read"] + mb["kv_write"]
    step_gbs = step_bytes / (args.gpus / jobs) / (ms_per_step * 1e-3) / 1e9   # per-GPU share of the algorithmic bytes

    attn = time_attention_kernel(nvr, eng, mc, args.attn_reps)
    achieved = attn["alg_bytes"] / (attn["us_per_launch"] * 1e-6) / 1e9
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_attn_latest.json")
    if os.path.exists(pmc_path):
        try:
            # PMC pass (profiles/): HBM bytes / algorithmic bytes of this kernel, scaled to this run's launch
            traffic = int(round(json.load(open(pmc_path))["traffic_over_algorithmic"] * attn["alg_bytes"]))
        except Exception:
            traffic = None

    if rank == 0:
        out = {
            "metric": "decode tokens/s + %HBM-roofline, Qwen3-0.6B bs=32 seq=1024, 1/2/4/8 GPU",
            "value": round(tokens_per_s, 2), "unit": "tokens/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": "Qwen3-0.6B fp16 random-init, bs=32 x 1024-token prompts, greedy paged-attention decode, "
                                   "block_size=256, hipGraph decode steps (BASELINE.json configs[1])",
                       "batch": BATCH, "prompt_len": PROMPT_LEN, "mean_context": ctx_mean,
                       "parallelism": parallelism, "hipgraph": not args.eager,
                       "logits": "materialised every step" if args.materialize_logits else "greedy arg-max fused into the LM head; f32 logits on demand"},
            "prefill": {"tokens": BATCH * PROMPT_LEN, "seconds": round(t_prefill, 4), "tokens_per_s": round(BATCH * PROMPT_LEN / t_prefill, 1),
                        "tflop_per_s": round(prefill_flop / t_prefill / 1e12, 1), "mfma_frac_of_2500": round(prefill_flop / t_prefill / 2.5e15, 4),
                        "mfma_busy_frac_pmc": _pmc_prefill_busy(),
                        "note": "one untimed engine prefill step (wall clock, includes host input preparation and upload); "
                                "mfma_busy_frac_pmc = matrix-pipe busy cycles / available cycles at the clock the chip held, from the "
                                "committed counter pass profiles/pmc_mfma_prefill_latest.json (tp1 kernels)"},
            "step_hbm_frac": round(step_gbs / HBM_PEAK_GBS, 4),
            "step_algorithmic_bytes": int(step_bytes),
            "roofline": {"kernel": "attn_rows_kernel (paged decode attention, K9)", "bound": "hbm", "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "us_per_launch": round(attn["us_per_launch"], 2), "launches_timed": attn["launches"],
                         "algorithmic_bytes_per_launch": int(attn["alg_bytes"])},
        }
        if replicas is not None:
            out["replicas"] = replicas
        if tensor_parallel is not None:
            out["tensor_parallel"] = tensor_parallel
        if args.gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    del eng
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
        # Two ROCm stacks live in a multi-rank process (this image's 7.2 libraries behind libnvr.so and the 7.0 copies
        # bundled with torch); their static destructors abort at interpreter exit ("double free") after all work is
        # done.  Everything is flushed and released above: leave without running them.
        if watchdog is not None:
            watchdog.cancel()
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
