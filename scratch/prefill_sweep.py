"""BASELINE configs[2]: Qwen3-0.6B fp16, 256 sequences at L in {128..4096}, token budget 32768 per prefill step
(the reference batches whole sequences, scheduler.rs:135-138): wall time of the prefill steps and model FLOP/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nvr_import
nvr = nvr_import.load()
mc = nvr.ModelConfig("qwen3-0.6b")
V, Hd, I, Lyr, H, D = 151936, 1024, 3072, 28, 16, 128
gemm_flop_per_token = 2 * (Hd * (H + 16) * D + H * D * Hd + 2 * I * Hd + I * Hd) * Lyr          # qkv(4096) + o + gate_up + down
for L in [int(x) for x in (sys.argv[1:] or [128, 256, 512, 1024, 2048, 4096])]:
    nseq = 256
    nblk = (L + 255) // 256 + 1
    cfg = nvr.Config(max_num_seqs=256, max_num_batched_tokens=32768, max_model_len=L + 64, kvcache_block_size=256, num_kvcache_blocks=nseq * nblk + 8)
    nvr.lib().nvr_seq_reset_id_counter()
    eng = nvr.LLMEngine(cfg, mc)
    for i in range(nseq):
        eng.add_request(nvr.synthetic_tokens(L, 1, i, V).tolist(), nvr.SamplingParams(temperature=0.0, max_tokens=2, ignore_eos=True))
    nvr.synchronize(); steps = 0; rows = 0; dt = 0.0
    while True:
        t0 = time.perf_counter()
        rec = eng.step()                      # a step ends with the D2H of its tokens: the device is idle when it returns
        t1 = time.perf_counter()
        if not rec["is_prefill"]: break       # the first decode step (graph capture) is not part of the prefill time
        steps += 1; rows += rec["num_tokens"]; dt += t1 - t0
    attn = 4 * H * D * (L + 1) / 2 * Lyr                                  # causal, per token
    flop = rows * (gemm_flop_per_token + attn) + nseq * 2 * V * Hd
    print(f"L={L:5d}: {steps:3d} prefill steps, {rows} tokens, {dt * 1e3:8.1f} ms -> {rows / dt / 1e3:7.1f} k tok/s, {flop / dt / 1e12:6.1f} TFLOP/s = {flop / dt / 2.5e15 * 100:4.1f} % of 2.5 PF", flush=True)
    del eng
