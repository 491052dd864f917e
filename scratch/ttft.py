"""Prefill latency (time to first token) of small batches, Qwen3-0.6B: one engine step, wall clock after a warm-up step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nvr_import
nvr = nvr_import.load()
mc = nvr.ModelConfig("qwen3-0.6b")
for B, L in [(1, 128), (1, 1024), (1, 4096), (4, 1024), (8, 1024), (8, 128), (64, 128)]:
    eng = nvr.LLMEngine(nvr.Config(max_num_seqs=2 * B, max_num_batched_tokens=max(32768, B * L), max_model_len=L + 64, kvcache_block_size=256,
                                   num_kvcache_blocks=2 * B * (L // 256 + 2)), mc)
    ts = []
    for rep in range(3):
        for i in range(B):
            eng.add_request(nvr.synthetic_tokens(L, 1, 100 * rep + i, 151936).tolist(), nvr.SamplingParams(temperature=0.0, max_tokens=1, ignore_eos=True))
        nvr.synchronize(); t0 = time.perf_counter()
        rec = eng.step()
        nvr.synchronize(); ts.append(time.perf_counter() - t0)
        assert rec["is_prefill"]
        while not eng.is_finished(): eng.step()
        eng.take_finished()
    t = min(ts[1:])
    flop = B * L * 880.8e6 + B * 114688 * L * (L + 1) / 2 + B * 2 * 151936 * 1024
    print(f"bs={B:3d} L={L:5d}: {t * 1e3:7.3f} ms  {B * L / t / 1e3:8.1f} k tok/s  {flop / t / 1e12:7.1f} TFLOP/s", flush=True)
    del eng
os._exit(0)
