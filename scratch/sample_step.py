"""Decode step time with stochastic sampling (temperature / top-k / top-p) vs greedy: Qwen3-0.6B, bs=32, ctx 1024."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nvr_import
nvr = nvr_import.load()
mc = nvr.ModelConfig("qwen3-0.6b")
for name, kw in [("greedy", dict(temperature=0.0)), ("temperature 0.8", dict(temperature=0.8)), ("top_k 50", dict(temperature=0.8, top_k=50)),
                 ("top_p 0.9", dict(temperature=0.8, top_p=0.9)), ("top_k 50 + top_p 0.9", dict(temperature=0.8, top_k=50, top_p=0.9))]:
    eng = nvr.LLMEngine(nvr.Config(max_num_seqs=32, max_num_batched_tokens=32768, max_model_len=1200, kvcache_block_size=256, num_kvcache_blocks=200), mc)
    for i in range(32):
        eng.add_request(nvr.synthetic_tokens(1024, 1, i, 151936).tolist(), nvr.SamplingParams(max_tokens=100, ignore_eos=True, **kw))
    eng.step()
    for _ in range(8): eng.step()
    nvr.synchronize(); t0 = time.perf_counter()
    for _ in range(48): eng.step()
    nvr.synchronize(); dt = time.perf_counter() - t0
    print(f"{name:24s}: {dt / 48 * 1e3:.3f} ms/step", flush=True)
    del eng
if not os.environ.get("NVR_NO_EXIT"): os._exit(0)
