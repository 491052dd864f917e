"""BASELINE configs[4] geometry in decode: 512 sequences sharing a 512-token prefix + 64 own tokens, Qwen3-0.6B."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nvr_import
nvr = nvr_import.load()
mc = nvr.ModelConfig("qwen3-0.6b")
B = 512
eng = nvr.LLMEngine(nvr.Config(max_num_seqs=B, max_num_batched_tokens=65536, max_model_len=1024, kvcache_block_size=256, num_kvcache_blocks=B * 2 + 16), mc)
shared = nvr.synthetic_tokens(512, 2, 0, 151936).tolist()
for i in range(B):
    eng.add_request(shared + nvr.synthetic_tokens(64, 1, i, 151936).tolist(), nvr.SamplingParams(temperature=0.0, max_tokens=60, ignore_eos=True))
t0 = time.perf_counter(); npre = 0
while True:
    rec = eng.step()
    if not rec["is_prefill"]: break
    npre += 1
nvr.synchronize(); print(f"prefill: {npre} steps, {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
for _ in range(4): eng.step()
nvr.synchronize(); t0 = time.perf_counter()
for _ in range(24): eng.step()
nvr.synchronize(); dt = (time.perf_counter() - t0) / 24
print(f"bs={B} ctx~600 (512 shared): {dt * 1e3:.3f} ms/step  {B / dt:.0f} tok/s", flush=True)
if not os.environ.get("NVR_NO_EXIT"): os._exit(0)
