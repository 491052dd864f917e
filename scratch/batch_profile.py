"""One decode configuration for rocprofv3: B sequences at context CTX (env), Qwen3-0.6B."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nvr_import
nvr = nvr_import.load()
mc = nvr.ModelConfig("qwen3-0.6b")
B, ctx = int(os.environ.get("B", "128")), int(os.environ.get("CTX", "512"))
eng = nvr.LLMEngine(nvr.Config(max_num_seqs=B, max_num_batched_tokens=max(32768, B * ctx), max_model_len=ctx + 128, kvcache_block_size=256,
                               num_kvcache_blocks=B * (ctx // 256 + 2)), mc)
for i in range(B):
    eng.add_request(nvr.synthetic_tokens(ctx, 1, i, 151936).tolist(), nvr.SamplingParams(temperature=0.0, max_tokens=100, ignore_eos=True))
while eng.step()["is_prefill"]: pass
for _ in range(4): eng.step()
nvr.synchronize(); t0 = time.perf_counter()
for _ in range(16): eng.step()
nvr.synchronize(); dt = (time.perf_counter() - t0) / 16
print(f"bs={B} ctx={ctx}: {dt * 1e3:.3f} ms/step  {B / dt:.0f} tok/s", flush=True)
if not os.environ.get("NVR_NO_EXIT"): os._exit(0)
