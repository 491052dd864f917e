"""Qwen3-8B shapes (BASELINE configs[3] geometry) on ONE GPU: bs=32, 2048-token prompts, greedy decode.  TP=N in the environment runs
ONE rank of a tensor-parallel degree N without a communicator (all-reduces skipped, results meaningless): the compute-only floor."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nvr_import
nvr = nvr_import.load()
mc = nvr.ModelConfig("qwen3-8b")
B, P, steps = int(os.environ.get("BATCH", "32")), int(os.environ.get("PROMPT", "2048")), 24
TP = int(os.environ.get("TP", "1"))
t0 = time.perf_counter()
eng = nvr.LLMEngine(nvr.Config(max_num_seqs=B, max_num_batched_tokens=32768, max_model_len=P + 64, kvcache_block_size=256,
                               num_kvcache_blocks=B * (P // 256 + 2), tensor_parallel_size=TP, tensor_parallel_rank=0), mc)
nvr.synchronize(); print(f"init {time.perf_counter() - t0:.1f} s", flush=True)
for i in range(B):
    eng.add_request(nvr.synthetic_tokens(P, 1, i, 151936).tolist(), nvr.SamplingParams(temperature=0.0, max_tokens=steps + 8, ignore_eos=True))
t0 = time.perf_counter(); npre = 0
while True:
    rec = eng.step()
    if not rec["is_prefill"]: break
    npre += 1
nvr.synchronize(); tp = time.perf_counter() - t0
for _ in range(4): eng.step()
nvr.synchronize(); t0 = time.perf_counter()
for _ in range(steps): eng.step()
nvr.synchronize(); dt = time.perf_counter() - t0
w_bytes = 2 * (36 * (4096 * 6144 + 4096 * 4096 + 2 * 12288 * 4096 + 12288 * 4096) + 151936 * 4096)
kv_bytes = B * (P + 16) * 36 * 2 * 8 * 128 * 2
print(f"tp={TP} prefill {B * P} tokens in {npre} steps: {tp * 1e3:.1f} ms ({B * P / tp / 1e3:.1f} k tok/s); decode {dt / steps * 1e3:.3f} ms/step = {B * steps / dt:.0f} tok/s; "
      f"algorithmic {(w_bytes + kv_bytes) / 1e9:.1f} GB/step -> {(w_bytes + kv_bytes) / (dt / steps) / 1e12:.2f} TB/s", flush=True)
if not os.environ.get("NVR_NO_EXIT"): os._exit(0)
