"""ms per decode step of the float32 path at bs 1 (BASELINE configs[0]: Qwen3-0.6B, 128-token prompt), hipGraph steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nvr_import
nvr = nvr_import.load()
mc = nvr.ModelConfig("qwen3-0.6b")
eng = nvr.LLMEngine(nvr.Config(max_num_seqs=1, max_num_batched_tokens=256, max_model_len=512, kvcache_block_size=256, num_kvcache_blocks=4, dtype="float32"), mc)
eng.add_request(nvr.synthetic_tokens(128, 1, 0, mc.c.vocab_size).tolist(), nvr.SamplingParams(temperature=0.0, max_tokens=120, ignore_eos=True))
while eng.step()["is_prefill"]: pass
for _ in range(8): eng.step()
nvr.synchronize(); t0 = time.perf_counter()
for _ in range(64): eng.step()
nvr.synchronize(); print(f"float32 bs 1: {(time.perf_counter() - t0) / 64 * 1e3:.3f} ms/step", flush=True)
os._exit(0)
