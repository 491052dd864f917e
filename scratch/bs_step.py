"""ms per decode step at a given batch (Qwen3-0.6B, ctx 1024): python3 scratch/bs_step.py <bs>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nvr_import
nvr = nvr_import.load()
B = int(sys.argv[1]); P = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
mc = nvr.ModelConfig("qwen3-0.6b")
eng = nvr.LLMEngine(nvr.Config(max_num_seqs=B, max_num_batched_tokens=32768, max_model_len=P + 120, kvcache_block_size=256, num_kvcache_blocks=B * (P // 256 + 2), async_decode=1), mc)
for i in range(B):
    eng.add_request(nvr.synthetic_tokens(P, 1, i, 151936).tolist(), nvr.SamplingParams(temperature=0.0, max_tokens=80, ignore_eos=True))
while eng.step()["is_prefill"]: pass
for _ in range(8): eng.step()
nvr.synchronize(); t0 = time.perf_counter()
for _ in range(40): eng.step()
nvr.synchronize(); print(f"bs {B} x {P}: {(time.perf_counter() - t0) / 40 * 1e3:.3f} ms/step", flush=True)
