"""Host gap between decode steps: n engine steps (host round trip per step) vs the same captured graph replayed n times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nvr_import
nvr = nvr_import.load(); L = nvr.lib()
mc = nvr.ModelConfig("qwen3-0.6b")
eng = nvr.LLMEngine(nvr.Config(max_num_seqs=32, max_num_batched_tokens=32768, max_model_len=1400, kvcache_block_size=256, num_kvcache_blocks=200), mc)
for i in range(32):
    eng.add_request(nvr.synthetic_tokens(1024, 1, i, 151936).tolist(), nvr.SamplingParams(temperature=0.0, max_tokens=300, ignore_eos=True))
eng.step()
for _ in range(8): eng.step()
n = 64
nvr.synchronize(); t0 = time.perf_counter()
for _ in range(n): eng.step()
nvr.synchronize(); t_eng = (time.perf_counter() - t0) / n
nvr.check(L.nvr_runner_replay_last_decode_graph(eng.model_runner.h, 8)); nvr.synchronize()
t0 = time.perf_counter()
nvr.check(L.nvr_runner_replay_last_decode_graph(eng.model_runner.h, n)); nvr.synchronize()
t_gpu = (time.perf_counter() - t0) / n
print(f"engine step {t_eng * 1e6:.1f} us (mean ctx ~{1024 + 9 + n // 2}); graph replay {t_gpu * 1e6:.1f} us (ctx {1024 + 9 + n}); difference {1e6 * (t_eng - t_gpu):.1f} us", flush=True)
os._exit(0)
