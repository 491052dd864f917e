"""GPU-only decode step time (execute_model replayed back to back, no sampling/host sync) vs engine.step() wall time."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nvr_import
nvr = nvr_import.load(); l = nvr.lib(); nvr.check(l.nvr_device_set(0))
cfg = nvr.Config(max_num_seqs=32, max_num_batched_tokens=32 * 1024, max_model_len=1200, kvcache_block_size=256, num_kvcache_blocks=200)
mc = nvr.ModelConfig("qwen3-0.6b")
eng = nvr.LLMEngine(cfg, mc)
for i in range(32):
    eng.add_request(nvr.synthetic_tokens(1024, 1, i, 151936).tolist(), nvr.SamplingParams(temperature=0.0, max_tokens=100, ignore_eos=True))
eng.step()
for _ in range(8): eng.step()
nvr.synchronize(); t0 = time.perf_counter()
for _ in range(32): eng.step()
nvr.synchronize(); wall = (time.perf_counter() - t0) / 32
seqs = eng.last_batch()
r = eng.model_runner
r.execute_model(seqs, False); nvr.synchronize()
t0 = time.perf_counter()
for _ in range(32): r.execute_model(seqs, False)
nvr.synchronize(); gpu = (time.perf_counter() - t0) / 32
print(f"engine.step wall {wall*1e3:.4f} ms   execute_model back-to-back {gpu*1e3:.4f} ms   host/sync gap {1e3*(wall-gpu):.4f} ms")
