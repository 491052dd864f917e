"""Compute-only time of ONE rank's decode step at tensor-parallel degree tp (no communicator: the all-reduces are skipped,
results are meaningless) — the floor a perfect all-reduce would leave.  Qwen3-0.6B, bs=32, ctx 1024."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nvr_import
nvr = nvr_import.load()
tp = int(sys.argv[1])
mc = nvr.ModelConfig("qwen3-0.6b")
eng = nvr.LLMEngine(nvr.Config(max_num_seqs=32, max_num_batched_tokens=32768, max_model_len=1200, kvcache_block_size=256, num_kvcache_blocks=200,
                               tensor_parallel_size=tp, tensor_parallel_rank=0), mc)
for i in range(32):
    eng.add_request(nvr.synthetic_tokens(1024, 1, i, 151936).tolist(), nvr.SamplingParams(temperature=0.0, max_tokens=80, ignore_eos=True))
eng.step()
for _ in range(8): eng.step()
nvr.synchronize(); t0 = time.perf_counter()
for _ in range(32): eng.step()
nvr.synchronize(); dt = time.perf_counter() - t0
print(f"tp={tp}: {dt / 32 * 1e3:.3f} ms/step compute only (one rank, collectives skipped)", flush=True)
if not os.environ.get("NVR_NO_EXIT"): os._exit(0)
