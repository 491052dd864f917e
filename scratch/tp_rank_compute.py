"""Compute-only time of ONE rank's decode step at tensor-parallel degree tp (NVR_TP_NO_COMM=1: the exchanges are skipped, results are
meaningless) — the floor a free all-reduce would leave; DESIGN.md §6 prices the collectives on top of it.
usage: python3 scratch/tp_rank_compute.py <tp> [qwen3-0.6b | qwen3-8b]   (bs 32 x 1024 / bs 32 x 2048, the BASELINE configs[1] / [3] shapes)"""
import os, sys, time
os.environ["NVR_TP_NO_COMM"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nvr_import
nvr = nvr_import.load()
tp = int(sys.argv[1])
model = sys.argv[2] if len(sys.argv) > 2 else "qwen3-0.6b"
P = 1024 if model == "qwen3-0.6b" else 2048
mc = nvr.ModelConfig(model)
eng = nvr.LLMEngine(nvr.Config(max_num_seqs=32, max_num_batched_tokens=32768, max_model_len=P + 120, kvcache_block_size=256, num_kvcache_blocks=32 * (P // 256 + 2),
                               tensor_parallel_size=tp, tensor_parallel_rank=0), mc)
for i in range(32):
    eng.add_request(nvr.synthetic_tokens(P, 1, i, 151936).tolist(), nvr.SamplingParams(temperature=0.0, max_tokens=80, ignore_eos=True))
while eng.step()["is_prefill"]:
    pass
for _ in range(8): eng.step()
nvr.synchronize(); t0 = time.perf_counter()
for _ in range(32): eng.step()
nvr.synchronize(); dt = time.perf_counter() - t0
print(f"{model} tp={tp}: {dt / 32 * 1e3:.3f} ms/step compute only (one rank, collectives skipped, synchronous steps)", flush=True)
del eng
