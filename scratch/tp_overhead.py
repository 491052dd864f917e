"""What do the 57 RCCL collectives of a tensor-parallel decode step cost in launch overhead alone?  One rank, collectives
kept in the step (NVR_TP_FORCE_COMM=1): Qwen3-0.6B bs=32 ctx 1024, graph replay vs eager, with and without collectives."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
force = sys.argv[1] == "1"; eager = sys.argv[2] == "1"
os.environ["NVR_TP_FORCE_COMM"] = "1" if force else "0"
import nvr_import
nvr = nvr_import.load()
mc = nvr.ModelConfig("qwen3-0.6b")
eng = nvr.LLMEngine(nvr.Config(max_num_seqs=32, max_num_batched_tokens=32768, max_model_len=1200, kvcache_block_size=256, num_kvcache_blocks=200, enforce_eager=int(eager)), mc)
if force:
    eng.model_runner.init_comm(nvr.comm_unique_id())
for i in range(32):
    eng.add_request(nvr.synthetic_tokens(1024, 1, i, 151936).tolist(), nvr.SamplingParams(temperature=0.0, max_tokens=80, ignore_eos=True))
eng.step()
for _ in range(8): eng.step()
nvr.synchronize(); t0 = time.perf_counter()
for _ in range(32): eng.step()
nvr.synchronize(); dt = time.perf_counter() - t0
print(f"collectives={'on ' if force else 'off'} {'eager' if eager else 'graph'}: {dt / 32 * 1e3:.3f} ms/step", flush=True)
os._exit(0)
