"""Wall time of the 32 x 1024 prefill step: first in the process, after a 512-token warm-up, and repeated (host gap vs kernels)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nvr_import
nvr = nvr_import.load()
mc = nvr.ModelConfig("qwen3-0.6b")
V = mc.c.vocab_size
warm = int(sys.argv[1]) if len(sys.argv) > 1 else 512
eng = nvr.LLMEngine(nvr.Config(max_num_seqs=32, max_num_batched_tokens=32768, max_model_len=1100, kvcache_block_size=256, num_kvcache_blocks=200), mc)
def prefill(n, L, tag):
    for i in range(n):
        eng.add_request(nvr.synthetic_tokens(L, 1, i, V).tolist(), nvr.SamplingParams(temperature=0.0, max_tokens=1, ignore_eos=True))
    nvr.synchronize(); t0 = time.perf_counter()
    rec = eng.step()
    nvr.synchronize(); dt = time.perf_counter() - t0
    assert rec["is_prefill"]
    while not eng.is_finished(): eng.step()
    eng.take_finished()
    print(f"{tag}: {n} x {L}: {dt * 1e3:.2f} ms", flush=True)
if warm: prefill(1, warm, "warm-up")
for r in range(4): prefill(32, 1024, f"prefill {r}")
