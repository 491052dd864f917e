"""ms per prefill step over token counts around the GEMM routing boundaries (one engine, prompts of T/8 tokens x 8 sequences, warm):
python scratch/prefill_scan.py qwen3-0.6b 256 512 513 ...   -> tokens, ms, k tok/s"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nvr_import
nvr = nvr_import.load()
preset = sys.argv[1]
mc = nvr.ModelConfig(preset)
S = 8
eng = nvr.LLMEngine(nvr.Config(max_num_seqs=S, max_num_batched_tokens=32768, max_model_len=4200, kvcache_block_size=256, num_kvcache_blocks=S * 18 * 2), mc)
prev = None
for T in map(int, sys.argv[2:]):
    lens = [T // S + (1 if i < T % S else 0) for i in range(S)]
    best = 1e9
    for rep in range(3):
        for i, n in enumerate(lens):
            if n: eng.add_request(nvr.synthetic_tokens(n, 7 + rep, i, 151936).tolist(), nvr.SamplingParams(temperature=0.0, max_tokens=1, ignore_eos=True))
        nvr.synchronize(); t0 = time.perf_counter()
        while not eng.is_finished(): eng.step()
        nvr.synchronize(); best = min(best, time.perf_counter() - t0)
        eng.take_finished()
    flag = "" if prev is None or best >= prev[1] * 0.98 else "   <-- faster than the smaller step"
    print(f"{preset}: {T:6d} tokens  {best * 1e3:8.3f} ms  {T / best / 1e3:8.1f} k tok/s{flag}", flush=True)
    prev = (T, best)
if not os.environ.get("NVR_NO_EXIT"): os._exit(0)
