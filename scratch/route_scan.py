"""ms per decode step over batch sizes around every routing boundary (short contexts: the GEMM routes show, not the attention stream):
python scratch/route_scan.py qwen3-0.6b 256 16 24 32 33 40 ...   -> batch, ms/step, us per sequence"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nvr_import
nvr = nvr_import.load()
preset, P = sys.argv[1], int(sys.argv[2])
mc = nvr.ModelConfig(preset)
prev = None
for B in map(int, sys.argv[3:]):
    eng = nvr.LLMEngine(nvr.Config(max_num_seqs=B, max_num_batched_tokens=32768, max_model_len=P + 64, kvcache_block_size=256, num_kvcache_blocks=B * (P // 256 + 2),
                                   tensor_parallel_size=int(os.environ.get("TP", "1")), tensor_parallel_rank=0), mc)   # TP=n with NVR_TP_NO_COMM=1: ONE rank's compute, exchanges skipped
    for i in range(B):
        eng.add_request(nvr.synthetic_tokens(P, 1, i, 151936).tolist(), nvr.SamplingParams(temperature=0.0, max_tokens=40, ignore_eos=True))
    while eng.step()["is_prefill"]: pass
    for _ in range(5): eng.step()
    nvr.synchronize(); t0 = time.perf_counter()
    for _ in range(20): eng.step()
    nvr.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
    flag = "" if prev is None or ms >= prev[1] * 0.995 else "   <-- faster than the smaller batch"
    if prev is not None and (ms - prev[1]) / (B - prev[0]) * 1e3 > 60 and preset == "qwen3-0.6b": flag += "   <-- step up"
    print(f"{preset} ctx {P}: bs {B:4d}  {ms:7.3f} ms/step  {ms / B * 1e3:7.2f} us per sequence{flag}", flush=True)
    prev = (B, ms)
    del eng
if not os.environ.get("NVR_NO_EXIT"): os._exit(0)
