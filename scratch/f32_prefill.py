"""r06: prefill of the float32 path (Qwen3-0.6B): 1 x 128 tokens (BASELINE configs[0]) and 8 x 1024 tokens, warm, wall clock; under rocprofv3 for the kernel breakdown."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nvr_import
nvr = nvr_import.load()
mc = nvr.ModelConfig("qwen3-0.6b")
for B, P in ((1, 128), (8, 1024)):
    eng = nvr.LLMEngine(nvr.Config(max_num_seqs=B, max_num_batched_tokens=B * P, max_model_len=P + 16, kvcache_block_size=256, num_kvcache_blocks=2 * B * (P // 256 + 2), dtype="float32"), mc)
    for seed in (3, 1):
        for i in range(B):
            eng.add_request(nvr.synthetic_tokens(P, seed, i, mc.c.vocab_size).tolist(), nvr.SamplingParams(temperature=0.0, max_tokens=1, ignore_eos=True))
        nvr.synchronize(); t0 = time.perf_counter()
        while not eng.is_finished(): eng.step()
        nvr.synchronize(); dt = time.perf_counter() - t0
        eng.take_finished()
    print(f"float32 prefill {B} x {P}: {dt * 1e3:.2f} ms = {B * P / dt / 1e3:.1f} k tok/s ({B * P * 0.8808e9 / dt / 1e12:.1f} TFLOP/s of GEMM)", flush=True)
    del eng
if not os.environ.get("NVR_NO_EXIT"): os._exit(0)
