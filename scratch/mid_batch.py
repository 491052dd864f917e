"""Decode step time for batches of 96..512 sequences (Qwen3-0.6B): the LDS-tiled GEMM regime."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nvr_import
nvr = nvr_import.load()
mc = nvr.ModelConfig("qwen3-0.6b")
for B, ctx in [tuple(map(int, a.split("x"))) for a in sys.argv[1:]] or [(96, 256), (128, 256), (128, 512), (192, 256), (256, 256), (384, 256), (512, 256)]:
    eng = nvr.LLMEngine(nvr.Config(max_num_seqs=B, max_num_batched_tokens=max(32768, B * ctx), max_model_len=ctx + 128, kvcache_block_size=256,
                                   num_kvcache_blocks=B * (ctx // 256 + 2)), mc)
    for i in range(B):
        eng.add_request(nvr.synthetic_tokens(ctx, 1, i, 151936).tolist(), nvr.SamplingParams(temperature=0.0, max_tokens=100, ignore_eos=True))
    while eng.step()["is_prefill"]: pass
    for _ in range(8): eng.step()
    nvr.synchronize(); t0 = time.perf_counter()
    for _ in range(48): eng.step()
    nvr.synchronize(); dt = (time.perf_counter() - t0) / 48
    print(f"bs={B:4d} ctx={ctx:5d}: {dt * 1e3:.3f} ms/step  {B / dt:9.0f} tok/s", flush=True)
    del eng

