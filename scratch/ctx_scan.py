"""ms per decode step over context lengths at a fixed batch (the attention launch's partition rules): python scratch/ctx_scan.py qwen3-0.6b 32 64 128 256 ..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nvr_import
nvr = nvr_import.load()
preset, B = sys.argv[1], int(sys.argv[2])
mc = nvr.ModelConfig(preset)
kv_tok = mc.c.num_hidden_layers * 2 * mc.c.num_key_value_heads * (mc.c.head_dim or mc.c.hidden_size // mc.c.num_attention_heads) * 2
prev = None
for P in map(int, sys.argv[3:]):
    eng = nvr.LLMEngine(nvr.Config(max_num_seqs=B, max_num_batched_tokens=32768, max_model_len=P + 64, kvcache_block_size=256, num_kvcache_blocks=B * ((P + 64) // 256 + 2)), mc)
    for i in range(B):
        eng.add_request(nvr.synthetic_tokens(P, 1, i, 151936).tolist(), nvr.SamplingParams(temperature=0.0, max_tokens=40, ignore_eos=True))
    while eng.step()["is_prefill"]: pass
    for _ in range(5): eng.step()
    nvr.synchronize(); t0 = time.perf_counter()
    for _ in range(20): eng.step()
    nvr.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
    extra = "" if prev is None else f"   +{(ms - prev[1]) * 1e3 / max(1, P - prev[0]) / B * 1e3:7.2f} ns per added key and sequence ({kv_tok / 8e3:.1f} ns at 8 TB/s)"
    print(f"{preset} bs {B}: ctx {P:6d}  {ms:7.3f} ms/step{extra}", flush=True)
    prev = (P, ms)
    del eng
if not os.environ.get("NVR_NO_EXIT"): os._exit(0)
