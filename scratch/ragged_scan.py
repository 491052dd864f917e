"""ms per decode step of a RAGGED batch (context lengths spread geometrically between lo and hi): python scratch/ragged_scan.py qwen3-0.6b 32 64 4096"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nvr_import
nvr = nvr_import.load()
preset, B, lo, hi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
mc = nvr.ModelConfig(preset)
lens = [int(lo * (hi / lo) ** (i / max(1, B - 1))) for i in range(B)]
eng = nvr.LLMEngine(nvr.Config(max_num_seqs=B, max_num_batched_tokens=32768, max_model_len=hi + 64, kvcache_block_size=256, num_kvcache_blocks=sum((n + 64) // 256 + 2 for n in lens),
                               tensor_parallel_size=int(os.environ.get("TP", "1")), tensor_parallel_rank=0), mc)   # TP=n with NVR_TP_NO_COMM=1: one rank's compute
for i, n in enumerate(lens):
    eng.add_request(nvr.synthetic_tokens(n, 1, i, 151936).tolist(), nvr.SamplingParams(temperature=0.0, max_tokens=40, ignore_eos=True))
while eng.step()["is_prefill"]: pass
for _ in range(5): eng.step()
nvr.synchronize(); t0 = time.perf_counter()
for _ in range(20): eng.step()
nvr.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
print(f"{preset} bs {B}, contexts {lo}..{hi} (sum {sum(lens)}, max {max(lens)}): {ms:7.3f} ms/step   NVR_ATTN_SHARE={os.environ.get('NVR_ATTN_SHARE', '1')}", flush=True)
if not os.environ.get("NVR_NO_EXIT"): os._exit(0)
