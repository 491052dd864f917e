"""Decode steps of the float32 path (BASELINE configs[0]: Qwen3-0.6B, bs 1, 128-token prompt) for a kernel-time profile."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nvr_import
nvr = nvr_import.load()
mc = nvr.ModelConfig("qwen3-0.6b")
eng = nvr.LLMEngine(nvr.Config(max_num_seqs=1, max_num_batched_tokens=256, max_model_len=256, kvcache_block_size=256, num_kvcache_blocks=2, dtype="float32",
                               enforce_eager=int(os.environ.get("EAGER", "1"))), mc)
eng.add_request(nvr.synthetic_tokens(128, 1, 0, mc.c.vocab_size).tolist(), nvr.SamplingParams(temperature=0.0, max_tokens=40, ignore_eos=True))
while not eng.is_finished(): eng.step()
nvr.synchronize()
del eng
