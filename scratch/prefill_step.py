"""One engine prefill step of the bench workload (Qwen3-0.6B, 32 x 1024 tokens) + nothing else: the target of the counter passes of
tools/pmc_prefill.sh."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nvr_import
nvr = nvr_import.load()
mc = nvr.ModelConfig("qwen3-0.6b")
eng = nvr.LLMEngine(nvr.Config(max_num_seqs=32, max_num_batched_tokens=32768, max_model_len=1100, kvcache_block_size=256, num_kvcache_blocks=32 * 6), mc)
for rep in range(2):                      # the second prefill is the one evaluated (warm code objects)
    for i in range(32):
        eng.add_request(nvr.synthetic_tokens(1024, 1, i + 100 * rep, mc.c.vocab_size).tolist(), nvr.SamplingParams(temperature=0.0, max_tokens=1, ignore_eos=True))
    rec = eng.step(); assert rec["is_prefill"] and rec["num_seqs"] == 32
    nvr.synchronize(); eng.take_finished()
del eng
