import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import oracle
from oracle import model_oracle as mo, engine_oracle as eo
import nvr_import
nvr = nvr_import.load()
import test_engine_gpu as t
mcfg = mo.small(seed=9)
V = mcfg.vocab_size
ecfg = dict(max_num_seqs=12, max_num_batched_tokens=2048, max_model_len=512, kvcache_block_size=64, num_kvcache_blocks=60)
system = oracle.fill_tokens(150, 4, 7, V).tolist()
prompts = [system + oracle.fill_tokens(3 + 9 * i, 4, 100 + i, V).tolist() for i in range(9)]
p = nvr.LLMEngine(nvr.Config(skip_block_size_check=1, shared_prefix_min_seqs=4, **ecfg), t._model_cfgs(mcfg))
print("cfg field", p.config.c.shared_prefix_min_seqs)
for pr in prompts:
    p.add_request(pr, nvr.SamplingParams(temperature=0.0, max_tokens=5, ignore_eos=True))
for i in range(3):
    rec = p.step()
    print(rec["is_prefill"], rec["num_seqs"], p.model_runner.last_shared_prefix_len(), [s.block_table[:4] for s in p.last_batch()][:4])
