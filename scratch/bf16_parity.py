"""bfloat16 build of the engine (Config.dtype = "bfloat16") against the bf16-faithful oracle: small model end to end, then Qwen3-0.6B
4 x 256-token prompts + 3 decode steps (256^2 GEMMs, flash prefill, weight-streaming decode GEMMs, paged attention, LM head)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nvr_import
import oracle
from oracle import engine_oracle as eo, model_oracle as mo
nvr = nvr_import.load()
def run(mcfg, pm, ecfg, prompts, max_tokens, dtype):
    eo.reset_sequence_counter(); nvr.lib().nvr_seq_reset_id_counter()
    o = mo.OracleEngine(mcfg, eo.Config(**{k: v for k, v in ecfg.items() if k != "skip_block_size_check"}), fp16=(dtype == "float16"), bf16=(dtype == "bfloat16"), max_pos=ecfg["max_model_len"])
    p = nvr.LLMEngine(nvr.Config(dtype=dtype, **ecfg), pm)
    for pr in prompts:
        sp = dict(temperature=0.0, max_tokens=max_tokens, ignore_eos=True)
        o.add_request(pr, eo.SamplingParams(**sp)); p.add_request(pr, nvr.SamplingParams(**sp))
    worst, mism, steps = 0.0, 0, 0
    while not p.is_finished():
        rec = p.step(); lg = p.model_runner.logits(rec["num_seqs"])
        orec = o.step(forced_tokens=rec["tokens"])
        err = float(np.abs(lg - orec["logits"]).max()); worst = max(worst, err)
        srt = np.sort(orec["logits"], axis=1); margin = srt[:, -1] - srt[:, -2]
        for i, (a, b) in enumerate(zip(rec["tokens"], orec["tokens"])):
            if a != b: mism += 1; print("   step", steps, "row", i, "token", a, "vs", b, "margin", margin[i])
        steps += 1
    print(f"{dtype}: {steps} steps, max |dlogit| {worst:.5f}, logit std {orec['logits'].std():.3f}, token mismatches {mism}", flush=True)
m = mo.small(seed=3)
pm = nvr.ModelConfig(vocab_size=m.vocab_size, hidden_size=m.hidden_size, intermediate_size=m.intermediate_size, num_hidden_layers=m.num_hidden_layers,
                     num_attention_heads=m.num_attention_heads, num_key_value_heads=m.num_key_value_heads, head_dim=m.head_dim,
                     max_position_embeddings=m.max_position_embeddings, rms_norm_eps=m.rms_norm_eps, rope_theta=m.rope_theta,
                     tie_word_embeddings=m.tie_word_embeddings, init_std=m.init_std, seed=m.seed)
ecfg = dict(max_num_seqs=4, max_num_batched_tokens=256, max_model_len=128, kvcache_block_size=16, num_kvcache_blocks=32, skip_block_size_check=1)
prompts = [oracle.fill_tokens(n, 2, i, m.vocab_size).tolist() for i, n in enumerate([7, 30, 17])]
for dt in ("float16", "bfloat16"):
    run(m, pm, ecfg, prompts, 8, dt)
V = 151936
ecfg = dict(max_num_seqs=4, max_num_batched_tokens=1024, max_model_len=272, kvcache_block_size=256, num_kvcache_blocks=10)
prompts = [nvr.synthetic_tokens(256, 1, i, V).tolist() for i in range(4)]
for dt in ("float16", "bfloat16"):
    run(mo.qwen3_0_6b(), nvr.ModelConfig("qwen3-0.6b"), ecfg, prompts, 4, dt)
