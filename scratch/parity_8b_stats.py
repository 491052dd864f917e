"""Qwen3-8B full depth, 2 x 256-token prefill: product logits vs the fp16-faithful oracle vs the f32 oracle — how large is the
inherent fp16-pipeline noise at this width / depth (profiles/r03_parity_8b_stats.txt)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nvr_import
import oracle
from oracle import engine_oracle as eo, model_oracle as mo
nvr = nvr_import.load()
V = 151936
L = int(os.environ.get("LAYERS", "36"))
ecfg = dict(max_num_seqs=2, max_num_batched_tokens=512, max_model_len=272, kvcache_block_size=256, num_kvcache_blocks=6)
prompts = [nvr.synthetic_tokens(256, 1, i, V).tolist() for i in range(2)]
sp = dict(temperature=0.0, max_tokens=2, ignore_eos=True)
mcfg = mo.qwen3_8b(); mcfg.num_hidden_layers = L
pm = nvr.ModelConfig("qwen3-8b", num_hidden_layers=L)
nvr.lib().nvr_seq_reset_id_counter()
p = nvr.LLMEngine(nvr.Config(**ecfg), pm)
for pr in prompts: p.add_request(pr, nvr.SamplingParams(**sp))
rec = p.step(); lp = p.model_runner.logits(2).copy()
res = {}
for name, fp16 in (("fp16", True), ("f32", False)):
    t0 = time.time(); eo.reset_sequence_counter()
    o = mo.OracleEngine(mcfg, eo.Config(**ecfg), fp16=fp16, max_pos=272, compact=fp16)
    for pr in prompts: o.add_request(pr, eo.SamplingParams(**sp))
    res[name] = o.step()["logits"]; print(name, "oracle", round(time.time() - t0, 1), "s", flush=True)
    del o
print(f"layers {L}: logits std {lp.std():.4f} max|l| {np.abs(lp).max():.3f}")
for a, b, n in ((lp, res["fp16"], "product - oracle_fp16"), (lp, res["f32"], "product - oracle_f32"), (res["fp16"], res["f32"], "oracle_fp16 - oracle_f32")):
    d = np.abs(a - b)
    print(f"{n:28s}: max {d.max():.5f}  rms {np.sqrt((d ** 2).mean()):.6f}  p99.99 {np.quantile(d, 0.9999):.5f}  argmax equal {np.array_equal(a.argmax(1), b.argmax(1))}")
srt = np.sort(res["fp16"], axis=1); print("oracle top1-top2 margins", srt[:, -1] - srt[:, -2])
sys.stdout.flush(); os._exit(0)
