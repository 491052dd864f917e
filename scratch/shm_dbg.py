import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, nvr_import, oracle
from oracle import model_oracle as mo
nvr = nvr_import.load()
nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 300
mcfg = mo.small(seed=21, hidden_size=256, num_attention_heads=16, num_key_value_heads=8, head_dim=128, intermediate_size=512, num_hidden_layers=2)
V = mcfg.vocab_size
ecfg = dict(max_num_seqs=nseq, max_num_batched_tokens=1 << 17, max_model_len=320, kvcache_block_size=64, num_kvcache_blocks=nseq * 2 + 8)
system = oracle.fill_tokens(128, 4, 7, V).tolist()
prompts = [system + oracle.fill_tokens(3 + (7 * i) % 50, 4, 100 + i, V).tolist() for i in range(nseq)]
def model_cfgs(m):
    c = nvr.ModelConfig()
    for k, v in m.__dict__.items():
        if hasattr(c.c, k): setattr(c.c, k, type(getattr(c.c, k))(v))
    return c
from test_engine_gpu import _model_cfgs
def run(flag, ms=4):
    os.environ["NVR_ATTN_FUSED_MERGE"] = flag
    nvr.lib().nvr_seq_reset_id_counter()
    p = nvr.LLMEngine(nvr.Config(skip_block_size_check=1, shared_prefix_min_seqs=ms, async_decode=0, **ecfg), _model_cfgs(mcfg))
    os.environ.pop("NVR_ATTN_FUSED_MERGE", None)
    for pr in prompts: p.add_request(pr, nvr.SamplingParams(temperature=0.0, max_tokens=6, ignore_eos=True))
    out = []
    while not p.is_finished():
        rec = p.step()
        if not rec["is_prefill"]: out.append((rec["tokens"], p.model_runner.logits(rec["num_seqs"]).copy()))
    return out
a, b, a2, b2, pl = run("1"), run("0"), run("1"), run("0"), run("1", -1)
for i, ((ta, la), (tb, lb), (tc, lc), (td, ld), (te, le)) in enumerate(zip(a, b, a2, b2, pl)):
    d = np.abs(la - lb).max(axis=1); rows = np.nonzero(d)[0]
    d2 = np.abs(la - lc).max(axis=1)
    d3 = np.abs(lb - ld).max(axis=1)
    print(f"   unfused vs unfused again: {np.count_nonzero(d3)} rows; |fused - plain| max {np.abs(la - le).max():.2e}, |unfused - plain| max {np.abs(lb - le).max():.2e}; rows where they differ: fused-plain {np.abs(la - le).max(axis=1)[rows][:6]}, unfused-plain {np.abs(lb - le).max(axis=1)[rows][:6]}", flush=True)
    print(f"step {i}: fused vs unfused rows differing {len(rows)} (first {rows[:8]}), max {d.max():.2e}; fused vs fused again: {np.count_nonzero(d2)} rows, tokens equal {ta == tb}", flush=True)
os._exit(0)
