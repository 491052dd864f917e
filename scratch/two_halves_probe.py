"""Probe: do two half batches (2 x 16 sequences, each on its own engine / stream / host thread) finish decode steps faster than one batch of 32?  The attention
launch is an HBM stream, the GEMM / norm chain sits on launch floors: two independent pipelines could overlap one's stream with the other's chain."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nvr_import
nvr = nvr_import.load()
mc = nvr.ModelConfig("qwen3-0.6b")
P, STEPS = 1024, 60
def make(B, seed0):
    e = nvr.LLMEngine(nvr.Config(max_num_seqs=B, max_num_batched_tokens=32768, max_model_len=P + 128, kvcache_block_size=256, num_kvcache_blocks=B * 6), mc)
    for i in range(B):
        e.add_request(nvr.synthetic_tokens(P, 1, seed0 + i, 151936).tolist(), nvr.SamplingParams(temperature=0.0, max_tokens=STEPS + 20, ignore_eos=True))
    while e.step()["is_prefill"]: pass
    for _ in range(5): e.step()
    return e
def run(engs):
    nvr.synchronize()
    def drive(e):
        for _ in range(STEPS): e.step()
    ths = [threading.Thread(target=drive, args=(e,)) for e in engs]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    nvr.synchronize()
    return (time.perf_counter() - t0) / STEPS * 1e3
for label, engs in (("one engine, 32 sequences", lambda: [make(32, 0)]), ("one engine, 16 sequences (alone)", lambda: [make(16, 0)]),
                    ("two engines x 16, two host threads", lambda: [make(16, 0), make(16, 16)]),
                    ("four engines x 8, four host threads", lambda: [make(8, 8 * i) for i in range(4)])):
    es = engs()
    print(f"{label:40s} {run(es):.3f} ms per step of {len(es)} x {32 // len(es) if len(es) > 1 else (32 if '32' in label else 16)} tokens", flush=True)
    del es
os._exit(0)
