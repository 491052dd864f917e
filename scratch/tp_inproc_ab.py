"""tp in-process tensor-parallel ranks (threads, one GPU, the peer-to-peer kernels with a REAL peer on the same device) decoding the headline
batch: ms per step of the slowest rank.  The ranks share one GPU, so the absolute number means nothing; two builds / switches of the exchange
path compared on the same box do (NVR_TP_FUSED=0/1, NVR_DBG).   usage: python3 scratch/tp_inproc_ab.py <tp> [model] [steps]"""
import os, sys, time, threading
NOCOMM = os.environ.get("NOCOMM") == "1"          # the same two ranks WITHOUT their exchanges (NVR_TP_NO_COMM): what the protocol adds = the difference
if NOCOMM: os.environ["NVR_TP_NO_COMM"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nvr_import
nvr = nvr_import.load()
tp = int(sys.argv[1]); model = sys.argv[2] if len(sys.argv) > 2 else "qwen3-0.6b"; steps = int(sys.argv[3]) if len(sys.argv) > 3 else 48
P = 1024 if model == "qwen3-0.6b" else 2048
mc = nvr.ModelConfig(model)
group = None if NOCOMM else nvr.LocalGroup(tp, p2p=True)
engines = []
for r in range(tp):
    e = nvr.LLMEngine(nvr.Config(max_num_seqs=32, max_num_batched_tokens=32768, max_model_len=P + 200, kvcache_block_size=256, num_kvcache_blocks=32 * (P // 256 + 2),
                                 tensor_parallel_size=tp, tensor_parallel_rank=r, async_decode=int(os.environ.get("ASYNC", "1"))), mc)
    if group is not None: group.attach(e.model_runner)
    engines.append(e)
for e in engines:
    nvr.lib().nvr_seq_reset_id_counter()
    for i in range(32):
        e.add_request(nvr.synthetic_tokens(P, 1, i, 151936).tolist(), nvr.SamplingParams(temperature=0.0, max_tokens=steps + 40, ignore_eos=True))
times = [0.0] * tp; toks = [[] for _ in range(tp)]; errs = []
bar = threading.Barrier(tp)
def drive(r):
    try:
        e = engines[r]
        while e.step()["is_prefill"]: pass
        for _ in range(8): e.step()
        bar.wait(); t0 = time.perf_counter()
        for _ in range(steps): toks[r].append(e.step()["tokens"])
        nvr.synchronize(); times[r] = time.perf_counter() - t0
    except BaseException as ex: errs.append((r, ex))
import faulthandler; faulthandler.dump_traceback_later(80, exit=True)
th = [threading.Thread(target=drive, args=(r,)) for r in range(tp)]
[t.start() for t in th]; [t.join(600) for t in th]
assert not errs, errs
assert NOCOMM or all(t == toks[0] for t in toks), "ranks disagree"
import zlib
print(f"{model} tp={tp} in-process: {max(times) / steps * 1e3:.3f} ms/step (slowest rank), tokens crc {zlib.crc32(str(toks[0]).encode()):08x}  {'exchanges skipped (NVR_TP_NO_COMM)' if NOCOMM else 'peer-to-peer exchanges'}", flush=True)
os._exit(0)
