"""tp in-process ranks (one GPU, one thread per rank, peer-to-peer kernels over plain pointers) decoding bs 32: run under rocprofv3
--kernel-trace --stats to read the one-shot collectives' kernel durations.  usage: tp_inproc_prof.py <tp> <model> [steps]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nvr_import
nvr = nvr_import.load()
tp = int(sys.argv[1]); model = sys.argv[2]; steps = int(sys.argv[3]) if len(sys.argv) > 3 else 24
P = 1024 if model == "qwen3-0.6b" else 2048
mc = nvr.ModelConfig(model)
group = nvr.LocalGroup(tp)
engines = []
for r in range(tp):
    e = nvr.LLMEngine(nvr.Config(max_num_seqs=32, max_num_batched_tokens=32768, max_model_len=P + 64, kvcache_block_size=256, num_kvcache_blocks=32 * (P // 256 + 2),
                                 tensor_parallel_size=tp, tensor_parallel_rank=r), mc)
    group.attach(e.model_runner)
    nvr.lib().nvr_seq_reset_id_counter()
    for i in range(32):
        e.add_request(nvr.synthetic_tokens(P, 1, i, 151936).tolist(), nvr.SamplingParams(temperature=0.0, max_tokens=steps + 2, ignore_eos=True))
    engines.append(e)
times = [0.0] * tp
def drive(r):
    e = engines[r]
    while e.step()["is_prefill"]:
        pass
    for _ in range(4): e.step()
    nvr.synchronize(); t0 = time.perf_counter()
    for _ in range(steps - 6): e.step()
    nvr.synchronize(); times[r] = (time.perf_counter() - t0) / (steps - 6)
ths = [threading.Thread(target=drive, args=(r,)) for r in range(tp)]
for t in ths: t.start()
for t in ths: t.join()
print(f"{model} tp={tp} in-process on one GPU: {max(times) * 1e3:.3f} ms/step (all ranks share the GPU)", flush=True)
engines.clear()
