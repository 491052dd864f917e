"""Diagnostic (libnvr_occ.so): per-CU concurrency of the flash prefill launches INSIDE an engine prefill step (the runner's tile order);
the stamps are those of the last layer's launch."""
import ctypes as C, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("NVR_LIBNVR", os.path.join(ROOT, "nano-vllm-rs_amd", "libnvr_occ.so"))
sys.path.insert(0, ROOT)
import numpy as np, nvr_import
nvr = nvr_import.load()
mc = nvr.ModelConfig("qwen3-0.6b")
eng = nvr.LLMEngine(nvr.Config(max_num_seqs=32, max_num_batched_tokens=32768, max_model_len=1100, kvcache_block_size=256, num_kvcache_blocks=32 * 6), mc)
for rep in range(2):
    for i in range(32):
        eng.add_request(nvr.synthetic_tokens(1024, 1, i + 100 * rep, mc.c.vocab_size).tolist(), nvr.SamplingParams(temperature=0.0, max_tokens=1, ignore_eos=True))
    rec = eng.step(); assert rec["is_prefill"] and rec["num_seqs"] == 32
    nvr.synchronize(); eng.take_finished()
raw = C.CDLL(os.environ["NVR_LIBNVR"])
buf = (C.c_uint64 * (8192 * 4))(); assert raw.nvr_debug_flash_wg(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 4).astype(np.int64)
a = a[a[:, 0] > 0]
t0 = a[:, 0].min(); print(f"{len(a)} workgroups, launch span {(a[:, 1].max() - t0) / 100:.1f} us")
hw = a[:, 2]; xcc = a[:, 3] & 0xf
key = xcc * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 12) & 1) * 20 + ((hw >> 8) & 0xf)
cus = collections.defaultdict(list)
for k, s, e in zip(key, a[:, 0], a[:, 1]): cus[int(k)].append((int(s - t0), int(e - t0)))
conc = []
for k, iv in cus.items():
    ev = sorted([(s, 1) for s, e in iv] + [(e, -1) for s, e in iv])
    cur = 0; last = 0; area = 0; mx = 0; b = 0
    for t, d in ev:
        area += cur * (t - last); b += (t - last) if cur > 0 else 0; last = t; cur += d; mx = max(mx, cur)
    conc.append((area / max(b, 1), mx, len(iv), b / 100, area / 100))
c = np.asarray(conc)
print(f"{len(cus)} CUs; per CU: workgroups {c[:,2].mean():.1f} ({c[:,2].min():.0f}..{c[:,2].max():.0f}); concurrency while busy {c[:,0].mean():.2f}; busy {c[:,3].mean():.1f} us ({c[:,3].min():.1f}..{c[:,3].max():.1f}); workgroup-time per CU {c[:,4].mean():.1f} us ({c[:,4].min():.1f}..{c[:,4].max():.1f})")
del eng
