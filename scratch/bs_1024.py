"""bench.side_decode at batches beyond the sweep (Qwen3-0.6B bs 768 / 1024 x 1024, Qwen3-8B bs 256 x 2048): ms per step, share of the step roofline, and the first
decode tokens of sequences 0..511 against the bs-512 run's (same prompts)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, nvr_import
nvr = nvr_import.load()
for preset, bsz, plen in [(a.split(":")[0], int(a.split(":")[1]), int(a.split(":")[2])) for a in sys.argv[1:]]:
    try:
        r = bench.side_decode(nvr, preset, batch=bsz, prompt_len=plen, steps=8, warmup=3)
        print(json.dumps({k: r.get(k) for k in ("workload", "ms_per_step", "tokens_per_s", "step_hbm_frac_per_gpu", "prefill_plus_first_decode_seconds", "init_seconds", "decode_token_crc", "error")}), flush=True)
    except Exception as ex:
        print(json.dumps({"model": preset, "batch": bsz, "error": str(ex)[:300]}), flush=True)
os._exit(0)
