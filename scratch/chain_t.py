import time, sys, os
sys.path.insert(0, os.getcwd())
import bench, nvr_import
nvr = nvr_import.load(); nvr.check(nvr.lib().nvr_device_set(0))
mc = nvr.ModelConfig("qwen3-0.6b")
bench.BATCH = 32
for eng in (False, True, False, True):
    t0 = time.time(); r = bench.time_decode_chain(nvr, mc, mlp_engine=eng); print(eng, round(r["us_per_layer"], 2), "us/layer", round(time.time() - t0, 1), "s wall", flush=True)
