import sys; sys.path.insert(0,'.')
import numpy as np, nvr_import, oracle
nvr = nvr_import.load(); nvr.check(nvr.lib().nvr_device_set(0))
rows, cols, gcols = 37, 96, 512
key = oracle.weight_key(7, 1234); sc = oracle.weight_scale(0.02)
d = nvr.DeviceBuffer(rows*cols*2)
nvr.check(nvr.lib().nvr_fill_weight(d.ptr, rows, cols, cols, gcols, 5, 64, key, sc, None))
got = d.to_numpy((rows, cols), np.float16)
ref = oracle.fill_weight(rows, cols, gcols, 5, 64, key, sc, True)
ref32 = oracle.fill_weight(rows, cols, gcols, 5, 64, key, sc, False)
bad = np.argwhere(got.astype(np.float32) != ref)
print("fill mismatches", len(bad))
for r,c in bad[:10]:
    print(r,c, got[r,c], ref[r,c], ref32[r,c], ref32[r,c].hex() if hasattr(ref32[r,c],'hex') else float(ref32[r,c]).hex(), hex(got.view(np.uint16)[r,c]), hex(np.float16(ref[r,c]).view(np.uint16)))
