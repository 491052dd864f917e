import sys; sys.path.insert(0,'.')
import numpy as np, nvr_import, oracle
nvr = nvr_import.load(); nvr.check(nvr.lib().nvr_device_set(0))
F16=np.float16
H,KVH,D,T=16,8,128,32
rng = np.random.default_rng(5)
NB, bs, max_pos = 6, 16, 200
QKV = (H + 2 * KVH) * D
qkvb = rng.standard_normal((T, QKV)).astype(np.float32).astype(F16); qkv = qkvb.astype(np.float32)
pos = rng.integers(0, max_pos, T).astype(np.int64)
slots = rng.permutation(NB * bs)[:T].astype(np.int32)
cos, sin = oracle.rope_table(D, max_pos, 1e6)
d_cos, d_sin = nvr.DeviceBuffer.from_numpy(cos), nvr.DeviceBuffer.from_numpy(sin)
d_qkv = nvr.DeviceBuffer.from_numpy(qkvb); d_pos=nvr.DeviceBuffer.from_numpy(pos); d_sl=nvr.DeviceBuffer.from_numpy(slots)
d_k, d_v = nvr.DeviceBuffer(NB * bs * KVH * D * 2), nvr.DeviceBuffer(NB * bs * KVH * D * 2)
nvr.check(nvr.lib().nvr_rope_store_kv(d_qkv.ptr, d_pos.ptr, d_sl.ptr, T, H, KVH, D, d_cos.ptr, d_sin.ptr, d_k.ptr, d_v.ptr, None))
got = d_qkv.to_numpy((T, QKV), F16).astype(np.float32)[:, :H*D].reshape(T,H,D)
x = qkv[:, :H * D].reshape(T, H, D)
q32 = oracle.rope_apply(x, pos, cos, sin); q = oracle.round_f16(q32)
bad = np.argwhere(got != q); print("mismatch", len(bad), "of", q.size)
for t,h,d in bad[:8]:
    j = d % 64; c=cos[pos[t], j]; s=sin[pos[t], j]; x1=x[t,h,j]; x2=x[t,h,j+64]
    print(t,h,d,"got",got[t,h,d],"ref",q[t,h,d],"f32",float(q32[t,h,d]).hex(),"x1",x1,"x2",x2,"c",float(c).hex(),"s",float(s).hex(),
      "np", float(np.float32(x1)*np.float32(c) - np.float32(x2)*np.float32(s)).hex() if d<64 else float(np.float32(x2)*np.float32(c)+np.float32(x1)*np.float32(s)).hex())
