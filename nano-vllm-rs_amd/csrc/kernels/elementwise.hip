// elementwise.hip — the HBM-bound row kernels of the hot path for gfx950 (wave64):
// embedding gather (K1), RMSNorm and fused add+RMSNorm (K2, K11), RoPE + KV store (K5, K6),
// SiluAndMul (K13), last-token select (K15), greedy argmax (K17), synthetic weight fill.
// All activation traffic is 16 B per lane (8 fp16), one wave per row where a row reduction exists.
#include "kernels.h"
#include "device_utils.h"
#include "../common.h"

namespace nvr { namespace NVR_DT_NS {

#define LAUNCH_CHECK()                                                                                   \
    do {                                                                                                 \
        hipError_t _e = hipGetLastError();                                                               \
        if (_e != hipSuccess) return nvr::fail(NVR_ERR_HIP, "kernel launch failed: %s (%s:%d)",          \
                                               hipGetErrorString(_e), __FILE__, __LINE__);               \
    } while (0)

// ---------------------------------------------------------------- K1 embedding
// reference: VocabParallelEmbedding::forward, src/layers/embed_head.rs:77-97
__global__ void embedding_kernel(const int64_t *__restrict__ ids, const half_t *__restrict__ E, int Hd,
                                 half_t *__restrict__ out) {
    const int t = blockIdx.x;
    const half8_t *src = reinterpret_cast<const half8_t *>(E + (int64_t)ids[t] * Hd);
    half8_t *dst = reinterpret_cast<half8_t *>(out + (int64_t)t * Hd);
    for (int c = threadIdx.x; c < Hd / 8; c += blockDim.x) dst[c] = src[c];
}
int embedding(const int64_t *ids, int64_t T, const half_bits *E, int64_t Hd, half_bits *out, hipStream_t s) {
    if (Hd % 8) return nvr::fail(NVR_ERR_UNSUPPORTED, "embedding: hidden size %ld not a multiple of 8", (long)Hd);
    if (T == 0) return 0;
    int threads = Hd / 8 >= 128 ? 128 : 64;
    embedding_kernel<<<dim3((unsigned)T), dim3(threads), 0, s>>>(ids, (const half_t *)E, (int)Hd, (half_t *)out);
    LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------- K2 / K11 RMSNorm
// reference: RMSNorm::forward_simple, src/layers/layernorm.rs:58-75 — f32: rms = sqrt(mean(x^2)+eps),
// out = (x / rms) * w; fused variant: OptimizedRMSNorm::forward_with_residual, :170-176 —
// h <- fp16(h + y), out = rmsnorm(h).  One wave per row, 4 rows per workgroup.
template <bool ADD, int C>   // C = chunks of 512 elements per row kept in registers (0: re-read the row)
__global__ __launch_bounds__(256) void rmsnorm_kernel(half_t *__restrict__ h, const half_t *__restrict__ y,
                                                      const half_t *__restrict__ w, float eps, int T, int Hd,
                                                      half_t *__restrict__ out) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= T) return;
    half_t *hr = h + (int64_t)row * Hd;
    const half_t *yr = ADD ? y + (int64_t)row * Hd : nullptr;
    half_t *orow = out + (int64_t)row * Hd;
    float ss = 0.f;
    if (C > 0) {
        // single pass: the row (and the weight) stay in registers; all loads are issued before the reduction
        half8_t v[C > 0 ? C : 1], g[C > 0 ? C : 1], u[C > 0 ? C : 1];
#pragma unroll
        for (int i = 0; i < C; ++i) {
            const int c = lane * 8 + i * 512;
            if (c < Hd) {
                v[i] = *reinterpret_cast<const half8_t *>(hr + c);
                if (ADD) u[i] = *reinterpret_cast<const half8_t *>(yr + c);
                g[i] = *reinterpret_cast<const half8_t *>(w + c);
            }
        }
#pragma unroll
        for (int i = 0; i < C; ++i) {
            const int c = lane * 8 + i * 512;
            if (c < Hd) {
                if (ADD) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[i][j] = to_half_rn((float)v[i][j] + (float)u[i][j]);
                    *reinterpret_cast<half8_t *>(hr + c) = v[i];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) { float f = (float)v[i][j]; ss += f * f; }
            }
        }
        ss = wave_sum(ss);
        const float rms = sqrtf(ss / (float)Hd + eps);
#pragma unroll
        for (int i = 0; i < C; ++i) {
            const int c = lane * 8 + i * 512;
            if (c < Hd) {
                half8_t o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = to_half_rn(__fmul_rn(__fdiv_rn((float)v[i][j], rms), (float)g[i][j]));
                *reinterpret_cast<half8_t *>(orow + c) = o;
            }
        }
        return;
    }
    for (int c = lane * 8; c < Hd; c += 512) {
        half8_t v = *reinterpret_cast<const half8_t *>(hr + c);
        if (ADD) {
            half8_t u = *reinterpret_cast<const half8_t *>(yr + c);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = to_half_rn((float)v[j] + (float)u[j]);
            *reinterpret_cast<half8_t *>(hr + c) = v;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { float f = (float)v[j]; ss += f * f; }
    }
    ss = wave_sum(ss);
    const float rms = sqrtf(ss / (float)Hd + eps);
    for (int c = lane * 8; c < Hd; c += 512) {
        half8_t v = *reinterpret_cast<const half8_t *>(hr + c);   // own writes: same lane, L1/L2 hit
        half8_t g = *reinterpret_cast<const half8_t *>(w + c);
        half8_t o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = to_half_rn(__fmul_rn(__fdiv_rn((float)v[j], rms), (float)g[j]));
        *reinterpret_cast<half8_t *>(orow + c) = o;
    }
}
// decode-sized calls (T <= 64 rows): one row per workgroup, four waves per row — the launch is latency-bound and what counts
// is the arithmetic per wave between the loads and the stores (see add_rmsnorm_slabs_kernel below)
template <bool ADD, int C, int P>   // each thread owns one P-element piece of every (256*P)-element chunk, C chunks
__global__ __launch_bounds__(256) void rmsnorm_row4_kernel(half_t *__restrict__ h, const half_t *__restrict__ y,
                                                           const half_t *__restrict__ w, float eps, int Hd,
                                                           half_t *__restrict__ out) {
    typedef half_t hp_t __attribute__((ext_vector_type(P)));
    const int row = blockIdx.x, tid = threadIdx.x;
    half_t *hr = h + (int64_t)row * Hd;
    hp_t v[C], g[C];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < C; ++i) {
        const int c = tid * P + i * (256 * P);
        if (c < Hd) {
            v[i] = *reinterpret_cast<const hp_t *>(hr + c);
            g[i] = *reinterpret_cast<const hp_t *>(w + c);
            if (ADD) {
                const hp_t u = *reinterpret_cast<const hp_t *>(y + (int64_t)row * Hd + c);
#pragma unroll
                for (int j = 0; j < P; ++j) v[i][j] = to_half_rn((float)v[i][j] + (float)u[j]);
                *reinterpret_cast<hp_t *>(hr + c) = v[i];
            }
#pragma unroll
            for (int j = 0; j < P; ++j) { const float f = (float)v[i][j]; ss += f * f; }
        }
    }
    __shared__ float sm[4];
    ss = wave_sum(ss);
    if ((tid & 63) == 0) sm[tid >> 6] = ss;
    __syncthreads();
    const float rms = sqrtf((sm[0] + sm[1] + sm[2] + sm[3]) / (float)Hd + eps);
#pragma unroll
    for (int i = 0; i < C; ++i) {
        const int c = tid * P + i * (256 * P);
        if (c < Hd) {
            hp_t o;
#pragma unroll
            for (int j = 0; j < P; ++j) o[j] = to_half_rn(__fmul_rn(__fdiv_rn((float)v[i][j], rms), (float)g[i][j]));
            *reinterpret_cast<hp_t *>(out + (int64_t)row * Hd + c) = o;
        }
    }
}
// K1 + K2 of a decode-sized step in one launch: h[row] = E[ids[row]] (VocabParallelEmbedding::forward, embed_head.rs:77-97),
// out[row] = rmsnorm(h[row]) * w (the first layer's input norm) — same arithmetic as embedding followed by rmsnorm_row4_kernel
template <int C, int P>
__global__ __launch_bounds__(256) void embed_rmsnorm_kernel(const int64_t *__restrict__ ids, const half_t *__restrict__ E,
                                                            const half_t *__restrict__ w, float eps, int Hd,
                                                            half_t *__restrict__ h, half_t *__restrict__ out) {
    typedef half_t hp_t __attribute__((ext_vector_type(P)));
    const int row = blockIdx.x, tid = threadIdx.x;
    const half_t *src = E + ids[row] * (int64_t)Hd;
    hp_t v[C], g[C];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < C; ++i) {
        const int c = tid * P + i * (256 * P);
        if (c < Hd) {
            v[i] = *reinterpret_cast<const hp_t *>(src + c);
            g[i] = *reinterpret_cast<const hp_t *>(w + c);
            *reinterpret_cast<hp_t *>(h + (int64_t)row * Hd + c) = v[i];
#pragma unroll
            for (int j = 0; j < P; ++j) { const float f = (float)v[i][j]; ss += f * f; }
        }
    }
    __shared__ float sm[4];
    ss = wave_sum(ss);
    if ((tid & 63) == 0) sm[tid >> 6] = ss;
    __syncthreads();
    const float rms = sqrtf((sm[0] + sm[1] + sm[2] + sm[3]) / (float)Hd + eps);
#pragma unroll
    for (int i = 0; i < C; ++i) {
        const int c = tid * P + i * (256 * P);
        if (c < Hd) {
            hp_t o;
#pragma unroll
            for (int j = 0; j < P; ++j) o[j] = to_half_rn(__fmul_rn(__fdiv_rn((float)v[i][j], rms), (float)g[i][j]));
            *reinterpret_cast<hp_t *>(out + (int64_t)row * Hd + c) = o;
        }
    }
}
bool embedding_rmsnorm_ok(int64_t T, int64_t Hd) { return T >= 1 && T <= 64 && Hd % 8 == 0 && Hd <= 8192; }
int embedding_rmsnorm(const int64_t *ids, int64_t T, const half_bits *E, const half_bits *w, float eps, int64_t Hd, half_bits *h,
                      half_bits *out, hipStream_t s) {
    if (!embedding_rmsnorm_ok(T, Hd)) return nvr::fail(NVR_ERR_UNSUPPORTED, "embedding_rmsnorm: T=%ld (1..64), hidden size %ld", (long)T, (long)Hd);
    dim3 grid((unsigned)T), block(256);
    if (Hd <= 1024) embed_rmsnorm_kernel<1, 4><<<grid, block, 0, s>>>(ids, (const half_t *)E, (const half_t *)w, eps, (int)Hd, (half_t *)h, (half_t *)out);
    else if (Hd <= 2048) embed_rmsnorm_kernel<1, 8><<<grid, block, 0, s>>>(ids, (const half_t *)E, (const half_t *)w, eps, (int)Hd, (half_t *)h, (half_t *)out);
    else embed_rmsnorm_kernel<4, 8><<<grid, block, 0, s>>>(ids, (const half_t *)E, (const half_t *)w, eps, (int)Hd, (half_t *)h, (half_t *)out);
    LAUNCH_CHECK();
    return 0;
}

template <bool ADD>
static void launch_rmsnorm(half_t *h, const half_t *y, const half_t *w, float eps, int T, int Hd, half_t *out, hipStream_t s) {
    if (T <= 64 && Hd <= 8192) {
        dim3 grid((unsigned)T), block(256);
        if (Hd <= 1024) rmsnorm_row4_kernel<ADD, 1, 4><<<grid, block, 0, s>>>(h, y, w, eps, Hd, out);
        else if (Hd <= 2048) rmsnorm_row4_kernel<ADD, 1, 8><<<grid, block, 0, s>>>(h, y, w, eps, Hd, out);
        else rmsnorm_row4_kernel<ADD, 4, 8><<<grid, block, 0, s>>>(h, y, w, eps, Hd, out);
        return;
    }
    dim3 grid((unsigned)((T + 3) / 4)), block(256);
    if (Hd <= 1024) rmsnorm_kernel<ADD, 2><<<grid, block, 0, s>>>(h, y, w, eps, T, Hd, out);
    else if (Hd <= 4096) rmsnorm_kernel<ADD, 8><<<grid, block, 0, s>>>(h, y, w, eps, T, Hd, out);
    else rmsnorm_kernel<ADD, 0><<<grid, block, 0, s>>>(h, y, w, eps, T, Hd, out);
}
int rmsnorm(const half_bits *x, const half_bits *w, float eps, int64_t T, int64_t Hd, half_bits *out, hipStream_t s) {
    if (Hd % 8) return nvr::fail(NVR_ERR_UNSUPPORTED, "rmsnorm: hidden size %ld not a multiple of 8", (long)Hd);
    if (T == 0) return 0;
    launch_rmsnorm<false>((half_t *)x, nullptr, (const half_t *)w, eps, (int)T, (int)Hd, (half_t *)out, s);
    LAUNCH_CHECK();
    return 0;
}
int add_rmsnorm(half_bits *h, const half_bits *y, const half_bits *w, float eps, int64_t T, int64_t Hd,
                half_bits *out, hipStream_t s) {
    if (Hd % 8) return nvr::fail(NVR_ERR_UNSUPPORTED, "rmsnorm: hidden size %ld not a multiple of 8", (long)Hd);
    if (T == 0) return 0;
    launch_rmsnorm<true>((half_t *)h, (const half_t *)y, (const half_t *)w, eps, (int)T, (int)Hd, (half_t *)out, s);
    LAUNCH_CHECK();
    return 0;
}

// split-k consumer: y = fp16(slab0 + slab1 + ...) (f32, fixed order), h <- fp16(h + y), out = rmsnorm(h)*w.
// One row per workgroup (32 rows use 32 CUs), four waves per row: the launch is latency-bound (0.7 MB of data), what counts
// is the arithmetic per wave between the loads and the stores — a quarter of a one-wave-per-row kernel's (-40 us per
// Qwen3-0.6B decode step, profiles/r01_gemm_ablation.txt).  Each thread owns one P-element piece of every (256*P)-element
// chunk; the row's sum of squares crosses the waves through LDS.
template <int S, int WAVES, int C, int P = 8>   // P = 8 or 4 elements per thread and chunk
__global__ __launch_bounds__(WAVES * 64) void add_rmsnorm_slabs_kernel(half_t *__restrict__ h, const float *__restrict__ slabs,
                                                                            int64_t slab_stride, const half_t *__restrict__ w, float eps,
                                                                            int Hd, half_t *__restrict__ out) {
    typedef half_t hp_t __attribute__((ext_vector_type(P)));
    const int row = blockIdx.x, tid = threadIdx.x;
    half_t *hr = h + (int64_t)row * Hd;
    const float *sr = slabs + (int64_t)row * Hd;
    hp_t v[C], g[C];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < C; ++i) {
        const int c = tid * P + i * (WAVES * 64 * P);
        if (c < Hd) {
            v[i] = *reinterpret_cast<const hp_t *>(hr + c);
            g[i] = *reinterpret_cast<const hp_t *>(w + c);
            float4_t a[P / 4];
#pragma unroll
            for (int q = 0; q < P / 4; ++q) a[q] = *reinterpret_cast<const float4_t *>(sr + c + 4 * q);
#pragma unroll
            for (int z = 1; z < S; ++z)
#pragma unroll
                for (int q = 0; q < P / 4; ++q) a[q] += *reinterpret_cast<const float4_t *>(sr + z * slab_stride + c + 4 * q);
#pragma unroll
            for (int j = 0; j < P; ++j) {
                const float y = (float)to_half_rn(a[j / 4][j % 4]);
                v[i][j] = to_half_rn((float)v[i][j] + y);
                const float f = (float)v[i][j]; ss += f * f;
            }
            *reinterpret_cast<hp_t *>(hr + c) = v[i];
        }
    }
    __shared__ float sm[WAVES];
    ss = wave_sum(ss);
    if ((tid & 63) == 0) sm[tid >> 6] = ss;
    __syncthreads();
    float tot = sm[0];
#pragma unroll
    for (int k = 1; k < WAVES; ++k) tot += sm[k];
    const float rms = sqrtf(tot / (float)Hd + eps);
#pragma unroll
    for (int i = 0; i < C; ++i) {
        const int c = tid * P + i * (WAVES * 64 * P);
        if (c < Hd) {
            hp_t o;
#pragma unroll
            for (int j = 0; j < P; ++j) o[j] = to_half_rn(__fmul_rn(__fdiv_rn((float)v[i][j], rms), (float)g[i][j]));
            *reinterpret_cast<hp_t *>(out + (int64_t)row * Hd + c) = o;
        }
    }
}

int add_rmsnorm_slabs(half_bits *h, const float *slabs, int64_t S, const half_bits *w, float eps, int64_t T, int64_t Hd,
                      half_bits *out, hipStream_t s) {
    if (Hd % 8 || Hd > 8192) return nvr::fail(NVR_ERR_UNSUPPORTED, "add_rmsnorm_slabs: hidden size %ld must be a multiple of 8, <= 8192", (long)Hd);
    if (S != 2 && S != 4 && S != 8) return nvr::fail(NVR_ERR_UNSUPPORTED, "add_rmsnorm_slabs: S=%ld must be 2, 4 or 8", (long)S);
    if (T == 0) return 0;
#define NVR_SLABN(SS, CC, PP) add_rmsnorm_slabs_kernel<SS, 4, CC, PP><<<dim3((unsigned)T), dim3(256), 0, s>>>( \
        (half_t *)h, slabs, T * Hd, (const half_t *)w, eps, (int)Hd, (half_t *)out)
    if (Hd <= 1024) { if (S == 2) NVR_SLABN(2, 1, 4); else if (S == 4) NVR_SLABN(4, 1, 4); else NVR_SLABN(8, 1, 4); }
    else if (Hd <= 2048) { if (S == 2) NVR_SLABN(2, 1, 8); else if (S == 4) NVR_SLABN(4, 1, 8); else NVR_SLABN(8, 1, 8); }
    else { if (S == 2) NVR_SLABN(2, 4, 8); else if (S == 4) NVR_SLABN(4, 4, 8); else NVR_SLABN(8, 4, 8); }
#undef NVR_SLABN
    LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------- K13 SiluAndMul
// reference: SiluAndMul::forward, src/layers/activation.rs:46-63; silu(x) = x*sigmoid(x), :12-15
__global__ void silu_mul_kernel(const half_t *__restrict__ x, int I, half_t *__restrict__ out, int64_t total8) {
    const int per_row = I / 8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total8; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = i / per_row; const int c = (int)(i % per_row) * 8;
        half8_t g = *reinterpret_cast<const half8_t *>(x + t * 2 * I + c);
        half8_t u = *reinterpret_cast<const half8_t *>(x + t * 2 * I + I + c);
        half8_t o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float gf = (float)g[j];
            float sg = sigmoid_fast(gf);
            o[j] = to_half_rn(__fmul_rn(__fmul_rn(gf, sg), (float)u[j]));
        }
        *reinterpret_cast<half8_t *>(out + t * I + c) = o;
    }
}
int silu_and_mul(const half_bits *x, int64_t T, int64_t I, half_bits *out, hipStream_t s) {
    if (I % 8) return nvr::fail(NVR_ERR_UNSUPPORTED, "silu_and_mul: intermediate size %ld not a multiple of 8", (long)I);
    int64_t total8 = T * (I / 8);
    if (total8 == 0) return 0;
    int64_t blocks = (total8 + 255) / 256; if (blocks > 4096) blocks = 4096;
    silu_mul_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>((const half_t *)x, (int)I, (half_t *)out, total8);
    LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------- the rest of src/layers/activation.rs: silu (:12-15), gelu (:20-22, candle's tanh
// form 0.5 x (1 + tanh(sqrt(2/pi) x (1 + 0.044715 x^2)))), relu (:25-27), GeluAndMul (:74-100) and the ActivationType dispatch (:111-163).
// One launch for every type: 8 elements per thread, f32 inside, one rounding to the 16-bit type at the store (A-22).  kind: 0 silu, 1 gelu, 2 relu
// ([T, cols] -> [T, cols]); 3 SiluAndMul, 4 GeluAndMul ([T, 2 I] -> [T, I]: act(x[:, :I]) * x[:, I:]).  SiluAndMul here is the exact-division form of the
// oracle (the K13 kernel above keeps its v_exp + v_rcp sigmoid: the GEMM epilogues share it bit for bit).
__device__ __forceinline__ float act_one(int kind, float g) {
    if (kind == 1 || kind == 4) { const float inner = 0.7978845608028654f * g * (1.0f + 0.044715f * g * g); return 0.5f * g * (1.0f + tanhf(inner)); }
    if (kind == 2) return fmaxf(g, 0.0f);
    return g * (1.0f / (1.0f + expf(-g)));
}
__global__ void activation_kernel(int kind, const half_t *__restrict__ x, int cols_in, int cols_out, half_t *__restrict__ out, int64_t total8) {
    const int per_row = cols_out / 8;
    const bool mul = kind >= 3;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total8; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = i / per_row; const int c = (int)(i % per_row) * 8;
        const half8_t g = *reinterpret_cast<const half8_t *>(x + t * cols_in + c);
        half8_t u = g;
        if (mul) u = *reinterpret_cast<const half8_t *>(x + t * cols_in + cols_out + c);
        half8_t o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float a = act_one(kind, (float)g[j]);
            o[j] = to_half_rn(mul ? a * (float)u[j] : a);
        }
        *reinterpret_cast<half8_t *>(out + t * cols_out + c) = o;
    }
}
int activation(int kind, const half_bits *x, int64_t T, int64_t cols, half_bits *out, hipStream_t s) {
    if (kind < 0 || kind > 4) return nvr::fail(NVR_ERR_INVALID_ARG, "activation: unknown type %d", kind);
    if (kind >= 3 && cols % 2) return nvr::fail(NVR_ERR_INVALID_ARG, "Input dimension must be even for %s, got %ld", kind == 3 ? "SiluAndMul" : "GeluAndMul", (long)cols);
    const int64_t cols_out = kind >= 3 ? cols / 2 : cols;
    if (cols_out % 8) return nvr::fail(NVR_ERR_UNSUPPORTED, "activation: %ld output columns are not a multiple of 8", (long)cols_out);
    const int64_t total8 = T * (cols_out / 8);
    if (total8 == 0) return 0;
    int64_t blocks = (total8 + 255) / 256; if (blocks > 4096) blocks = 4096;
    activation_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(kind, (const half_t *)x, (int)cols, (int)cols_out, (half_t *)out, total8);
    LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------- bias of a projection (Qwen3Config::use_bias, qwen3.rs:54-55)
// reference: candle_nn::Linear::forward behind src/layers/linear.rs:12-24 — the matmul's 16-bit result, then broadcast_add(bias): y <- 16bit(y + b)
// (A-30: two tensor ops, two roundings; row kernels carry no FMA contraction, so this is the oracle's add bit for bit)
__global__ void add_bias_kernel(half_t *__restrict__ y, const half_t *__restrict__ b, int N, int64_t total8) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total8; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % (N / 8)) * 8;
        half8_t v = *reinterpret_cast<const half8_t *>(y + i * 8);
        const half8_t bb = *reinterpret_cast<const half8_t *>(b + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = to_half_rn(__fadd_rn((float)v[e], (float)bb[e]));
        *reinterpret_cast<half8_t *>(y + i * 8) = v;
    }
}
int add_bias(half_bits *y, const half_bits *b, int64_t T, int64_t N, hipStream_t s) {
    if (N % 8) return nvr::fail(NVR_ERR_UNSUPPORTED, "add_bias: %ld output features are not a multiple of 8", (long)N);
    const int64_t total8 = T * (N / 8);
    if (total8 == 0) return 0;
    int64_t blocks = (total8 + 255) / 256; if (blocks > 4096) blocks = 4096;
    add_bias_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>((half_t *)y, (const half_t *)b, (int)N, total8);
    LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------- K15 last-token select
// reference: ParallelLMHead::extract_last_tokens, src/layers/embed_head.rs:272-289
__global__ void select_last_kernel(const half_t *__restrict__ h, const int32_t *__restrict__ cu, int Hd,
                                   half_t *__restrict__ out) {
    const int b = blockIdx.x;
    const int64_t row = (int64_t)cu[b + 1] - 1;
    const half8_t *src = reinterpret_cast<const half8_t *>(h + row * Hd);
    half8_t *dst = reinterpret_cast<half8_t *>(out + (int64_t)b * Hd);
    for (int c = threadIdx.x; c < Hd / 8; c += blockDim.x) dst[c] = src[c];
}
int select_last_tokens(const half_bits *h, const int32_t *cu, int64_t B, int64_t Hd, half_bits *out, hipStream_t s) {
    if (Hd % 8) return nvr::fail(NVR_ERR_UNSUPPORTED, "select_last: hidden size %ld not a multiple of 8", (long)Hd);
    if (B == 0) return 0;
    select_last_kernel<<<dim3((unsigned)B), dim3(128), 0, s>>>((const half_t *)h, cu, (int)Hd, (half_t *)out);
    LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------- K5 + K6 RoPE + KV store
// reference: apply_rotary_emb_single, src/layers/rotary_embedding.rs:23-48 (halves split at D/2,
// out1 = x1*c - x2*s, out2 = x2*c + x1*s, f32 math per SURVEY A-14); store_kv_cache,
// src/layers/attention.rs:150-174 with slot = block*bs + offset (A-6).  One workgroup per token:
// work items are 8-wide pair chunks of the q and k heads followed by 8-wide chunks of v.
__global__ __launch_bounds__(256) void rope_store_kernel(half_t *__restrict__ qkv, const int64_t *__restrict__ pos,
                                                         const int32_t *__restrict__ slots, int H, int KVH, int D,
                                                         const float *__restrict__ cos_t, const float *__restrict__ sin_t,
                                                         half_t *__restrict__ kc, half_t *__restrict__ vc) {
    const int t = blockIdx.x;
    const int half_d = D / 2, cpp = half_d / 8;            // chunks per head (pairs)
    const int n_rope = (H + KVH) * cpp, n_v = KVH * D / 8;
    const int64_t ld = (int64_t)(H + 2 * KVH) * D;
    half_t *row = qkv + t * ld;
    const int64_t p = pos[t];
    const int slot = slots ? slots[t] : -1;
    const float *c = cos_t + p * half_d, *sn = sin_t + p * half_d;
    for (int i = threadIdx.x; i < n_rope + n_v; i += blockDim.x) {
        if (i < n_rope) {
            const int head = i / cpp, j = (i % cpp) * 8;
            half_t *x = row + head * D;
            half8_t x1 = *reinterpret_cast<half8_t *>(x + j), x2 = *reinterpret_cast<half8_t *>(x + j + half_d);
            half8_t o1, o2;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float a = (float)x1[e], b = (float)x2[e], cs = c[j + e], si = sn[j + e];
                o1[e] = to_half_rn(mul_sub_unfused(a, cs, b, si));
                o2[e] = to_half_rn(mul_add_unfused(b, cs, a, si));
            }
            *reinterpret_cast<half8_t *>(x + j) = o1;
            *reinterpret_cast<half8_t *>(x + j + half_d) = o2;
            if (head >= H && slot >= 0) {
                half_t *dst = kc + ((int64_t)slot * KVH + (head - H)) * D;
                *reinterpret_cast<half8_t *>(dst + j) = o1;
                *reinterpret_cast<half8_t *>(dst + j + half_d) = o2;
            }
        } else if (slot >= 0) {
            const int e = (i - n_rope) * 8;
            *reinterpret_cast<half8_t *>(vc + (int64_t)slot * KVH * D + e) =
                *reinterpret_cast<const half8_t *>(row + (int64_t)(H + KVH) * D + e);
        }
    }
}
// The same with an RMSNorm over head_dim on every q and k head in front of the rotation (real Qwen3 checkpoints:
// self_attn.q_norm / k_norm; an extension of the reference graph, DESIGN A-27): n = fp16((x / rms) * w) with the arithmetic of
// rmsnorm_kernel (RMSNorm::forward_simple, layernorm.rs:58-75), then RoPE on n.  One wave per head, one lane per rotation pair.
__global__ __launch_bounds__(256) void qknorm_rope_store_kernel(half_t *__restrict__ qkv, const int64_t *__restrict__ pos,
                                                                const int32_t *__restrict__ slots, int H, int KVH, int D,
                                                                const float *__restrict__ cos_t, const float *__restrict__ sin_t,
                                                                const half_t *__restrict__ qw, const half_t *__restrict__ kw, float eps,
                                                                half_t *__restrict__ kc, half_t *__restrict__ vc) {
    const int t = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int half_d = D / 2;
    const int64_t ld = (int64_t)(H + 2 * KVH) * D;
    half_t *row = qkv + t * ld;
    const int64_t p = pos[t];
    const int slot = slots ? slots[t] : -1;
    for (int head = wave; head < H + KVH; head += 4) {
        half_t *x = row + head * D;
        const half_t *g = head < H ? qw : kw;
        float ss = 0.f;
        for (int j = lane; j < half_d; j += 64) { const float a = (float)x[j], b = (float)x[j + half_d]; ss += a * a + b * b; }
        ss = wave_sum(ss);
        const float rms = sqrtf(ss / (float)D + eps);
        for (int j = lane; j < half_d; j += 64) {
            const float a = (float)to_half_rn(__fmul_rn(__fdiv_rn((float)x[j], rms), (float)g[j]));
            const float b = (float)to_half_rn(__fmul_rn(__fdiv_rn((float)x[j + half_d], rms), (float)g[j + half_d]));
            const float cs = cos_t[p * half_d + j], si = sin_t[p * half_d + j];
            const half_t o1 = to_half_rn(mul_sub_unfused(a, cs, b, si));
            const half_t o2 = to_half_rn(mul_add_unfused(b, cs, a, si));
            x[j] = o1; x[j + half_d] = o2;
            if (head >= H && slot >= 0) {
                half_t *dst = kc + ((int64_t)slot * KVH + (head - H)) * D;
                dst[j] = o1; dst[j + half_d] = o2;
            }
        }
    }
    if (slot >= 0)
        for (int e = threadIdx.x * 8; e < KVH * D; e += 256 * 8)
            *reinterpret_cast<half8_t *>(vc + (int64_t)slot * KVH * D + e) = *reinterpret_cast<const half8_t *>(row + (int64_t)(H + KVH) * D + e);
}

int rope_store_kv(half_bits *qkv, const int64_t *positions, const int32_t *slots, int64_t T, int64_t H, int64_t KVH,
                  int64_t D, const float *cos_t, const float *sin_t, half_bits *k_cache, half_bits *v_cache,
                  hipStream_t s, const half_bits *q_norm_w, const half_bits *k_norm_w, float eps) {
    if (D % 16) return nvr::fail(NVR_ERR_UNSUPPORTED, "rope: head_dim %ld not a multiple of 16", (long)D);
    if (T == 0) return 0;
    if (q_norm_w || k_norm_w) {
        if (!q_norm_w || !k_norm_w) return nvr::fail(NVR_ERR_INVALID_ARG, "rope_store_kv: q and k norm weights come together");
        qknorm_rope_store_kernel<<<dim3((unsigned)T), dim3(256), 0, s>>>((half_t *)qkv, positions, slots, (int)H, (int)KVH, (int)D, cos_t, sin_t,
                                                                         (const half_t *)q_norm_w, (const half_t *)k_norm_w, eps,
                                                                         (half_t *)k_cache, (half_t *)v_cache);
        LAUNCH_CHECK();
        return 0;
    }
    rope_store_kernel<<<dim3((unsigned)T), dim3(256), 0, s>>>((half_t *)qkv, positions, slots, (int)H, (int)KVH, (int)D,
                                                              cos_t, sin_t, (half_t *)k_cache, (half_t *)v_cache);
    LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------- K17 greedy argmax
// reference: Sampler::greedy_sample, src/layers/sampler.rs:109-112; ties -> lowest index (SURVEY A-12).
// One 1024-thread workgroup per row; (value, index) pairs reduced through the wave then LDS.
__device__ __forceinline__ void amax_merge(float &bv, int &bi, float v, int i) {
    if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
}
__global__ __launch_bounds__(1024) void argmax_kernel(const float *__restrict__ logits, int V, int64_t *__restrict__ out_idx,
                                                      float *__restrict__ out_val, int64_t idx_offset) {
    const float *x = logits + (int64_t)blockIdx.x * V;
    float bv = -INFINITY; int bi = 0x7fffffff;
    const int V4 = ((reinterpret_cast<uintptr_t>(x) & 15) == 0) ? V / 4 : 0;
    // eight 16-byte pieces requested before the first comparison (as a plain loop hipcc keeps one in flight per thread: 37 round trips for a
    // 151 936-column row, 24 us per launch — r05, the float32 path's greedy step); the winner does not depend on the order of the comparisons
    for (int i0 = threadIdx.x; i0 < V4; i0 += 8 * blockDim.x) {
        float4_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = min(i0 + u * (int)blockDim.x, V4 - 1); v[u] = *reinterpret_cast<const float4_t *>(x + 4 * i); }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * (int)blockDim.x;
            if (i < V4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) amax_merge(bv, bi, v[u][j], 4 * i + j);
            }
        }
    }
    for (int i = V4 * 4 + threadIdx.x; i < V; i += blockDim.x) amax_merge(bv, bi, x[i], i);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(bv, o, 64); int oi = __shfl_xor(bi, o, 64);
        amax_merge(bv, bi, ov, oi);
    }
    __shared__ float sv[16]; __shared__ int si[16];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { sv[wave] = bv; si[wave] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w2 = 1; w2 < (int)(blockDim.x >> 6); ++w2) amax_merge(bv, bi, sv[w2], si[w2]);
        out_idx[blockIdx.x] = (int64_t)bi + idx_offset;
        if (out_val) out_val[blockIdx.x] = bv;
    }
}
int argmax(const float *logits, int64_t B, int64_t V, int64_t *out_idx, float *out_val, int64_t idx_offset,
           hipStream_t s) {
    if (B == 0) return 0;
    argmax_kernel<<<dim3((unsigned)B), dim3(1024), 0, s>>>(logits, (int)V, out_idx, out_val, idx_offset);
    LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------- vocab-shard concat (C3)
// reference: ParallelLMHead::gather_logits, src/layers/embed_head.rs:321-336 (Tensor::cat along the vocab axis).
// gathered [tp][B][Vl] (all-gather order) -> full [B][tp*Vl]
__global__ void concat_vocab_kernel(const float *__restrict__ g, int tp, int B, int Vl, float *__restrict__ full) {
    const int64_t total = (int64_t)tp * B * Vl;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % Vl); const int64_t rb = i / Vl; const int b = (int)(rb % B), r = (int)(rb / B);
        full[(int64_t)b * tp * Vl + (int64_t)r * Vl + j] = g[i];
    }
}
int concat_vocab_shards(const float *gathered, int64_t tp, int64_t B, int64_t Vl, float *full, hipStream_t s) {
    if (tp * B * Vl == 0) return 0;
    concat_vocab_kernel<<<dim3(2048), dim3(256), 0, s>>>(gathered, (int)tp, (int)B, (int)Vl, full);
    LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------- decode-side weight layout
// The weight-streaming decode kernels feed v_mfma_f32_16x16x32_f16 with 16-row x 32-k tiles of W, 16 B per lane.  From the
// row-major [N][K] parameter one wave-instruction gathers 16 pieces of 64 B, 2·K bytes apart; from the TILED copy
// [N/16][K/32][16 rows][32 k] the same tile is 1 KiB contiguous, consecutive k-steps are consecutive KiB, and a workgroup's
// whole stream is one contiguous range (decode GEMM chain 28.0 -> 25.6 us per layer, profiles/r02_tiled_weights.txt).
// mode 0: tile t holds rows 16t..16t+15.  mode 1 (qkv with the RoPE epilogue): tile (head, c) of a q / k head holds columns
// c*8..c*8+7 of the first half and of the second half of the head (rotation partners in one MFMA tile), v heads as mode 0 —
// the row order linear_skinny_kernel<EPI_ROPE> reads.
__global__ void retile_weight_kernel(const half_t *__restrict__ src, half_t *__restrict__ dst, int N, int K, int mode, int H, int KVH, int D) {
    const int64_t chunks = (int64_t)N * K / 8;
    const int ksteps = K / 32;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < chunks; c += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(c & 3), r = (int)((c >> 2) & 15);
        const int64_t t2 = c >> 6;
        const int ks = (int)(t2 % ksteps), tile = (int)(t2 / ksteps);
        int row = tile * 16 + r;
        if (mode == 1) {
            const int tph = D / 16, head = tile / tph, cc = tile % tph;
            if (head < H + KVH) row = head * D + (r < 8 ? cc * 8 + r : D / 2 + cc * 8 + (r - 8));
        }
        *reinterpret_cast<half8_t *>(dst + c * 8) = *reinterpret_cast<const half8_t *>(src + (int64_t)row * K + ks * 32 + q * 8);
    }
}
int retile_weight(const half_bits *src, half_bits *dst, int64_t N, int64_t K, int mode, int64_t H, int64_t KVH, int64_t D, hipStream_t s) {
    if (N % 16 || K % 32 || (mode == 1 && (D % 16 || N != (H + 2 * KVH) * D)))
        return nvr::fail(NVR_ERR_UNSUPPORTED, "retile_weight: N=%ld (multiple of 16), K=%ld (multiple of 32), mode %d", (long)N, (long)K, mode);
    if (N * K == 0) return 0;
    int64_t blocks = (N * K / 8 + 255) / 256; if (blocks > 8192) blocks = 8192;
    retile_weight_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>((const half_t *)src, (half_t *)dst, (int)N, (int)K, mode, (int)H, (int)KVH, (int)D);
    LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------- synthetic weights
__global__ void fill_weight_kernel(half_t *__restrict__ dst, int64_t rows, int64_t cols, int64_t ld, int64_t gcols,
                                   int64_t row0, int64_t col0, uint64_t key, float scale) {
    const int64_t total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols, c = i % cols;
        dst[r * ld + c] = to_half_rn(weight_value(key, (uint64_t)((row0 + r) * gcols + (col0 + c)), scale));
    }
}
int fill_weight(half_bits *dst, int64_t rows, int64_t cols, int64_t ld, int64_t global_cols, int64_t row0,
                int64_t col0, uint64_t key, float scale, hipStream_t s) {
    if (rows * cols == 0) return 0;
    int64_t blocks = (rows * cols + 255) / 256; if (blocks > 8192) blocks = 8192;
    fill_weight_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>((half_t *)dst, rows, cols, ld, global_cols, row0, col0,
                                                                   key, scale);
    LAUNCH_CHECK();
    return 0;
}
__global__ void fill_const_kernel(half_t *__restrict__ dst, int64_t n, float v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = (half_t)v;
}
int fill_const(half_bits *dst, int64_t n, float v, hipStream_t s) {
    if (n == 0) return 0;
    int64_t blocks = (n + 255) / 256; if (blocks > 4096) blocks = 4096;
    fill_const_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>((half_t *)dst, n, v);
    LAUNCH_CHECK();
    return 0;
}

}}  // namespace nvr::k / nvr::kb (NVR_DT_NS)
