// f32_path.hip — Config.dtype = "float32" (reference src/config.rs:51,113-116; the reference's own CPU path computes in f32): every op of the
// Qwen3 graph on 4-byte storage, as plain FMA kernels.  This is the REFERENCE-PRECISION path of the product — outputs comparable with the
// reference's f32 CPU path at 1e-3 instead of through 16-bit rounding — a parity vehicle first: FMA kernels for the decode-sized steps, the f32
// matrix cores (v_mfma_f32_32x32x2_f32, 1/16 of the fp16 rate) for the GEMMs of prefill-sized steps since r06.  Each kernel cites the reference op it restates, like its 16-bit twin:
//   embedding   VocabParallelEmbedding::forward, src/layers/embed_head.rs:77-97
//   rmsnorm     RMSNorm::forward_simple, src/layers/layernorm.rs:58-75; add_rmsnorm: forward_with_residual :170-176
//   linear      Linear::forward x·Wᵀ(+b), src/layers/linear.rs:12-24
//   rope_store  apply_rotary_emb_single, src/layers/rotary_embedding.rs:23-48 (+ q/k head norm, A-27); store_kv_cache, attention.rs:150-174
//   attention   compute_attention: softmax_f32(q·Kᵀ·D^-½ + mask)·V, src/layers/attention.rs:238-261 (varlen :177-208, paged :225-318)
//   silu_mul    SiluAndMul, src/layers/activation.rs:46-63;  select_last  ParallelLMHead::extract_last_tokens, embed_head.rs:272-289
#include "kernels.h"
#include "device_utils.h"
#include "../common.h"

namespace nvr { namespace kf {

#define F32_LAUNCH_CHECK(what)                                                                                          \
    do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return nvr::fail(NVR_ERR_HIP, what " launch failed: %s", hipGetErrorString(e_)); } while (0)

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// sum / max over a workgroup of NW waves (result in every thread); sm: NW floats of LDS
template <bool MAX>
__device__ __forceinline__ float block_reduce(float v, float *sm, int nw) {
    v = MAX ? wmax(v) : wsum(v);
    __syncthreads();                                      // sm may still be read from the previous reduction
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = sm[0];
    for (int i = 1; i < nw; ++i) r = MAX ? fmaxf(r, sm[i]) : r + sm[i];
    return r;
}

// ---------------------------------------------------------------- synthetic weights (unrounded: the f32 oracle's values)
__global__ void fill_weight_kernel(float *__restrict__ dst, int64_t rows, int64_t cols, int64_t ld, int64_t gcols, int64_t row0, int64_t col0,
                                   uint64_t key, float scale) {
    const int64_t total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols, c = i % cols;
        dst[r * ld + c] = weight_value(key, (uint64_t)((row0 + r) * gcols + (col0 + c)), scale);
    }
}
int fill_weight(float *dst, int64_t rows, int64_t cols, int64_t ld, int64_t gcols, int64_t row0, int64_t col0, uint64_t key, float scale, hipStream_t s) {
    if (rows * cols == 0) return 0;
    int64_t blocks = (rows * cols + 255) / 256; if (blocks > 8192) blocks = 8192;
    fill_weight_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(dst, rows, cols, ld, gcols, row0, col0, key, scale);
    F32_LAUNCH_CHECK("f32 fill_weight");
    return 0;
}
__global__ void fill_const_kernel(float *__restrict__ dst, int64_t n, float v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = v;
}
int fill_const(float *dst, int64_t n, float v, hipStream_t s) {
    if (n == 0) return 0;
    int64_t blocks = (n + 255) / 256; if (blocks > 4096) blocks = 4096;
    fill_const_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(dst, n, v);
    F32_LAUNCH_CHECK("f32 fill_const");
    return 0;
}

// ---------------------------------------------------------------- embedding, last-token select
__global__ void gather_rows_kernel(const int64_t *__restrict__ ids, const int32_t *__restrict__ cu, const float *__restrict__ src, int Hd,
                                   float *__restrict__ out) {
    const int64_t row = ids ? ids[blockIdx.x] : (int64_t)cu[blockIdx.x + 1] - 1;
    const float4 *s4 = reinterpret_cast<const float4 *>(src + row * Hd);
    float4 *d4 = reinterpret_cast<float4 *>(out + (int64_t)blockIdx.x * Hd);
    for (int c = threadIdx.x; c < Hd / 4; c += blockDim.x) d4[c] = s4[c];
}
int embedding(const int64_t *ids, int64_t T, const float *E, int64_t Hd, float *out, hipStream_t s) {
    if (Hd % 4) return nvr::fail(NVR_ERR_UNSUPPORTED, "f32 embedding: hidden size %ld is not a multiple of 4", (long)Hd);
    if (T == 0) return 0;
    gather_rows_kernel<<<dim3((unsigned)T), dim3(256), 0, s>>>(ids, nullptr, E, (int)Hd, out);
    F32_LAUNCH_CHECK("f32 embedding");
    return 0;
}
int select_last_tokens(const float *h, const int32_t *cu, int64_t B, int64_t Hd, float *out, hipStream_t s) {
    if (Hd % 4) return nvr::fail(NVR_ERR_UNSUPPORTED, "f32 select_last: hidden size %ld is not a multiple of 4", (long)Hd);
    if (B == 0) return 0;
    gather_rows_kernel<<<dim3((unsigned)B), dim3(256), 0, s>>>(nullptr, cu, h, (int)Hd, out);
    F32_LAUNCH_CHECK("f32 select_last");
    return 0;
}

// ---------------------------------------------------------------- RMSNorm (+ residual add): one workgroup per row
template <bool ADD>
__global__ __launch_bounds__(256) void rmsnorm_kernel(float *__restrict__ h, const float *__restrict__ y, const float *__restrict__ w, float eps, int Hd,
                                                      float *__restrict__ out) {
    __shared__ float sm[4];
    float *hr = h + (int64_t)blockIdx.x * Hd;
    const float *yr = ADD ? y + (int64_t)blockIdx.x * Hd : nullptr;
    float ss = 0.f;
    for (int c = threadIdx.x; c < Hd; c += 256) {
        float v = hr[c];
        if (ADD) { v = v + yr[c]; hr[c] = v; }                       // h <- h + y (qwen3.rs:382,389)
        ss = fmaf(v, v, ss);                                         // (pinned: add_norm_rows_to_lds below promises these bits)
    }
    ss = block_reduce<false>(ss, sm, 4);
    const float rms = sqrtf(ss / (float)Hd + eps);
    for (int c = threadIdx.x; c < Hd; c += 256) out[(int64_t)blockIdx.x * Hd + c] = hr[c] / rms * w[c];
}
int rmsnorm(const float *x, const float *w, float eps, int64_t T, int64_t Hd, float *out, hipStream_t s) {
    if (T == 0) return 0;
    rmsnorm_kernel<false><<<dim3((unsigned)T), dim3(256), 0, s>>>(const_cast<float *>(x), nullptr, w, eps, (int)Hd, out);
    F32_LAUNCH_CHECK("f32 rmsnorm");
    return 0;
}
// tensor-parallel ranks: y = ((p_0 + p_1) + p_2) + ... over the ranks' partial sums parts[r * stride + ...] in rank order, then h <- h + y and
// the norm, as rmsnorm_kernel<true>
__global__ __launch_bounds__(256) void sum_ranks_rmsnorm_kernel(float *__restrict__ h, const float *__restrict__ parts, int nranks, int64_t stride,
                                                                const float *__restrict__ w, float eps, int Hd, float *__restrict__ out) {
    __shared__ float sm[4];
    float *hr = h + (int64_t)blockIdx.x * Hd;
    const float *pr = parts + (int64_t)blockIdx.x * Hd;
    float ss = 0.f;
    for (int c = threadIdx.x; c < Hd; c += 256) {
        float y = pr[c];
        for (int r = 1; r < nranks; ++r) y = y + pr[(int64_t)r * stride + c];
        const float v = hr[c] + y;
        hr[c] = v;
        ss = fmaf(v, v, ss);
    }
    ss = block_reduce<false>(ss, sm, 4);
    const float rms = sqrtf(ss / (float)Hd + eps);
    for (int c = threadIdx.x; c < Hd; c += 256) out[(int64_t)blockIdx.x * Hd + c] = hr[c] / rms * w[c];
}
int sum_ranks_add_rmsnorm(float *h, const float *parts, int nranks, int64_t stride, const float *w, float eps, int64_t T, int64_t Hd, float *out, hipStream_t s) {
    if (T == 0) return 0;
    sum_ranks_rmsnorm_kernel<<<dim3((unsigned)T), dim3(256), 0, s>>>(h, parts, nranks, stride, w, eps, (int)Hd, out);
    F32_LAUNCH_CHECK("f32 sum_ranks_add_rmsnorm");
    return 0;
}
int add_rmsnorm(float *h, const float *y, const float *w, float eps, int64_t T, int64_t Hd, float *out, hipStream_t s) {
    if (T == 0) return 0;
    rmsnorm_kernel<true><<<dim3((unsigned)T), dim3(256), 0, s>>>(h, y, w, eps, (int)Hd, out);
    F32_LAUNCH_CHECK("f32 add_rmsnorm");
    return 0;
}

// ---------------------------------------------------------------- y[T,N] = x[T,K]·W[N,K]ᵀ (+ b[N]): 64 x 64 output tile, 16-deep k slices in LDS,
// 4 x 4 outputs per thread, k ascending (one f32 FMA chain per output)
constexpr int LB = 64, LK = 16;
__global__ __launch_bounds__(256) void linear_kernel(const float *__restrict__ x, int64_t ldx, const float *__restrict__ W, int T, int K, int N,
                                                     const float *__restrict__ bias, float *__restrict__ y) {
    __shared__ float xs[LK][LB + 1], ws[LK][LB + 1];
    const int t0 = blockIdx.y * LB, n0 = blockIdx.x * LB;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;             // outputs: tokens t0 + ty*4 .. +3, columns n0 + tx*4 .. +3
    float acc[4][4] = {};
    for (int k0 = 0; k0 < K; k0 += LK) {
        // 64 rows x 16 k of each operand: thread i loads row i / 4, k = (i % 4) * 4 .. + 3
        const int r = threadIdx.x >> 2, kk = (threadIdx.x & 3) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = k0 + kk + e;
            xs[kk + e][r] = (t0 + r < T && k < K) ? x[(int64_t)(t0 + r) * ldx + k] : 0.f;
            ws[kk + e][r] = (n0 + r < N && k < K) ? W[(int64_t)(n0 + r) * K + k] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < LK; ++k) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = xs[k][ty * 4 + i]; b[i] = ws[k][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = t0 + ty * 4 + i, n = n0 + tx * 4 + j;
            if (t < T && n < N) y[(int64_t)t * N + n] = bias ? acc[i][j] + bias[n] : acc[i][j];
        }
}
// ---------------------------------------------------------------- the same GEMM on the matrix cores (r06): prefill-sized steps
// (T > 8) of the float32 path ran the FMA kernel above at ~4 TFLOP/s (4.6 k prompt tokens/s on Qwen3-0.6B: 200 x under the fp16 build).
// v_mfma_f32_32x32x2_f32: A = 32 W rows x 2 k, B = 2 k x 32 tokens, 16 f32 accumulators per lane; workgroup = 4 waves = 64 W rows x 64 tokens, BK = 32:
// both operand tiles go global -> registers -> LDS as 16-byte pieces (row stride 36 floats: the 8-row groups of a ds_read_b128 land on different
// banks), two LDS buffers, the next K-tile's global loads in flight under the MFMAs.  A lane owns ONE row of each operand (row = lane % 32) and the k
// range [16 h, 16 h + 16) of the K-tile (h = lane / 32): four 16-byte LDS reads per operand give it the sixteen (a, b) pairs of sixteen MFMA steps — the
// two halves of a wave cover k and k + 16 in the same instruction.  Sums are f32 throughout, in another order than the FMA chain's (the f32 oracle's
// tolerance, 2e-4, is for exactly that); bias as before.  C layout: lane (h, token t = lane % 32) holds features 8 b + 4 h + e (b = reg / 4, e = reg % 4).
constexpr int MB = 64, MKT = 32, MLD = MKT + 4;
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void linear_mfma_kernel(const float *__restrict__ x, int64_t ldx, const float *__restrict__ W, int T, int K, int N,
                                                          const float *__restrict__ bias, float *__restrict__ y) {
    __shared__ __attribute__((aligned(16))) float as[2][MB][MLD], bs[2][MB][MLD];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l32 = lane & 31, h = lane >> 5;
    const int wn = wave >> 1, wm = wave & 1;                              // this wave: W rows n0 + 32 wn .., tokens t0 + 32 wm ..
    const int t0 = blockIdx.y * MB, n0 = blockIdx.x * MB;
    // staging: thread -> rows r and r + 32 of each tile, 16-byte piece c of the 128-byte K-tile row
    const int r = tid >> 3, c = tid & 7;
    const float *wsrc[2], *xsrc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        wsrc[i] = W + (int64_t)min(n0 + r + 32 * i, N - 1) * K + c * 4;
        xsrc[i] = x + (int64_t)min(t0 + r + 32 * i, T - 1) * ldx + c * 4;
    }
    float4 wv[2], xv[2];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) { wv[i] = *reinterpret_cast<const float4 *>(wsrc[i] + k0); xv[i] = *reinterpret_cast<const float4 *>(xsrc[i] + k0); }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<float4 *>(&as[buf][r + 32 * i][c * 4]) = wv[i];
            *reinterpret_cast<float4 *>(&bs[buf][r + 32 * i][c * 4]) = xv[i];
        }
    };
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int KT = K / MKT;
    fetch(0);
    stash(0);
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < KT) fetch((kt + 1) * MKT);
        float4 a4[4], b4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a4[j] = *reinterpret_cast<const float4 *>(&as[cur][wn * 32 + l32][h * 16 + j * 4]);
            b4[j] = *reinterpret_cast<const float4 *>(&bs[cur][wm * 32 + l32][h * 16 + j * 4]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j].x, b4[j].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j].y, b4[j].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j].z, b4[j].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j].w, b4[j].w, acc, 0, 0, 0);
        }
        if (kt + 1 < KT) stash(cur ^ 1);                                  // (the other buffer: last read in step kt - 1, before the barrier that ended it)
        __syncthreads();
    }
    const int t = t0 + wm * 32 + l32;
    if (t < T) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int n = n0 + wn * 32 + 8 * b + 4 * h;
            if (n + 3 < N) {
                float4 o = {acc[4 * b], acc[4 * b + 1], acc[4 * b + 2], acc[4 * b + 3]};
                if (bias) { o.x += bias[n]; o.y += bias[n + 1]; o.z += bias[n + 2]; o.w += bias[n + 3]; }
                *reinterpret_cast<float4 *>(y + (int64_t)t * N + n) = o;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (n + e < N) y[(int64_t)t * N + n + e] = bias ? acc[4 * b + e] + bias[n + e] : acc[4 * b + e];
            }
        }
    }
}
// h + y and its RMSNorm for the T <= TT rows of a decode-sized step, INTO LDS (xs [TT][K]) — what rmsnorm_kernel<true> writes to memory, with its
// arithmetic (a thread's columns c = tid, tid + 256, ..; fma chain of squares; wave sums, then the four waves in order; v / rms * w): the same bits.
// Every workgroup of a consumer GEMV does this for itself (K floats per row: nothing beside the weight rows it streams), so the add + norm launch in
// front of the GEMV disappears; h_out (given to ONE workgroup) receives the new residual stream — a second buffer, the others are still reading h_in.
template <int TT>
__device__ __forceinline__ void add_norm_rows_to_lds(const float *__restrict__ h_in, const float *__restrict__ y, const float *__restrict__ nw, float eps, int T,
                                                     int K, float *__restrict__ h_out, float *xs, float *sm) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    float ss[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) ss[t] = 0.f;
    for (int c = tid; c < K; c += 256) {
        float hv[TT], yv[TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) { const int64_t o = (int64_t)(t < T ? t : 0) * K + c; hv[t] = h_in[o]; yv[t] = y[o]; }
#pragma unroll
        for (int t = 0; t < TT; ++t) { const float v = hv[t] + yv[t]; xs[t * K + c] = v; ss[t] = fmaf(v, v, ss[t]); }
    }
#pragma unroll
    for (int t = 0; t < TT; ++t) { const float v = wsum(ss[t]); if (lane == 0) sm[t * 4 + wave] = v; }
    __syncthreads();
    float rms[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        float r = sm[t * 4];
        for (int i = 1; i < 4; ++i) r = r + sm[t * 4 + i];
        rms[t] = sqrtf(r / (float)K + eps);
    }
    for (int c = tid; c < K; c += 256) {
        const float w = nw[c];
#pragma unroll
        for (int t = 0; t < TT; ++t)
            if (t < T) {
                const float v = xs[t * K + c];
                if (h_out) h_out[(int64_t)t * K + c] = v;
                xs[t * K + c] = v / rms[t] * w;
            }
    }
    __syncthreads();
}
// decode-sized steps (T <= 8 rows): one wave per output column, the lanes stride over k with 16-byte loads (the weight row is read once, coalesced:
// this is the f32 path's memory-bound regime), the T partial sums meet by a wave reduction
template <int TT>
__global__ __launch_bounds__(256) void gemv_kernel(const float *__restrict__ x, int64_t ldx, const float *__restrict__ W, int T, int K, int N,
                                                   const float *__restrict__ bias, float *__restrict__ y) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    const float4 *w4 = reinterpret_cast<const float4 *>(W + (int64_t)n * K);
    float acc[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) acc[t] = 0.f;
    // four 16-byte pieces of the weight row (and of every activation row) requested before the first FMA — a plain loop waits for each piece's own
    // round trip (r04: 7.3 us per launch at K = 1024); the FMA chain of an output keeps its order: the same bits
    const int n4 = K / 4;
    for (int k0 = lane; k0 < n4; k0 += 256) {
        float4 w[4], a[TT][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k4 = min(k0 + 64 * u, n4 - 1);
            w[u] = w4[k4];
#pragma unroll
            for (int t = 0; t < TT; ++t) a[t][u] = reinterpret_cast<const float4 *>(x + (int64_t)(t < T ? t : 0) * ldx)[k4];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (k0 + 64 * u < n4) {
#pragma unroll
                for (int t = 0; t < TT; ++t)
                    if (t < T) acc[t] = fmaf(a[t][u].x, w[u].x, fmaf(a[t][u].y, w[u].y, fmaf(a[t][u].z, w[u].z, fmaf(a[t][u].w, w[u].w, acc[t]))));
            }
    }
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        const float v = wsum(acc[t]);
        if (lane == 0 && t < T) y[(int64_t)t * N + n] = bias ? v + bias[n] : v;
    }
}
__device__ __forceinline__ float silu_mul_one(float g, float u) { return g / (1.0f + expf(-g)) * u; }          // SiluAndMul, activation.rs:46-63
// gate_up projection + SiluAndMul of a decode-sized step in one launch (activation.rs:46-63 behind linear.rs:437-439): wave n forms gate column n and
// up column I + n of every row with gemv_kernel's loads and FMA chains, then act = silu(g) * u as silu_mul_kernel writes it — the same bits as the two
// launches (bias: added to g and u first, like Linear::forward)
// NORM: x = RMSNorm(h_in + y) formed by the workgroup itself (add_norm_rows_to_lds); workgroup 0 writes the new residual stream to h_out
template <int TT, bool NORM = false>
__global__ __launch_bounds__(256) void gemv_silu_kernel(const float *__restrict__ x, int64_t ldx, const float *__restrict__ W, int T, int K, int I,
                                                        const float *__restrict__ bias, float *__restrict__ act, const float *__restrict__ h_in = nullptr,
                                                        const float *__restrict__ y = nullptr, const float *__restrict__ nw = nullptr, float eps = 0.f,
                                                        float *__restrict__ h_out = nullptr) {
    extern __shared__ float xs[];                                       // NORM: [TT][K] normalised rows, then 4 * TT of reduction scratch
    const int n_ = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int n = n_ < I ? n_ : I - 1;                                  // (a wave past the end still helps with the norm)
    const float4 *g4 = reinterpret_cast<const float4 *>(W + (int64_t)n * K), *u4 = reinterpret_cast<const float4 *>(W + (int64_t)(I + n) * K);
    const int n4 = K / 4;
    float4 wg0[4], wu0[4];                                              // NORM: the first weight pieces are requested in front of the norm (they depend on nothing)
    if (NORM) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int k4 = min(lane + 64 * u, n4 - 1); wg0[u] = g4[k4]; wu0[u] = u4[k4]; }
        add_norm_rows_to_lds<TT>(h_in, y, nw, eps, T, K, blockIdx.x == 0 ? h_out : nullptr, xs, xs + TT * K); x = xs; ldx = K;
    }
    if (n_ >= I) return;
    float ag[TT], au[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) ag[t] = au[t] = 0.f;
    for (int k0 = lane; k0 < n4; k0 += 256) {
        float4 wg[4], wu[4], a[TT][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k4 = min(k0 + 64 * u, n4 - 1);
            if (NORM && k0 == lane) { wg[u] = wg0[u]; wu[u] = wu0[u]; } else { wg[u] = g4[k4]; wu[u] = u4[k4]; }
#pragma unroll
            for (int t = 0; t < TT; ++t) a[t][u] = reinterpret_cast<const float4 *>(x + (int64_t)(t < T ? t : 0) * ldx)[k4];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (k0 + 64 * u < n4) {
#pragma unroll
                for (int t = 0; t < TT; ++t)
                    if (t < T) {
                        ag[t] = fmaf(a[t][u].x, wg[u].x, fmaf(a[t][u].y, wg[u].y, fmaf(a[t][u].z, wg[u].z, fmaf(a[t][u].w, wg[u].w, ag[t]))));
                        au[t] = fmaf(a[t][u].x, wu[u].x, fmaf(a[t][u].y, wu[u].y, fmaf(a[t][u].z, wu[u].z, fmaf(a[t][u].w, wu[u].w, au[t]))));
                    }
            }
    }
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        float g = wsum(ag[t]), u = wsum(au[t]);
        if (lane == 0 && t < T) {
            if (bias) { g = g + bias[n]; u = u + bias[I + n]; }
            act[(int64_t)t * I + n] = silu_mul_one(g, u);
        }
    }
}
bool linear_silu_ok(int64_t T, int64_t K, int64_t ldx) { return T >= 1 && T <= 8 && K % 4 == 0 && ldx % 4 == 0; }
int linear_silu_mul(const float *x, int64_t ldx, const float *W, int64_t T, int64_t K, int64_t I, const float *bias, float *act, hipStream_t s) {
    if (!linear_silu_ok(T, K, ldx)) return nvr::fail(NVR_ERR_UNSUPPORTED, "f32 linear_silu_mul: T=%ld K=%ld (decode-sized steps)", (long)T, (long)K);
    const dim3 grid((unsigned)((I + 3) / 4));
    if (T == 1) gemv_silu_kernel<1><<<grid, dim3(256), 0, s>>>(x, ldx, W, (int)T, (int)K, (int)I, bias, act);
    else if (T <= 4) gemv_silu_kernel<4><<<grid, dim3(256), 0, s>>>(x, ldx, W, (int)T, (int)K, (int)I, bias, act);
    else gemv_silu_kernel<8><<<grid, dim3(256), 0, s>>>(x, ldx, W, (int)T, (int)K, (int)I, bias, act);
    F32_LAUNCH_CHECK("f32 gemv + silu");
    return 0;
}
// rows of the launch's template instance (1 / 4 / 8) and whether their normalised image fits the LDS a kernel gets without an opt-in
static int gemv_rows(int64_t T) { return T == 1 ? 1 : T <= 4 ? 4 : 8; }
bool fused_norm_ok(int64_t T, int64_t K) { return T >= 1 && T <= 8 && K % 4 == 0 && (size_t)gemv_rows(T) * (K + 4) * 4 <= 48 * 1024; }
int add_norm_linear_silu_mul(const float *h_in, const float *y, const float *nw, float eps, float *h_out, const float *W, int64_t T, int64_t K, int64_t I,
                             const float *bias, float *act, hipStream_t s) {
    if (!fused_norm_ok(T, K)) return nvr::fail(NVR_ERR_UNSUPPORTED, "f32 add + norm + gate_up + silu: T=%ld K=%ld", (long)T, (long)K);
    const dim3 grid((unsigned)((I + 3) / 4));
    const size_t lds = (size_t)gemv_rows(T) * (K + 4) * 4;
#define NVR_GS(TT_) gemv_silu_kernel<TT_, true><<<grid, dim3(256), lds, s>>>(nullptr, 0, W, (int)T, (int)K, (int)I, bias, act, h_in, y, nw, eps, h_out)
    if (T == 1) NVR_GS(1); else if (T <= 4) NVR_GS(4); else NVR_GS(8);
#undef NVR_GS
    F32_LAUNCH_CHECK("f32 add + norm + gemv + silu");
    return 0;
}
int linear(const float *x, int64_t ldx, const float *W, int64_t T, int64_t K, int64_t N, const float *bias, float *y, hipStream_t s) {
    if (T == 0 || N == 0) return 0;
    if (T <= 8 && K % 4 == 0 && ldx % 4 == 0) {
        const dim3 grid((unsigned)((N + 3) / 4));
        if (T == 1) gemv_kernel<1><<<grid, dim3(256), 0, s>>>(x, ldx, W, (int)T, (int)K, (int)N, bias, y);
        else if (T <= 4) gemv_kernel<4><<<grid, dim3(256), 0, s>>>(x, ldx, W, (int)T, (int)K, (int)N, bias, y);
        else gemv_kernel<8><<<grid, dim3(256), 0, s>>>(x, ldx, W, (int)T, (int)K, (int)N, bias, y);
        F32_LAUNCH_CHECK("f32 gemv");
        return 0;
    }
    if (K % MKT == 0 && ldx % 4 == 0 && N % 4 == 0 && ((uintptr_t)x | (uintptr_t)W | (uintptr_t)y) % 16 == 0) {     // matrix cores (r06)
        linear_mfma_kernel<<<dim3((unsigned)((N + MB - 1) / MB), (unsigned)((T + MB - 1) / MB)), dim3(256), 0, s>>>(x, ldx, W, (int)T, (int)K, (int)N, bias, y);
        F32_LAUNCH_CHECK("f32 linear (mfma)");
        return 0;
    }
    linear_kernel<<<dim3((unsigned)((N + LB - 1) / LB), (unsigned)((T + LB - 1) / LB)), dim3(256), 0, s>>>(x, ldx, W, (int)T, (int)K, (int)N, bias, y);
    F32_LAUNCH_CHECK("f32 linear");
    return 0;
}

// x1 c - x2 s and x2 c + x1 s with every product rounded on its own (rotary_embedding.rs:36-44 as the oracle evaluates it; the same bits in the two kernels
// below, whatever hipcc's per-kernel choice between fma and multiply + add would have been: see attention.hip mul_then_add)
__device__ __forceinline__ float rot_sub(float a, float b, float c, float d) {
#pragma clang fp contract(off)
    const float x = a * b, y = c * d;
    return x - y;
}
__device__ __forceinline__ float rot_add(float a, float b, float c, float d) {
#pragma clang fp contract(off)
    const float x = a * b, y = c * d;
    return x + y;
}
// qkv projection + RoPE + KV store of a decode-sized step in one launch (linear.rs:354-356, rotary_embedding.rs:23-48, attention.rs:150-174; models
// without q / k head norms): wave w forms columns j and D/2 + j of one head — the two halves of a rotation pair — with gemv_kernel's loads and FMA
// chains, rotates them (value heads pass through) and writes the qkv row and the cache row: the bits of gemv_kernel + rope_store_kernel.
template <int TT, bool NORM = false>
__global__ __launch_bounds__(256) void gemv_rope_kernel(const float *__restrict__ x, int64_t ldx, const float *__restrict__ W, int T, int K, int H, int KVH, int D,
                                                        const float *__restrict__ bias, const int64_t *__restrict__ pos, const int32_t *__restrict__ slots,
                                                        const float *__restrict__ cos_t, const float *__restrict__ sin_t, float *__restrict__ qkv,
                                                        float *__restrict__ kc, float *__restrict__ vc, const float *__restrict__ h_in = nullptr,
                                                        const float *__restrict__ y = nullptr, const float *__restrict__ nw = nullptr, float eps = 0.f,
                                                        float *__restrict__ h_out = nullptr) {
    extern __shared__ float xs[];                                       // NORM: as gemv_silu_kernel
    const int half = D / 2, N = (H + 2 * KVH) * D;
    const int pair_ = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int pair = pair_ < N / 2 ? pair_ : N / 2 - 1;
    const int hd = pair / half, j = pair - hd * half;
    const int n1 = hd * D + j, n2 = n1 + half;
    // position / slot of every row: requested with the weights, used after the reduction
    int64_t p[TT]; int sl[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) { p[t] = pos[t < T ? t : 0]; sl[t] = slots ? slots[t < T ? t : 0] : -1; }
    const float4 *g4 = reinterpret_cast<const float4 *>(W + (int64_t)n1 * K), *u4 = reinterpret_cast<const float4 *>(W + (int64_t)n2 * K);
    const int n4 = K / 4;
    float4 w10[4], w20[4];
    if (NORM) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int k4 = min(lane + 64 * u, n4 - 1); w10[u] = g4[k4]; w20[u] = u4[k4]; }
        add_norm_rows_to_lds<TT>(h_in, y, nw, eps, T, K, blockIdx.x == 0 ? h_out : nullptr, xs, xs + TT * K); x = xs; ldx = K;
    }
    if (pair_ >= N / 2) return;
    float a1[TT], a2[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) a1[t] = a2[t] = 0.f;
    for (int k0 = lane; k0 < n4; k0 += 256) {
        float4 w1[4], w2[4], a[TT][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k4 = min(k0 + 64 * u, n4 - 1);
            if (NORM && k0 == lane) { w1[u] = w10[u]; w2[u] = w20[u]; } else { w1[u] = g4[k4]; w2[u] = u4[k4]; }
#pragma unroll
            for (int t = 0; t < TT; ++t) a[t][u] = reinterpret_cast<const float4 *>(x + (int64_t)(t < T ? t : 0) * ldx)[k4];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (k0 + 64 * u < n4) {
#pragma unroll
                for (int t = 0; t < TT; ++t)
                    if (t < T) {
                        a1[t] = fmaf(a[t][u].x, w1[u].x, fmaf(a[t][u].y, w1[u].y, fmaf(a[t][u].z, w1[u].z, fmaf(a[t][u].w, w1[u].w, a1[t]))));
                        a2[t] = fmaf(a[t][u].x, w2[u].x, fmaf(a[t][u].y, w2[u].y, fmaf(a[t][u].z, w2[u].z, fmaf(a[t][u].w, w2[u].w, a2[t]))));
                    }
            }
    }
    const bool is_k = hd >= H && hd < H + KVH, is_v = hd >= H + KVH;
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        float x1 = wsum(a1[t]), x2 = wsum(a2[t]);
        if (lane == 0 && t < T) {
            if (bias) { x1 = x1 + bias[n1]; x2 = x2 + bias[n2]; }
            float o1 = x1, o2 = x2;
            if (!is_v) {
                const float c = cos_t[p[t] * half + j], sn = sin_t[p[t] * half + j];
                o1 = rot_sub(x1, c, x2, sn); o2 = rot_add(x2, c, x1, sn);
            }
            float *row = qkv + (int64_t)t * N;
            row[n1] = o1; row[n2] = o2;
            if (hd >= H && sl[t] >= 0) {
                float *dst = (is_k ? kc : vc) + ((int64_t)sl[t] * KVH + (is_k ? hd - H : hd - H - KVH)) * D;
                dst[j] = o1; dst[half + j] = o2;
            }
        }
    }
}
bool linear_qkv_rope_ok(int64_t T, int64_t K, int64_t D, int64_t ldx) { return T >= 1 && T <= 8 && K % 4 == 0 && ldx % 4 == 0 && D % 2 == 0; }
int linear_qkv_rope_store(const float *x, int64_t ldx, const float *W, int64_t T, int64_t K, int64_t H, int64_t KVH, int64_t D, const float *bias,
                          const int64_t *pos, const int32_t *slots, const float *cos_t, const float *sin_t, float *qkv, float *kc, float *vc, hipStream_t s) {
    if (!linear_qkv_rope_ok(T, K, D, ldx)) return nvr::fail(NVR_ERR_UNSUPPORTED, "f32 linear_qkv_rope_store: T=%ld K=%ld D=%ld (decode-sized steps)", (long)T, (long)K, (long)D);
    const int64_t pairs = (H + 2 * KVH) * D / 2;
    const dim3 grid((unsigned)((pairs + 3) / 4));
#define NVR_GR(TT_) gemv_rope_kernel<TT_><<<grid, dim3(256), 0, s>>>(x, ldx, W, (int)T, (int)K, (int)H, (int)KVH, (int)D, bias, pos, slots, cos_t, sin_t, qkv, kc, vc)
    if (T == 1) NVR_GR(1); else if (T <= 4) NVR_GR(4); else NVR_GR(8);
#undef NVR_GR
    F32_LAUNCH_CHECK("f32 gemv + rope + store");
    return 0;
}

int add_norm_linear_qkv_rope_store(const float *h_in, const float *y, const float *nw, float eps, float *h_out, const float *W, int64_t T, int64_t K, int64_t H,
                                   int64_t KVH, int64_t D, const float *bias, const int64_t *pos, const int32_t *slots, const float *cos_t, const float *sin_t,
                                   float *qkv, float *kc, float *vc, hipStream_t s) {
    if (!fused_norm_ok(T, K) || D % 2) return nvr::fail(NVR_ERR_UNSUPPORTED, "f32 add + norm + qkv + rope: T=%ld K=%ld D=%ld", (long)T, (long)K, (long)D);
    const int64_t pairs = (H + 2 * KVH) * D / 2;
    const dim3 grid((unsigned)((pairs + 3) / 4));
    const size_t lds = (size_t)gemv_rows(T) * (K + 4) * 4;
#define NVR_GR(TT_) gemv_rope_kernel<TT_, true><<<grid, dim3(256), lds, s>>>(nullptr, 0, W, (int)T, (int)K, (int)H, (int)KVH, (int)D, bias, pos, slots, cos_t, sin_t, \
                                                                            qkv, kc, vc, h_in, y, nw, eps, h_out)
    if (T == 1) NVR_GR(1); else if (T <= 4) NVR_GR(4); else NVR_GR(8);
#undef NVR_GR
    F32_LAUNCH_CHECK("f32 add + norm + gemv + rope + store");
    return 0;
}

// ---------------------------------------------------------------- [q/k head norm,] RoPE, KV store: one workgroup per token, one wave per head in turn
__global__ __launch_bounds__(256) void rope_store_kernel(float *__restrict__ qkv, const int64_t *__restrict__ pos, const int32_t *__restrict__ slots, int H,
                                                         int KVH, int D, const float *__restrict__ cos_t, const float *__restrict__ sin_t,
                                                         float *__restrict__ kc, float *__restrict__ vc, const float *__restrict__ qn,
                                                         const float *__restrict__ kn, float eps) {
    const int t = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, half = D / 2;
    float *row = qkv + (int64_t)t * (H + 2 * KVH) * D;
    const int64_t p = pos[t];
    const int slot = slots ? slots[t] : -1;
    for (int hd = wave; hd < H + 2 * KVH; hd += 4) {
        float *x = row + (int64_t)hd * D;
        const bool is_k = hd >= H && hd < H + KVH, is_v = hd >= H + KVH;
        float *dst = (hd >= H && slot >= 0) ? (is_k ? kc : vc) + ((int64_t)slot * KVH + (is_k ? hd - H : hd - H - KVH)) * D : nullptr;
        if (is_v) {                                                     // value rows pass through (every lane copies what it read itself)
            if (dst) for (int j = lane; j < D; j += 64) dst[j] = x[j];
            continue;
        }
        const float *nw = hd < H ? qn : kn;
        float rms = 1.f;
        if (nw) {                                                       // A-27: RMSNorm over head_dim before the rotation
            float ss = 0.f;
            for (int j = lane; j < D; j += 64) ss += x[j] * x[j];
            rms = sqrtf(wsum(ss) / (float)D + eps);
        }
        for (int j = lane; j < half; j += 64) {
            float x1 = x[j], x2 = x[half + j];
            if (nw) { x1 = x1 / rms * nw[j]; x2 = x2 / rms * nw[half + j]; }
            const float c = cos_t[p * half + j], sn = sin_t[p * half + j];
            const float o1 = rot_sub(x1, c, x2, sn), o2 = rot_add(x2, c, x1, sn);   // rotary_embedding.rs:36-44
            x[j] = o1; x[half + j] = o2;
            if (dst) { dst[j] = o1; dst[half + j] = o2; }
        }
    }
}
int rope_store_kv(float *qkv, const int64_t *pos, const int32_t *slots, int64_t T, int64_t H, int64_t KVH, int64_t D, const float *cos_t,
                  const float *sin_t, float *kc, float *vc, const float *q_norm, const float *k_norm, float eps, hipStream_t s) {
    if (T == 0) return 0;
    if (D % 2) return nvr::fail(NVR_ERR_UNSUPPORTED, "f32 rope: head_dim %ld is odd", (long)D);
    rope_store_kernel<<<dim3((unsigned)T), dim3(256), 0, s>>>(qkv, pos, slots, (int)H, (int)KVH, (int)D, cos_t, sin_t, kc, vc, q_norm, k_norm, eps);
    F32_LAUNCH_CHECK("f32 rope_store_kv");
    return 0;
}

// ---------------------------------------------------------------- SiluAndMul
__global__ void silu_mul_kernel(const float *__restrict__ gu, int64_t T, int I, float *__restrict__ out) {
    const int64_t total = T * I;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = i / I; const int c = (int)(i % I);
        const float g = gu[t * 2 * I + c], u = gu[t * 2 * I + I + c];
        out[i] = silu_mul_one(g, u);
    }
}
int silu_and_mul(const float *gu, int64_t T, int64_t I, float *out, hipStream_t s) {
    if (T * I == 0) return 0;
    int64_t blocks = (T * I + 255) / 256; if (blocks > 8192) blocks = 8192;
    silu_mul_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(gu, T, (int)I, out);
    F32_LAUNCH_CHECK("f32 silu_and_mul");
    return 0;
}

// ---------------------------------------------------------------- silu / gelu / relu / SiluAndMul / GeluAndMul (activation.rs:12-27,74-100,147-159), f32
__global__ void activation_kernel(int kind, const float *__restrict__ x, int cols_in, int cols_out, float *__restrict__ out, int64_t total) {
    const bool mul = kind >= 3;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = i / cols_out; const int c = (int)(i % cols_out);
        const float g = x[t * cols_in + c];
        float a;
        if (kind == 1 || kind == 4) { const float inner = 0.7978845608028654f * g * (1.0f + 0.044715f * g * g); a = 0.5f * g * (1.0f + tanhf(inner)); }
        else if (kind == 2) a = fmaxf(g, 0.0f);
        else a = g * (1.0f / (1.0f + expf(-g)));
        out[i] = mul ? a * x[t * cols_in + cols_out + c] : a;
    }
}
int activation(int kind, const float *x, int64_t T, int64_t cols, float *out, hipStream_t s) {
    if (kind < 0 || kind > 4) return nvr::fail(NVR_ERR_INVALID_ARG, "activation: unknown type %d", kind);
    if (kind >= 3 && cols % 2) return nvr::fail(NVR_ERR_INVALID_ARG, "Input dimension must be even for %s, got %ld", kind == 3 ? "SiluAndMul" : "GeluAndMul", (long)cols);
    const int64_t cols_out = kind >= 3 ? cols / 2 : cols, total = T * cols_out;
    if (total == 0) return 0;
    int64_t blocks = (total + 255) / 256; if (blocks > 8192) blocks = 8192;
    activation_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(kind, x, (int)cols, (int)cols_out, out, total);
    F32_LAUNCH_CHECK("f32 activation");
    return 0;
}

// ---------------------------------------------------------------- attention: one workgroup per (query row, head).  Keys of the row: ctx_lens[t] of them,
// either rows kv_base[t] + j of a contiguous K / V (stride ldkv; the prefill step's own qkv buffer) or cache rows through block table seq_of_q[t]
// (paged decode).  Scores of all keys in LDS (two-pass softmax as compute_attention writes it), then out[d] = sum_j p_j v_j[d] / sum.
template <int NV>   // output columns per lane: head_dim <= 64 * NV
__global__ __launch_bounds__(256) void attention_kernel(AttnArgsF a, int paged) {
    extern __shared__ float sc[];                                       // [max_ctx] scores, then probabilities | q [D] | partial outputs [4][D] | 8 of reduction scratch
    const int t = blockIdx.x, hd = blockIdx.y, g = hd / (a.H / a.KVH), D = a.D;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *qs = sc + a.max_ctx, *part = qs + D, *sm = part + 4 * D;
    // At a small batch this launch is a chain of dependent round trips (r05 stamps, bs 1: ctx -> table entry -> V rows issued, table entry -> K rows, ..,
    // table entry -> second batch of V rows: six of ~2 us).  What only needs the block table is therefore looked up BEFORE the context length is
    // known: the rows of the first 256 keys (scores) and of the first two V batches.  A key index past the context reads a table entry that may be
    // stale; the row offset made from it is never dereferenced (key 0's row takes its place once ctx is in).  Trips: (ctx, q, table) -> (K, V).
    const int32_t *bt = paged ? a.block_tables + (int64_t)(a.seq_of_q ? a.seq_of_q[t] : t) * a.max_blocks : nullptr;
    const int64_t base = paged ? 0 : (int64_t)a.kv_base[t];
    auto row_of = [&](int j) -> int64_t {                               // element offset of key j's row of this kv head (any j >= 0)
        if (paged) return (((int64_t)max(bt[min(j / a.block_size, a.max_blocks - 1)], 0) * a.block_size + j % a.block_size) * a.KVH + g) * D;
        return (base + j) * a.ldkv + (int64_t)g * D;
    };
    // P.V: wave w takes keys w, w + 4, ...; a lane holds output columns lane, lane + 64, ... .  KB of the wave's keys are requested per round trip (their
    // rows' block-table entries looked up by KB lanes at once); the FIRST batch goes out in front of the scores (the FMA order of an output is unchanged)
    constexpr int KB = 64 / NV;                                          // 64 registers of V per lane
    // (paged: the loads below stay unconditional — a key past the context reads key 0's row, and what it returns is replaced by 0; the entries behind a
    //  sequence's blocks are read but never turned into an address that is used.  Contiguous K / V: the rows are arithmetic, formed once ctx is known)
    int64_t k_row0 = paged ? row_of(threadIdx.x) : 0;
    int64_t v_row0 = paged ? row_of(wave + 4 * (lane < KB ? lane : 0)) : 0;
    int64_t v_next = paged ? row_of(wave + 4 * KB + 4 * (lane < KB ? lane : 0)) : 0;
    const int64_t key0_row = paged ? row_of(0) : 0;
    const int ctx = a.ctx_lens[t];
    if (!paged) {
        const int last = max(ctx - 1, 0);
        k_row0 = row_of(min((int)threadIdx.x, last));
        v_row0 = row_of(min(wave + 4 * (lane < KB ? lane : 0), last));
        v_next = row_of(min(wave + 4 * KB + 4 * (lane < KB ? lane : 0), last));
    } else {                                                            // a key past the context: key 0's row instead of what its (unused) table entry says
        if ((int)threadIdx.x >= ctx) k_row0 = key0_row;
        if (wave + 4 * (lane < KB ? lane : 0) >= ctx) v_row0 = key0_row;
        if (wave + 4 * KB + 4 * (lane < KB ? lane : 0) >= ctx) v_next = key0_row;
    }
    const float *q = a.q + (int64_t)t * a.ldq + (int64_t)hd * D;
    for (int j = threadIdx.x; j < D; j += 256) qs[j] = q[j];
    __syncthreads();
    float vv[KB][NV], vv2[KB][NV];                                      // the first two V batches (a one-workgroup-per-CU launch: registers are free)
    auto request_v = [&](float (&dst)[KB][NV], int j0, int64_t myrow) { // myrow: lane u's row of key j0 + 4 u
#pragma unroll
        for (int u = 0; u < KB; ++u) {
            const float *vr = a.v + __shfl(myrow, u, 64);
#pragma unroll
            for (int i = 0; i < NV; ++i) { const int e = lane + 64 * i; dst[u][i] = vr[e < D ? e : 0]; }
        }
    };
    auto mask_v = [&](float (&dst)[KB][NV], int j0) {                   // (after the loads have been issued: what a dead key returned becomes 0)
#pragma unroll
        for (int u = 0; u < KB; ++u) {
            const bool live = j0 + 4 * u < ctx;
#pragma unroll
            for (int i = 0; i < NV; ++i) if (!live || lane + 64 * i >= D) dst[u][i] = 0.f;
        }
    };
    request_v(vv, wave, v_row0);
    const bool two = wave + 4 * KB < ctx;                               // a second batch exists: requested now too (ctx <= 8 KB keys then needs no third trip)
    if (two) request_v(vv2, wave + 4 * KB, v_next);
    // scores: a thread per key (256 keys in flight per workgroup: a decode step has one query row per sequence, so the parallelism has to come from the keys)
    float mx = -INFINITY;
    for (int j = threadIdx.x; j < ctx; j += 256) {
        const float4 *kr = reinterpret_cast<const float4 *>(a.k + (j == (int)threadIdx.x ? k_row0 : row_of(j)));
        float d = 0.f;
        // sixteen 16-byte pieces of the row requested before the first FMA (as a plain loop every piece waited for its own round trip to HBM: 32 in
        // a row at head_dim 128 — 29.6 us per launch at bs 1, r04); the FMA chain keeps its order: the same bits
        for (int e0 = 0; e0 < D / 4; e0 += 16) {
            float4 kk[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) kk[u] = kr[min(e0 + u, D / 4 - 1)];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int e = e0 + u;
                if (e < D / 4) d = fmaf(qs[4 * e], kk[u].x, fmaf(qs[4 * e + 1], kk[u].y, fmaf(qs[4 * e + 2], kk[u].z, fmaf(qs[4 * e + 3], kk[u].w, d))));
            }
        }
        d *= a.scale;
        sc[j] = d; mx = fmaxf(mx, d);
    }
    mx = block_reduce<true>(mx, sm, 4);
    float sum = 0.f;
    for (int j = threadIdx.x; j < ctx; j += 256) { const float p = expf(sc[j] - mx); sc[j] = p; sum += p; }
    sum = block_reduce<false>(sum, sm, 4);                              // (its barriers also publish the probabilities)
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    auto consume = [&](const float (&src)[KB][NV], int j0) {
#pragma unroll
        for (int u = 0; u < KB; ++u) {
            const int j = j0 + 4 * u;
            const float pp = j < ctx ? sc[j] : 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) acc[i] = fmaf(pp, src[u][i], acc[i]);
        }
    };
    if (wave < ctx) {
        if (wave + 8 * KB < ctx) v_next = row_of(min(wave + 8 * KB + 4 * (lane < KB ? lane : 0), ctx - 1));   // a third batch: rows looked up under the first two's FMAs
        mask_v(vv, wave); consume(vv, wave);
        if (two) { mask_v(vv2, wave + 4 * KB); consume(vv2, wave + 4 * KB); }
    }
    for (int j0 = wave + 8 * KB; j0 < ctx; j0 += 4 * KB) {               // keys 8 KB.. : one batch per round trip, the next batch's rows looked up under the FMAs
        request_v(vv, j0, v_next);
        if (j0 + 4 * KB < ctx) v_next = row_of(min(j0 + 4 * KB + 4 * (lane < KB ? lane : 0), ctx - 1));
        mask_v(vv, j0); consume(vv, j0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int e = lane + 64 * i; if (e < D) part[wave * D + e] = acc[i]; }
    __syncthreads();
    for (int d = threadIdx.x; d < D; d += 256) {
        const float o = (part[d] + part[D + d]) + (part[2 * D + d] + part[3 * D + d]);
        a.out[((int64_t)t * a.H + hd) * D + d] = ctx > 0 ? o / sum : 0.f;
    }
}
// opt the attention kernel in to > 64 KiB of dynamic LDS (runner init: never inside a stream capture)
int prepare() {
    const void *fns[] = {reinterpret_cast<const void *>(&attention_kernel<1>), reinterpret_cast<const void *>(&attention_kernel<2>),
                         reinterpret_cast<const void *>(&attention_kernel<4>)};
    for (const void *f : fns)
        if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return nvr::fail(NVR_ERR_HIP, "f32 attention: hipFuncSetAttribute failed");
    return 0;
}
int attention(const AttnArgsF &a, bool paged, hipStream_t s) {
    if (a.nq == 0) return 0;
    if (a.D > 256 || a.D % 4) return nvr::fail(NVR_ERR_UNSUPPORTED, "f32 attention: head_dim %d (multiples of 4 up to 256)", a.D);
    const size_t lds = ((size_t)a.max_ctx + 5 * (size_t)a.D + 8) * 4;
    if (lds > 160 * 1024) return nvr::fail(NVR_ERR_UNSUPPORTED, "f32 attention: context %d does not fit the score buffer (%zu bytes of LDS)", a.max_ctx, lds);
    const dim3 grid((unsigned)a.nq, (unsigned)a.H);
    if (a.D <= 64) attention_kernel<1><<<grid, dim3(256), lds, s>>>(a, paged ? 1 : 0);
    else if (a.D <= 128) attention_kernel<2><<<grid, dim3(256), lds, s>>>(a, paged ? 1 : 0);
    else attention_kernel<4><<<grid, dim3(256), lds, s>>>(a, paged ? 1 : 0);
    F32_LAUNCH_CHECK("f32 attention");
    return 0;
}

}}  // namespace nvr::kf
