// sampler.hip — per-row temperature / top-k / top-p / Gumbel-max sampling (K18) on gfx950.
// reference: Sampler::sample_single, src/layers/sampler.rs:71-106; apply_top_k :115-148 (stable
// descending sort, keep the first k); apply_top_p :151-188 (softmax, stable descending sort, keep the
// prefix up to and including the first index whose cumulative probability >= p); multinomial_sample /
// sample_gumbel :191-218 (argmax(logits + (-log(-log u))), u clamped to [1e-8, 1-1e-8]).
// Decisions: SURVEY.md A-18 (k == 0 disables top-k), A-19 (ties keep the lower index), A-20
// (counter-based RNG keyed by (seed, seq_id, step) instead of candle's global RNG).
//
// The reference sorts all 151 936 logits per row on the host.  Here one 1024-thread workgroup owns a
// row and never sorts: both filters are 4-pass 8-bit radix *selects* over the row (LDS histograms),
// followed by one index-ordered pass that resolves ties at the threshold exactly as a stable sort
// would.  top-p sums probabilities in 2^-40 fixed point so the LDS atomics are order-independent
// (bit-reproducible run to run).  The row (608 KB of f32) stays L2-resident between passes.
#include "kernels.h"
#include "device_utils.h"
#include "../common.h"

namespace nvr { namespace k {

namespace {

constexpr int kThreads = 1024;

struct Shared {
    unsigned int hist[256];
    unsigned long long hsum[256];
    unsigned int wave_cnt[16];
    float redf[16];
    int redi[16];
    unsigned int sel, need, run_base, flag;
    unsigned long long cum;
    float bcast_f;
};

__device__ __forceinline__ unsigned int ordered_key(float f) {        // larger float <-> larger uint
    unsigned int u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ float block_max(float v, Shared &sh) {
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) sh.redf[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) { float m = sh.redf[0]; for (int i = 1; i < kThreads / 64; ++i) m = fmaxf(m, sh.redf[i]); sh.bcast_f = m; }
    __syncthreads();
    float r = sh.bcast_f;
    __syncthreads();
    return r;
}
__device__ float block_sum(float v, Shared &sh) {                      // fixed tree: deterministic
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh.redf[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) { float s = 0.f; for (int i = 0; i < kThreads / 64; ++i) s += sh.redf[i]; sh.bcast_f = s; }
    __syncthreads();
    float r = sh.bcast_f;
    __syncthreads();
    return r;
}

// Keep elements with key > thr, plus the first `need` (in index order) with key == thr; the rest -> -inf.
template <class KeyFn>
__device__ void filter_row(float *w, int V, unsigned int thr, unsigned int need, unsigned int cnt_eq, KeyFn key, Shared &sh) {
    if (need >= cnt_eq) {                                              // no tie to break: one unordered pass
        for (int i = threadIdx.x; i < V; i += kThreads) if (key(i) < thr) w[i] = -INFINITY;
        __syncthreads();
        return;
    }
    if (threadIdx.x == 0) sh.run_base = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = 0; c < V; c += kThreads) {
        const int i = c + threadIdx.x;
        const unsigned int u = i < V ? key(i) : 0u;
        const bool eq = (i < V) && (u == thr);
        const unsigned long long bal = __ballot(eq);
        const unsigned int below = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) sh.wave_cnt[wave] = __popcll(bal);
        __syncthreads();
        unsigned int pre = sh.run_base;
        for (int w2 = 0; w2 < wave; ++w2) pre += sh.wave_cnt[w2];
        const unsigned int rank = pre + below;
        if (i < V && !(u > thr || (eq && rank < need))) w[i] = -INFINITY;
        __syncthreads();
        if (threadIdx.x == 0) { unsigned int t = 0; for (int w2 = 0; w2 < kThreads / 64; ++w2) t += sh.wave_cnt[w2]; sh.run_base += t; }
        __syncthreads();
    }
}

}  // namespace

__global__ __launch_bounds__(kThreads) void sample_kernel(const float *__restrict__ logits, int V,
                                                          const float *__restrict__ temperature,
                                                          const int64_t *__restrict__ top_k, const float *__restrict__ top_p,
                                                          const uint64_t *__restrict__ keys, int64_t *__restrict__ out,
                                                          float *__restrict__ ws) {
    __shared__ Shared sh;
    const int row = blockIdx.x;
    const float *x = logits + (int64_t)row * V;
    float *w = ws + (int64_t)row * V;
    const float temp = temperature[row];
    const bool greedy = (temp == 0.0f);                                // sampler.rs:78-81

    if (!greedy) {
        // temperature scaling, sampler.rs:84-88
        for (int i = threadIdx.x; i < V; i += kThreads) w[i] = (temp != 1.0f) ? x[i] / temp : x[i];
        __syncthreads();

        // ---- top-k, sampler.rs:115-148 --------------------------------------------------------
        long long k = top_k ? top_k[row] : 0;
        if (k > 0 && k < V) {
            auto key = [&](int i) { return ordered_key(w[i]); };
            unsigned int prefix = 0, mask = 0, need = (unsigned int)k, cnt_eq = 0;
            for (int pass = 0; pass < 4; ++pass) {
                const int shift = 24 - 8 * pass;
                for (int b = threadIdx.x; b < 256; b += kThreads) sh.hist[b] = 0;
                __syncthreads();
                for (int i = threadIdx.x; i < V; i += kThreads) {
                    const unsigned int u = key(i);
                    if ((u & mask) == prefix) atomicAdd(&sh.hist[(u >> shift) & 255u], 1u);
                }
                __syncthreads();
                if (threadIdx.x == 0) {
                    unsigned int nd = need, b = 255;
                    for (;; --b) { if (nd > sh.hist[b]) nd -= sh.hist[b]; else break; if (b == 0) break; }
                    sh.sel = b; sh.need = nd; sh.flag = sh.hist[b];
                }
                __syncthreads();
                prefix |= sh.sel << shift; mask |= 255u << shift; need = sh.need; cnt_eq = sh.flag;
                __syncthreads();
            }
            filter_row(w, V, prefix, need, cnt_eq, key, sh);
        }

        // ---- top-p, sampler.rs:151-188 --------------------------------------------------------
        const float p = top_p ? top_p[row] : -1.0f;
        if (p >= 0.0f) {
            float lm = -INFINITY;
            for (int i = threadIdx.x; i < V; i += kThreads) lm = fmaxf(lm, w[i]);
            const float m = block_max(lm, sh);
            float ls = 0.f;
            for (int i = threadIdx.x; i < V; i += kThreads) ls += __expf(w[i] - m);
            const float inv = 1.0f / block_sum(ls, sh);
            auto prob = [&](int i) { return __expf(w[i] - m) * inv; };
            auto key = [&](int i) { return __float_as_uint(prob(i)); };   // probs >= 0: bits are monotone
            const double FX = 1099511627776.0;                            // 2^40
            const unsigned long long P = (unsigned long long)((double)p * FX);
            unsigned int prefix = 0, mask = 0, cnt_eq = 0;
            unsigned long long cum = 0;
            bool keep_all = false;
            for (int pass = 0; pass < 4 && !keep_all; ++pass) {
                const int shift = 24 - 8 * pass;
                for (int b = threadIdx.x; b < 256; b += kThreads) { sh.hist[b] = 0; sh.hsum[b] = 0ull; }
                __syncthreads();
                for (int i = threadIdx.x; i < V; i += kThreads) {
                    const float pr = prob(i);
                    const unsigned int u = __float_as_uint(pr);
                    if ((u & mask) == prefix) {
                        const unsigned int b = (u >> shift) & 255u;
                        atomicAdd(&sh.hist[b], 1u);
                        atomicAdd(&sh.hsum[b], (unsigned long long)((double)pr * FX));
                    }
                }
                __syncthreads();
                if (threadIdx.x == 0) {
                    unsigned long long c = cum; int sel = -1;
                    for (int b = 255; b >= 0; --b) {
                        if (sh.hist[b] == 0) continue;
                        if (c + sh.hsum[b] < P) c += sh.hsum[b]; else { sel = b; break; }
                    }
                    sh.flag = (sel < 0) ? 1u : 0u;                      // cumulative mass never reaches p: keep everything
                    sh.sel = sel < 0 ? 0u : (unsigned int)sel; sh.cum = c; sh.need = sel < 0 ? 0u : sh.hist[sel];
                }
                __syncthreads();
                keep_all = sh.flag != 0; cum = sh.cum; cnt_eq = sh.need;
                prefix |= sh.sel << shift; mask |= 255u << shift;
                __syncthreads();
            }
            if (!keep_all) {
                const unsigned long long qv = (unsigned long long)((double)__uint_as_float(prefix) * FX);
                unsigned long long need = 1;
                if (qv > 0) { need = (P - cum + qv - 1) / qv; if (need < 1) need = 1; }
                if (need > cnt_eq) need = cnt_eq;
                filter_row(w, V, prefix, (unsigned int)need, cnt_eq, key, sh);
            }
        }
    }

    // ---- Gumbel-max (or plain argmax when greedy), sampler.rs:109-112,191-203 ----------------------
    const uint64_t rkey = keys ? keys[row] : 0ull;
    float bv = -INFINITY; int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < V; i += kThreads) {
        float v;
        if (greedy) v = x[i];
        else {
            v = w[i];
            if (v == -INFINITY) continue;
            const uint64_t r = splitmix64(rkey ^ (uint64_t)i);
            float u = ((float)(r >> 40) + 0.5f) * (1.0f / 16777216.0f);
            u = fminf(fmaxf(u, 1e-8f), 1.0f - 1e-8f);
            v += -logf(-logf(u));
        }
        if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { sh.redf[threadIdx.x >> 6] = bv; sh.redi[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < kThreads / 64; ++i)
            if (sh.redf[i] > bv || (sh.redf[i] == bv && sh.redi[i] < bi)) { bv = sh.redf[i]; bi = sh.redi[i]; }
        out[row] = (bi == 0x7fffffff) ? 0 : (int64_t)bi;
    }
}

size_t sample_workspace_bytes(int64_t B, int64_t V) { return (size_t)(B * V) * sizeof(float) + 256; }

int sample(const float *logits, int64_t B, int64_t V, const float *temperature, const int64_t *top_k,
           const float *top_p, const uint64_t *keys, int64_t *out_ids, void *workspace, hipStream_t s) {
    if (B == 0) return 0;
    if (!workspace || !temperature) return nvr::fail(NVR_ERR_INVALID_ARG, "sample: workspace and temperature are required");
    sample_kernel<<<dim3((unsigned)B), dim3(kThreads), 0, s>>>(logits, (int)V, temperature, top_k, top_p, keys, out_ids,
                                                              (float *)workspace);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return nvr::fail(NVR_ERR_HIP, "sample launch failed: %s", hipGetErrorString(e));
    return 0;
}

}}  // namespace nvr::k
