// sampler.hip — per-row temperature / top-k / top-p / Gumbel-max sampling (K18) on gfx950.
// reference: Sampler::sample_single, src/layers/sampler.rs:71-106; apply_top_k :115-148 (stable
// descending sort, keep the first k); apply_top_p :151-188 (softmax, stable descending sort, keep the
// prefix up to and including the first index whose cumulative probability >= p); multinomial_sample /
// sample_gumbel :191-218 (argmax(logits + (-log(-log u))), u clamped to [1e-8, 1-1e-8]).
// Decisions: SURVEY.md A-18 (k == 0 disables top-k), A-19 (ties keep the lower index), A-20
// (counter-based RNG keyed by (seed, seq_id, step) instead of candle's global RNG).
//
// The reference sorts all 151 936 logits per row on the host.  Here nothing is sorted: both filters are 4-pass 8-bit radix
// *selects* over the row (LDS histograms), followed by an index-ordered pass that resolves ties at the threshold exactly as
// a stable sort would.  top-p sums probabilities in 2^-40 fixed point so the atomics are order-independent (bit-reproducible
// run to run).  Two forms: sample_rows_kernel (<= 64 rows: up to 8 workgroups share a row, each holding its slice in
// registers for all passes; histograms and softmax partials are combined through agent-scope atomics and a bounded arrival
// barrier) and sample_kernel (one 1024-thread workgroup per row re-reading the L2-resident row every pass: more rows, or
// rows too long for the register form).  The shared form needs its <= 256 workgroups co-resident: two such launches running
// CONCURRENTLY on one GPU (two engines on two streams) could starve each other until the spin bound trips (RowHdr::timeout);
// one engine issues its sampler launches in stream order.
#include <cstdlib>
#include "kernels.h"
#include "device_utils.h"
#include "../common.h"

namespace nvr { namespace NVR_DT_NS {

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

// ------------------------------------------------------------------------------------------------------------------
// Row-sharing form (B <= 64 rows): P workgroups own one row and each keeps its slice of the row in REGISTERS from the
// first load to the final arg-max, so the 8-13 passes of the filters touch no memory; what crosses workgroups are the
// 256-bin histograms of the radix selects (agent-scope atomics into a zeroed header in the workspace), softmax partials
// and an arrival counter.  All P workgroups of a row take the same decisions from the same combined data, so they agree on
// the number of synchronisations; every spin is bounded.  The grid (<= 256 workgroups of 1024 threads) is co-resident on
// the 256 CUs; workgroup w of a row group is placed so that the P sharers have equal w % 8 (same XCD, same L2).
// Same arithmetic as sample_kernel (temperature division, __expf probabilities, 2^-40 fixed-point masses, index-ordered
// tie breaks); only the f32 softmax denominator is summed per slice and then over slices.
constexpr int kMaxParts = 8;
struct alignas(256) RowHdr {
    unsigned int cnt[8][256];                  // [0..3] top-k passes, [4..7] top-p passes
    unsigned long long mass[4][256];           // top-p passes: fixed-point probability mass per bin
    unsigned long long best;                   // (ordered value, ~index) of the Gumbel arg-max
    unsigned int arrive, done, timeout, pad;
    unsigned int eq[2][kMaxParts];             // elements equal to the threshold per slice (tie break), top-k / top-p
    unsigned int part_m[kMaxParts], part_s[kMaxParts];   // softmax partials per slice (f32 bits)
};
struct RShared {
    unsigned int hist[256];
    unsigned long long hsum[256];
    unsigned long long scan[2][256];
    unsigned int wave_cnt[16];
    unsigned long long red64[16];
    float redf[16];
    int sel;
    unsigned int run_base, eq_cnt;
    float bcast_f;
};

__device__ __forceinline__ void row_sync(RowHdr *h, unsigned int &epoch, int P) {
    if (P == 1) { __syncthreads(); return; }
    // everything that crosses workgroups is an agent-scope atomic (performed at the coherent level, sc1): waiting for this
    // thread's outstanding ones is all the "release" needed — an agent-scope fence would write back / invalidate the whole L2
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ++epoch;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(&h->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned int target = epoch * (unsigned int)P;
        unsigned int spins = 0;
        while (__hip_atomic_load(&h->arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 22)) { __hip_atomic_store(&h->timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
    }
    __syncthreads();
}

// Inclusive suffix sums of 256 bin values (threads 0..255 hold v): S[t] = sum_{b >= t} v[b]; every thread may read sc[.] after.
__device__ __forceinline__ const unsigned long long *suffix_scan(unsigned long long v, RShared &sh) {
    const int t = threadIdx.x;
    if (t < 256) sh.scan[0][t] = v;
    __syncthreads();
    int cur = 0;
#pragma unroll
    for (int d = 1; d < 256; d <<= 1) {
        if (t < 256) sh.scan[cur ^ 1][t] = sh.scan[cur][t] + (t + d < 256 ? sh.scan[cur][t + d] : 0ull);
        __syncthreads();
        cur ^= 1;
    }
    return sh.scan[cur];          // 8 steps: cur == 0
}

template <int EPT>
__global__ __launch_bounds__(kThreads) void sample_rows_kernel(const float *__restrict__ logits, int V, int B, int P, int rows_per_xcd,
                                                               int chunk, const float *__restrict__ temperature,
                                                               const int64_t *__restrict__ top_k, const float *__restrict__ top_p,
                                                               const uint64_t *__restrict__ keys, int64_t *__restrict__ out,
                                                               float *__restrict__ ws, RowHdr *__restrict__ hdrs, int store_w) {
    __shared__ RShared sh;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int row = xcd * rows_per_xcd + slot / P, part = slot % P;
    if (slot / P >= rows_per_xcd || row >= B) return;
    RowHdr *h = hdrs + row;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int base = part * chunk, end = min(V, base + chunk);
    const float *x = logits + (int64_t)row * V;
    const float temp = temperature[row];
    const bool greedy = (temp == 0.0f);                                // sampler.rs:78-81
    unsigned int epoch = 0;

    float w[EPT];
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int i = base + j * kThreads + tid;
        w[j] = i < end ? x[i] : -INFINITY;
        if (!greedy && temp != 1.0f) w[j] = w[j] / temp;               // sampler.rs:84-88
    }
    auto valid = [&](int j) { return base + j * kThreads + tid < end; };

    // combine the workgroup's LDS histogram(s) into the row's, wait for the sharers, read the totals back (threads < 256)
    auto combine = [&](int cslot, int mslot, unsigned int &c_out, unsigned long long &m_out) {
        c_out = 0; m_out = 0;
        if (P == 1) {
            if (tid < 256) { c_out = sh.hist[tid]; if (mslot >= 0) m_out = sh.hsum[tid]; }
            __syncthreads();
            return;
        }
        if (tid < 256) {
            if (sh.hist[tid]) __hip_atomic_fetch_add(&h->cnt[cslot][tid], sh.hist[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (mslot >= 0 && sh.hsum[tid]) __hip_atomic_fetch_add(&h->mass[mslot][tid], sh.hsum[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        row_sync(h, epoch, P);
        if (tid < 256) {
            c_out = __hip_atomic_load(&h->cnt[cslot][tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (mslot >= 0) m_out = __hip_atomic_load(&h->mass[mslot][tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };

    // Keep elements with key > thr plus the first `need` (in index order over the whole row) with key == thr
    auto filter = [&](auto key, unsigned int thr, unsigned int need, unsigned int cnt_eq, int which) {
        if (need >= cnt_eq) {                                          // no tie to break
#pragma unroll
            for (int j = 0; j < EPT; ++j) if (valid(j) && key(j) < thr) w[j] = -INFINITY;
            return;
        }
        unsigned int before = 0;                                       // equal elements in the slices ahead of this one
        if (P > 1) {
            if (tid == 0) sh.eq_cnt = 0;
            __syncthreads();
            unsigned int mine = 0;
#pragma unroll
            for (int j = 0; j < EPT; ++j) mine += (valid(j) && key(j) == thr) ? 1u : 0u;
            if (mine) atomicAdd(&sh.eq_cnt, mine);
            __syncthreads();
            if (tid == 0) __hip_atomic_store(&h->eq[which][part], sh.eq_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            row_sync(h, epoch, P);
            for (int q = 0; q < part; ++q) before += __hip_atomic_load(&h->eq[which][q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid == 0) sh.run_base = before;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            const bool ok = valid(j);
            const unsigned int u = ok ? key(j) : 0u;
            const bool eq = ok && (u == thr);
            const unsigned long long bal = __ballot(eq);
            const unsigned int below = __popcll(bal & ((1ull << lane) - 1ull));
            if (lane == 0) sh.wave_cnt[wave] = __popcll(bal);
            __syncthreads();
            unsigned int pre = sh.run_base;
            for (int w2 = 0; w2 < wave; ++w2) pre += sh.wave_cnt[w2];
            if (ok && !(u > thr || (eq && pre + below < need))) w[j] = -INFINITY;
            __syncthreads();
            if (tid == 0) { unsigned int t = 0; for (int w2 = 0; w2 < kThreads / 64; ++w2) t += sh.wave_cnt[w2]; sh.run_base += t; }
            __syncthreads();
        }
    };

    if (!greedy) {
        // ---- top-k, sampler.rs:115-148: 4-pass radix select of the k-th largest key ------------------
        const long long k = top_k ? top_k[row] : 0;
        if (k > 0 && k < V) {
            auto key = [&](int j) { return ordered_key(w[j]); };
            unsigned int prefix = 0, mask = 0, need = (unsigned int)k, cnt_eq = 0;
            for (int pass = 0; pass < 4; ++pass) {
                const int shift = 24 - 8 * pass;
                if (tid < 256) sh.hist[tid] = 0;
                __syncthreads();
                unsigned int rb = 0xffffffffu, rc = 0;                 // run of equal bins in this thread's elements
#pragma unroll
                for (int j = 0; j < EPT; ++j) {
                    if (!valid(j)) continue;
                    const unsigned int u = key(j);
                    if ((u & mask) != prefix) continue;
                    const unsigned int b = (u >> shift) & 255u;
                    if (b != rb) { if (rc) atomicAdd(&sh.hist[rb], rc); rb = b; rc = 0; }
                    ++rc;
                }
                if (rc) atomicAdd(&sh.hist[rb], rc);
                __syncthreads();
                unsigned int c; unsigned long long unused;
                combine(pass, -1, c, unused);
                const unsigned long long *S = suffix_scan((unsigned long long)c, sh);
                if (tid == 0) sh.sel = 0;
                __syncthreads();
                // highest bin whose inclusive suffix count reaches `need` (the serial scan of sample_kernel)
                if (tid < 256 && S[tid] >= need && (tid == 255 || S[tid + 1] < need)) sh.sel = tid;
                __syncthreads();
                const int b = sh.sel;
                const unsigned int above = b < 255 ? (unsigned int)S[b + 1] : 0u;
                cnt_eq = (unsigned int)(S[b] - (b < 255 ? S[b + 1] : 0ull));
                need -= above;
                prefix |= (unsigned int)b << shift; mask |= 255u << shift;
                __syncthreads();
            }
            filter(key, prefix, need, cnt_eq, 0);
        }

        // ---- top-p, sampler.rs:151-188 ------------------------------------------------------------------
        const float p = top_p ? top_p[row] : -1.0f;
        if (p >= 0.0f) {
            float lm = -INFINITY;
#pragma unroll
            for (int j = 0; j < EPT; ++j) lm = fmaxf(lm, w[j]);
            lm = wave_max(lm);
            if (lane == 0) sh.redf[wave] = lm;
            __syncthreads();
            if (tid == 0) { float m = sh.redf[0]; for (int i = 1; i < kThreads / 64; ++i) m = fmaxf(m, sh.redf[i]); sh.bcast_f = m; }
            __syncthreads();
            const float pm = sh.bcast_f;                               // maximum of this slice
            __syncthreads();
            float ls = 0.f;
            if (pm != -INFINITY) {
#pragma unroll
                for (int j = 0; j < EPT; ++j) ls += __expf(w[j] - pm);
            }
            ls = wave_sum(ls);
            if (lane == 0) sh.redf[wave] = ls;
            __syncthreads();
            if (tid == 0) { float sacc = 0.f; for (int i = 0; i < kThreads / 64; ++i) sacc += sh.redf[i]; sh.bcast_f = sacc; }
            __syncthreads();
            const float ps = sh.bcast_f;
            __syncthreads();
            float m = pm, Z = ps;
            if (P > 1) {
                if (tid == 0) {
                    __hip_atomic_store(&h->part_m[part], __float_as_uint(pm), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&h->part_s[part], __float_as_uint(ps), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                row_sync(h, epoch, P);
                float qm[kMaxParts], qs[kMaxParts];
                m = -INFINITY;
                for (int q = 0; q < P; ++q) {
                    qm[q] = __uint_as_float(__hip_atomic_load(&h->part_m[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    qs[q] = __uint_as_float(__hip_atomic_load(&h->part_s[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    m = fmaxf(m, qm[q]);
                }
                Z = 0.f;
                for (int q = 0; q < P; ++q) if (qm[q] != -INFINITY) Z += qs[q] * __expf(qm[q] - m);
            }
            const float inv = 1.0f / Z;
            auto prob = [&](int j) { return __expf(w[j] - m) * inv; };
            auto key = [&](int j) { return __float_as_uint(prob(j)); };   // probs >= 0: bits are monotone
            const double FX = 1099511627776.0;                            // 2^40
            const unsigned long long Pfx = (unsigned long long)((double)p * FX);
            unsigned int prefix = 0, mask = 0, cnt_eq = 0;
            unsigned long long cum = 0;
            bool keep_all = false;
            for (int pass = 0; pass < 4 && !keep_all; ++pass) {
                const int shift = 24 - 8 * pass;
                if (tid < 256) { sh.hist[tid] = 0; sh.hsum[tid] = 0ull; }
                __syncthreads();
                unsigned int rb = 0xffffffffu, rc = 0; unsigned long long rm = 0;
#pragma unroll
                for (int j = 0; j < EPT; ++j) {
                    if (!valid(j)) continue;
                    const float pr = prob(j);
                    const unsigned int u = __float_as_uint(pr);
                    if ((u & mask) != prefix) continue;
                    const unsigned int b = (u >> shift) & 255u;
                    if (b != rb) {
                        if (rc) { atomicAdd(&sh.hist[rb], rc); atomicAdd(&sh.hsum[rb], rm); }
                        rb = b; rc = 0; rm = 0;
                    }
                    ++rc; rm += (unsigned long long)((double)pr * FX);
                }
                if (rc) { atomicAdd(&sh.hist[rb], rc); atomicAdd(&sh.hsum[rb], rm); }
                __syncthreads();
                unsigned int c; unsigned long long ms;
                combine(4 + pass, pass, c, ms);
                const unsigned long long *S = suffix_scan(ms, sh);
                if (tid == 0) sh.sel = -1;
                __syncthreads();
                // first non-empty bin from the top where the cumulative mass reaches p (the serial scan of sample_kernel)
                if (tid < 256 && c > 0 && cum + S[tid] >= Pfx) atomicMax(&sh.sel, tid);
                __syncthreads();
                const int b = sh.sel;
                if (tid < 256 && tid == (b < 0 ? 0 : b)) sh.eq_cnt = c;   // count of the selected bin
                __syncthreads();
                keep_all = b < 0;                                       // cumulative mass never reaches p: keep everything
                if (!keep_all) {
                    cum += b < 255 ? S[b + 1] : 0ull;
                    cnt_eq = sh.eq_cnt;
                    prefix |= (unsigned int)b << shift; mask |= 255u << shift;
                }
                __syncthreads();
            }
            if (!keep_all) {
                const unsigned long long qv = (unsigned long long)((double)__uint_as_float(prefix) * FX);
                unsigned long long need = 1;
                if (qv > 0) { need = (Pfx - cum + qv - 1) / qv; if (need < 1) need = 1; }
                if (need > cnt_eq) need = cnt_eq;
                filter(key, prefix, (unsigned int)need, cnt_eq, 1);
            }
        }
    }
    if (store_w && !greedy) {
        float *wr = ws + (int64_t)row * V;
#pragma unroll
        for (int j = 0; j < EPT; ++j) if (valid(j)) wr[base + j * kThreads + tid] = w[j];
    }

    // ---- Gumbel-max (or plain argmax when greedy), sampler.rs:109-112,191-203 ----------------------
    const uint64_t rkey = keys ? keys[row] : 0ull;
    unsigned long long best = 0;                                       // (ordered value << 32) | ~index; 0 = nothing kept
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        if (!valid(j)) continue;
        const int i = base + j * kThreads + tid;
        float v = w[j];
        if (!greedy) {
            if (v == -INFINITY) continue;
            const uint64_t r = splitmix64(rkey ^ (uint64_t)i);
            float u = ((float)(r >> 40) + 0.5f) * (1.0f / 16777216.0f);
            u = fminf(fmaxf(u, 1e-8f), 1.0f - 1e-8f);
            v += -logf(-logf(u));
        }
        if (!(v == v)) continue;                                       // NaN never wins (v > bv is false for it)
        v += 0.0f;                                                     // -0 -> +0: equal values tie on the index
        const unsigned long long cand = ((unsigned long long)ordered_key(v) << 32) | (unsigned int)~(unsigned int)i;
        best = cand > best ? cand : best;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long ob = __shfl_xor(best, o, 64);
        best = ob > best ? ob : best;
    }
    if (lane == 0) sh.red64[wave] = best;
    __syncthreads();
    if (tid == 0) {
        for (int i = 1; i < kThreads / 64; ++i) best = sh.red64[i] > best ? sh.red64[i] : best;
        bool last = true;
        if (P > 1) {
            if (best) __hip_atomic_fetch_max(&h->best, best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the maximum is in before this slice counts as done
            last = __hip_atomic_fetch_add(&h->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned int)P - 1;
            if (last) best = __hip_atomic_load(&h->best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (last) out[row] = best ? (int64_t)(~(unsigned int)(best & 0xffffffffull)) : 0;
    }
}

static size_t sample_header_bytes(int64_t B) { return (size_t)B * sizeof(RowHdr); }
static size_t sample_header_offset(int64_t B, int64_t V) { return ((size_t)(B * V) * sizeof(float) + 255) / 256 * 256; }

size_t sample_workspace_bytes(int64_t B, int64_t V) { return sample_header_offset(B, V) + 256 + sample_header_bytes(B); }

int sample(const float *logits, int64_t B, int64_t V, const float *temperature, const int64_t *top_k,
           const float *top_p, const uint64_t *keys, int64_t *out_ids, void *workspace, hipStream_t s, bool store_filtered) {
    if (B == 0) return 0;
    if (!workspace || !temperature) return nvr::fail(NVR_ERR_INVALID_ARG, "sample: workspace and temperature are required");
    if (B <= 64) {
        const int rpx = (int)(B + 7) / 8;                            // rows per XCD
        // the sharers of a row wait for each other: every workgroup of the grid must be resident at once, one 1024-thread
        // workgroup per CU (128 VGPRs) -> no more workgroups than the device has CUs (256 on an MI355X, fewer in a partitioned mode)
        static const int cus = [] { int d = 0; hipDeviceProp_t pr; return (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&pr, d) == hipSuccess) ? pr.multiProcessorCount : 0; }();
        int P = 1;
        while (P * 2 <= kMaxParts && P * 2 * rpx * 8 <= cus) P *= 2;
        while (P > 1 && V / P < 4096) P /= 2;
        const int chunk = (int)(((V + P - 1) / P + kThreads - 1) / kThreads * kThreads);
        const int ept = chunk / kThreads;
        if (ept <= 40) {
            RowHdr *hdrs = reinterpret_cast<RowHdr *>(reinterpret_cast<char *>(workspace) + sample_header_offset(B, V));
            if (P > 1) NVR_HIP_CHECK(hipMemsetAsync(hdrs, 0, sample_header_bytes(B), s));
            const dim3 grid((unsigned)(8 * rpx * P)), block(kThreads);
#define NVR_SROWS(E) sample_rows_kernel<E><<<grid, block, 0, s>>>(logits, (int)V, (int)B, P, rpx, chunk, temperature, top_k, top_p, keys, \
                                                                 out_ids, (float *)workspace, hdrs, store_filtered ? 1 : 0)
            if (ept <= 4) NVR_SROWS(4); else if (ept <= 20) NVR_SROWS(20); else NVR_SROWS(40);
#undef NVR_SROWS
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return nvr::fail(NVR_ERR_HIP, "sample launch failed: %s", hipGetErrorString(e));
            return 0;
        }
    }
    sample_kernel<<<dim3((unsigned)B), dim3(kThreads), 0, s>>>(logits, (int)V, temperature, top_k, top_p, keys, out_ids,
                                                              (float *)workspace);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return nvr::fail(NVR_ERR_HIP, "sample launch failed: %s", hipGetErrorString(e));
    return 0;
}

}}  // namespace nvr::k / nvr::kb (NVR_DT_NS)
