// attention.hip — paged decode attention (K9), varlen causal prefill attention (K7) and the
// prefix-cached paged prefill variant (K8) for gfx950.
// reference: Attention::flash_attention_decode / compute_attention_with_cache / gather_cached_kv,
// src/layers/attention.rs:225-235,264-318; flash_attention_varlen + causal mask :177-208,321-339;
// GQA mapping kv = h / (H/KVH), :419-435; scale 1/sqrt(D), :45.  Semantics per SURVEY.md A-8/A-9:
// query t sees exactly ctx_lens[t] keys, softmax in f32.
//
// The decode step is HBM-bound (114 688 B of K/V per cached token per step for
// Qwen3-0.6B, 76 % of all decode bytes at ctx 1024), so the kernel is built around the load path:
//  * K and V rows of one kv head are 2·D contiguous bytes; a wave reads whole rows, 16 B per lane,
//    D/8 lanes per row (4 rows = 1 KiB per wave-instruction at D=128) — full 128-B lines, straight
//    to VGPRs (no LDS round trip for a once-read stream), U such loads of K and of V in flight.
//  * q·k: v_dot2_f32_f16 on the lane's 8-element slice, then a butterfly over the D/8 lanes of the
//    row; softmax is online (running max / sum per wave, f32), p·v accumulates the lane's slice.
//  * split-KV: grid = (partitions, KVH, queries); the 4 waves of a workgroup interleave row groups of
//    one partition and merge through LDS; partitions are merged by a second tiny kernel.
//  * all G = H/KVH query heads of a kv head are processed together so K/V are read once.
#include "kernels.h"
#include "device_utils.h"
#include "../common.h"

namespace nvr { namespace k {

struct AttnParams {
    const half_t *q; int64_t ldq;
    const half_t *k, *v; int64_t ldkv;
    const int32_t *ctx_lens, *seq_of_q, *kv_base, *block_tables;
    int32_t max_blocks, block_size, bs_shift;
    int32_t H, KVH;
    float scale;
    int32_t part_size, num_parts;
    float *part_o; float *part_ml;     // [nq, H, num_parts, D], [nq, H, num_parts, 2]
    half_t *out;                       // [nq, H, D]
};

template <int D, int G, bool PAGED, bool DIRECT_OUT>
__global__ __launch_bounds__(256) void attn_rows_kernel(AttnParams p) {
    constexpr int LPR = D / 8;          // lanes per K/V row
    constexpr int RPI = 64 / LPR;       // rows per wave-instruction
    constexpr int U = (D == 128) ? 4 : 2;   // row-groups in flight: 16 tokens per wave iteration
    constexpr int TPI = RPI * U;
    const int part = blockIdx.x % p.num_parts, g = (blockIdx.x / p.num_parts) % p.KVH;
    const int t = blockIdx.x / (p.num_parts * p.KVH);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int dc = lane % LPR, tg = lane / LPR;

    const int ctx = p.ctx_lens[t];
    const int p0 = part * p.part_size;
    if (p0 >= ctx && !(DIRECT_OUT)) return;          // empty partition: the merge kernel skips it too
    const int pend = min(ctx, p0 + p.part_size);

    // q slice of this lane for the G heads of kv head g (fp16 pairs for v_dot2)
    half2_t qv[G][4];
    {
        const half_t *qrow = p.q + (int64_t)t * p.ldq + (int64_t)g * G * D + dc * 8;
#pragma unroll
        for (int i = 0; i < G; ++i) {
            half8_t h = *reinterpret_cast<const half8_t *>(qrow + i * D);
#pragma unroll
            for (int j = 0; j < 4; ++j) qv[i][j] = (half2_t){h[2 * j], h[2 * j + 1]};
        }
    }
    const int32_t *bt = PAGED ? p.block_tables + (int64_t)(p.seq_of_q ? p.seq_of_q[t] : t) * p.max_blocks : nullptr;
    const int64_t base_row = PAGED ? 0 : (int64_t)p.kv_base[t];

    float m[G], l[G], acc[G][8];
#pragma unroll
    for (int i = 0; i < G; ++i) {
        m[i] = -INFINITY; l[i] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
    }

    for (int tb = p0 + wave * TPI; tb < pend; tb += 4 * TPI) {
        half8_t kk[U], vv[U];
        bool valid[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int tok = tb + u * RPI + tg;
            valid[u] = tok < pend;
            const int tc = valid[u] ? tok : pend - 1;
            int64_t off;
            if (PAGED) {
                int bi, bo;
                if (p.bs_shift >= 0) { bi = tc >> p.bs_shift; bo = tc & (p.block_size - 1); }
                else { bi = tc / p.block_size; bo = tc - bi * p.block_size; }
                const int64_t row = (int64_t)bt[bi] * p.block_size + bo;
                off = (row * p.KVH + g) * D + dc * 8;
            } else {
                off = (base_row + tc) * p.ldkv + (int64_t)g * D + dc * 8;
            }
            kk[u] = *reinterpret_cast<const half8_t *>(p.k + off);
            vv[u] = *reinterpret_cast<const half8_t *>(p.v + off);
        }
        float s[U][G];
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int i = 0; i < G; ++i) {
                float d = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    d = __builtin_amdgcn_fdot2((half2_t){kk[u][2 * j], kk[u][2 * j + 1]}, qv[i][j], d, false);
#pragma unroll
                for (int o = 1; o < LPR; o <<= 1) d += __shfl_xor(d, o, 64);
                s[u][i] = valid[u] ? d * p.scale : -INFINITY;
            }
        }
#pragma unroll
        for (int i = 0; i < G; ++i) {
            float mx = s[0][i];
#pragma unroll
            for (int u = 1; u < U; ++u) mx = fmaxf(mx, s[u][i]);
#pragma unroll
            for (int o = LPR; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
            const float mn = fmaxf(m[i], mx);            // finite: token tb (u=0, tg=0) is valid
            const float alpha = __expf(m[i] - mn);       // m = -inf on first use -> 0
            m[i] = mn;
            float ps = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] *= alpha;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float pr = __expf(s[u][i] - mn);
                ps += pr;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] = fmaf(pr, (float)vv[u][j], acc[i][j]);
            }
            l[i] = l[i] * alpha + ps;
        }
    }

    // merge the RPI row groups of the wave (m is wave-uniform), then the 4 waves through LDS
    __shared__ float sm_acc[4][G][D];
    __shared__ float sm_ml[4][G][2];
#pragma unroll
    for (int i = 0; i < G; ++i) {
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1) {
            l[i] += __shfl_xor(l[i], o, 64);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] += __shfl_xor(acc[i][j], o, 64);
        }
        if (tg == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) sm_acc[wave][i][dc * 8 + j] = acc[i][j];
            if (dc == 0) { sm_ml[wave][i][0] = m[i]; sm_ml[wave][i][1] = l[i]; }
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < G * D; idx += 256) {
        const int i = idx / D, d = idx % D;
        float M = sm_ml[0][i][0];
#pragma unroll
        for (int w2 = 1; w2 < 4; ++w2) M = fmaxf(M, sm_ml[w2][i][0]);
        float o = 0.f, L = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) {
            const float mw = sm_ml[w2][i][0];
            const float wgt = (mw == -INFINITY) ? 0.f : __expf(mw - M);
            o += wgt * sm_acc[w2][i][d];
            L += wgt * sm_ml[w2][i][1];
        }
        const int h = g * G + i;
        if (DIRECT_OUT) {
            p.out[((int64_t)t * p.H + h) * D + d] = (half_t)(L > 0.f ? o / L : 0.f);
        } else {
            const int64_t slot = ((int64_t)t * p.H + h) * p.num_parts + part;
            p.part_o[slot * D + d] = o;
            if (d == 0) { p.part_ml[slot * 2] = M; p.part_ml[slot * 2 + 1] = L; }
        }
    }
}

// merge split-KV partitions: out = sum_p e^(m_p-M) o_p / sum_p e^(m_p-M) l_p
template <int D>
__global__ void attn_merge_kernel(const float *__restrict__ part_o, const float *__restrict__ part_ml,
                                  const int32_t *__restrict__ ctx_lens, int H, int part_size, int num_parts,
                                  half_t *__restrict__ out) {
    const int h = blockIdx.x % H, t = blockIdx.x / H, d = threadIdx.x;
    const int np = min(num_parts, (ctx_lens[t] + part_size - 1) / part_size);
    const int64_t base = ((int64_t)t * H + h) * num_parts;
    float M = -INFINITY;
    for (int i = 0; i < np; ++i) M = fmaxf(M, part_ml[(base + i) * 2]);
    float o = 0.f, L = 0.f;
    for (int i = 0; i < np; ++i) {
        const float w = __expf(part_ml[(base + i) * 2] - M);
        o += w * part_o[(base + i) * D + d];
        L += w * part_ml[(base + i) * 2 + 1];
    }
    out[((int64_t)t * H + h) * D + d] = (half_t)(L > 0.f ? o / L : 0.f);
}

static inline int parts_for(int64_t nq, int64_t KVH, int64_t max_ctx, int *part_size) {
    // aim at >= ~1024 workgroups (4 per CU); partitions are multiples of 64 tokens (4 waves x 16)
    int64_t want = (1024 + nq * KVH - 1) / (nq * KVH);
    int64_t max_parts = (max_ctx + 63) / 64;
    if (want > max_parts) want = max_parts;
    if (want < 1) want = 1;
    int64_t ps = ((max_ctx + want - 1) / want + 63) / 64 * 64;
    if (ps < 64) ps = 64;
    *part_size = (int)ps;
    return (int)((max_ctx + ps - 1) / ps);
}

size_t attn_workspace_bytes(int64_t nq, int64_t H, int64_t D, int64_t max_ctx) {
    int64_t max_parts = (max_ctx + 63) / 64;
    if (max_parts > 1024) max_parts = 1024;
    if (max_parts < 1) max_parts = 1;
    return (size_t)(nq * H * max_parts * (D + 2) * sizeof(float));
}

template <int D, int G>
static int launch_attn(const AttnArgs &a, bool paged, hipStream_t s) {
    AttnParams p{};
    p.q = (const half_t *)a.q; p.ldq = a.ldq; p.k = (const half_t *)a.k; p.v = (const half_t *)a.v; p.ldkv = a.ldkv;
    p.ctx_lens = a.ctx_lens; p.seq_of_q = a.seq_of_q; p.kv_base = a.kv_base; p.block_tables = a.block_tables;
    p.max_blocks = a.max_blocks; p.block_size = a.block_size;
    p.bs_shift = (a.block_size > 0 && (a.block_size & (a.block_size - 1)) == 0) ? __builtin_ctz(a.block_size) : -1;
    p.H = a.H; p.KVH = a.KVH; p.scale = a.scale; p.out = (half_t *)a.out;
    int part_size = 0;
    int np = a.workspace ? parts_for(a.nq, a.KVH, a.max_ctx > 0 ? a.max_ctx : 1, &part_size) : 1;
    if (np <= 1) {
        p.part_size = 0x3fffffff; p.num_parts = 1;
        dim3 grid((unsigned)((int64_t)a.KVH * a.nq));
        if (paged) attn_rows_kernel<D, G, true, true><<<grid, dim3(256), 0, s>>>(p);
        else attn_rows_kernel<D, G, false, true><<<grid, dim3(256), 0, s>>>(p);
    } else {
        p.part_size = part_size; p.num_parts = np;
        p.part_o = (float *)a.workspace;
        p.part_ml = p.part_o + (int64_t)a.nq * a.H * np * D;
        dim3 grid((unsigned)((int64_t)np * a.KVH * a.nq));
        if (paged) attn_rows_kernel<D, G, true, false><<<grid, dim3(256), 0, s>>>(p);
        else attn_rows_kernel<D, G, false, false><<<grid, dim3(256), 0, s>>>(p);
        attn_merge_kernel<D><<<dim3((unsigned)((int64_t)a.H * a.nq)), dim3(D), 0, s>>>(p.part_o, p.part_ml, a.ctx_lens, a.H,
                                                                                  part_size, np, p.out);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return nvr::fail(NVR_ERR_HIP, "attention launch failed: %s", hipGetErrorString(e));
    return 0;
}

int attention(const AttnArgs &a, bool paged, hipStream_t s) {
    if (a.nq == 0) return 0;
    if (a.H % a.KVH) return nvr::fail(NVR_ERR_INVALID_ARG, "attention: H=%d not a multiple of KVH=%d", a.H, a.KVH);
    const int G = a.H / a.KVH;
#define NVR_ATTN_CASE(DD, GG) if (a.D == DD && G == GG) return launch_attn<DD, GG>(a, paged, s);
    NVR_ATTN_CASE(128, 1) NVR_ATTN_CASE(128, 2) NVR_ATTN_CASE(128, 4) NVR_ATTN_CASE(128, 8)
    NVR_ATTN_CASE(64, 1) NVR_ATTN_CASE(64, 2) NVR_ATTN_CASE(64, 4) NVR_ATTN_CASE(64, 8)
#undef NVR_ATTN_CASE
    return nvr::fail(NVR_ERR_UNSUPPORTED, "attention: unsupported head_dim=%d / group=%d (D in {64,128}, G in {1,2,4,8})",
                     a.D, G);
}

}}  // namespace nvr::k
