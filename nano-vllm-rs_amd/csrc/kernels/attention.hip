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
//    to VGPRs (no LDS round trip for a once-read stream), U such loads of K and of V in flight
//    (2U with PREFETCH: the next iteration's rows are requested before this iteration's math).
//  * q·k: v_dot2_f32_f16 on the lane's 8-element slice, then a butterfly over the D/8 lanes of the
//    row; softmax is online (running max / sum per wave, f32), p·v accumulates the lane's slice.
//  * split-KV: one workgroup = WAVES waves owns (query, kv head, partition); its waves interleave
//    row groups and merge through LDS.  With one partition the workgroup writes the fp16 result
//    itself; otherwise f32 partials go to a workspace and a second tiny kernel merges them.
//  * all G = H/KVH query heads of a kv head are processed together so K/V are read once.
#include <cstdio>
#include <cstdlib>
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

template <bool NT>
__device__ __forceinline__ half8_t load_row16(const half_t *p) {
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const half8_t *>(p));
    return *reinterpret_cast<const half8_t *>(p);
}

// sum over the LPR (= 16 or 8) consecutive lanes that hold one K row, result in every lane of the group: DPP row
// rotations / quad permutes fused into v_add_f32 (no LDS traffic; __shfl_xor compiles to ds_bpermute_b32)
template <int CTRL>
__device__ __forceinline__ float dpp_add(float x) {
    const int y = __builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false);
    return x + __int_as_float(y);
}
template <int LPR>
__device__ __forceinline__ float row_sum(float x) {
    if (LPR == 16) { x = dpp_add<0x128>(x); x = dpp_add<0x124>(x); }     // row_ror:8, row_ror:4
    else x = dpp_add<0x141>(x);                                            // row_half_mirror (8 lanes: i <-> 7-i)
    x = dpp_add<0x4E>(x);                                                  // quad_perm [2,3,0,1]
    x = dpp_add<0xB1>(x);                                                  // quad_perm [1,0,3,2]
    return x;
}

// U: row groups (wave-instructions of K and of V) per iteration; WAVES: waves per workgroup;
// PREFETCH: request the next iteration's K/V rows before computing on the current ones; NT: non-temporal loads.
template <int D, int G, bool PAGED, bool DIRECT_OUT, int U, int WAVES, bool PREFETCH, bool NT>
__global__ __launch_bounds__(WAVES * 64) void attn_rows_kernel(AttnParams p) {
    constexpr int LPR = D / 8;          // lanes per K/V row
    constexpr int RPI = 64 / LPR;       // rows per wave-instruction
    constexpr int TPI = RPI * U;        // tokens per wave iteration
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

    // element offset of row `tok` (clamped into the partition) of this lane's 16-byte slice
    auto row_off = [&](int tok) -> int64_t {
        const int tc = tok < pend ? tok : pend - 1;
        if (PAGED) {
            int bi, bo;
            if (p.bs_shift >= 0) { bi = tc >> p.bs_shift; bo = tc & (p.block_size - 1); }
            else { bi = tc / p.block_size; bo = tc - bi * p.block_size; }
            const int64_t row = (int64_t)bt[bi] * p.block_size + bo;
            return (row * p.KVH + g) * D + dc * 8;
        }
        return (base_row + tc) * p.ldkv + (int64_t)g * D + dc * 8;
    };

    constexpr int STRIDE = WAVES * TPI;
    int tb = p0 + wave * TPI;
    half8_t kn[U], vn[U];
    if (PREFETCH && tb < pend) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t off = row_off(tb + u * RPI + tg);
            kn[u] = load_row16<NT>(p.k + off); vn[u] = load_row16<NT>(p.v + off);
        }
    }
    for (; tb < pend; tb += STRIDE) {
        half8_t kk[U], vv[U];
        if (PREFETCH) {
#pragma unroll
            for (int u = 0; u < U; ++u) { kk[u] = kn[u]; vv[u] = vn[u]; }
            if (tb + STRIDE < pend) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int64_t off = row_off(tb + STRIDE + u * RPI + tg);
                    kn[u] = load_row16<NT>(p.k + off); vn[u] = load_row16<NT>(p.v + off);
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t off = row_off(tb + u * RPI + tg);
                kk[u] = load_row16<NT>(p.k + off); vv[u] = load_row16<NT>(p.v + off);
            }
        }
        float s[U][G];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool valid = (tb + u * RPI + tg) < pend;
#pragma unroll
            for (int i = 0; i < G; ++i) {
                float d = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    d = __builtin_amdgcn_fdot2((half2_t){kk[u][2 * j], kk[u][2 * j + 1]}, qv[i][j], d, false);
                d = row_sum<LPR>(d);
                s[u][i] = valid ? d * p.scale : -INFINITY;
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

    // merge the RPI row groups of the wave (m is wave-uniform), then the waves through LDS
    __shared__ float sm_acc[WAVES][G][D];
    __shared__ float sm_ml[WAVES][G][2];
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
    for (int idx = threadIdx.x; idx < G * D; idx += WAVES * 64) {
        const int i = idx / D, d = idx % D;
        float M = sm_ml[0][i][0];
#pragma unroll
        for (int w2 = 1; w2 < WAVES; ++w2) M = fmaxf(M, sm_ml[w2][i][0]);
        float o = 0.f, L = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < WAVES; ++w2) {
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

// ---- launch configuration ------------------------------------------------------------------------
struct Tune { int U, waves, prefetch, nt, parts; };          // parts: 0 = automatic
static Tune env_tune() {
    Tune v{0, 0, 0, 0, 0};
    if (const char *e = std::getenv("NVR_ATTN_TUNE")) std::sscanf(e, "%d,%d,%d,%d,%d", &v.U, &v.waves, &v.prefetch, &v.nt, &v.parts);
    return v;
}

static inline int parts_for(int64_t nq, int64_t KVH, int64_t max_ctx, int waves, int *part_size) {
    // aim at ~4096 waves over the 256 CUs; partitions are multiples of 64 tokens
    const int64_t want_wgs = 4096 / waves;
    int64_t want = (want_wgs + nq * KVH - 1) / (nq * KVH);
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

template <int D, int G, int U, int WAVES, bool PF, bool NT>
static void launch_cfg(const AttnParams &p, bool paged, bool direct, int64_t nwg, hipStream_t s) {
    dim3 grid((unsigned)nwg), block(WAVES * 64);
    if (paged) {
        if (direct) attn_rows_kernel<D, G, true, true, U, WAVES, PF, NT><<<grid, block, 0, s>>>(p);
        else attn_rows_kernel<D, G, true, false, U, WAVES, PF, NT><<<grid, block, 0, s>>>(p);
    } else {
        if (direct) attn_rows_kernel<D, G, false, true, U, WAVES, PF, NT><<<grid, block, 0, s>>>(p);
        else attn_rows_kernel<D, G, false, false, U, WAVES, PF, NT><<<grid, block, 0, s>>>(p);
    }
}

template <int D, int G>
static int launch_attn(const AttnArgs &a, bool paged, hipStream_t s) {
    AttnParams p{};
    p.q = (const half_t *)a.q; p.ldq = a.ldq; p.k = (const half_t *)a.k; p.v = (const half_t *)a.v; p.ldkv = a.ldkv;
    p.ctx_lens = a.ctx_lens; p.seq_of_q = a.seq_of_q; p.kv_base = a.kv_base; p.block_tables = a.block_tables;
    p.max_blocks = a.max_blocks; p.block_size = a.block_size;
    p.bs_shift = (a.block_size > 0 && (a.block_size & (a.block_size - 1)) == 0) ? __builtin_ctz(a.block_size) : -1;
    p.H = a.H; p.KVH = a.KVH; p.scale = a.scale; p.out = (half_t *)a.out;
    constexpr int DU = (D == 128) ? 4 : 2;           // default: 16 tokens per wave iteration
    const Tune tn = env_tune();
    // Geometry (measured on MI355X at B=32, ctx 1030, KVH=8, D=128, profiles/r01_attn_tune.txt): one
    // 16-wave workgroup per (query, kv head) with non-temporal K/V loads and no split (25.4 us, 5.3 TB/s)
    // beats 4-wave workgroups over 4 partitions + merge kernel (33.0 us).  So: when there is enough
    // work for >= ~192 sixteen-wave workgroups use them (splitting only to reach ~256 workgroups),
    // otherwise fall back to 4-wave workgroups over 64-token-granular partitions.
    const int64_t mc = a.max_ctx > 0 ? a.max_ctx : 1;
    const int64_t pairs = (int64_t)a.nq * a.KVH;
    int waves = tn.waves ? tn.waves : ((paged && a.workspace && pairs * ((mc + 255) / 256) >= 192) ? 16 : 4);
    int part_size = 0x3fffffff, np = 1;
    if (a.workspace) {
        if (tn.parts > 0) { part_size = (int)(((mc + tn.parts - 1) / tn.parts + 63) / 64 * 64); np = (int)((mc + part_size - 1) / part_size); }
        else if (waves == 16) {
            int64_t want = pairs >= 96 ? 1 : (256 + pairs - 1) / pairs;
            int64_t ps = ((mc + want - 1) / want + 255) / 256 * 256;
            part_size = (int)ps; np = (int)((mc + ps - 1) / ps);
        } else np = parts_for(a.nq, a.KVH, mc, waves, &part_size);
    }
    const bool direct = np <= 1;
    if (direct) { p.part_size = 0x3fffffff; p.num_parts = 1; }
    else {
        p.part_size = part_size; p.num_parts = np;
        p.part_o = (float *)a.workspace;
        p.part_ml = p.part_o + (int64_t)a.nq * a.H * np * D;
    }
    const int64_t nwg = (int64_t)p.num_parts * a.KVH * a.nq;
    bool done = false;
#ifdef NVR_ATTN_EXPERIMENTS
    if (D == 128 && G == 2 && paged && tn.U) {
#define NVR_TRY(UU, WW, PP, NN)                                                                          \
        if (!done && tn.U == UU && waves == WW && tn.prefetch == PP && tn.nt == NN) {                    \
            launch_cfg<D, G, UU, WW, PP != 0, NN != 0>(p, paged, direct, nwg, s); done = true; }
        NVR_TRY(8, 4, 0, 0) NVR_TRY(4, 4, 1, 0) NVR_TRY(4, 8, 0, 0) NVR_TRY(4, 16, 0, 0) NVR_TRY(4, 16, 1, 0)
        NVR_TRY(4, 4, 0, 1) NVR_TRY(4, 16, 0, 1) NVR_TRY(8, 16, 0, 0) NVR_TRY(4, 8, 1, 0) NVR_TRY(2, 16, 1, 0)
        NVR_TRY(2, 8, 1, 0) NVR_TRY(4, 16, 1, 1) NVR_TRY(4, 8, 1, 1) NVR_TRY(2, 4, 1, 0)
#undef NVR_TRY
        if (!done) return nvr::fail(NVR_ERR_UNSUPPORTED, "NVR_ATTN_TUNE names a variant that is not compiled in");
    }
#endif
    if (!done) {
        if (waves == 16) launch_cfg<D, G, DU, 16, false, true>(p, paged, direct, nwg, s);   // K/V streamed once: nt loads
        else if (paged) launch_cfg<D, G, DU, 4, false, true>(p, paged, direct, nwg, s);
        else launch_cfg<D, G, DU, 4, false, false>(p, paged, direct, nwg, s);             // prefill: rows re-read from L2
    }
    if (!direct)
        attn_merge_kernel<D><<<dim3((unsigned)((int64_t)a.H * a.nq)), dim3(D), 0, s>>>(p.part_o, p.part_ml, a.ctx_lens, a.H,
                                                                                  part_size, np, p.out);
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
