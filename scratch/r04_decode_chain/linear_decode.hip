// linear_decode.hip — the decode step's GEMM chain in FOUR launches per layer instead of six:
//   K_B  qkv      : RMSNorm(h)·W_qkvᵀ -> RoPE -> KV store          (input norm folded into the GEMM's prologue)
//   K_A  o_proj   : h <- fp16(h + fp16(attn·W_oᵀ))                  (split-k over S workgroups; the LAST ARRIVER of a tile
//   K_B  gate_up  : SiluAndMul(RMSNorm(h)·W_guᵀ)                      sums the f32 slabs in fixed order and adds the residual)
//   K_A  down_proj: h <- fp16(h + fp16(act·W_dᵀ))
// reference call sites: Qwen3DecoderLayer::forward src/models/qwen3.rs:372-392 (norm :378,:385; residual :382,:389),
// Qwen3Attention :208-240, Qwen3MLP :305-314; RMSNorm::forward_simple src/layers/layernorm.rs:58-75; RowParallelLinear
// src/layers/linear.rs:228-239; QKVParallelLinear :354-356; MergedColumnParallelLinear :437-439; RoPE
// src/layers/rotary_embedding.rs:23-48; store_kv_cache src/layers/attention.rs:150-174; SiluAndMul activation.rs:46-63.
//
// Why: every kernel of the chain is latency-bound (4-13 MB of weights, ~4.7 us of wall clock each incl. its boundary,
// profiles/r01_decode_step_breakdown_final.txt); the two add+RMSNorm launches per layer moved 0.6 MB each.  Here
//  * the residual add rides on the split-k reduction it follows: each k-slice workgroup publishes its f32 partial tile
//    write-through (sc1), drains, and draws a ticket from the tile's counter; the workgroup that draws S-1 reads the S slabs
//    back with sc1 loads, sums them in slab order (bit-identical to add_rmsnorm_slabs), adds the residual tile and writes h.
//    Nobody polls: the other workgroups exit (cdna_hip_programming.md §5 "In-launch split-K reduction").
//  * the RMSNorm rides in the prologue of the GEMM that consumes it: every workgroup already reads the whole [T, K] block of
//    its B operand; it now reads h instead of the normalised rows, reduces the row sums of squares across its waves through
//    LDS while its weight loads are in flight, and normalises its fragments in registers:
//    n = fp16(fp16-exact(h) * (1/rms) * w) — the oracle's rounding point (fp16 n) is kept, the quotient h/rms is formed as
//    h * (1/rms) (one IEEE division per row; normalised rows within 1 fp16 ulp of the oracle's, DESIGN A-25).
// Weight streaming itself is the r01 kernel's: v_mfma_f32_16x16x32_f16 with the 16-row weight tile as the A operand (16 B per
// lane straight from HBM, non-temporal), tokens as B, k interleaved over the waves of a workgroup, f32 partials reduced
// through LDS in wave order (same accumulation order as linear.hip => same bits for the same inputs).
#include <cstdlib>
#include "kernels.h"
#include "device_utils.h"
#include "../common.h"

namespace nvr { namespace NVR_DT_NS {

enum { DEPI_SILU = 0, DEPI_ROPE = 1, DEPI_RESID = 2 };

struct DecEpi {
    // DEPI_ROPE
    const int64_t *pos; const int32_t *slots; const float *cos_t, *sin_t;
    half_t *kc, *vc;
    int32_t H, KVH, D;
    // DEPI_RESID: blockIdx.z owns k in [z*kslice, (z+1)*kslice); S = gridDim.z slabs of [T][N] f32; cnt[tile] tickets
    int32_t kslice; int64_t slab_stride; float *slabs; half_t *h; unsigned int *cnt;
    // NORM prologue: x is the residual stream h; wn the RMSNorm weight
    const half_t *wn; float eps;
    // W is the tiled copy [N/16][K/32][16][32] (retile_weight; DEPI_ROPE: mode 1, tiles already in dec_w_row order)
    int32_t tiled;
};

template <int EPI>
__device__ __forceinline__ int dec_w_row(int bx, int i, int NT, int r, int N, const DecEpi &e) {
    if (EPI == DEPI_SILU) return (i == 0 ? 0 : N) + bx * 16 + r;            // N == I: gate rows [0,I), up rows [I,2I)
    if (EPI == DEPI_ROPE) {
        const int tph = e.D / 16, head = bx / tph, c = bx % tph;            // rotation partners x1[j], x2[j] in one tile
        if (head < e.H + e.KVH) return head * e.D + (r < 8 ? c * 8 + r : e.D / 2 + c * 8 + (r - 8));
        return head * e.D + c * 16 + r;
    }
    const int n = (bx * NT + i) * 16 + r;
    return n < N ? n : N - 1;
}

// normalise 8 fp16 elements: n = fp16(h * inv * w).  Contraction is allowed here (v_fma_mix_f32 / v_fma_mixlo_f16 read the fp16
// operands directly: two instructions per element): the product h*inv is rounded to f32, (that)*w once to fp16 — within 1 fp16
// ulp of the oracle's (h / rms) * w, like the division-free quotient itself (DESIGN A-25).
__device__ __forceinline__ half8_t norm8(half8_t v, float inv, half8_t g) {
#pragma clang fp contract(fast)
    half8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float t = (float)v[e] * inv; o[e] = (half_t)(t * (float)g[e]); }
    return o;
}
__device__ __forceinline__ float sumsq8(half8_t v, float s) {
#pragma clang fp contract(fast)
#pragma unroll
    for (int e = 0; e < 8; ++e) s += (float)v[e] * (float)v[e];
    return s;
}

// KI: k-steps of 32 per wave (all in flight at once); the k range of the workgroup is <= 32*WAVES*KI (EXACT: equal, no guards).
// RT (DEPI_RESID, no k split): weight rows per workgroup, 16 (a whole MFMA tile), 8 or 4 (the MFMA's rows RT..15 repeat rows
// 0..RT-1 — same addresses, no extra traffic — and are dropped), so that the N = hidden GEMMs reach N/RT workgroups without
// cutting k: no slabs, no tickets, nothing to hand over inside the launch.
template <int NT, int MT, int WAVES, int KI, int EPI, bool NORM, bool EXACT, int RT>
__global__ __launch_bounds__(WAVES * 64) void decode_linear_kernel(const half_t *__restrict__ x, int64_t ldx,
                                                                   const half_t *__restrict__ W, int T, int K, int N,
                                                                   half_t *__restrict__ y, DecEpi epi) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.y * (16 * MT);
    constexpr int KS = 32 * WAVES;
    const int kbeg = EPI == DEPI_RESID ? blockIdx.z * epi.kslice : 0;
    const int kend = EPI == DEPI_RESID ? min(K, kbeg + epi.kslice) : K;

    const half_t *wrow[NT];
    const half_t *xrow[MT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        if (epi.tiled) {          // one 16x32 operand tile = 1 KiB contiguous; consecutive k tiles of a row tile follow each other
            constexpr int PER = 16 / RT;                     // workgroups sharing one 16-row tile
            const int tile = RT < 16 ? blockIdx.x / PER : EPI == DEPI_SILU ? (i == 0 ? 0 : N / 16) + blockIdx.x : EPI == DEPI_ROPE ? blockIdx.x : blockIdx.x * NT + i;
            const int rr = RT < 16 ? (blockIdx.x % PER) * RT + (r & (RT - 1)) : r;
            wrow[i] = W + (int64_t)tile * (K / 32) * 512 + rr * 32 + q * 8;
        } else {
            const int row = RT < 16 ? min(blockIdx.x * RT + (r & (RT - 1)), N - 1) : dec_w_row<EPI>(blockIdx.x, i, NT, r, N, epi);
            wrow[i] = W + (int64_t)row * K + q * 8;
        }
    }
    const int kmul = epi.tiled ? 16 : 1;                     // k advances 32 halfs in a row, 512 halfs (one tile) in the tiled copy
#pragma unroll
    for (int j = 0; j < MT; ++j) { int m = m0 + j * 16 + r; if (m > T - 1) m = T - 1; xrow[j] = x + (int64_t)m * ldx + q * 8; }

    // Request order = the order the data is needed in (s_waitcnt vmcnt counts in issue order): the RoPE position, the activation
    // rows and norm weights (L2), the residual tile, then the weights (HBM) — all back to back, nothing waited for in between:
    // the norm prologue then runs under the weight stream's latency.
    float4_t rope_cs = (float4_t){0.f, 0.f, 0.f, 0.f}, rope_sn = rope_cs;
    int rope_slot = -1;
    int64_t rope_pos = 0;
    // every wave requests a position and (below) its cos / sin row — waves >= MT redundantly — so that no load sits under a
    // divergent branch: s_waitcnt counts at a join assume the shorter queue and over-wait on the longer one (r02: the norm
    // prologue of the two epilogue waves waited for two weight loads)
    if (EPI == DEPI_ROPE) {
        const int m = m0 + (wave % MT) * 16 + r, mc = m < T ? m : T - 1;
        rope_pos = epi.pos[mc];
        if (wave < MT && blockIdx.x / (epi.D / 16) >= epi.H && epi.slots && m < T) rope_slot = epi.slots[m];
    }
    half8_t a[KI][NT], b[KI][MT], gw[NORM ? KI : 1];
#pragma unroll
    for (int u = 0; u < KI; ++u) {
        const int kk = kbeg + wave * 32 + u * KS;
#pragma unroll
        for (int j = 0; j < MT; ++j) b[u][j] = (EXACT || kk < kend) ? *reinterpret_cast<const half8_t *>(xrow[j] + kk) : (half8_t)(half_t)0;
        if constexpr (NORM) gw[u] = (EXACT || kk < kend) ? *reinterpret_cast<const half8_t *>(epi.wn + kk + q * 8) : (half8_t)(half_t)0;
    }
    // DEPI_RESID: the residual tile this workgroup adds IF it turns out to be the tile's last arriver (nobody else writes it)
    constexpr int TT = (NT * MT + WAVES - 1) / WAVES;          // output tiles per wave in the epilogue
    half4_t hres[TT];
    if (EPI == DEPI_RESID) {
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            const int tile = wave + tt * WAVES;
            const int i = tile / MT, j = tile % MT;
            const int n = RT < 16 ? blockIdx.x * RT + (q & (RT / 4 - 1)) * 4 : (blockIdx.x * NT + i) * 16 + q * 4, m = m0 + j * 16 + r;
            hres[tt] = (tile < NT * MT && m < T && n < N) ? *reinterpret_cast<const half4_t *>(epi.h + (int64_t)m * N + n)
                                                          : (half4_t){(half_t)0, (half_t)0, (half_t)0, (half_t)0};
        }
    }
#pragma unroll
    for (int u = 0; u < KI; ++u) {
        const int kk = kbeg + wave * 32 + u * KS;
#pragma unroll
        for (int i = 0; i < NT; ++i)
            a[u][i] = (EXACT || kk < kend) ? __builtin_nontemporal_load(reinterpret_cast<const half8_t *>(wrow[i] + (int64_t)kk * kmul)) : (half8_t)(half_t)0;
    }
    // every request is out before anything is waited for (hipcc otherwise sinks loads below the first use of an earlier one:
    // weight loads behind the norm's waits, or one HBM round trip per MFMA)
    __builtin_amdgcn_sched_barrier(0);
    // RoPE cos / sin rows: their address hangs on the position loaded first, so they are requested only now — behind the
    // weights in issue order (waiting for the position BEFORE the weight requests would start the HBM stream one L2 round trip
    // late: +1.0 us on the qkv launch, r02).  The position is the oldest request and arrives with the activation rows the norm
    // prologue waits for anyway; cos / sin then fly under the prologue, the weight wait, the MFMAs and the LDS reduction.
    if (EPI == DEPI_ROPE) {                                   // (v heads load a row too and never use it)
        const int c = blockIdx.x % (epi.D / 16), half_d = epi.D / 2, jj = c * 8 + (q & 1) * 4;
        rope_cs = *reinterpret_cast<const float4_t *>(epi.cos_t + rope_pos * half_d + jj);
        rope_sn = *reinterpret_cast<const float4_t *>(epi.sin_t + rope_pos * half_d + jj);
        __builtin_amdgcn_sched_barrier(0);
    }

    __shared__ float4_t part[WAVES][NT * MT][64];
    __shared__ float red[WAVES][MT * 16];
    __shared__ unsigned int ticket_s;

    if constexpr (NORM) {
        // RMSNorm::forward_simple, layernorm.rs:58-75: rms = sqrt(mean(x^2) + eps); the row sum crosses the 4 k-quads of a
        // wave on lane swaps and the waves through LDS (fixed order: every workgroup of the launch forms the same rms)
        float ss[MT];
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            float s = 0.f;
#pragma unroll
            for (int u = 0; u < KI; ++u) s = sumsq8(b[u][j], s);
            ss[j] = xor32_partner_sum(xor16_partner_sum(s));
        }
        if (q == 0) {
#pragma unroll
            for (int j = 0; j < MT; ++j) red[wave][j * 16 + r] = ss[j];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            float tot = red[0][j * 16 + r];
#pragma unroll
            for (int w2 = 1; w2 < WAVES; ++w2) tot += red[w2][j * 16 + r];
            const float inv = __fdiv_rn(1.0f, sqrtf(tot / (float)K + epi.eps));
#pragma unroll
            for (int u = 0; u < KI; ++u) b[u][j] = norm8(b[u][j], inv, gw[u]);
        }
    }

    float4_t acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < KI; ++u)
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j)
                acc[i][j] = mfma16(a[u][i], b[u][j], acc[i][j]);

    // cross-wave reduction through LDS (wave order)
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) part[wave][i * MT + j][lane] = acc[i][j];
    __syncthreads();
    auto reduce = [&](int tile) {
        float4_t s = part[0][tile][lane];
#pragma unroll
        for (int w2 = 1; w2 < WAVES; ++w2) { float4_t pz = part[w2][tile][lane]; s += pz; }
        return s;
    };

    // C layout of the 16x16 MFMA: row (n) = q*4 + reg, col (token) = r
    if (EPI == DEPI_RESID) {
        const int S = gridDim.z;
        typedef unsigned int u4 __attribute__((ext_vector_type(4)));
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(epi.slabs, 0, (int)(S * epi.slab_stride * 4), 0x00020000);
        if (S == 1) {                                   // no k split: the accumulator is the GEMM row
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) {
                const int tile = wave + tt * WAVES;
                if (tile >= NT * MT) continue;
                const float4_t s = reduce(tile);
                const int i = tile / MT, j = tile % MT;
                const int n = RT < 16 ? blockIdx.x * RT + q * 4 : (blockIdx.x * NT + i) * 16 + q * 4, m = m0 + j * 16 + r;
                if (m < T && n < N && q < RT / 4) {
                    half4_t o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = to_half_rn((float)hres[tt][e] + (float)to_half_rn(s[e]));
                    *reinterpret_cast<half4_t *>(epi.h + (int64_t)m * N + n) = o;
                }
            }
            return;
        }
        // (1) publish the partial tile write-through (sc1: no release fence, cdna guide G16 R1) ...
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            const int tile = wave + tt * WAVES;
            if (tile >= NT * MT) continue;
            const float4_t sv = reduce(tile);
            const int i = tile / MT, j = tile % MT;
            const int n = (blockIdx.x * NT + i) * 16 + q * 4, m = m0 + j * 16 + r;
            if (m < T && n < N)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, sv), rs,
                                                       (int)((blockIdx.z * epi.slab_stride + (int64_t)m * N + n) * 4), 0, 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // ... every storing wave drains its stores ...
        __syncthreads();
        unsigned int *cnt = epi.cnt + blockIdx.y * gridDim.x + blockIdx.x;
        if (threadIdx.x == 0) ticket_s = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... one ticket
        __syncthreads();
        if (ticket_s != (unsigned)(S - 1)) return;                       // not the last k-slice of this tile: done
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");           // keeps the sc1 loads below the ticket
        // (2) last arriver: y = fp16(slab_0 + slab_1 + ...) in slab order, h <- fp16(h + y)
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            const int tile = wave + tt * WAVES;
            if (tile >= NT * MT) continue;
            const int i = tile / MT, j = tile % MT;
            const int n = (blockIdx.x * NT + i) * 16 + q * 4, m = m0 + j * 16 + r;
            if (m < T && n < N) {
                float4_t pz[4];                                       // S <= 4: all slab loads in flight together
#pragma unroll
                for (int z = 0; z < 4; ++z)
                    pz[z] = z < S ? __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(
                                        rs, (int)((z * epi.slab_stride + (int64_t)m * N + n) * 4), 0, 16))
                                  : (float4_t){0.f, 0.f, 0.f, 0.f};
                float4_t sum = pz[0];
#pragma unroll
                for (int z = 1; z < 4; ++z) if (z < S) sum += pz[z];
                half4_t o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = to_half_rn((float)hres[tt][e] + (float)to_half_rn(sum[e]));
                *reinterpret_cast<half4_t *>(epi.h + (int64_t)m * N + n) = o;
            }
        }
        if (threadIdx.x == 0) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm for the next launch
    } else if (EPI == DEPI_SILU) {
        // tiles (0,j) = gate, (1,j) = up for the same 16 columns: act = fp16(silu(fp16 g) * fp16 u)
        for (int j = wave; j < MT; j += WAVES) {
            const float4_t g4 = reduce(j), u4 = reduce(MT + j);
            const int n = blockIdx.x * 16 + q * 4, m = m0 + j * 16 + r;
            if (m < T) {
                half4_t h;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float gf = (float)to_half_rn(g4[e]), uf = (float)to_half_rn(u4[e]);
                    const float sg = sigmoid_fast(gf);
                    h[e] = to_half_rn(__fmul_rn(__fmul_rn(gf, sg), uf));
                }
                *reinterpret_cast<half4_t *>(y + (int64_t)m * N + n) = h;
            }
        }
    } else {   // DEPI_ROPE: NT == 1, MT <= WAVES
        const int tph = epi.D / 16, head = blockIdx.x / tph, c = blockIdx.x % tph, half_d = epi.D / 2;
        const int64_t ldq = (int64_t)(epi.H + 2 * epi.KVH) * epi.D;
        if (wave < MT) {
            const int j = wave;
            const float4_t s = reduce(j);
            const int m = m0 + j * 16 + r;
            float v[4], pv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = (float)to_half_rn(s[e]); pv[e] = __shfl_xor(v[e], 32, 64); }
            half4_t h;
            int col;                                           // first of the lane's 4 consecutive head columns
            if (head < epi.H + epi.KVH) {
                const int jj = c * 8 + (q & 1) * 4;            // index inside the half dimension
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // rotary_embedding.rs:36-44: out1 = x1*c - x2*s ; out2 = x2*c + x1*s
                    h[e] = (q < 2) ? to_half_rn(__fsub_rn(__fmul_rn(v[e], rope_cs[e]), __fmul_rn(pv[e], rope_sn[e])))
                                   : to_half_rn(__fadd_rn(__fmul_rn(v[e], rope_cs[e]), __fmul_rn(pv[e], rope_sn[e])));
                }
                col = (q < 2) ? jj : half_d + jj;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = to_half_rn(s[e]);
                col = c * 16 + q * 4;
            }
            if (m < T) {
                *reinterpret_cast<half4_t *>(y + (int64_t)m * ldq + head * epi.D + col) = h;
                if (rope_slot >= 0 && head >= epi.H) {
                    const bool is_k = head < epi.H + epi.KVH;
                    const int kvh = is_k ? head - epi.H : head - epi.H - epi.KVH;
                    half_t *dst = (is_k ? epi.kc : epi.vc) + ((int64_t)rope_slot * epi.KVH + kvh) * epi.D + col;
                    *reinterpret_cast<half4_t *>(dst) = h;
                }
            }
        }
    }
}

static int dec_launch_check(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return nvr::fail(NVR_ERR_HIP, "%s launch failed: %s", what, hipGetErrorString(e));
    return 0;
}

template <int NT, int MT, int WAVES, int KI, int EPI, bool NORM, int RT = 16>
static void dec_launch(const half_t *x, int64_t ldx, const half_t *W, int T, int K, int N, half_t *y, const DecEpi &e, unsigned gx,
                       unsigned gz, hipStream_t s) {
    dim3 grid(gx, (unsigned)((T + 16 * MT - 1) / (16 * MT)), gz);
    const int kr = EPI == DEPI_RESID ? e.kslice : K;
    if (kr == 32 * WAVES * KI) decode_linear_kernel<NT, MT, WAVES, KI, EPI, NORM, true, RT><<<grid, dim3(WAVES * 64), 0, s>>>(x, ldx, W, T, K, N, y, e);
    else decode_linear_kernel<NT, MT, WAVES, KI, EPI, NORM, false, RT><<<grid, dim3(WAVES * 64), 0, s>>>(x, ldx, W, T, K, N, y, e);
}

// (waves, k-steps per wave) for a workgroup k range: the r01 geometry (8 waves at K = 1024, 16 at 2048, 4-wave k-slices)
#define NVR_DEC_GEOM(KR, BODY)                                                   \
    do {                                                                         \
        if ((KR) <= 256) { BODY(4, 2); }                                         \
        else if ((KR) <= 512) { BODY(4, 4); }                                    \
        else if ((KR) <= 768) { BODY(4, 6); }                                    \
        else if ((KR) <= 1024) { BODY(8, 4); }                                   \
        else { BODY(16, 4); }                                                    \
    } while (0)

bool decode_chain_ok(int64_t T, int64_t Hd, int64_t qkv_rows, int64_t I, int64_t D) {
    return T >= 1 && T <= 64 && Hd % 32 == 0 && Hd <= 2048 && qkv_rows % 16 == 0 && I % 16 == 0 && D % 16 == 0;
}

int decode_splitk_slices(int64_t T, int64_t K, int64_t N) {
    // reach ~256 workgroups with k-slices of >= 256 columns that are multiples of 64 (the r01 rule of row_parallel_norm)
    int64_t S = 1;
    const int64_t tiles = (N / 16) * ((T + 31) / 32);
    while (S < 4 && tiles * S < 256 && K % (32 * S * 2) == 0 && K / (S * 2) >= 128) S *= 2;
    return (int)S;
}

// h[T,N] <- fp16(h + fp16(x[T,K]·W[N,K]ᵀ)); S k-slices, slabs [S][T][N] f32, cnt: one zeroed counter per (column tile, token tile)
int linear_resid(const half_bits *x, int64_t ldx, const half_bits *W, int64_t T, int64_t K, int64_t N, int64_t S, float *slabs,
                 unsigned int *cnt, half_bits *h, hipStream_t s, const half_bits *Wt) {
    const bool half_tiles = S == 0;            // S = 0: 8-row weight tiles, no k split, no slabs, no tickets
    if (half_tiles) S = 1;
    if (S < 1 || S > 4 || K % (32 * S) || N % 16 || ldx % 8 || T > 64 || (!half_tiles && K / S > 2048))
        return nvr::fail(NVR_ERR_UNSUPPORTED, "linear_resid: K=%ld S=%ld N=%ld T=%ld ldx=%ld", (long)K, (long)S, (long)N, (long)T, (long)ldx);
    if (T == 0) return 0;
    DecEpi e{};
    e.kslice = (int32_t)(K / S); e.slab_stride = T * N; e.slabs = slabs; e.h = (half_t *)h; e.cnt = cnt;
    if (Wt && N % 16 == 0) { W = Wt; e.tiled = 1; }
    if (half_tiles) {
        // 8-row weight tiles, 16 tokens per workgroup, no k split: N/8 x ceil(T/16) workgroups, no slabs, no tickets.
        // (4-row tiles with both token tiles in one workgroup — every weight byte requested once — measured 6.98 / 9.25 us
        // against 5.18 / 6.50 us for o_proj / down_proj: each workgroup then pulls the WHOLE [32, K] activation block, 128-192 KB,
        // through its CU's 64 B/clk L1 fill path; profiles/r02_decode_chain_ablation.txt)
        if (N % 8 || K > 4096) return nvr::fail(NVR_ERR_UNSUPPORTED, "linear_resid (row tiles): K=%ld N=%ld", (long)K, (long)N);
        e.kslice = (int32_t)K;
        const unsigned gx8 = (unsigned)(N / 8);
#define HBODY(WV, KI_) dec_launch<1, 1, WV, KI_, DEPI_RESID, false, 8>((const half_t *)x, ldx, (const half_t *)W, (int)T, (int)K, (int)N, nullptr, e, gx8, 1, s)
        if (K <= 512) HBODY(4, 4);
        else if (K <= 1024) HBODY(8, 4);
        else if (K <= 2048) HBODY(16, 4);
        else if (K <= 3072) HBODY(16, 6);
        else HBODY(16, 8);
#undef HBODY
        return dec_launch_check("linear_resid (row tiles)");
    }
    const unsigned gx = (unsigned)(N / 16);
    const int64_t kr = K / S;
#define BODY(WV, KI_)                                                                                                              \
    if (T <= 16) dec_launch<1, 1, WV, KI_, DEPI_RESID, false>((const half_t *)x, ldx, (const half_t *)W, (int)T, (int)K, (int)N, nullptr, e, gx, (unsigned)S, s); \
    else dec_launch<1, 2, WV, KI_, DEPI_RESID, false>((const half_t *)x, ldx, (const half_t *)W, (int)T, (int)K, (int)N, nullptr, e, gx, (unsigned)S, s)
    NVR_DEC_GEOM(kr, BODY);
#undef BODY
    return dec_launch_check("linear_resid");
}

// act[T,I] = SiluAndMul(RMSNorm(h; wn)·W[2I,K]ᵀ)
int linear_silu_mul_normed(const half_bits *h, int64_t ldx, const half_bits *wn, float eps, const half_bits *W, int64_t T, int64_t K,
                           int64_t I, half_bits *out, hipStream_t s, const half_bits *Wt) {
    if (K % 32 || I % 16 || ldx % 8 || T > 64 || K > 2048)
        return nvr::fail(NVR_ERR_UNSUPPORTED, "linear_silu_mul_normed: K=%ld I=%ld T=%ld", (long)K, (long)I, (long)T);
    if (T == 0) return 0;
    DecEpi e{};
    e.wn = (const half_t *)wn; e.eps = eps;
    if (Wt) { W = Wt; e.tiled = 1; }
    const unsigned gx = (unsigned)(I / 16);
#define BODY(WV, KI_)                                                                                                              \
    if (T <= 16) dec_launch<2, 1, WV, KI_, DEPI_SILU, true>((const half_t *)h, ldx, (const half_t *)W, (int)T, (int)K, (int)I, (half_t *)out, e, gx, 1, s); \
    else dec_launch<2, 2, WV, KI_, DEPI_SILU, true>((const half_t *)h, ldx, (const half_t *)W, (int)T, (int)K, (int)I, (half_t *)out, e, gx, 1, s)
    NVR_DEC_GEOM(K, BODY);
#undef BODY
    return dec_launch_check("linear_silu_mul_normed");
}

// qkv[T,(H+2KVH)D] = RoPE(RMSNorm(h; wn)·Wᵀ) (+ k, v rows stored at slots)
int linear_qkv_rope_store_normed(const half_bits *h, int64_t ldx, const half_bits *wn, float eps, const half_bits *W, int64_t T, int64_t K,
                                 int64_t H, int64_t KVH, int64_t D, const int64_t *positions, const int32_t *slots, const float *cos_t,
                                 const float *sin_t, half_bits *qkv, half_bits *k_cache, half_bits *v_cache, hipStream_t s,
                                 const half_bits *Wt) {
    if (K % 32 || D % 16 || ldx % 8 || T > 64 || K > 2048)
        return nvr::fail(NVR_ERR_UNSUPPORTED, "linear_qkv_rope_store_normed: K=%ld D=%ld T=%ld", (long)K, (long)D, (long)T);
    if (T == 0) return 0;
    DecEpi e{};
    e.pos = positions; e.slots = slots; e.cos_t = cos_t; e.sin_t = sin_t; e.kc = (half_t *)k_cache; e.vc = (half_t *)v_cache;
    e.H = (int32_t)H; e.KVH = (int32_t)KVH; e.D = (int32_t)D;
    e.wn = (const half_t *)wn; e.eps = eps;
    if (Wt) { W = Wt; e.tiled = 1; }
    const int N = (int)((H + 2 * KVH) * D);
    const unsigned gx = (unsigned)(N / 16);
#define BODY(WV, KI_)                                                                                                              \
    if (T <= 16) dec_launch<1, 1, WV, KI_, DEPI_ROPE, true>((const half_t *)h, ldx, (const half_t *)W, (int)T, (int)K, N, (half_t *)qkv, e, gx, 1, s); \
    else dec_launch<1, 2, WV, KI_, DEPI_ROPE, true>((const half_t *)h, ldx, (const half_t *)W, (int)T, (int)K, N, (half_t *)qkv, e, gx, 1, s)
    NVR_DEC_GEOM(K, BODY);
#undef BODY
    return dec_launch_check("linear_qkv_rope_store_normed");
}

}}  // namespace nvr::k / nvr::kb (NVR_DT_NS)
