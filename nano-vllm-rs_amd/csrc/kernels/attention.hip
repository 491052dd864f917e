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
//    to VGPRs (no LDS round trip for a once-read stream), U such loads of K and of V requested back to back.
//  * q·k: v_dot2_f32_f16 on the lane's 8-element slice, then a butterfly over the D/8 lanes of the
//    row; softmax is online (running max / sum per wave, f32), p·v accumulates the lane's slice.
//  * split-KV: one workgroup = WAVES waves owns (query, kv head, partition); its waves interleave
//    row groups and merge through LDS.  With one partition the workgroup writes the fp16 result
//    itself; otherwise f32 partials go to a workspace and a second tiny kernel merges them.
//  * all G = H/KVH query heads of a kv head are processed together so K/V are read once.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include "kernels.h"
#include "device_utils.h"
#include "../common.h"

namespace nvr { namespace NVR_DT_NS {

struct AttnParams {
    const half_t *q; int64_t ldq;
    const half_t *k, *v; int64_t ldkv;
    const int32_t *ctx_lens, *seq_of_q, *kv_base, *block_tables;
    int32_t max_blocks, block_size, bs_shift;
    int32_t H, KVH;
    float scale;
    int32_t part_size, num_parts;
    int32_t part0, kv0;                // shared-prefix decode: partitions 0..part0-1 hold the batch's shared keys [0, kv0) (written by
                                       // flash_shared_prefix); this launch covers partition part0 + i = keys [kv0 + i*part_size, ..)
    const int32_t *kv0_rows;           // per-query kv0 (shared-prefix groups: members shared_len, others 0) or null = p.kv0 for all
    float *part_o; float *part_ml;     // [nq, H, num_parts, D], [nq, H, num_parts, 2]
    half_t *out;                       // [nq, H, D]
    unsigned int *tickets;             // FUSE: one arrival counter per (query, kv head), zero between launches (the last arriver re-arms it)
};

// a * b rounded, then + c rounded: never contracted into an fma, whatever kernel this is inlined into
__device__ __forceinline__ float mul_then_add(float a, float b, float c) {
#pragma clang fp contract(off)
    return a * b + c;
}

template <bool NT>
__device__ __forceinline__ half8_t load_row16(const half_t *p) {
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const half8_t *>(p));
    return *reinterpret_cast<const half8_t *>(p);
}

// sum / max over the 64/LPR row groups of the wave (lanes with equal lane % LPR), result in every lane
template <int LPR>
__device__ __forceinline__ float groups_sum(float x) {
    if (LPR == 8) x = dpp_add<0x128>(x);                                   // row_ror:8 inside the 16-lane row
    return xor32_partner_sum(xor16_partner_sum(x));
}
template <int LPR>
__device__ __forceinline__ float groups_max(float x) {
    if (LPR == 8) x = fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x128, 0xF, 0xF, false)));
    return xor32_partner_max(xor16_partner_max(x));
}

// A wave-instruction reads one ROW GROUP: RPI = 64/(D/8) consecutive tokens of one kv head, 16 B per lane (1 KiB).
// The partition is cut into chunks of U row groups (TPI = U*RPI consecutive tokens); in a FULL round every wave of
// the workgroup takes one complete chunk (chunk r*WAVES + wave: the waves' loads are neighbours in memory) and
// requests its U groups of K and of V back to back before any arithmetic, with no validity tests.  What is left
// after the last full round (< WAVES*U groups, the last one possibly partial) is dealt out group by group
// (group j -> wave j % WAVES) and requested BEFORE the arithmetic of the last full round, so the remainder costs no
// extra exposed memory round trip and no wave has more than one group more than another.
// NT: non-temporal loads.  UB: block_size is a multiple of TPI, so a chunk lies inside ONE cache block: its
// block-table entry is a scalar read (v_readlane) from a register copy of the table (lane j holds entry
// 64*c + j) and the row address is scalar base + small lane offset -- no dependent table load sits between the row
// loads.
// out = sum_p e^(m_p-M) o_p / sum_p e^(m_p-M) l_p over partitions [0, np) of ONE (query, head), 4 columns per thread: the partials of up
// to 8 partitions are requested together (a runtime-count loop of dependent loads would pay one round trip per partition); sums run over
// the partitions in ascending order for every column.  load_ml(i) / load_o(i): (max, sum) and the 4 columns of partition i — plain loads
// in the merge kernel, sc1 loads in the last arriver of the fused form (the same arithmetic: the two forms agree bit for bit).
template <class LoadML, class LoadO>
__device__ __forceinline__ half4_t merge_partitions(int np, LoadML load_ml, LoadO load_o) {
    float M = -INFINITY;
    float4_t o = {0.f, 0.f, 0.f, 0.f};
    float L = 0.f;
    for (int i0 = 0; i0 < np; i0 += 8) {
        float2_t ml[8]; float4_t po[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = min(i0 + u, np - 1);
            ml[u] = load_ml(i);
            po[u] = load_o(i);
        }
        float Mn = M;
#pragma unroll
        for (int u = 0; u < 8; ++u) if (i0 + u < np) Mn = fmaxf(Mn, ml[u][0]);
        if (i0 > 0 && Mn != M) { const float r = __expf(M - Mn); o *= r; L *= r; }     // (more than 8 partitions: rescale what is summed)
        M = Mn;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (i0 + u < np) {
                const float w = __expf(ml[u][0] - M);
                // (spelled out as fused multiply-adds: under -ffp-contract=fast hipcc decides per INSTANTIATION whether "a += w * b" becomes an fma,
                //  and this function is inlined into three kernels whose results are promised to agree bit for bit; fma is what the merge kernel got)
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = __fmaf_rn(w, po[u][e], o[e]);
                L = __fmaf_rn(w, ml[u][1], L);
            }
        }
    }
    half4_t hv;
#pragma unroll
    for (int e = 0; e < 4; ++e) hv[e] = (half_t)(L > 0.f ? o[e] / L : 0.f);
    return hv;
}

// FUSE (split-KV, no shared-prefix pass): the merge of a (query, kv head)'s partitions rides on the LAST ARRIVER of its partition workgroups
// instead of a second launch (cdna guide Guideline 16, counter form): every workgroup publishes its partials write-through (sc1), drains,
// and one lane draws a ticket from the pair's counter; the workgroup whose ticket says it came last reads all partials back with sc1 loads
// and merges them with merge_partitions — nobody polls.  On a tensor-parallel rank (1-2 kv heads) every kernel of the decode step sits on the
// launch floor: one launch less per layer.
// SHM (shared-prefix decode where EVERY query shares the prefix and its own keys fit one partition — BASELINE configs[4]): the partials of the shared
// partitions [0, part0) were written by the launch in front of this one (flash_shared_prefix), so the single own-partition workgroup of a (query, kv head) IS
// the last arriver by stream order: it merges the pair itself (own partial through LDS, the same merge_partitions: the merge launch's bits) — no merge launch.
template <int D, int G, bool PAGED, bool DIRECT_OUT, int U, int WAVES, bool NT, bool UB, bool FUSE = false, bool SHM = false>
__global__ __launch_bounds__(WAVES * 64) void attn_rows_kernel(AttnParams p) {
    constexpr int LPR = D / 8;          // lanes per K/V row
    constexpr int RPI = 64 / LPR;       // rows per wave-instruction (= tokens per row group)
    constexpr int TPI = RPI * U;        // tokens per chunk
    const int lparts = p.num_parts - p.part0;        // partitions of this launch
    const int part = p.part0 + blockIdx.x % lparts, g = (blockIdx.x / lparts) % p.KVH;
    const int t = blockIdx.x / (lparts * p.KVH);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int dc = lane % LPR, tg = lane / LPR;
    const int p0 = (p.kv0_rows ? p.kv0_rows[t] : p.kv0) + (part - p.part0) * p.part_size;

    const int32_t *bt = PAGED ? p.block_tables + (int64_t)(p.seq_of_q ? p.seq_of_q[t] : t) * p.max_blocks : nullptr;
    int bt_chunk = -1, bt_reg = 0;                   // UB: register copy of 64 block-table entries
    auto load_bt_chunk = [&](int c) {
        bt_chunk = c;
        const int idx = (c << 6) + lane;
        bt_reg = idx < p.max_blocks ? bt[idx] : 0;
    };
    if (PAGED && UB) {                               // requested together with ctx and q, ahead of their use
        const int b0 = p.bs_shift >= 0 ? p0 >> p.bs_shift : p0 / p.block_size;
        load_bt_chunk(b0 >> 6);
    }

    // SHM: the shared partitions' partials [G heads][part0][D] (+ their (max, sum) pairs) go straight into LDS by LDS-DMA, requested HERE, with ctx and
    // the block table: as loads of the merge at the end of the kernel they were a third dependent round trip in the life of a workgroup whose own
    // keys are two chunks per wave (configs[4]: 4096 workgroups of ~94 own tokens, 43 us per layer = 0.57 of HBM; profiles/r05_priced_levers.txt 7.)
    constexpr int SHP = 4;                            // shared partitions whose partials are staged (more: the merge reads them from global memory)
    __shared__ __attribute__((aligned(16))) float shm_po[SHM ? G * SHP * D : 4];
    __shared__ float shm_pml[SHM ? G * SHP * 2 : 1];
    const bool staged = SHM && p.part0 <= SHP;
    if (SHM && staged) {
        const int per_head = p.part0 * (D / 4), total = G * per_head;            // 16-byte pieces, lane-linear in LDS: [head][partition][D]
        const float *src0 = p.part_o + ((int64_t)t * p.H + (int64_t)g * G) * p.num_parts * D;
        for (int x0 = wave * 64; x0 < total; x0 += WAVES * 64) {
            const int x = x0 + lane;
            if (x < total) {
                const int i = x / per_head, rem = x - i * per_head;
                __builtin_amdgcn_global_load_lds(src0 + ((int64_t)i * p.num_parts * (D / 4) + rem) * 4,
                                                 (__attribute__((address_space(3))) void *)(shm_po + x0 * 4), 16, 0, 0);
            }
        }
        if (wave == WAVES - 1 && lane < G * p.part0 * 2) {
            const int i = lane / (p.part0 * 2), rem = lane - i * (p.part0 * 2);
            __builtin_amdgcn_global_load_lds(p.part_ml + (((int64_t)t * p.H + (int64_t)g * G + i) * p.num_parts) * 2 + rem,
                                             (__attribute__((address_space(3))) void *)shm_pml, 4, 0, 0);
        }
    }
    // q slice of this lane for the G heads of kv head g (fp16 pairs for v_dot2); requested in front of ctx's first use
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
    const int ctx = p.ctx_lens[t];
    if (p0 >= ctx && !(DIRECT_OUT) && !SHM) return;  // empty partition: the merge kernel skips it too (SHM: this workgroup still merges the shared partitions)
    const int pend = max(p0, min(ctx, p0 + p.part_size));
    const int64_t base_row = PAGED ? 0 : (int64_t)p.kv_base[t];
    const int row_elems = p.KVH * D;

    // element offset of (scalar token tb, kv head g, d = 0) through the register copy of the block table
    auto block_base = [&](int tb) -> int64_t {
        int bi, bo;
        if (p.bs_shift >= 0) { bi = tb >> p.bs_shift; bo = tb & (p.block_size - 1); }
        else { bi = tb / p.block_size; bo = tb - bi * p.block_size; }
        if ((bi >> 6) != bt_chunk) load_bt_chunk(bi >> 6);
        const int blk = __builtin_amdgcn_readlane(bt_reg, bi & 63);
        return (((int64_t)blk * p.block_size + bo) * p.KVH + g) * D;
    };
    // per-lane form (contiguous K/V, or block sizes that are no multiple of the chunk)
    auto lane_row = [&](int tok) -> int64_t {
        if (PAGED) {
            int bi, bo;
            if (p.bs_shift >= 0) { bi = tok >> p.bs_shift; bo = tok & (p.block_size - 1); }
            else { bi = tok / p.block_size; bo = tok - bi * p.block_size; }
            return (((int64_t)bt[bi] * p.block_size + bo) * p.KVH + g) * D + dc * 8;
        }
        return (base_row + tok) * p.ldkv + (int64_t)g * D + dc * 8;
    };

    // online softmax state per head AND per row-group slot of the wave (the 64/LPR lane groups never exchange data
    // inside the loop; they are merged once at the end)
    float m[G], l[G], acc[G][8];
#pragma unroll
    for (int i = 0; i < G; ++i) {
        m[i] = -INFINITY; l[i] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
    }

    // scores, online softmax and p.v for U row groups; CHECK: slot u holds token tok0 + u*WAVES*RPI + tg, valid < pend
    auto process = [&](auto check, const half8_t (&kk)[U], const half8_t (&vv)[U], int tok0) {
        constexpr bool CHECK = decltype(check)::value;
        float s[U][G];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool valid = !CHECK || tok0 + u * (WAVES * RPI) + tg < pend;
#pragma unroll
            for (int i = 0; i < G; ++i) {
                float d = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    d = dot2((half2_t){kk[u][2 * j], kk[u][2 * j + 1]}, qv[i][j], d);
                d = row_sum<LPR>(d);
                s[u][i] = valid ? d * p.scale : -INFINITY;
            }
        }
#pragma unroll
        for (int i = 0; i < G; ++i) {
            float mx = s[0][i];
#pragma unroll
            for (int u = 1; u < U; ++u) mx = fmaxf(mx, s[u][i]);
            const float mn = fmaxf(m[i], mx);
            const float ms = (CHECK && mn == -INFINITY) ? 0.f : mn;   // a slot that has seen no valid token yet
            const float alpha = __expf(m[i] - ms);                    // m = -inf on first use -> 0
            m[i] = mn;
            float ps = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] *= alpha;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float pr = __expf(s[u][i] - ms);
                ps += pr;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] = fmaf(pr, (float)vv[u][j], acc[i][j]);
            }
            l[i] = __fmaf_rn(l[i], alpha, ps);                         // (pinned: see the cross-wave merge below)
        }
    };

    const int ng = (pend - p0 + RPI - 1) / RPI;      // row groups of this partition, the last one possibly partial
    const int R = (pend - p0) / (TPI * WAVES);       // full rounds: every wave gets a complete chunk
    const int gt0 = R * (WAVES * U) + wave;          // this wave's remainder groups: gt0, gt0 + WAVES, ..
    auto issue_remainder = [&](half8_t (&kt)[U], half8_t (&vt)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int gt = gt0 + u * WAVES;
            if (gt < ng) {                           // uniform
                const int tb = p0 + gt * RPI;
                if (PAGED && UB) {
                    const int64_t base = block_base(tb);
                    const unsigned lo = (unsigned)(min(tg, pend - 1 - tb) * row_elems + dc * 8);
                    kt[u] = load_row16<NT>(p.k + base + lo); vt[u] = load_row16<NT>(p.v + base + lo);
                } else {
                    const int64_t off = lane_row(min(tb + tg, pend - 1));
                    kt[u] = load_row16<NT>(p.k + off); vt[u] = load_row16<NT>(p.v + off);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) { kt[u][j] = (half_t)0.f; vt[u][j] = (half_t)0.f; }
            }
        }
    };

    half8_t kt[U], vt[U];
    if (R == 0) issue_remainder(kt, vt);
    for (int r = 0; r < R; ++r) {
        const int tb = p0 + (r * WAVES + wave) * TPI;
        half8_t kk[U], vv[U];
        if (PAGED && UB) {
            const int64_t base = block_base(tb);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const unsigned lo = (unsigned)((u * RPI + tg) * row_elems + dc * 8);
                kk[u] = load_row16<NT>(p.k + base + lo); vv[u] = load_row16<NT>(p.v + base + lo);
            }
        } else {
            int64_t off[U];
#pragma unroll
            for (int u = 0; u < U; ++u) off[u] = lane_row(tb + u * RPI + tg);
#pragma unroll
            for (int u = 0; u < U; ++u) { kk[u] = load_row16<NT>(p.k + off[u]); vv[u] = load_row16<NT>(p.v + off[u]); }
        }
        if (r == R - 1) issue_remainder(kt, vt);
        process(std::false_type{}, kk, vv, 0);
    }
    if (gt0 < ng) process(std::true_type{}, kt, vt, p0 + gt0 * RPI);

    // bring the row-group slots of the wave to their common max and sum them, then merge the waves through LDS
    constexpr bool ONE = SHM && WAVES == 1;                               // one wave per workgroup: its sums ARE the partition's partial (the cross-wave
                                                                          // merge below would multiply them by exp(0) = 1 and add them to 0)
    __shared__ float sm_acc[ONE ? 1 : WAVES][ONE ? 1 : G][ONE ? 1 : D];
    __shared__ float sm_ml[WAVES][G][2];
    __shared__ float shm_o[SHM ? G : 1][SHM ? D : 1];                     // SHM: this workgroup's own partial, merged below with the shared partitions'
    __shared__ float shm_ml[SHM ? G : 1][2];
#pragma unroll
    for (int i = 0; i < G; ++i) {
        const float M = groups_max<LPR>(m[i]);
        const float w = (m[i] == -INFINITY) ? 0.f : __expf(m[i] - M);
        l[i] = groups_sum<LPR>(l[i] * w);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = groups_sum<LPR>(acc[i][j] * w);
        if (tg == 0) {
            if constexpr (ONE) {
#pragma unroll
                for (int j = 0; j < 8; ++j) shm_o[i][dc * 8 + j] = acc[i][j];
                if (dc == 0) { shm_ml[i][0] = M; shm_ml[i][1] = l[i]; }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) sm_acc[wave][i][dc * 8 + j] = acc[i][j];
                if (dc == 0) { sm_ml[wave][i][0] = M; sm_ml[wave][i][1] = l[i]; }
            }
        }
    }
    if constexpr (!ONE) __syncthreads();
    // the pair's partials are contiguous: [G heads][num_parts][D] (and [..][2]); buffer offsets stay small whatever the workspace size
    const int64_t pair0 = DIRECT_OUT ? 0 : ((int64_t)t * p.H + (int64_t)g * G) * p.num_parts;
    const auto rs_o = __builtin_amdgcn_make_buffer_rsrc(DIRECT_OUT ? nullptr : p.part_o + pair0 * D, 0, DIRECT_OUT ? 0 : (int)(G * p.num_parts * D * 4), 0x00020000);
    const auto rs_ml = __builtin_amdgcn_make_buffer_rsrc(DIRECT_OUT ? nullptr : p.part_ml + pair0 * 2, 0, DIRECT_OUT ? 0 : (int)(G * p.num_parts * 2 * 4), 0x00020000);
    constexpr int AUX = FUSE ? 16 : 0;                                    // sc1: write-through, so the hand-off needs no release fence
    if constexpr (!ONE)
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
            // (multiply, THEN add, spelled out: left to -ffp-contract=fast the SHM instantiation of this loop got fmas where the instantiation in front of
            //  the merge launch got v_pk_mul + v_add — a 1-ulp difference that flipped ~1 output in 10^5 between two forms promised to be bit-identical
            //  (r05, found at 300 sequences; the 9-sequence test was too small to see it).  Unfused is what the partial-writing instantiations always were;
            //  HIP's __fmul_rn / __fadd_rn are plain * and + and contract like them, hence the helper with contraction switched off.)
            o = mul_then_add(wgt, sm_acc[w2][i][d], o);
            L = mul_then_add(wgt, sm_ml[w2][i][1], L);
        }
        const int h = g * G + i;
        if (DIRECT_OUT) {
            p.out[((int64_t)t * p.H + h) * D + d] = (half_t)(L > 0.f ? o / L : 0.f);
        } else if (SHM) {
            shm_o[i][d] = o;
            if (d == 0) { shm_ml[i][0] = M; shm_ml[i][1] = L; }
        } else {
            const int slot = i * p.num_parts + part;                     // inside the pair's region
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o), rs_o, (slot * D + d) * 4, 0, AUX);
            if (d == 0) { __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(M), rs_ml, slot * 8, 0, AUX);
                          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(L), rs_ml, slot * 8 + 4, 0, AUX); }
        }
    }
    if constexpr (!DIRECT_OUT) {
        constexpr int TPH = D / 4;                                // merging threads per head: 4 columns each (merge_partitions)
        static_assert(G * TPH <= WAVES * 64, "one pass over the (head, column group) pairs");
        if constexpr (SHM) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the LDS-DMA requests of the shared partitions' partials (long landed)
            __syncthreads();
            if (threadIdx.x < G * TPH) {
                const int i = threadIdx.x / TPH, d = (threadIdx.x % TPH) * 4;
                const int own = part;                                         // = part0: the one partition of this launch
                const int np = own + (ctx > p0 ? 1 : 0);                      // shared partitions, then the own one if the query has own keys
                const int64_t base = ((int64_t)t * p.H + g * G + i) * p.num_parts;
                const float2_t own_ml = {shm_ml[i][0], shm_ml[i][1]};
                const float4_t own_o = {shm_o[i][d], shm_o[i][d + 1], shm_o[i][d + 2], shm_o[i][d + 3]};
                half4_t hv;
                if (staged)
                    hv = merge_partitions(np,
                        [&](int pi) { return pi < own ? *reinterpret_cast<const float2_t *>(shm_pml + (i * own + pi) * 2) : own_ml; },
                        [&](int pi) { return pi < own ? *reinterpret_cast<const float4_t *>(shm_po + (i * own + pi) * D + d) : own_o; });
                else
                    hv = merge_partitions(np,
                        [&](int pi) { return pi < own ? *reinterpret_cast<const float2_t *>(p.part_ml + (base + pi) * 2) : own_ml; },
                        [&](int pi) { return pi < own ? *reinterpret_cast<const float4_t *>(p.part_o + (base + pi) * D + d) : own_o; });
                *reinterpret_cast<half4_t *>(p.out + ((int64_t)t * p.H + g * G + i) * D + d) = hv;
            }
        }
        if constexpr (FUSE) {
            __shared__ unsigned int ticket_s;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // every storing wave drains its stores ...
            __syncthreads();
            unsigned int *cnt = p.tickets + (int64_t)t * p.KVH + g;
            if (threadIdx.x == 0) ticket_s = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... one ticket
            __syncthreads();
            const int np = min(p.num_parts, (ctx + p.part_size - 1) / p.part_size);   // non-empty partitions of this query
            if (ticket_s != (unsigned)(np - 1)) return;                       // not the last partition of this pair to finish: done
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");            // keeps the sc1 loads below the ticket
            if (threadIdx.x < G * TPH) {
                const int i = threadIdx.x / TPH, d = (threadIdx.x % TPH) * 4;
                const half4_t hv = merge_partitions(np,
                    [&](int pi) { return __builtin_bit_cast(float2_t, __builtin_amdgcn_raw_buffer_load_b64(rs_ml, (i * p.num_parts + pi) * 8, 0, 16)); },
                    [&](int pi) { return __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(rs_o, ((i * p.num_parts + pi) * D + d) * 4, 0, 16)); });
                *reinterpret_cast<half4_t *>(p.out + ((int64_t)t * p.H + g * G + i) * D + d) = hv;
            }
            if (threadIdx.x == 0) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm for the next launch
        }
    }
}

// merge split-KV partitions: out = sum_p e^(m_p-M) o_p / sum_p e^(m_p-M) l_p
// 32 (D = 128) or 16 (D = 64) threads per (query, head), 4 columns each; the partials of up to 8 partitions are requested together
// (a runtime-count loop of dependent loads would pay one round trip per partition).  Sums run over the partitions in ascending
// order for every column, as before.
template <int D>
__global__ __launch_bounds__(256) void attn_merge_kernel(const float *__restrict__ part_o, const float *__restrict__ part_ml,
                                                         const int32_t *__restrict__ ctx_lens, int H, int part_size, int num_parts,
                                                         int part0, int kv0, const int32_t *__restrict__ kv0_rows, int64_t pairs,
                                                         half_t *__restrict__ out) {
    constexpr int TPP = D / 4;                                   // threads per (query, head) pair
    const int64_t pair = (int64_t)blockIdx.x * (256 / TPP) + threadIdx.x / TPP;
    if (pair >= pairs) return;
    const int t = (int)(pair / H), d = (threadIdx.x % TPP) * 4;
    const int my_kv0 = kv0_rows ? kv0_rows[t] : kv0;
    const int first = (part0 > 0 && my_kv0 == 0) ? part0 : 0;   // a query outside the sharing group has no partials in the shared slots
    const int np = min(num_parts, part0 + (max(ctx_lens[t] - my_kv0, 0) + part_size - 1) / part_size) - first;
    const int64_t base = pair * num_parts + first;
    const half4_t hv = merge_partitions(np,
        [&](int i) { return *reinterpret_cast<const float2_t *>(part_ml + (base + i) * 2); },
        [&](int i) { return *reinterpret_cast<const float4_t *>(part_o + (base + i) * D + d); });
    *reinterpret_cast<half4_t *>(out + pair * D + d) = hv;
}

// ---- work-balanced form (r06) ---------------------------------------------------------------------------------------
// attn_rows_kernel gives every (query, kv head) pair its own workgroup(s): 264 pairs on 256 CUs (33 sequences x 8 kv heads) run as one full round and a tail,
// 35.6 us per layer where 256 pairs take 24.5 (profiles/r06_priced_levers.txt 9.).  Here the launch is nw workgroups (one per CU) and the WORK is cut evenly: the
// keys of all pairs, in units of CH = WAVES * TPI keys (one full round of a workgroup), pair after pair (query-major), form one line of U_tot units; workgroup w
// walks the units [w * U_tot / nw, (w + 1) * U_tot / nw) — a suffix of one pair, whole pairs, a prefix of another.  A pair that lies inside one share is written
// directly; a pair cut by share boundaries leaves one f32 partial per share (slot = share index - first share of the pair) and the LAST of its shares to finish
// merges them (the ticket scheme of FUSE above: sc1 partials, one counter per pair, nobody polls; merge_partitions: the same arithmetic).
// Paged K/V through block tables whose block size is a power of two >= TPI (the scalar block-table form), nt loads; per-segment arithmetic = attn_rows_kernel's.
template <int D, int G, int U, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void attn_share_kernel(AttnParams p, int nq, int nw) {
    constexpr int LPR = D / 8, RPI = 64 / LPR, TPI = RPI * U, CH = TPI * WAVES;
    static_assert(G * D <= WAVES * 64 && G * (D / 4) <= WAVES * 64, "one pass over the (head, column) pairs");
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int dc = lane % LPR, tg = lane / LPR;
    __shared__ int pre[1025];                               // pre[t] = units of the queries in front of t (a query's KVH pairs have the same count)
    __shared__ int cx[1024];                                // ctx_lens[t]
    __shared__ float sm_acc[WAVES][G][D];
    __shared__ float sm_ml[WAVES][G][2];
    __shared__ unsigned int ticket_s;
    if (wave == 0) {
        int base = 0;
        for (int k = 0; k < nq; k += 64) {
            const int t = k + lane;
            const int c = t < nq ? p.ctx_lens[t] : 0;
            int un = (c + CH - 1) / CH;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(un, o, 64); if (lane >= o) un += v; }
            if (t < nq) { pre[t + 1] = base + un; cx[t] = c; }
            base += __shfl(un, 63, 64);
        }
        if (lane == 0) pre[0] = 0;
    }
    __syncthreads();
    const long long utot = (long long)p.KVH * pre[nq];
    const long long ub = (long long)blockIdx.x * utot / nw, ue = (long long)(blockIdx.x + 1) * utot / nw;
    // a query without keys owns no unit: its output row is zero, as in attn_rows_kernel (L == 0)
    for (int t0 = blockIdx.x; t0 < nq; t0 += nw)
        if (cx[t0] == 0)
            for (int x = threadIdx.x; x < p.H * D; x += WAVES * 64) p.out[(int64_t)t0 * p.H * D + x] = (half_t)0.f;
    if (ub >= ue) return;                                   // (fewer units than workgroups: an empty share)
    const int row_elems = p.KVH * D;

    // ---- state of the segment being streamed (set by `open`, which also REQUESTS its first K/V rows: a segment is opened under the tail of the one in front) ----
    int t = 0, g = 0, np = 1, slot = 0, p0 = 0, pend = 0, R = 0, ng = 0, gt0 = 0, bt_chunk = -1, bt_reg = 0;
    const int32_t *bt = p.block_tables;
    half2_t qv[G][4];
    half8_t k0[U], v0[U], kt[U], vt[U];                     // round 0 of the segment; its remainder groups
    auto block_base = [&](int tb) -> int64_t {              // element offset of (scalar token tb, kv head g, d = 0)
        const int bi = tb >> p.bs_shift, bo = tb & (p.block_size - 1);
        if ((bi >> 6) != bt_chunk) { bt_chunk = bi >> 6; const int idx = (bt_chunk << 6) + lane; bt_reg = idx < p.max_blocks ? bt[idx] : 0; }
        const int blk = __builtin_amdgcn_readlane(bt_reg, bi & 63);
        return (((int64_t)blk * p.block_size + bo) * p.KVH + g) * D;
    };
    auto issue_remainder = [&]() {
#pragma unroll
        for (int x = 0; x < U; ++x) {
            const int gt = gt0 + x * WAVES;
            if (gt < ng) {
                const int tb = p0 + gt * RPI;
                const int64_t base = block_base(tb);
                const unsigned off = (unsigned)(min(tg, pend - 1 - tb) * row_elems + dc * 8);
                kt[x] = load_row16<true>(p.k + base + off); vt[x] = load_row16<true>(p.v + base + off);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) { kt[x][j] = (half_t)0.f; vt[x][j] = (half_t)0.f; }
            }
        }
    };
    auto load_round = [&](int r, half8_t (&kk)[U], half8_t (&vv)[U]) {
        const int64_t base = block_base(p0 + (r * WAVES + wave) * TPI);
#pragma unroll
        for (int x = 0; x < U; ++x) {
            const unsigned off = (unsigned)((x * RPI + tg) * row_elems + dc * 8);
            kk[x] = load_row16<true>(p.k + base + off); vv[x] = load_row16<true>(p.v + base + off);
        }
    };
    // opens the segment that starts at unit u (u < ue); returns its length in units
    auto open = [&](long long u) -> int {
        int lo = 0, hi = nq;                                // the query of unit u: the last t with KVH * pre[t] <= u (queries without keys own no unit)
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((long long)p.KVH * pre[mid] <= u) lo = mid; else hi = mid; }
        t = lo;
        const int ut = pre[t + 1] - pre[t], rem = (int)(u - (long long)p.KVH * pre[t]);
        g = rem / ut;
        const int lu = rem - g * ut, seg = (int)min((long long)(ut - lu), ue - u);
        // shares that touch this pair: unit x belongs to share floor(((x + 1) * nw - 1) / utot)
        const long long S = (long long)p.KVH * pre[t] + (long long)g * ut, E = S + ut - 1;
        const int wf = (int)(((S + 1) * nw - 1) / utot), wl = (int)(((E + 1) * nw - 1) / utot);
        np = wl - wf + 1; slot = (int)blockIdx.x - wf;
        p0 = lu * CH; pend = min(cx[t], (lu + seg) * CH);
        ng = (pend - p0 + RPI - 1) / RPI; R = (pend - p0) / CH; gt0 = R * (WAVES * U) + wave;
        bt = p.block_tables + (int64_t)t * p.max_blocks;
        bt_chunk = (p0 >> p.bs_shift) >> 6;
        { const int idx = (bt_chunk << 6) + lane; bt_reg = idx < p.max_blocks ? bt[idx] : 0; }
        const half_t *qrow = p.q + (int64_t)t * p.ldq + (int64_t)g * G * D + dc * 8;
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const half8_t h = *reinterpret_cast<const half8_t *>(qrow + i * D);
#pragma unroll
            for (int j = 0; j < 4; ++j) qv[i][j] = (half2_t){h[2 * j], h[2 * j + 1]};
        }
        if (R >= 1) load_round(0, k0, v0);
        if (R <= 1) issue_remainder();                      // (requested before the arithmetic of the last full round, as in attn_rows_kernel)
        asm volatile("" ::: "memory");                      // the requests stay HERE: hipcc would sink them to their first use, behind the tail of the segment in front
        return seg;
    };

    long long u = ub;
    u += open(u);
    for (;;) {
        float m[G], l[G], acc[G][8];
#pragma unroll
        for (int i = 0; i < G; ++i) {
            m[i] = -INFINITY; l[i] = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
        }
        auto process = [&](auto check, const half8_t (&kk)[U], const half8_t (&vv)[U], int tok0) {
            constexpr bool CHECK = decltype(check)::value;
            float sc[U][G];
#pragma unroll
            for (int x = 0; x < U; ++x) {
                const bool valid = !CHECK || tok0 + x * (WAVES * RPI) + tg < pend;
#pragma unroll
                for (int i = 0; i < G; ++i) {
                    float d = 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) d = dot2((half2_t){kk[x][2 * j], kk[x][2 * j + 1]}, qv[i][j], d);
                    d = row_sum<LPR>(d);
                    sc[x][i] = valid ? d * p.scale : -INFINITY;
                }
            }
#pragma unroll
            for (int i = 0; i < G; ++i) {
                float mx = sc[0][i];
#pragma unroll
                for (int x = 1; x < U; ++x) mx = fmaxf(mx, sc[x][i]);
                const float mn = fmaxf(m[i], mx);
                const float ms = (CHECK && mn == -INFINITY) ? 0.f : mn;
                const float alpha = __expf(m[i] - ms);
                m[i] = mn;
                float ps = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] *= alpha;
#pragma unroll
                for (int x = 0; x < U; ++x) {
                    const float pr = __expf(sc[x][i] - ms);
                    ps += pr;
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[i][j] = fmaf(pr, (float)vv[x][j], acc[i][j]);
                }
                l[i] = __fmaf_rn(l[i], alpha, ps);
            }
        };
        if (R >= 1) process(std::false_type{}, k0, v0, 0);
        for (int r = 1; r < R; ++r) {
            half8_t kk[U], vv[U];
            load_round(r, kk, vv);
            if (r == R - 1) issue_remainder();
            process(std::false_type{}, kk, vv, 0);
        }
        if (gt0 < ng) process(std::true_type{}, kt, vt, p0 + gt0 * RPI);

        // the segment is in the accumulators: what its tail needs of the state, then the NEXT segment is opened (its query, block table and first K/V rows
        // are on their way while this one is merged and handed over)
        const int st = t, sg = g, snp = np, sslot = slot;
        const bool more = u < ue;
        if (more) u += open(u);

        // row-group slots of the wave -> the wave's sums; waves -> the segment's partial (attn_rows_kernel's arithmetic)
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const float M = groups_max<LPR>(m[i]);
            const float w = (m[i] == -INFINITY) ? 0.f : __expf(m[i] - M);
            l[i] = groups_sum<LPR>(l[i] * w);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = groups_sum<LPR>(acc[i][j] * w);
            if (tg == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) sm_acc[wave][i][dc * 8 + j] = acc[i][j];
                if (dc == 0) { sm_ml[wave][i][0] = M; sm_ml[wave][i][1] = l[i]; }
            }
        }
        __syncthreads();
        const int64_t pair0 = ((int64_t)st * p.H + (int64_t)sg * G) * p.num_parts;
        const auto rs_o = __builtin_amdgcn_make_buffer_rsrc(p.part_o + pair0 * D, 0, (int)(G * p.num_parts * D * 4), 0x00020000);
        const auto rs_ml = __builtin_amdgcn_make_buffer_rsrc(p.part_ml + pair0 * 2, 0, (int)(G * p.num_parts * 2 * 4), 0x00020000);
        if (threadIdx.x < G * D) {
            const int i = threadIdx.x / D, d = threadIdx.x % D;
            float M = sm_ml[0][i][0];
#pragma unroll
            for (int w2 = 1; w2 < WAVES; ++w2) M = fmaxf(M, sm_ml[w2][i][0]);
            float o = 0.f, L = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < WAVES; ++w2) {
                const float mw = sm_ml[w2][i][0];
                const float wgt = (mw == -INFINITY) ? 0.f : __expf(mw - M);
                o = mul_then_add(wgt, sm_acc[w2][i][d], o);
                L = mul_then_add(wgt, sm_ml[w2][i][1], L);
            }
            if (snp == 1) {
                p.out[((int64_t)st * p.H + sg * G + i) * D + d] = (half_t)(L > 0.f ? o / L : 0.f);
            } else {
                const int sl = i * p.num_parts + sslot;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o), rs_o, (sl * D + d) * 4, 0, 16);
                if (d == 0) { __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(M), rs_ml, sl * 8, 0, 16);
                              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(L), rs_ml, sl * 8 + 4, 0, 16); }
            }
        }
        if (snp > 1) {                                                        // (workgroup-uniform)
            constexpr int TPH = D / 4;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // every storing wave drains its stores (and the next segment's first rows land) ...
            __syncthreads();
            unsigned int *cnt = p.tickets + (int64_t)st * p.KVH + sg;
            if (threadIdx.x == 0) ticket_s = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... one ticket
            __syncthreads();
            if (ticket_s == (unsigned)(snp - 1)) {                            // the last of the pair's shares to finish merges it
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");        // keeps the sc1 loads below the ticket
                if (threadIdx.x < G * TPH) {
                    const int i = threadIdx.x / TPH, d = (threadIdx.x % TPH) * 4;
                    const half4_t hv = merge_partitions(snp,
                        [&](int pi) { return __builtin_bit_cast(float2_t, __builtin_amdgcn_raw_buffer_load_b64(rs_ml, (i * p.num_parts + pi) * 8, 0, 16)); },
                        [&](int pi) { return __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(rs_o, ((i * p.num_parts + pi) * D + d) * 4, 0, 16)); });
                    *reinterpret_cast<half4_t *>(p.out + ((int64_t)st * p.H + sg * G + i) * D + d) = hv;
                }
                if (threadIdx.x == 0) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm for the next launch
            }
        }
        if (!more) break;
        __syncthreads();                                                      // sm_acc / sm_ml / ticket_s are the next segment's too
    }
}

// ---- launch configuration ------------------------------------------------------------------------
static int share_mode() { static const int v = getenv("NVR_ATTN_SHARE") ? atoi(getenv("NVR_ATTN_SHARE")) : 1; return v; }   // 0: never, 1: by the rule, 2: whenever it can run (probes)
static bool share_enabled() { return share_mode() != 0; }
static int share_workgroups() {                                           // one workgroup per CU of the current device
    static const int v = [] { int dev = 0, n = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256; return n; }();
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

template <int D, int G, int U, int WAVES, bool NT>
static void launch_cfg(const AttnParams &p, bool paged, bool direct, int64_t nwg, hipStream_t s, bool fuse = false, bool shm = false) {
    dim3 grid((unsigned)nwg), block(WAVES * 64);
    constexpr int TPI = 64 / (D / 8) * U;
    const bool ub = paged && p.block_size % TPI == 0 && (p.num_parts == 1 || p.part_size % TPI == 0);
    if constexpr (WAVES <= 4 && NT) {                                     // (only instantiated where launch_attn can ask for it)
        if (shm) {                                                        // (paged, split, every query behind the shared prefix: launch_attn)
            if (ub) attn_rows_kernel<D, G, true, false, U, WAVES, NT, true, false, true><<<grid, block, 0, s>>>(p);
            else attn_rows_kernel<D, G, true, false, U, WAVES, NT, false, false, true><<<grid, block, 0, s>>>(p);
            return;
        }
    }
    if constexpr (NT) {                                                   // the cache through block tables (every paged call site streams: nt loads)
        if (ub) {
            if (direct) attn_rows_kernel<D, G, true, true, U, WAVES, NT, true><<<grid, block, 0, s>>>(p);
            else if (fuse) attn_rows_kernel<D, G, true, false, U, WAVES, NT, true, true><<<grid, block, 0, s>>>(p);
            else attn_rows_kernel<D, G, true, false, U, WAVES, NT, true><<<grid, block, 0, s>>>(p);
        } else if (paged) {
            if (direct) attn_rows_kernel<D, G, true, true, U, WAVES, NT, false><<<grid, block, 0, s>>>(p);
            else if (fuse) attn_rows_kernel<D, G, true, false, U, WAVES, NT, false, true><<<grid, block, 0, s>>>(p);
            else attn_rows_kernel<D, G, true, false, U, WAVES, NT, false><<<grid, block, 0, s>>>(p);
        }
    } else {                                           // contiguous K / V (prefill): rows re-read from L2, 4-wave workgroups only
        if (direct) attn_rows_kernel<D, G, false, true, U, WAVES, NT, false><<<grid, block, 0, s>>>(p);
        else attn_rows_kernel<D, G, false, false, U, WAVES, NT, false><<<grid, block, 0, s>>>(p);
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
    // Geometry (measured on MI355X at B=32, ctx 1044, KVH=8, D=128, profiles/r01_attn_tune.txt): one 8-wave
    // workgroup per (query, kv head) = one per CU, each wave with 2 row groups of K and of V in flight (32 KiB per
    // CU), non-temporal loads and no split (23.6 us, 5.8 TB/s) beats 16 waves x 4 groups (25.0 us: more bytes in
    // flight per CU is slower, as for a plain streaming read) and 4-wave workgroups over 4 partitions + merge
    // kernel (29.6 us).  So: when there is enough work for >= ~192 such workgroups use them (splitting only to reach
    // ~256 workgroups), otherwise fall back to 4-wave workgroups over 64-token-granular partitions.
    const int64_t mc = a.max_ctx > 0 ? a.max_ctx : 1;
    const int64_t pairs = (int64_t)a.nq * a.KVH;
    // 8-wave workgroups whenever ~256 of them can be given >= 64 tokens each; with fewer than ~192 (query, kv head) pairs
    // (small batches, tensor-parallel ranks that hold 1-4 kv heads) the context is cut into ceil(256 / pairs)
    // 64-token-granular partitions + merge kernel (B=32, ctx 1044: KVH=1 10.9 us vs 14.6 us for the 4-wave path, KVH=2
    // 12.7 vs 14.6, KVH=4 17.5 vs 19.7; scratch/attn_tp_shape.py)
    int waves = (paged && a.workspace && pairs * ((mc + 63) / 64) >= 256) ? 8 : 4;
    // Waves per workgroup when there are >= 2048 pairs (batches of >= 256 sequences at 8 kv heads): enough for the chip's ~4096 waves, and by the bound
    // on the keys a workgroup walks (the 256-token context bucket: the contexts themselves are shorter) — with eight waves on a short context a
    // workgroup's life is mostly start-up, barriers and the cross-wave merge, not K/V requests (scratch/attn_batch_shape.py, ragged contexts, us per
    // launch at 8 / 4 / 2 / 1 waves: 512 x <= 100: 45.0 / 37.0 / 35.7 / 32.8; 1024 x <= 100: 83.9 / 70.1 / 64.7 / 63.4; 256 x <= 200: 37.9 / 35.9 /
    // 33.8 / 42.3; 512 x <= 300: 92.3 / 89.2 / 90.0 / 91.1; 256 x 1024: 149 / 153 / 152 / 161; configs[4]'s own partitions, bound 256: 2 waves 3.23, 1 wave 3.13 ms / step)
    constexpr int min_waves = (G * (D / 4) + 63) / 64;                    // the merge takes one pass over (head, 4-column group)
    auto waves_for = [&](int64_t wgs, int64_t keys, int cap) {
        const int64_t want = std::max<int64_t>({(4096 + wgs - 1) / wgs, keys <= 256 ? 1 : keys <= 512 ? 4 : 8, (int64_t)min_waves});
        int w = 1;
        while (w < want && w < cap) w *= 2;
        return w;
    };
    if (paged && a.workspace && a.shared_len <= 0 && pairs >= 2048) waves = waves_for(pairs, mc, 8);
    int part_size = 0x3fffffff, np = 1, sparts = 0, shared_part = 0;
    const bool shared = a.shared_len > 0;
    if (shared) {
        // partitions of shared_len tokens: number 0 (the same K/V for every query) goes through the MFMA kernel, 1.. through
        // the row kernel below; always merged (np >= 2 keeps the partial format even when nobody has own tokens yet)
        if (!paged || !a.workspace || a.seq_of_q || a.shared_len % a.block_size || a.shared_len % 64)
            return nvr::fail(NVR_ERR_INVALID_ARG, "attention: shared_len needs paged decode with a workspace (shared_len %d, block_size %d)", a.shared_len, a.block_size);
        // the shared keys [0, shared_len) are cut into sparts equal partitions (64-key granularity) until the MFMA launch has ~256
        // workgroups; the keys behind them are partitioned like a context of their own (one partition per pair when there are
        // enough pairs, else ~256 workgroups)
        const int qb = flash_tile_positions(a.H, a.KVH);
        const int64_t wgs = (int64_t)((a.nq + qb - 1) / qb) * a.KVH, n64 = a.shared_len / 64;
        int64_t c = std::min<int64_t>(std::max<int64_t>(1, (256 + wgs - 1) / wgs), n64);
        while (n64 % c) --c;
        sparts = (int)c;
        const bool grouped = a.shared_rows != nullptr;
        if (grouped && (!a.shared_kv0 || !a.shared_count)) return nvr::fail(NVR_ERR_INVALID_ARG, "attention: shared_rows / shared_kv0 / shared_count come together");
        const int64_t rest = std::max<int64_t>(grouped ? mc : mc - a.shared_len, 64);   // a query outside the group reads its whole context
        // the workspace holds cap partials per (query, head) (nvr_paged_attn_workspace_bytes: one per 64 tokens of the context bound):
        // the shared partitions and the remainder's together must fit
        const int64_t cap = a.workspace_bytes ? (int64_t)(a.workspace_bytes / ((size_t)a.nq * a.H * (D + 2) * sizeof(float))) : (mc + 63) / 64;
        if (cap < 2) return nvr::fail(NVR_ERR_INVALID_ARG, "attention: workspace too small for the shared-prefix pass");
        while (sparts > 1 && sparts > cap / 2) { --sparts; while (n64 % sparts) --sparts; }
        shared_part = a.shared_len / sparts;
        int64_t want = pairs >= 192 ? 1 : (256 + pairs - 1) / pairs;
        want = std::max<int64_t>(1, std::min<int64_t>(want, cap - sparts));
        part_size = (int)(((rest + want - 1) / want + 63) / 64 * 64);
        np = sparts + (int)((rest + part_size - 1) / part_size);
        // waves per (query, kv head, own partition) workgroup: 4 while that gives the chip its ~4096 waves; with more workgroups than that
        // (configs[4]: 512 x 8 pairs of ~94 own keys) a workgroup's life is mostly the part with no K/V request in flight — cross-wave merge,
        // two barriers, the partition merge behind 24 keys per wave (r05 stamps: ~4 of ~9 us, four rounds of 1024 workgroups) — so one wave
        // streams the whole partition and 16 independent waves per CU cover each other's tails
        const int64_t own_wgs = pairs * (np - sparts);
        waves = own_wgs >= 2048 ? waves_for(own_wgs, part_size, 4) : 4;
    } else if (a.workspace) {
        if (waves >= 8) {
            int64_t want = pairs >= 192 ? 1 : (256 + pairs - 1) / pairs;
            int64_t ps = ((mc + want - 1) / want + 63) / 64 * 64;
            part_size = (int)ps; np = (int)((mc + ps - 1) / ps);
        } else np = parts_for(a.nq, a.KVH, mc, waves, &part_size);
    }
    // Work-balanced form (attn_share_kernel): a pair count whose last round of one-workgroup-per-pair launches would be mostly empty — 257..2047 pairs with
    // >= 15 % of the rounds' slots unused (33..37 and 65..74 sequences at 8 kv heads, ...), contexts long enough for the stream to matter — or a RAGGED batch
    // (balance_hint: the caller knows the sum of the contexts; per-pair launches size every pair's partitions by the LONGEST context and leave the short sequences'
    // workgroups idle, or — with several pairs per CU — end on the long sequences' workgroups: 32 sequences of 256..8192 keys 4.09 -> 2.43 ms per Qwen3-0.6B step,
    // 64 of 64..4096 keys 3.26 -> 2.78).  NVR_ATTN_SHARE=0: off.
    if constexpr (G * D <= 512)
    if (!shared && paged && a.workspace && a.tickets && !a.seq_of_q && waves == 8 && p.bs_shift >= 0 && a.block_size >= 8 && a.nq <= 1024 && share_enabled()) {
        const int64_t cus = share_workgroups(), rounds = (pairs + cus - 1) / cus, units = (mc + 63) / 64;
        const int64_t cap = a.workspace_bytes ? (int64_t)(a.workspace_bytes / ((size_t)a.nq * a.H * (D + 2) * sizeof(float))) : units;
        const bool by_count = pairs > cus && rounds * cus * 100 >= pairs * 115 && mc >= 256;
        // shares: one per CU for the pair-count case (uniform contexts: a second segment per share buys nothing); for ragged batches the caller's count — two per CU
        // (co-resident: twice the rows in flight, and the dispatcher evens out what the static cut leaves), three from 2 pairs per CU on, fewer while a share would
        // hold less than four 64-key units (measured: profiles/r06_priced_levers.txt 14.)
        const int64_t nw = a.balance_hint > 0 ? a.balance_hint : share_mode() == 2 ? cus * (pairs <= 2 * cus ? 2 : 3) : cus;
        if ((by_count || a.balance_hint || share_mode() == 2) && pairs < 2048 && cap >= units) {
            p.part_size = 64; p.num_parts = (int32_t)cap;
            p.part_o = (float *)a.workspace; p.part_ml = p.part_o + (int64_t)a.nq * a.H * cap * D;
            p.part0 = 0; p.kv0 = 0; p.kv0_rows = nullptr; p.tickets = a.tickets;
            attn_share_kernel<D, G, DU / 2, 8><<<dim3((unsigned)nw), dim3(512), 0, s>>>(p, (int)a.nq, (int)nw);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return nvr::fail(NVR_ERR_HIP, "attention (work-balanced) launch failed: %s", hipGetErrorString(e));
            return 0;
        }
    }
    const bool direct = np <= 1;
    if (!direct && a.workspace_bytes && (size_t)a.nq * a.H * np * (D + 2) * sizeof(float) > a.workspace_bytes)
        return nvr::fail(NVR_ERR_INVALID_ARG, "attention: %d partitions of %d queries x %d heads need %zu workspace bytes, %zu given",
                         np, a.nq, a.H, (size_t)a.nq * a.H * np * (D + 2) * sizeof(float), a.workspace_bytes);
    if (direct) { p.part_size = 0x3fffffff; p.num_parts = 1; }
    else {
        p.part_size = part_size; p.num_parts = np;
        p.part_o = (float *)a.workspace;
        p.part_ml = p.part_o + (int64_t)a.nq * a.H * np * D;
    }
    p.part0 = sparts; p.kv0 = shared ? a.shared_len : 0; p.kv0_rows = shared ? a.shared_kv0 : nullptr;
    const int64_t nwg = (int64_t)(p.num_parts - p.part0) * a.KVH * a.nq;
    // split-KV without a shared-prefix pass: the last partition workgroup of a (query, kv head) to finish merges the pair (no merge launch)
    const bool fuse = !direct && !shared && paged && a.tickets != nullptr;
    p.tickets = a.tickets;
    // ... and with a shared-prefix pass in front whose group is the whole batch, one own partition per pair: that workgroup merges (SHM)
    const bool shm = shared && paged && !a.shared_rows && np - sparts == 1 && a.tickets != nullptr;
    if (shared)
        if (int rc = flash_shared_prefix(a.q, a.ldq, a.k, a.v, a.block_tables, a.max_blocks, a.block_size, a.nq, a.H, a.KVH, a.D, a.scale,
                                         shared_part, sparts, np, p.part_o, p.part_ml, s, a.shared_rows, a.shared_count)) return rc;
    {
        if (waves == 8) launch_cfg<D, G, DU / 2, 8, true>(p, paged, direct, nwg, s, fuse);  // K/V streamed once: nt loads
        else if (paged && waves == 1) { if constexpr (G * (D / 4) <= 64) launch_cfg<D, G, DU, 1, true>(p, paged, direct, nwg, s, fuse, shm); }   // (U = 2: 3.18-3.20 against 3.14-3.17 ms per configs[4] step)
        else if (paged && waves == 2) { if constexpr (G * (D / 4) <= 128) launch_cfg<D, G, DU, 2, true>(p, paged, direct, nwg, s, fuse, shm); }
        else if (paged) launch_cfg<D, G, DU, 4, true>(p, paged, direct, nwg, s, fuse, shm);
        else launch_cfg<D, G, DU, 4, false>(p, paged, direct, nwg, s);                     // prefill: rows re-read from L2
    }
    if (!direct && !fuse && !shm)
        attn_merge_kernel<D><<<dim3((unsigned)(((int64_t)a.H * a.nq + 256 / (D / 4) - 1) / (256 / (D / 4)))), dim3(256), 0, s>>>(
            p.part_o, p.part_ml, a.ctx_lens, a.H, part_size, np, p.part0, p.kv0, p.kv0_rows, (int64_t)a.H * a.nq, p.out);
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

}}  // namespace nvr::k / nvr::kb (NVR_DT_NS)
