// flash_prefill.hip — varlen causal prefill attention (K7) and its prefix-cached paged variant (K8) on MFMA.
// reference: Attention::flash_attention_varlen + compute_attention + causal mask, src/layers/attention.rs:177-208,
// 238-261,321-339; flash_attention_varlen_with_cache / gather_cached_kv :211-222,264-318; GQA :419-435.
// Semantics (SURVEY A-9): per head softmax_f32(q·Kᵀ·D^-½ + causal mask)·V, query at absolute position p sees keys 0..p.
// Bound: MFMA (4·D flop per query-key pair and head).
//
// One workgroup = FLASH_WAVES (4) waves = one (tile of 128/G query positions, kv head) = 128 query rows; a wave owns TWO 16-row
// query tiles (G=1: two position blocks of the head; G=2: both heads of one position block; G=4: two of the four heads),
// so every K fragment (ds_read_b128) and every V fragment (ds_read_b64_tr_b16) it reads from LDS feeds two MFMAs, and the
// K/V tiles staged in LDS (global_load_lds, 64 keys per step, double buffered) are shared by all G heads.
// Both products run with the QUERY on the lane:
//   Sᵀ[key, q] = K·Qᵀ   A = K rows from LDS (ds_read_b128, XOR-swizzled image), B = the lane's Q row (registers);
//                       the accumulator holds 4 keys x 1 query per 16-key tile, so the row max / sum are in-lane
//                       plus two cross-group shuffles (no LDS, cdna guide §5.5 T12 "swapped QKᵀ");
//   Oᵀ[d, q]  += Vᵀ·Pᵀ  B = exp'd Sᵀ accumulators converted to fp16 in place (k-slot (g,j) <-> key 16·(j/4)+4g+j%4:
//                       the same permutation is used for A), A = Vᵀ read with ds_read_b64_tr_b16 (hardware
//                       transpose of 4 keys x 16 d); the O accumulator again has the query on the lane, so the
//                       online-softmax rescale is lane-local (and skipped for a whole wave when no lane's max moved).
// LDS images: K [key][16-byte chunks], chunk' = chunk ^ (key & (chunks-1)) — conflict-free for the 4 x 16 lane groups of
// ds_read_b128; V [key][chunks] with the chunk PAIR index XORed by the key (256-byte rows: (key & 7) << 1; 128-byte
// rows: ((key >> 1) & 3) << 1), so the 8 keys a 32-lane group of ds_read_b64_tr_b16 touches sit on 8 different
// 32-byte bank segments (un-swizzled, all 8 keys share one segment: 8-way conflicts, the first version's bottleneck).
// Steps whose 64 keys all precede the tile's first query skip the causal compare.
#include <type_traits>
#include <vector>
#include <map>
#include <algorithm>
#include <cstdio>
#include "kernels.h"
#include "device_utils.h"
#include "../common.h"

namespace nvr { namespace NVR_DT_NS {

struct FlashParams {
    const half_t *q; int64_t ldq;
    const half_t *k, *v; int64_t ldkv;
    const int32_t *block_tables; int32_t max_blocks, block_size, bs_shift;
    const FlashTile *tiles;
    const int32_t *lanes; int32_t nlanes;    // flash2: per-workgroup tile lists, [nlanes + 1 starts | tile indices] (flash_lanes)
    int32_t H, KVH;
    float scale;
    half_t *out;
    // SHARED (decode over a prefix every sequence of the batch shares): queries = one row per sequence, keys = the first
    // shared tokens through block-table row 0, shared_len of them per blockIdx.y; results are split-KV partials (slot (row, head,
    // partition blockIdx.y))
    int32_t nq_total, shared_len, num_parts;
    float *part_o, *part_ml;
    const int32_t *srows, *scount;     // optional: only the rows srows[0 .. *scount) of the batch share the prefix (row srows[0] names its blocks)
};


template <int D>
__device__ __forceinline__ int v_swz(int row) { return D == 128 ? ((row & 7) << 1) : (((row >> 1) & 3) << 1); }

// One K/V tile into LDS by LDS-DMA (global_load_lds_dwordx4: 16 B per lane, 1 KiB per wave-instruction, lane-linear at M0), written as
// inline asm ON PURPOSE: hipcc treats the builtin as a pending write to LDS and drains it (s_waitcnt vmcnt(0)) in front of the first
// transposing read (ds_read_b64_tr_b16) of the tile IN USE, i.e. in the middle of every step, 1-2 k cycles after the requests went
// out (r02 .s: the stall behind the 27 % MFMA-busy figure).  As asm the requests are invisible to its wait insertion; they are
// counted by hand at the end of the step (s_waitcnt vmcnt + barrier in step()), one whole step after they were issued.  N pieces of
// STRIDE-apart LDS destinations (STRIDE = 16 B x the workgroup's threads) starting at `lds` (wave-uniform byte address); M0 is saved and
// restored inside the statement.
template <int N, int STRIDE>
__device__ __forceinline__ void glds_pieces(const half_t *const (&src)[N], unsigned lds) {
    static_assert(N == 2 || N == 4 || N == 8, "1, 2 or 4 pieces per operand");
    unsigned keep;
    if constexpr (N == 8) {
        asm volatile("s_mov_b32 %0, m0\n\t"
                     "s_mov_b32 m0, %9\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                     "s_add_u32 m0, m0, %[st]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
                     "s_add_u32 m0, m0, %[st]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off\n\t"
                     "s_add_u32 m0, m0, %[st]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, off\n\t"
                     "s_add_u32 m0, m0, %[st]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, off\n\t"
                     "s_add_u32 m0, m0, %[st]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %6, off\n\t"
                     "s_add_u32 m0, m0, %[st]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %7, off\n\t"
                     "s_add_u32 m0, m0, %[st]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %8, off\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(src[0]), "v"(src[1]), "v"(src[2]), "v"(src[3]), "v"(src[4]), "v"(src[5]), "v"(src[6]), "v"(src[7]), "s"(lds), [st] "n"(STRIDE)
                     : "memory", "scc");
    } else if constexpr (N == 2) {
        asm volatile("s_mov_b32 %0, m0\n\t"
                     "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                     "s_add_u32 m0, m0, %[st]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(src[0]), "v"(src[1]), "s"(lds), [st] "n"(STRIDE)
                     : "memory", "scc");
    } else {
        asm volatile("s_mov_b32 %0, m0\n\t"
                     "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                     "s_add_u32 m0, m0, %[st]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
                     "s_add_u32 m0, m0, %[st]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off\n\t"
                     "s_add_u32 m0, m0, %[st]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, off\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(src[0]), "v"(src[1]), "v"(src[2]), "v"(src[3]), "s"(lds), [st] "n"(STRIDE)
                     : "memory", "scc");
    }
}

// max over the four 16-lane rows of a wave (the 4 key quads of a query), in the vector ALU: v_permlane16_swap / v_permlane32_swap of a
// value with itself leave {own, partner} in the two results (an LDS round trip per ds_bpermute before: two dependent ones per step)
__device__ __forceinline__ float max_over_rows(float v) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

#ifndef NVR_FLASH_KT
#define NVR_FLASH_KT 64
#define NVR_FLASH_NBUF 2
#endif
#ifndef NVR_FLASH_WAVES
#define NVR_FLASH_WAVES 4              // measured: 8 waves (one 256-row workgroup per CU) 252 us vs 4 waves x 2 workgroups 240 us per layer
#endif
constexpr int FLASH_WAVES = NVR_FLASH_WAVES;     // waves per workgroup: 32 query rows each share the staged K/V tiles
// UB (paged only): block_size is a power of two and a multiple of the 64-key step, so a step lies inside ONE cache block: its
// block-table entry is a scalar read (v_readlane) from a register copy of the table (lane j holds entry 64·c + j) — no
// dependent table load in front of the LDS-DMA requests (r02, 32 x 1024 through the block tables: 348 us per layer with a lookup per
// piece, 260 us with this; the contiguous form: 250 us).
// SHARED: the decode step's attention over a prefix that EVERY sequence of the batch holds in the same cache blocks (BASELINE
// configs[4]: 512 sequences behind one 512-token system prompt).  The row kernel re-reads those K/V rows once per sequence (from
// L2, but latency-paced: 113 us per layer at 512 x ~600); here the batch's query rows form the M dimension of the same MFMA
// schedule — 128/G sequences per workgroup, no causal mask, every key of [0, shared_len) visible to every row — and the result
// leaves as the split-KV partial of partition 0 (unnormalised f32 o, scaled max, sum), merged with the per-sequence remainder
// by attn_merge_kernel.
template <int D, int G, bool PAGED, bool UB = false, bool SHARED = false>
__global__ __launch_bounds__(64 * FLASH_WAVES, 8 / FLASH_WAVES) void flash_prefill_kernel(FlashParams p) {   // 2 waves per SIMD: <= 256 registers per lane
    constexpr int NT = 64 * FLASH_WAVES;         // threads
    constexpr int KT = NVR_FLASH_KT;             // keys per step
    constexpr int NBUF = NVR_FLASH_NBUF;         // K/V tiles in the LDS ring: NBUF-1 in flight ahead of the one in use
    constexpr int CPR = D / 8;                   // 16-byte chunks per K/V row
    constexpr int PIECES = KT * CPR / NT;        // 16-byte pieces per thread and operand
    constexpr int NKS = D / 32, NDT = D / 16, NQT = 2, NMT = KT / 16, NK2 = KT / 32;
    constexpr int STAGE = 2 * KT * D * 2;        // bytes of one ring slot [K | V]
    __shared__ __attribute__((aligned(16))) char smem[NBUF * STAGE];

    const int g = blockIdx.x % p.KVH;
    FlashTile tile;
    if (SHARED) {
        constexpr int QB = 32 * FLASH_WAVES / G;                      // sequences per workgroup
        tile.q_row0 = (int)(blockIdx.x / p.KVH) * QB;
        tile.nq = min(QB, (p.scount ? *p.scount : p.nq_total) - tile.q_row0);
        if (tile.nq <= 0) return;                                     // (workgroup-uniform: the launch is sized for the whole batch)
        tile.pos0 = 0x3fffffff; tile.kv_ref = p.srows ? p.srows[0] : 0;
    } else tile = p.tiles[blockIdx.x / p.KVH];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 15, g4 = lane >> 4;
    // SHARED: blockIdx.y cuts the shared keys into partitions of shared_len tokens (more workgroups than 8 per kv head at 512 sequences)
    const int kv_start = SHARED ? (int)blockIdx.y * p.shared_len : 0;
    const int kv_end = SHARED ? kv_start + p.shared_len : tile.pos0 + tile.nq;

    // the wave's two query tiles: head and position block
    int qi[NQT], head[NQT], qpos[NQT], qrow[NQT]; bool qvalid[NQT];
    half8_t qf[NQT][NKS];
#pragma unroll
    for (int t = 0; t < NQT; ++t) {
        int pblk;
        if (G == 1) { head[t] = g; pblk = wave * 2 + t; }
        else if (G == 2) { head[t] = g * 2 + t; pblk = wave; }
        else { head[t] = g * 4 + (wave & 1) * 2 + t; pblk = wave >> 1; }
        qi[t] = pblk * 16 + r;
        qvalid[t] = qi[t] < tile.nq;
        const int qc = qvalid[t] ? qi[t] : tile.nq - 1;
        qpos[t] = SHARED ? 0x3fffffff : tile.pos0 + qc;           // absolute position = last visible key
        qrow[t] = (SHARED && p.srows) ? p.srows[tile.q_row0 + qc] : tile.q_row0 + qc;
        const half_t *qptr = p.q + (int64_t)qrow[t] * p.ldq + (int64_t)head[t] * D + g4 * 8;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) qf[t][ks] = *reinterpret_cast<const half8_t *>(qptr + ks * 32);
    }

    // last key any lane of this wave may attend to: a step that starts beyond it does no arithmetic in this wave
    const int wave_last = SHARED ? kv_end - 1
        : __builtin_amdgcn_readfirstlane(tile.pos0 + min(tile.nq - 1, (G == 1 ? wave * 2 + 1 : (G == 2 ? wave : (wave >> 1))) * 16 + 15));
    int bt_reg = 0, bt_chunk = -1;                                    // UB: register copy of 64 block-table entries
    // The chunk is fetched by an inline-asm load that waits for itself: a compiler-visible load inside the step loop (the reload of a
    // context longer than 64 blocks) makes hipcc put s_waitcnt vmcnt(0) in front of EVERY v_readlane of bt_reg, i.e. into every step —
    // and that wait also drains the K/V tiles in flight by LDS-DMA (which the compiler does not see): the ring then runs one tile deep
    // (paged form 258 vs 217 us per layer at 32 x 1024).  The reload itself is rare (once per 64 blocks) and may block.
    auto load_bt_chunk = [&](int c) {
        bt_chunk = c;
        const int idx = min((c << 6) + lane, p.max_blocks - 1);                  // (entries past the table are never selected)
        const int32_t *src = p.block_tables + (int64_t)tile.kv_ref * p.max_blocks + idx;
        asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(bt_reg) : "v"(src) : "memory");
    };
    // (the first chunk as an ordinary load, overlapped with the query loads, was measured: hipcc then waits for it in every step again,
    // 36.7 ms per 32 x 1024 prefill against 35.1 with the blocking asm load)
    if (PAGED && UB) load_bt_chunk(0);

    // Source addresses of a tile.  Contiguous K/V and block-aligned paged steps (UB) are "scalar base of the step + a per-thread
    // constant": one 64-bit add per 16-byte piece (the general form below costs a 64-bit multiply chain per piece, ~50 VALU
    // instructions per step next to 68 MFMAs).  Keys beyond the last visible one (the final step of a tile) are clamped to it.
    constexpr int V_OFF = KT * D * 2;            // V image behind the K image of a ring slot
    static_assert(V_OFF == PIECES * NT * 16, "glds_pieces walks the pieces of the K image, then of the V image, NT * 16 bytes apart");
    const unsigned lds_ring = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    int64_t kconst[PIECES], vconst[PIECES];      // element offsets of this thread's pieces inside a step
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
        const int idx = i * NT + threadIdx.x, row = idx / CPR, c = idx % CPR;
        const int vsw = D == 128 ? ((row & 7) << 1) : (((row >> 1) & 3) << 1);      // = v_swz<D>(row), spelled out: a call here makes hipcc drop the host stub
        const int64_t rowoff = (PAGED && UB) ? ((int64_t)row * p.KVH + g) * D : (int64_t)row * p.ldkv + (int64_t)g * D;
        kconst[i] = rowoff + (c ^ (row & (CPR - 1))) * 8;
        vconst[i] = rowoff + (c ^ vsw) * 8;
    }
    // the query fragments must have LANDED before the first LDS-DMA goes out: hipcc would otherwise keep its own wait for them inside the
    // step loop (it cannot prove they arrived before the loop), and in hardware that vmcnt(0) also waits for the hidden LDS-DMA
#pragma unroll
    for (int t = 0; t < NQT; ++t)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) asm volatile("" :: "v"(qf[t][ks]));
    auto stage = [&](int buf, int kt) {
        const half_t *src[2 * PIECES];
        const bool whole = kt + KT <= kv_end;                             // uniform: no key of this step is clamped
        if ((!PAGED || UB) && whole) {
            int64_t sbase;
            if (PAGED) {
                const int bi = kt >> p.bs_shift;                              // every key of the step is in this block
                if ((bi >> 6) != bt_chunk) load_bt_chunk(bi >> 6);
                sbase = ((int64_t)__builtin_amdgcn_readlane(bt_reg, bi & 63) * p.block_size + (kt & (p.block_size - 1))) * p.KVH * D;
            } else sbase = (int64_t)(tile.kv_ref + kt) * p.ldkv;
#pragma unroll
            for (int i = 0; i < PIECES; ++i) { src[i] = p.k + sbase + kconst[i]; src[PIECES + i] = p.v + sbase + vconst[i]; }
        } else {
            int64_t blk_row0 = 0;
            if (PAGED && UB) {
                const int bi = kt >> p.bs_shift;
                if ((bi >> 6) != bt_chunk) load_bt_chunk(bi >> 6);
                blk_row0 = (int64_t)__builtin_amdgcn_readlane(bt_reg, bi & 63) * p.block_size;
            }
#pragma unroll
            for (int i = 0; i < PIECES; ++i) {
                const int idx = i * NT + threadIdx.x, row = idx / CPR, c = idx % CPR;
                int key = kt + row; if (key > kv_end - 1) key = kv_end - 1;
                int64_t off;
                if (PAGED && UB) {
                    off = ((blk_row0 + (key & (p.block_size - 1))) * p.KVH + g) * D;
                } else if (PAGED) {
                    int bi, bo;
                    if (p.bs_shift >= 0) { bi = key >> p.bs_shift; bo = key & (p.block_size - 1); }
                    else { bi = key / p.block_size; bo = key - bi * p.block_size; }
                    const int64_t rr = (int64_t)p.block_tables[(int64_t)tile.kv_ref * p.max_blocks + bi] * p.block_size + bo;
                    off = (rr * p.KVH + g) * D;
                } else {
                    off = (int64_t)(tile.kv_ref + key) * p.ldkv + (int64_t)g * D;
                }
                const int vsw = D == 128 ? ((row & 7) << 1) : (((row >> 1) & 3) << 1);
                src[i] = p.k + off + (c ^ (row & (CPR - 1))) * 8;
                src[PIECES + i] = p.v + off + (c ^ vsw) * 8;
            }
        }
        glds_pieces<2 * PIECES, NT * 16>(src, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_ring + buf * STAGE + wave_u * 1024)));
    };

    // The softmax runs on RAW scores: m is the running max of q·k (the scale is positive), p = 2^((s - m)·c) with
    // c = scale·log2(e) is one v_fma + one v_exp per element, and the row sums come from the matrix core (an all-ones A
    // fragment against the same fp16 P fragments that multiply V), so the VALU work per 64-key step is ~55 instructions
    // per query tile next to 36 MFMAs.
    float4_t o[NQT][NDT], ol[NQT];
    float m[NQT];
#pragma unroll
    for (int t = 0; t < NQT; ++t) {
        m[t] = -INFINITY; ol[t] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NDT; ++i) o[t][i] = (float4_t){0.f, 0.f, 0.f, 0.f};
    }
    const float c2 = p.scale * 1.44269504088896340736f;
    half8_t ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (half_t)1.0f;

    const int nsteps = (kv_end - kv_start + KT - 1) / KT;
    // step `it` computes on ring slot it % NBUF while tile it+NBUF-1 is requested into the slot step it-1 just released;
    // before its closing barrier every wave waits until only the loads of tiles it+2.. are outstanding, so tile it+1 has
    // landed for all waves after the barrier and the global loads never drain inside the stream.
    auto step = [&](auto cur_c, int it) {
        constexpr int cur = decltype(cur_c)::value;
        const int kt = kv_start + it * KT;
        const bool more = it + NBUF - 1 < nsteps;
        if (more) stage((cur + NBUF - 1) % NBUF, kt + (NBUF - 1) * KT);
        const char *kl = smem + cur * STAGE, *vl = kl + KT * D * 2;
        if (kt <= wave_last) {

        // Sᵀ tiles: NMT x (16 keys x 16 queries) per query tile; one K fragment read feeds both
        float4_t s[NQT][NMT];
        // K fragments of key tile mt+1 are requested before the MFMAs of key tile mt (two fragment sets in registers):
        // without this the compiler emits {2 reads, wait, 4 MFMAs} eight times and every LDS latency is exposed
        half8_t kf[2][NKS];
        auto read_k = [&](int mt, half8_t (&dst)[NKS]) {
            const int row = mt * 16 + r;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
                dst[ks] = *reinterpret_cast<const half8_t *>(kl + (row * CPR + ((ks * 4 + g4) ^ (row & (CPR - 1)))) * 16);
        };
        read_k(0, kf[0]);
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) {
            if (mt + 1 < NMT) read_k(mt + 1, kf[(mt + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NQT; ++t) s[t][mt] = (float4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                for (int t = 0; t < NQT; ++t) s[t][mt] = mfma16(kf[mt & 1][ks], qf[t][ks], s[t][mt]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // causal mask only on steps that reach past the tile's first query (keys kt + mt*16 + g4*4 + e)
        if (!SHARED && kt + KT - 1 > tile.pos0) {
#pragma unroll
            for (int t = 0; t < NQT; ++t)
#pragma unroll
                for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (kt + mt * 16 + g4 * 4 + e > qpos[t]) s[t][mt][e] = -INFINITY;
        }
        half8_t pf[NQT][NK2];
        bool moved = false;
        float alpha[NQT];
#pragma unroll
        for (int t = 0; t < NQT; ++t) {
            float mx = fmaxf(fmaxf(s[t][0][0], s[t][0][1]), fmaxf(s[t][0][2], s[t][0][3]));
#pragma unroll
            for (int mt = 1; mt < NMT; ++mt) mx = fmaxf(fmaxf(mx, s[t][mt][0]), fmaxf(fmaxf(s[t][mt][1], s[t][mt][2]), s[t][mt][3]));
            mx = max_over_rows(mx);
            const float mn = fmaxf(m[t], mx);                   // finite from the first step on (key 0 <= qpos)
            alpha[t] = __builtin_amdgcn_exp2f((m[t] - mn) * c2);
            moved |= mn != m[t];
            m[t] = mn;
            const float mc = -mn * c2;
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    pf[t][mt >> 1][(mt & 1) * 4 + e] = (half_t)__builtin_amdgcn_exp2f(fmaf(s[t][mt][e], c2, mc));
        }
        if (__any(moved)) {                                     // wave-uniform: the running max settles after a few steps
#pragma unroll
            for (int t = 0; t < NQT; ++t) {
                ol[t] *= alpha[t];
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) o[t][dt] *= alpha[t];
            }
        }
        // Oᵀ += Vᵀ·Pᵀ per 32-key half; one V fragment (two transposing reads) feeds both query tiles
#pragma unroll
        for (int k2 = 0; k2 < NK2; ++k2) {
            const int row0 = (2 * k2) * 16 + g4 * 4 + (r >> 2), row1 = row0 + 16;
            const int sw = v_swz<D>(row0);                      // same for row1 (row1 = row0 + 16)
#pragma unroll
            for (int t = 0; t < NQT; ++t) ol[t] = mfma16(ones, pf[t][k2], ol[t]);
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const int cb = ((((dt * 2 + ((r & 3) >> 1)) ^ sw) << 4) | ((r & 1) << 3));   // byte offset inside the row
                const half4_t a0 = lds_read_tr16(vl + row0 * (D * 2) + cb), a1 = lds_read_tr16(vl + row1 * (D * 2) + cb);
                half8_t vf;
#pragma unroll
                for (int e = 0; e < 4; ++e) { vf[e] = a0[e]; vf[4 + e] = a1[e]; }
#pragma unroll
                for (int t = 0; t < NQT; ++t) o[t][dt] = mfma16(vf, pf[t][k2], o[t][dt]);
            }
        }
        }   // kt <= wave_last
        if (it + NBUF < nsteps) {                               // NBUF-1 younger tiles were requested: leave NBUF-2 of them flying
            if (NBUF == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (NBUF == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 * 2 * PIECES) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * 2 * PIECES) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
    };

#pragma unroll
    for (int b = 0; b < NBUF - 1; ++b)
        if (b < nsteps) stage(b, kv_start + b * KT);
    if (NBUF - 1 < nsteps) {
        if (NBUF == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (NBUF == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 * 2 * PIECES) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * 2 * PIECES) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    for (int it = 0; it < nsteps; it += NBUF) {                 // ring slot as a compile-time constant: LDS addresses fold
        step(std::integral_constant<int, 0>{}, it);
        if (it + 1 < nsteps) step(std::integral_constant<int, 1 % NBUF>{}, it + 1);
        if (NBUF > 2 && it + 2 < nsteps) step(std::integral_constant<int, 2 % NBUF>{}, it + 2);
        if (NBUF > 3 && it + 3 < nsteps) step(std::integral_constant<int, 3 % NBUF>{}, it + 3);
    }

    if (SHARED) {                                               // partial of partition 0: (o, m * scale, l) as attn_rows_kernel writes them
#pragma unroll
        for (int t = 0; t < NQT; ++t) {
            if (!qvalid[t]) continue;
            const int64_t slot = ((int64_t)qrow[t] * p.H + head[t]) * p.num_parts + blockIdx.y;
            float *po = p.part_o + slot * D + g4 * 4;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) *reinterpret_cast<float4_t *>(po + dt * 16) = o[t][dt];
            if (g4 == 0) { p.part_ml[slot * 2] = m[t] * p.scale; p.part_ml[slot * 2 + 1] = ol[t][0]; }
        }
        return;
    }
#pragma unroll
    for (int t = 0; t < NQT; ++t) {
        if (qvalid[t]) {
            const float ls = ol[t][0];                          // every row of the ones-product holds the query's sum
            const float inv = ls > 0.f ? 1.0f / ls : 0.f;
            half_t *orow = p.out + ((int64_t)qrow[t] * p.H + head[t]) * D + g4 * 4;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                half4_t hv = {(half_t)(o[t][dt][0] * inv), (half_t)(o[t][dt][1] * inv), (half_t)(o[t][dt][2] * inv), (half_t)(o[t][dt][3] * inv)};
                *reinterpret_cast<half4_t *>(orow + dt * 16) = hv;
            }
        }
    }
}


// ======================================================================================================================================
// Second-generation prefill kernel (r04), head_dim 128: PERSISTENT workgroups of 256 query rows, 32x32x16 MFMAs, a 4-slot LDS-DMA ring.
//
// Why (profiles/r04_flash_prefill.txt): the kernel above moves 64 KiB of K/V per CU and 64-key step from L2 into LDS for 2 x 128 query
// rows with ONE tile in flight per workgroup (request at the start of a step, s_waitcnt vmcnt(0) at its end) and runs QKᵀ -> softmax ->
// P·V strictly one after the other in every wave on 16x16x32 MFMAs, which hold the SIMD's issue port for 8 of their 16 cycles: 32 %
// matrix-pipe busy.  Here
//   * one workgroup = 8 waves = 256 query rows (256 / G positions x the G heads of a kv head): half the L2 -> LDS bytes per FLOP;
//   * a wave owns 32 query rows and works on 32x32x16 MFMAs (24 of 32 cycles free for vector instructions of either wave of the SIMD):
//       Sᵀ[key, q] = K·Qᵀ   A = K rows (ds_read_b128, chunk ^ (row & 15): conflict-free), B = the lane's Q row (registers, 8 fragments)
//                           accumulator: query on the lane (l & 31), 16 keys of a 32-key half in the registers, the other 16 in lane l ^ 32
//       Oᵀ[d, q] += Vᵀ·Pᵀ   B = the exp'd accumulators packed to 16 bits IN PLACE (registers 8s..8s+7 = k-step s, k order
//                           16s + 8(j>>2) + 4h + (j&3); cdna guide §3 "an accumulator tile as the next MFMA's operand"), A = Vᵀ by two
//                           ds_read_b64_tr_b16 in that same key order (V image: 16-byte chunk ^ ((key & 3) << 2): the 32 lanes of a half
//                           touch 32 different 8-byte bank slots);
//     row max = 15 in-lane v_max + one v_permlane32_swap, row sum = in-lane f32 adds (halves combined once in the epilogue);
//   * software pipeline over 32-key halves with two named score states: QKᵀ of half u+1 is issued next to the exp / pack of half u and
//     P·V of half u, every operand fragment read from LDS one group of four MFMAs ahead (guide T15); the running max is only raised
//     when a row's maximum grew by more than 2^6 (guide T13: one wave-uniform, rarely taken rescale branch instead of 64 multiplies);
//   * K/V tiles by LDS-DMA into NBUF slots of 32 KiB, requested NBUF-1 steps ahead, counted vmcnt + ONE barrier per 64-key step;
//   * PERSISTENT: a workgroup is one (lane, kv head) of the launch and walks the lane's list of tiles (built on the host: tiles dealt to the
//     least-loaded lane in list order, flash_lanes).  The first build launched one workgroup per tile: 8.0 us between two workgroups of
//     a CU (128 KiB of LDS per workgroup), 10 k cycles of prologue per tile and a 39 us tail of idle CUs in a 311 us launch.  Here the
//     K/V requests run ahead ACROSS tile boundaries (a request cursor of its own) and the next tile's Q rows are loaded during the
//     last step of the current one.
constexpr int F2_ROWS = 256;                 // query rows per workgroup
constexpr float F2_DEFER = 6.0f;             // log2 units a row maximum may run ahead of the running max before the rescale branch is taken

__device__ __forceinline__ float16_t mfma32(half8_t a, half8_t b, float16_t c) {
#ifdef NVR_BF16
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#endif
}

#ifdef NVR_F2_STAMPS   // diagnostic build only (tools/build_variant.sh): shader-clock sums per phase of wave 0 of every workgroup
__device__ unsigned long long f2_stamp_buf[1024 * 12];
#define F2_STAMP(var) unsigned long long var; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0)
#else
#define F2_STAMP(var)
#endif
// MODE 0: K/V rows contiguous (p.ldkv apart, first row tile.kv_ref); 1: block tables, any block size; 2: block tables, block size a
// power of two and a multiple of the 64-key step (a step lies inside one block)
template <int G, int MODE, int NBUF>
__global__ __launch_bounds__(256, 1) void flash2_kernel(FlashParams p) {   // ONE wave per SIMD: the whole 512-register file per wave
    constexpr int D = 128, NT = 256, KT = 64, CPR = 16, PIECES = 4, NQ = 2;
    constexpr int STAGE = 2 * KT * D * 2, V_OFF = KT * D * 2;
    constexpr int PPW = 32 / G;                                       // query positions per 32-row query tile
    constexpr bool PAGED = MODE != 0, UB = MODE == 2;
    static_assert(V_OFF == PIECES * NT * 16, "the pieces of the K image, then of the V image, NT * 16 bytes apart");
    static_assert(NBUF == 4, "ring depth: K of step t+1 is read during step t (two steps ahead must have landed), one more tile may fly");
    extern __shared__ __attribute__((aligned(16))) char smem[];       // NBUF x [K image | V image]

    F2_STAMP(ts0);
    const int g = blockIdx.x % p.KVH, my_lane = blockIdx.x / p.KVH;
    const int it0 = p.lanes[my_lane], it1 = p.lanes[my_lane + 1];     // this workgroup's tiles: items[it0 .. it1)
    if (it0 >= it1) return;
    const int32_t *items = p.lanes + p.nlanes + 1;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = lane & 31, h = lane >> 5;
    const int head = g * G + c / PPW;                                 // a query tile's 32 columns are PPW positions x G heads
    int qi[NQ];                                                       // the lane's query positions inside a workgroup tile (query tiles 2 wave, 2 wave + 1)
#pragma unroll
    for (int j = 0; j < NQ; ++j) qi[j] = (wave * NQ + j) * PPW + c % PPW;

    // ---- request cursor: the K/V tile of (item rq_i, step rq_t) goes to ring slot rq_gs % NBUF -------------------------------------------
    int rq_i = it0, rq_t = 0, rq_gs = 0;
    FlashTile rq = p.tiles[items[it0]];
    int rq_end = rq.pos0 + rq.nq, rq_steps = (rq_end + KT - 1) / KT;
    int bt_reg = 0, bt_chunk = -1;                                    // UB: register copy of 64 block-table entries of the request cursor's sequence
    auto load_bt_chunk = [&](int cidx) {
        bt_chunk = cidx;
        const int idx = min((cidx << 6) + lane, p.max_blocks - 1);
        const int32_t *src = p.block_tables + (int64_t)rq.kv_ref * p.max_blocks + idx;
        asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(bt_reg) : "v"(src) : "memory");
    };
    // the ring starts on a 256-byte boundary (an LDS bank row = one K/V row): the read addresses below are formed with XORs
    const unsigned lds_ring = ((unsigned)(size_t)(__attribute__((address_space(3))) char *)smem + 255u) & ~255u;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    int kconst[PIECES], vconst[PIECES];                               // element offsets of this thread's pieces inside a step (< 2^31: checked by the launcher)
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
        const int idx = i * NT + threadIdx.x, row = idx / CPR, cc = idx % CPR;
        const int rowoff = UB ? (row * p.KVH + g) * D : row * (int)p.ldkv + g * D;
        kconst[i] = rowoff + (cc ^ (row & 15)) * 8;
        vconst[i] = rowoff + (cc ^ ((row & 3) << 2)) * 8;
    }
    // A request = the eight 1-KiB-per-wave pieces of one K/V tile: addressed by stage(), sent as its K half and its V half (request_k / request_v:
    // the loop puts one half behind each of a step's two halves)
    const half_t *src[2 * PIECES];
    unsigned rq_dst = 0;
    auto stage = [&](int buf, int kt) {                               // the request cursor's tile, keys kt .. kt+63, into ring slot buf
        const bool whole = kt + KT <= rq_end;                         // uniform: no key of this step is clamped
        if ((!PAGED || UB) && whole) {
            int64_t sbase;
            if (PAGED) {
                const int bi = kt >> p.bs_shift;
                if ((bi >> 6) != bt_chunk) load_bt_chunk(bi >> 6);
                sbase = ((int64_t)__builtin_amdgcn_readlane(bt_reg, bi & 63) * p.block_size + (kt & (p.block_size - 1))) * p.KVH * D;
            } else sbase = (int64_t)(rq.kv_ref + kt) * p.ldkv;
#pragma unroll
            for (int i = 0; i < PIECES; ++i) { src[i] = p.k + sbase + kconst[i]; src[PIECES + i] = p.v + sbase + vconst[i]; }
        } else {
            int64_t blk_row0 = 0;
            if (UB) {
                const int bi = kt >> p.bs_shift;
                if ((bi >> 6) != bt_chunk) load_bt_chunk(bi >> 6);
                blk_row0 = (int64_t)__builtin_amdgcn_readlane(bt_reg, bi & 63) * p.block_size;
            }
#pragma unroll
            for (int i = 0; i < PIECES; ++i) {
                const int idx = i * NT + threadIdx.x, row = idx / CPR, cc = idx % CPR;
                int key = kt + row; if (key > rq_end - 1) key = rq_end - 1;
                int64_t off;
                if (UB) {
                    off = ((blk_row0 + (key & (p.block_size - 1))) * p.KVH + g) * D;
                } else if (PAGED) {
                    int bi, bo;
                    if (p.bs_shift >= 0) { bi = key >> p.bs_shift; bo = key & (p.block_size - 1); }
                    else { bi = key / p.block_size; bo = key - bi * p.block_size; }
                    const int64_t rr = (int64_t)p.block_tables[(int64_t)rq.kv_ref * p.max_blocks + bi] * p.block_size + bo;
                    off = (rr * p.KVH + g) * D;
                } else {
                    off = (int64_t)(rq.kv_ref + key) * p.ldkv + (int64_t)g * D;
                }
                src[i] = p.k + off + (cc ^ (row & 15)) * 8;
                src[PIECES + i] = p.v + off + (cc ^ ((row & 3) << 2)) * 8;
            }
        }
        rq_dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_ring + buf * STAGE + wave_u * 1024));
    };
    bool rq_open = false;                                             // a staged request whose V half has not gone out yet
    auto request_k = [&]() {                                          // one virtual step further (across tile boundaries): address it, send the K half
        rq_open = rq_i < it1;
        if (!rq_open) return;
        stage(__builtin_amdgcn_readfirstlane(rq_gs % NBUF), rq_t * KT);
        ++rq_gs;
        if (++rq_t == rq_steps) {
            rq_t = 0; bt_chunk = -1;
            if (++rq_i < it1) { rq = p.tiles[items[rq_i]]; rq_end = rq.pos0 + rq.nq; rq_steps = (rq_end + KT - 1) / KT; }
        }
        const half_t *const ks[PIECES] = {src[0], src[1], src[2], src[3]};
        glds_pieces<PIECES, NT * 16>(ks, rq_dst);
    };
    auto request_v = [&]() {
        if (!rq_open) return;
        const half_t *const vs[PIECES] = {src[PIECES], src[PIECES + 1], src[PIECES + 2], src[PIECES + 3]};
        glds_pieces<PIECES, NT * 16>(vs, rq_dst + V_OFF);
        rq_open = false;
    };

    // ---- per-tile lane data and the Q fragments ----------------------------------------------------------------------------------------------
    FlashTile tile = rq;                                              // the compute cursor's tile (= the first one)
    half8_t qf[NQ][8];
    int qpos[NQ], qrow[NQ];
    auto lane_of_tile = [&](const FlashTile &t) {
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const int qc = qi[j] < t.nq ? qi[j] : t.nq - 1;           // rows past the tile's last query are copies of that query
            qpos[j] = t.pos0 + qc;                                    // absolute position = last visible key
            qrow[j] = t.q_row0 + qc;
        }
    };
    // The Q rows are loaded by inline asm STRAIGHT INTO ACCUMULATOR REGISTERS (the MFMA takes its B operand from there) and waited for by hand
    // (F2_Q_WAIT): as ordinary loads hipcc places the wait itself and, unable to tell the first tile (nothing behind the loads) from the later ones
    // (the previous tile's 16 output stores behind them), waits for vmcnt(0) at every tile boundary — the stores' round trip to HBM.  Nothing
    // touches qf between the two statements (guide §5.7 form (ii)).
    auto load_q = [&]() {
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const half_t *qptr = p.q + (int64_t)qrow[j] * p.ldq + (int64_t)head * D + h * 8;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(qf[j][ks]) : "v"(qptr + ks * 16) : "memory");
        }
    };
#define F2_Q_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")" : "+a"(qf[0][0]), "+a"(qf[0][1]), "+a"(qf[0][2]), "+a"(qf[0][3]), "+a"(qf[0][4]), "+a"(qf[0][5]), "+a"(qf[0][6]), "+a"(qf[0][7]), \
                                                     "+a"(qf[1][0]), "+a"(qf[1][1]), "+a"(qf[1][2]), "+a"(qf[1][3]), "+a"(qf[1][4]), "+a"(qf[1][5]), "+a"(qf[1][6]), "+a"(qf[1][7]) :: "memory")
    lane_of_tile(tile);
    load_q();
    if (UB) load_bt_chunk(0);
#pragma unroll
    for (int b = 0; b < NBUF - 1; ++b) { request_k(); request_v(); }

    // LDS read addresses of this lane inside ring slot 0
    const unsigned kbase = lds_ring + c * 256 + ((h ^ (c & 15)) << 4);   // K row c (+ 32 mt), 16-byte chunk (2 ks + h) ^ (row & 15): ^ (ks << 5)
    const int vq = (lane & 15) >> 2, vp = lane & 3, vg = (lane >> 4) & 1;
    const unsigned vbase = lds_ring + V_OFF + (4 * h + vq) * 256 + ((((vq << 2) | (vg << 1) | (vp >> 1)) << 4) | ((vp & 1) << 3));   // ^ (dt << 6)

    float16_t o[NQ][4];
    float m[NQ], mneg[NQ], lsum[NQ][4];                               // (four partial row sums: no 16-deep dependent add chain per half)
    const float c2 = p.scale * 1.44269504088896340736f;

    // causal mask of 32 keys from k0 on (only where they reach past the tile's first query: a small wave-uniform branch that touches S alone)
    auto mask = [&](float16_t &S, int j, int k0) {
        if (k0 + 31 > tile.pos0) {
            const int lim = qpos[j] - k0 - 4 * h;                     // key k0 + (e & 3) + 8 (e >> 2) + 4 h is visible iff <= qpos
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if ((e & 3) + 8 * (e >> 2) > lim) S[e] = -INFINITY;
        }
    };
    // row maximum and the rare raise of the running maximum; first (wave-uniform): the tile's first keys — nothing accumulated yet, the maximum is
    // taken as it is (key 0 is visible to every query: it is finite)
    auto stats = [&](bool first, const float16_t &S, int j) {
        float mx = fmaxf(S[0], S[1]);
#pragma unroll
        for (int e = 2; e < 16; e += 2) mx = fmaxf(fmaxf(mx, S[e]), S[e + 1]);
        mx = xor32_partner_max(mx);
        const bool need = !first && (mx - m[j]) * c2 > F2_DEFER;
        if (__any(need)) {                                            // wave-uniform, rare
            const float mn = fmaxf(m[j], mx);
            const float alpha = __builtin_amdgcn_exp2f((m[j] - mn) * c2);
            m[j] = mn; mneg[j] = -mn * c2;
#pragma unroll
            for (int i = 0; i < 4; ++i) lsum[j][i] *= alpha;
            // (the empty statements pin the accumulator-file reads and writes INSIDE this rare branch: hipcc otherwise hoists the 64
            //  v_accvgpr_read of a tile's output above the branch, into every half)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                asm volatile("s_nop 7\n\ts_nop 3" : "+a"(o[j][dt]));
#pragma unroll
                for (int e = 0; e < 16; ++e) o[j][dt][e] *= alpha;
                asm volatile("" : "+a"(o[j][dt]));
            }
        }
        if (first) { m[j] = mx; mneg[j] = -mx * c2; }
    };
    // ---- the matrix work is issued by inline asm -----------------------------------------------------------------------------------------------
    // One wave per SIMD owns 512 registers, but hipcc's allocator does not split them by role: as builtins the MFMAs got their score
    // accumulators in the accumulator file and the Q operands in the vector file, and ~2000 v_accvgpr moves per step shuttled scores to the
    // vector ALU.  As asm statements the classes are stated: O and Q live in accumulator registers ("a"), the scores, the K / V fragments and P in
    // vector registers ("v").  An asm statement is also a scheduling boundary for hipcc: the source order below IS the issue order — one MFMA,
    // then the slice of vector work and LDS reads that its 32 cycles hide (guide: <= 5 single-issue instructions per gap, one of them a v_exp).
    // Hazards hipcc does not see (guide §5.7 item 2): a P fragment written by the vector ALU just before the MFMA that reads it (s_nop 1 opens
    // the P·V statements); MFMA results read by anything but the next MFMA of the chain need 12 wait states (mfma_settle / instruction distance).
#ifdef NVR_BF16
#define F2_MFMA "v_mfma_f32_32x32x16_bf16"
#else
#define F2_MFMA "v_mfma_f32_32x32x16_f16"
#endif
    half8_t kf[8], vf[2][4];                                          // K fragments of one key half; V fragments of one key half ([k-step][d tile])
    float16_t S[NQ];                                                  // ONE score state per query tile
    half2_t pw[NQ][2][4];                                             // 16-bit P fragments per query tile ([k-step][dword])
    auto mfma_settle = [&]() { asm volatile("s_nop 11" ::: "memory"); };
    // (sched_barrier(0) in front of every MFMA: the vector work written behind the previous one stays there)
    auto qk_first = [&](int j) { __builtin_amdgcn_sched_barrier(0); asm volatile(F2_MFMA " %0, %1, %2, 0" : "=&v"(S[j]) : "v"(kf[0]), "a"(qf[j][0])); };
    auto qk_more = [&](int j, int ks) { __builtin_amdgcn_sched_barrier(0); asm volatile(F2_MFMA " %0, %1, %2, %0" : "+v"(S[j]) : "v"(kf[ks]), "a"(qf[j][ks])); };
    auto pv_one = [&](int j, int s2, int dt) {
        half8_t pfrag;
#pragma unroll
        for (int w = 0; w < 4; ++w) { pfrag[2 * w] = pw[j][s2][w][0]; pfrag[2 * w + 1] = pw[j][s2][w][1]; }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 1\n\t" F2_MFMA " %0, %1, %2, %0" : "+a"(o[j][dt]) : "v"(vf[s2][dt]), "v"(pfrag));
    };
    auto read_k = [&](int ks, int slot, int mt) {                     // K fragment ks of key half (slot, mt)
        const unsigned kl = kbase + slot * STAGE + mt * 8192;
        kf[ks] = *lds_ptr<const half8_t>(kl ^ (unsigned)(ks << 5));
    };
    auto read_v = [&](int s2, int dt, int slot, int mt) {             // Vᵀ fragment (k-step s2, d tile dt) of key half (slot, mt): two transposing reads
        const unsigned a = (vbase + slot * STAGE + (32 * mt + 16 * s2) * 256) ^ (unsigned)(dt << 6);
        const half4_t a0 = lds_read_tr16(a), a1 = lds_read_tr16(a + 8 * 256);
#pragma unroll
        for (int e = 0; e < 4; ++e) { vf[s2][dt][e] = a0[e]; vf[s2][dt][4 + e] = a1[e]; }
    };
    // p = 2^((s - m) c2) of accumulator registers e, e+1 of tile j (k-step e >> 3), their row sums, packed into the P fragment
    auto expo2 = [&](int j, int e) {
        const float p0 = __builtin_amdgcn_exp2f(fmaf(S[j][e], c2, mneg[j])), p1 = __builtin_amdgcn_exp2f(fmaf(S[j][e + 1], c2, mneg[j]));
        lsum[j][e & 3] += p0; lsum[j][(e + 1) & 3] += p1;
        pw[j][e >> 3][(e & 7) >> 1] = (half2_t){(half_t)p0, (half_t)p1};
    };
    // One wave per SIMD has no partner to fill its matrix pipe while it does the softmax, so its TWO query tiles take turns: they run half a
    // 32-key half apart, each with ONE score state, and every group of eight MFMAs of one tile carries half of the other tile's softmax:
    //   g0: S1 = QKᵀ(q1, u)        | softmax(q0, u) second part: exp of registers 8..15                  | reads: V(u), 16 transposing reads
    //   g1: O0 += P·V(q0, u)       | softmax(q1, u) first part: mask, row max (rare rescale), exp 0..7    | reads: K(u+1), 8 b128
    //   g2: S0 = QKᵀ(q0, u+1)      | softmax(q1, u) second part                                           | (+ the K half of a K/V request)
    //   g3: O1 += P·V(q1, u)       | softmax(q0, u+1) first part                                          | (+ the V half)
    // In g1 / g3 the scores were finished by the group before: the first two gaps carry the LDS reads / the request, the row max follows.
    auto half_step = [&](bool first, auto has_next_c, auto dma_c, int slot, int mt, int slot_n, int mtn, int k0, int k0n) {
        constexpr bool HAS_NEXT = decltype(has_next_c)::value;
        constexpr int DMA = decltype(dma_c)::value;
        // g0
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            if (i == 0) qk_first(1); else qk_more(1, i);
            read_v(i >> 2, i & 3, slot, mt);
            expo2(0, 8 + i);                                          // (registers 8+i, 9+i: one pair per two gaps)
            qk_more(1, i + 1);
            read_v((i + 1) >> 2, (i + 1) & 3, slot, mt);
        }
        // g1
        pv_one(0, 0, 0);
        if constexpr (HAS_NEXT) { read_k(0, slot_n, mtn); read_k(1, slot_n, mtn); read_k(2, slot_n, mtn); read_k(3, slot_n, mtn); }
        pv_one(0, 0, 1);
        if constexpr (HAS_NEXT) { read_k(4, slot_n, mtn); read_k(5, slot_n, mtn); read_k(6, slot_n, mtn); read_k(7, slot_n, mtn); }
        pv_one(0, 0, 2);
        mask(S[1], 1, k0);
        stats(first, S[1], 1);
        pv_one(0, 0, 3);
        expo2(1, 0);
        pv_one(0, 1, 0);
        expo2(1, 2);
        pv_one(0, 1, 1);
        expo2(1, 4);
        pv_one(0, 1, 2);
        expo2(1, 6);
        pv_one(0, 1, 3);
        // g2
        if constexpr (HAS_NEXT) {
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                if (i == 0) qk_first(0); else qk_more(0, i);
                expo2(1, 8 + i);
                qk_more(0, i + 1);
                if (DMA == 1 && i == 2) request_k();
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; i += 2) expo2(1, 8 + i);
            if constexpr (DMA == 1) request_k();
        }
        // g3
        pv_one(1, 0, 0);
        if constexpr (DMA == 2) request_v();
        pv_one(1, 0, 1);
        pv_one(1, 0, 2);
        if constexpr (HAS_NEXT) { mask(S[0], 0, k0n); stats(false, S[0], 0); }
        pv_one(1, 0, 3);
        if constexpr (HAS_NEXT) expo2(0, 0);
        pv_one(1, 1, 0);
        if constexpr (HAS_NEXT) expo2(0, 2);
        pv_one(1, 1, 1);
        if constexpr (HAS_NEXT) expo2(0, 4);
        pv_one(1, 1, 2);
        if constexpr (HAS_NEXT) expo2(0, 6);
        pv_one(1, 1, 3);
    };
    auto slot_of = [&](int gs) { return __builtin_amdgcn_readfirstlane(gs % NBUF); };

    // the first three K/V tiles and the first Q rows have landed for every wave
    F2_Q_WAIT(0);                                                     // (all but the N youngest vector-memory operations of this wave are done)
    __builtin_amdgcn_s_barrier();

#ifdef NVR_F2_STAMPS
    unsigned long long acc_bound = 0, acc_h0 = 0, acc_h1 = 0, acc_wait = 0, acc_last = 0, acc_epi = 0, n_steps = 0;
#endif
    int gs = 0;                                                       // global (virtual) step of the compute cursor
    for (int it = it0; it < it1; ++it) {
        F2_STAMP(tb0);
        const int kv_end = tile.pos0 + tile.nq, nsteps = (kv_end + KT - 1) / KT;
        // Tile boundary.  The K/V tile of the first step landed long ago (the wait of the previous tile's last loop step, or the one above); the Q
        // rows were requested behind the previous tile's last matrix work, in front of its 16 output stores: vmcnt(16) leaves exactly those in flight.
        if (it != it0) F2_Q_WAIT(16);
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
#pragma unroll
            for (int i = 0; i < 4; ++i) lsum[j][i] = 0.f;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[j][dt][e] = 0.f;
        }
        {   // query tile 0 runs ahead: its scores of the first half and the first part of their softmax
            const int sl = slot_of(gs);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) read_k(ks, sl, 0);
            qk_first(0);
#pragma unroll
            for (int ks = 1; ks < 8; ++ks) qk_more(0, ks);
            mfma_settle();
            mask(S[0], 0, 0);
            stats(true, S[0], 0);
#pragma unroll
            for (int e = 0; e < 8; e += 2) expo2(0, e);
        }
        F2_STAMP(tb1);
        for (int t = 0; t + 1 < nsteps; ++t, ++gs) {
            F2_STAMP(ta);
            const int sl = slot_of(gs), sn = slot_of(gs + 1);
            half_step(t == 0, std::true_type{}, std::integral_constant<int, 1>{}, sl, 0, sl, 1, t * KT, t * KT + 32);
            F2_STAMP(tc);
            half_step(false, std::true_type{}, std::integral_constant<int, 2>{}, sl, 1, sn, 0, t * KT + 32, (t + 1) * KT);
            F2_STAMP(td);
            // end of step: virtual step gs + 2 (K of the step after the next) has landed for this wave; the request sent during this step may fly
            if (rq_gs - 1 > gs + 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PIECES) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // (this wave's LDS reads have executed: the next request into the slot they read follows the barrier)
            __builtin_amdgcn_s_barrier();
            F2_STAMP(te);
#ifdef NVR_F2_STAMPS
            acc_h0 += tc - ta; acc_h1 += td - tc; acc_wait += te - td; ++n_steps;
#endif
        }
        F2_STAMP(tl0);
        {   // last step of the tile: no scores behind its second half
            const int sl = slot_of(gs);
            half_step(nsteps == 1, std::true_type{}, std::integral_constant<int, 1>{}, sl, 0, sl, 1, (nsteps - 1) * KT, (nsteps - 1) * KT + 32);
            half_step(false, std::false_type{}, std::integral_constant<int, 2>{}, sl, 1, 0, 0, (nsteps - 1) * KT + 32, 0);
            ++gs;
        }
        mfma_settle();                                                // (the output accumulators are read by ordinary code from here on)
        F2_STAMP(tl1);
        // the next tile's Q rows (requested before this tile's output stores: see the boundary wait); Q is dead since the last QKᵀ
        int done_row[NQ];
#pragma unroll
        for (int j = 0; j < NQ; ++j) done_row[j] = qrow[j];
        if (it + 1 < it1) {
            tile = p.tiles[items[it + 1]];
            lane_of_tile(tile);
            load_q();
        }
        // Every lane stores (rows past the tile's last query are copies of that query: identical bytes to the same address), so that a wave
        // issues exactly 16 stores: the boundary wait counts on it.  A query's row is split over lanes l and l ^ 32 (d 8b+4h .. +3 each): one
        // v_permlane32_swap per dword of two neighbouring groups leaves 16 contiguous bytes in each lane (guide T21).
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const float ls = xor32_partner_sum((lsum[j][0] + lsum[j][1]) + (lsum[j][2] + lsum[j][3]));   // (the halves of a query swap their sums)
            const float inv = ls > 0.f ? 1.0f / ls : 0.f;
            char *orow = reinterpret_cast<char *>(p.out + ((int64_t)done_row[j] * p.H + head) * D) + 16 * h;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int b = 0; b < 4; b += 2) {                      // registers 4b..4b+3 = d 32 dt + 8 b + 4 h + 0..3
                    union { half4_t v; unsigned u[2]; } ga, gb;
                    ga.v = (half4_t){(half_t)(o[j][dt][4 * b] * inv), (half_t)(o[j][dt][4 * b + 1] * inv), (half_t)(o[j][dt][4 * b + 2] * inv), (half_t)(o[j][dt][4 * b + 3] * inv)};
                    gb.v = (half4_t){(half_t)(o[j][dt][4 * b + 4] * inv), (half_t)(o[j][dt][4 * b + 5] * inv), (half_t)(o[j][dt][4 * b + 6] * inv), (half_t)(o[j][dt][4 * b + 7] * inv)};
                    const auto r0 = __builtin_amdgcn_permlane32_swap(ga.u[0], gb.u[0], false, false);
                    const auto r1 = __builtin_amdgcn_permlane32_swap(ga.u[1], gb.u[1], false, false);
                    // lanes 0..31: d 8b .. 8b+7 = [own group b | partner's group b]; lanes 32..63: d 8b+8 .. 8b+15 = [partner's group b+1 | own group b+1]
                    uint4 w; w.x = r0[0]; w.y = r1[0]; w.z = r0[1]; w.w = r1[1];
                    *reinterpret_cast<uint4 *>(orow + (32 * dt + 8 * b) * 2) = w;
                }
        }
        F2_STAMP(tl2);
#ifdef NVR_F2_STAMPS
        acc_bound += tb1 - tb0; acc_last += tl1 - tl0; acc_epi += tl2 - tl1;
#endif
        // the slot of this tile's last step is requested again by the next tile's first step: its reads (above) are ordered before that by a barrier
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
#ifdef NVR_F2_STAMPS
    F2_STAMP(ts9);
    if (threadIdx.x == 0 && blockIdx.x < 1024) {
        unsigned long long *d = f2_stamp_buf + (size_t)blockIdx.x * 12;
        d[0] = acc_bound; d[1] = 0; d[2] = acc_h0; d[3] = acc_h1; d[4] = acc_wait; d[5] = acc_last; d[6] = acc_epi; d[7] = n_steps; d[8] = it1 - it0; d[9] = ts9 - ts0;
    }
#endif
}

bool flash_prefill_ok(int D, int H, int KVH) {
    if (KVH <= 0 || H % KVH) return false;
    const int G = H / KVH;
    return (D == 64 || D == 128) && (G == 1 || G == 2 || G == 4);
}
#ifndef NVR_F2_ENABLE
#define NVR_F2_ENABLE 1
#endif
static bool flash2_shape(int D) { return NVR_F2_ENABLE && D == 128; }
// query positions of one tile of the prefill kernels
int flash_tile_positions(int H, int KVH, int D) { return (flash2_shape(D) ? F2_ROWS : 32 * FLASH_WAVES) / (H / KVH); }
// sequences per workgroup of the shared-prefix pass (flash_prefill_kernel<.., SHARED>)
int flash_shared_rows(int H, int KVH) { return 32 * FLASH_WAVES / (H / KVH); }

// Per-workgroup tile lists of the persistent kernel.  The launch has nlanes x KVH workgroups, workgroup w = (lane w / KVH, kv head w % KVH):
// with KVH = 8 a kv head's workgroups share one XCD (workgroups are dealt round-robin to the 8 XCDs: speed only), so the tiles of one
// sequence, which read the same K/V rows, meet in one L2.  Tiles are dealt IN LIST ORDER (the caller keeps sequences together) to the lane
// with the least work so far — cost = 64-key steps + 2 for the tile boundary — so that every lane ends at about the same time and a lane's
// tiles follow the list's time order.  out: [nlanes + 1 starts | ntiles tile indices], flash_lanes_ints(ntiles) ints; 0 when D takes the
// non-persistent kernel.
size_t flash_lanes_ints(int ntiles) { return (size_t)ntiles + 258; }
int flash_lanes(const FlashTile *tiles, int ntiles, int KVH, int D, int ncu, int32_t *out) {
    if (!flash2_shape(D) || ntiles <= 0) return 0;
    int nlanes = ncu / (KVH > 0 ? KVH : 1);
    if (nlanes > 256) nlanes = 256;
    if (nlanes > ntiles) nlanes = ntiles;
    if (nlanes < 1) nlanes = 1;
    std::vector<int64_t> load(nlanes, 0);
    std::vector<int32_t> owner(ntiles), count(nlanes, 0);
    // (a binary heap keyed by (load, lane): ntiles x log nlanes)
    std::vector<std::pair<int64_t, int>> heap(nlanes);
    for (int j = 0; j < nlanes; ++j) heap[j] = {0, j};
    auto cmp = [](const std::pair<int64_t, int> &a, const std::pair<int64_t, int> &b) { return a > b; };
    std::make_heap(heap.begin(), heap.end(), cmp);
    for (int i = 0; i < ntiles; ++i) {
        std::pop_heap(heap.begin(), heap.end(), cmp);
        auto &top = heap.back();
        owner[i] = top.second; ++count[top.second];
        top.first += (tiles[i].pos0 + tiles[i].nq + NVR_FLASH_KT - 1) / NVR_FLASH_KT + 2;
        std::push_heap(heap.begin(), heap.end(), cmp);
    }
    out[0] = 0;
    for (int j = 0; j < nlanes; ++j) out[j + 1] = out[j] + count[j];
    std::vector<int32_t> fill(out, out + nlanes);
    for (int i = 0; i < ntiles; ++i) out[nlanes + 1 + fill[owner[i]]++] = i;
    return nlanes;
}

template <int G, int MODE>
static int flash2_launch(const FlashParams &p, dim3 grid, hipStream_t s) {
    constexpr int NBUF = 4;
    constexpr int LDS = NBUF * 2 * NVR_FLASH_KT * 128 * 2 + 256;
    static bool ready = false;                                        // > 64 KiB of dynamic LDS: opt-in once per kernel
    if (!ready) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&flash2_kernel<G, MODE, NBUF>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return nvr::fail(NVR_ERR_HIP, "flash_prefill: hipFuncSetAttribute: %s", hipGetErrorString(e));
        ready = true;
    }
    flash2_kernel<G, MODE, NBUF><<<grid, dim3(256), LDS, s>>>(p);
#ifdef NVR_F2_STAMPS
    {
        static int calls = 0;
        hipStreamSynchronize(s);
        if (++calls == 8) {
            const size_t n = grid.x < 1024 ? grid.x : 1024;
            std::vector<unsigned long long> hbuf(n * 12);
            hipMemcpyFromSymbol(hbuf.data(), HIP_SYMBOL(f2_stamp_buf), n * 12 * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
            double sum[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, life_max = 0;
            for (size_t i = 0; i < n; ++i) { for (int j = 0; j < 10; ++j) sum[j] += (double)hbuf[i * 12 + j]; life_max = std::max(life_max, (double)hbuf[i * 12 + 9]); }
            std::fprintf(stderr, "[f2 stamps] %zu workgroups, %.1f tiles each, %.1f loop steps each | per tile: boundary %.0f, last step %.0f, epilogue %.0f | per loop step (wave 0): half0 %.0f half1 %.0f wait+barrier %.0f | lifetime mean %.0f max %.0f cycles\n",
                         n, sum[8] / n, sum[7] / n, sum[0] / sum[8], sum[5] / sum[8], sum[6] / sum[8], sum[2] / sum[7], sum[3] / sum[7], sum[4] / sum[7], sum[9] / n, life_max);
        }
    }
#endif
    return 0;
}

int flash_prefill(const FlashArgs &a, bool paged, hipStream_t s) {
    if (a.ntiles == 0) return 0;
    if (!flash_prefill_ok(a.D, a.H, a.KVH)) return nvr::fail(NVR_ERR_UNSUPPORTED, "flash_prefill: D=%d H=%d KVH=%d", a.D, a.H, a.KVH);
    FlashParams p{};
    p.q = (const half_t *)a.q; p.ldq = a.ldq; p.k = (const half_t *)a.k; p.v = (const half_t *)a.v; p.ldkv = a.ldkv;
    p.block_tables = a.block_tables; p.max_blocks = a.max_blocks; p.block_size = a.block_size;
    p.bs_shift = (a.block_size > 0 && (a.block_size & (a.block_size - 1)) == 0) ? __builtin_ctz(a.block_size) : -1;
    p.tiles = a.tiles; p.H = a.H; p.KVH = a.KVH; p.scale = a.scale; p.out = (half_t *)a.out;
    const int G = a.H / a.KVH;
    const bool ub = paged && p.bs_shift >= 0 && a.block_size % NVR_FLASH_KT == 0;
    if (flash2_shape(a.D)) {
        if (!a.lanes || a.nlanes <= 0) return nvr::fail(NVR_ERR_INVALID_ARG, "flash_prefill: head_dim 128 takes per-workgroup tile lists (flash_lanes)");
        p.lanes = a.lanes; p.nlanes = a.nlanes;
        dim3 grid((unsigned)((int64_t)a.nlanes * a.KVH));
        int rc = 0;
#define NVR_FLASH2(GG)                                                              \
        if (G == GG) rc = ub ? flash2_launch<GG, 2>(p, grid, s) : paged ? flash2_launch<GG, 1>(p, grid, s) : flash2_launch<GG, 0>(p, grid, s);
        NVR_FLASH2(1) NVR_FLASH2(2) NVR_FLASH2(4)
#undef NVR_FLASH2
        if (rc) return rc;
    } else {
    dim3 grid((unsigned)((int64_t)a.ntiles * a.KVH)), block(64 * FLASH_WAVES);
#define NVR_FLASH(DD, GG)                                                                             \
    if (a.D == DD && G == GG) {                                                                       \
        if (ub) flash_prefill_kernel<DD, GG, true, true><<<grid, block, 0, s>>>(p);                   \
        else if (paged) flash_prefill_kernel<DD, GG, true><<<grid, block, 0, s>>>(p);                 \
        else flash_prefill_kernel<DD, GG, false><<<grid, block, 0, s>>>(p);                           \
    }
#if NVR_F2_ENABLE
    NVR_FLASH(64, 1) NVR_FLASH(64, 2) NVR_FLASH(64, 4)
#else
    NVR_FLASH(128, 1) NVR_FLASH(128, 2) NVR_FLASH(128, 4) NVR_FLASH(64, 1) NVR_FLASH(64, 2) NVR_FLASH(64, 4)
#endif
#undef NVR_FLASH
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return nvr::fail(NVR_ERR_HIP, "flash_prefill launch failed: %s", hipGetErrorString(e));
    return 0;
}

// Partitions 0..sparts-1 (part_len tokens each, a multiple of the 64-key step) of a decode step whose sequences all share their
// first sparts*part_len cached tokens (block-table row 0 names the blocks): part_o [nq, H, num_parts, D] / part_ml [nq, H,
// num_parts, 2] slots (row, head, partition).
int flash_shared_prefix(const half_bits *q, int64_t ldq, const half_bits *k_cache, const half_bits *v_cache, const int32_t *block_tables,
                        int32_t max_blocks, int32_t block_size, int32_t nq, int32_t H, int32_t KVH, int32_t D, float scale,
                        int32_t part_len, int32_t sparts, int32_t num_parts, float *part_o, float *part_ml, hipStream_t s,
                        const int32_t *rows, const int32_t *count) {
    if (nq == 0) return 0;
    const bool pow2 = block_size > 0 && (block_size & (block_size - 1)) == 0;
    const int32_t shared_len = part_len;
    if (!flash_prefill_ok(D, H, KVH) || !pow2 || block_size % NVR_FLASH_KT || part_len <= 0 || part_len % NVR_FLASH_KT || sparts < 1 || sparts > num_parts)
        return nvr::fail(NVR_ERR_UNSUPPORTED, "flash_shared_prefix: D=%d H=%d KVH=%d block_size=%d part_len=%d x %d", D, H, KVH, block_size, part_len, sparts);
    FlashParams p{};
    p.q = (const half_t *)q; p.ldq = ldq; p.k = (const half_t *)k_cache; p.v = (const half_t *)v_cache;
    p.block_tables = block_tables; p.max_blocks = max_blocks; p.block_size = block_size; p.bs_shift = __builtin_ctz(block_size);
    p.H = H; p.KVH = KVH; p.scale = scale; p.nq_total = nq; p.shared_len = shared_len; p.num_parts = num_parts;
    p.part_o = part_o; p.part_ml = part_ml; p.srows = rows; p.scount = count;
    const int G = H / KVH, qb = flash_shared_rows(H, KVH);
    dim3 grid((unsigned)((int64_t)((nq + qb - 1) / qb) * KVH), (unsigned)sparts), block(64 * FLASH_WAVES);
#define NVR_FLASH_S(DD, GG) if (D == DD && G == GG) flash_prefill_kernel<DD, GG, true, true, true><<<grid, block, 0, s>>>(p);
    NVR_FLASH_S(128, 1) NVR_FLASH_S(128, 2) NVR_FLASH_S(128, 4) NVR_FLASH_S(64, 1) NVR_FLASH_S(64, 2) NVR_FLASH_S(64, 4)
#undef NVR_FLASH_S
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return nvr::fail(NVR_ERR_HIP, "flash_shared_prefix launch failed: %s", hipGetErrorString(e));
    return 0;
}

}}  // namespace nvr::k / nvr::kb (NVR_DT_NS)
