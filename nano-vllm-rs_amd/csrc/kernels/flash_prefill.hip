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
#include "kernels.h"
#include "device_utils.h"
#include "../common.h"

namespace nvr { namespace NVR_DT_NS {

struct FlashParams {
    const half_t *q; int64_t ldq;
    const half_t *k, *v; int64_t ldkv;
    const int32_t *block_tables; int32_t max_blocks, block_size, bs_shift;
    const FlashTile *tiles;
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

bool flash_prefill_ok(int D, int H, int KVH) {
    if (KVH <= 0 || H % KVH) return false;
    const int G = H / KVH;
    return (D == 64 || D == 128) && (G == 1 || G == 2 || G == 4);
}
int flash_tile_positions(int H, int KVH) { return 32 * FLASH_WAVES / (H / KVH); }

int flash_prefill(const FlashArgs &a, bool paged, hipStream_t s) {
    if (a.ntiles == 0) return 0;
    if (!flash_prefill_ok(a.D, a.H, a.KVH)) return nvr::fail(NVR_ERR_UNSUPPORTED, "flash_prefill: D=%d H=%d KVH=%d", a.D, a.H, a.KVH);
    FlashParams p{};
    p.q = (const half_t *)a.q; p.ldq = a.ldq; p.k = (const half_t *)a.k; p.v = (const half_t *)a.v; p.ldkv = a.ldkv;
    p.block_tables = a.block_tables; p.max_blocks = a.max_blocks; p.block_size = a.block_size;
    p.bs_shift = (a.block_size > 0 && (a.block_size & (a.block_size - 1)) == 0) ? __builtin_ctz(a.block_size) : -1;
    p.tiles = a.tiles; p.H = a.H; p.KVH = a.KVH; p.scale = a.scale; p.out = (half_t *)a.out;
    const int G = a.H / a.KVH;
    dim3 grid((unsigned)((int64_t)a.ntiles * a.KVH)), block(64 * FLASH_WAVES);
    const bool ub = paged && p.bs_shift >= 0 && a.block_size % NVR_FLASH_KT == 0;
#define NVR_FLASH(DD, GG)                                                                             \
    if (a.D == DD && G == GG) {                                                                       \
        if (ub) flash_prefill_kernel<DD, GG, true, true><<<grid, block, 0, s>>>(p);                   \
        else if (paged) flash_prefill_kernel<DD, GG, true><<<grid, block, 0, s>>>(p);                 \
        else flash_prefill_kernel<DD, GG, false><<<grid, block, 0, s>>>(p);                           \
    }
    NVR_FLASH(128, 1) NVR_FLASH(128, 2) NVR_FLASH(128, 4) NVR_FLASH(64, 1) NVR_FLASH(64, 2) NVR_FLASH(64, 4)
#undef NVR_FLASH
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
    const int G = H / KVH, qb = flash_tile_positions(H, KVH);
    dim3 grid((unsigned)((int64_t)((nq + qb - 1) / qb) * KVH), (unsigned)sparts), block(64 * FLASH_WAVES);
#define NVR_FLASH_S(DD, GG) if (D == DD && G == GG) flash_prefill_kernel<DD, GG, true, true, true><<<grid, block, 0, s>>>(p);
    NVR_FLASH_S(128, 1) NVR_FLASH_S(128, 2) NVR_FLASH_S(128, 4) NVR_FLASH_S(64, 1) NVR_FLASH_S(64, 2) NVR_FLASH_S(64, 4)
#undef NVR_FLASH_S
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return nvr::fail(NVR_ERR_HIP, "flash_shared_prefix launch failed: %s", hipGetErrorString(e));
    return 0;
}

}}  // namespace nvr::k / nvr::kb (NVR_DT_NS)
