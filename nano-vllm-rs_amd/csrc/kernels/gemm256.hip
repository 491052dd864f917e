// gemm256.hip — 256x256x64 eight-wave MFMA GEMM for the prefill regime (T >= 256 tokens): y[T,N] = x[T,K]·W[N,K]^T,
// with the three epilogues of the decode kernels (plain fp16, gate_up -> SiluAndMul, qkv -> RoPE + KV store).
// reference call sites as in linear.hip (linear.rs:354-356,228-239,437-439; activation.rs:46-63;
// rotary_embedding.rs:23-48; attention.rs:150-174).  Bound: MFMA (880.8 MFLOP per prefill token).
//
// Structure (after cdna_hip_programming.md "The 256² 8-phase template"; the half-tile / phase assignment is this file's):
//   * workgroup = 8 waves (2 along n x 4 along m) = 256 W rows (A operand) x 256 tokens (B operand), BK = 64.
//     Each operand tile is two HALF-TILES of 128 rows (16 KiB: [row][8 chunks of 16 B], chunk' = chunk ^ (row & 7),
//     conflict-free for the 4 x 16 lane groups of ds_read_b128).  LDS = 2 K-tile buffers x {A0, A1, B0, B1} = 128 KiB.
//   * a wave owns 64 rows of EACH A half and 32 rows of EACH B half, so its 128 x 64 output splits into four
//     quadrants (A half, B half) of 16 MFMAs (v_mfma_f32_16x16x32_f16, 4 n-tiles x 2 m-tiles x 2 k-steps) — one
//     quadrant per PHASE, and each phase needs only the half-tiles named below.
//   * a K-tile is four phases; every phase = R { ds_read the fragments it is missing; global_load_lds ONE half-tile of the NEXT
//     K-tile (2 x 16 B per thread) into the other buffer; s_waitcnt vmcnt(4); s_barrier } + M { 16 MFMAs; s_barrier }, the two wave
//     groups one barrier interval apart (reads and LDS-DMA issue of one overlap MFMAs of the other; r03: the requests moved from
//     M to R, where the wave only waits for the other group's MFMAs: -2..3 % per K-tile):
//         phase 0: quadrant (A0,B0)  reads A0 (8 fragments) + B0 (4)   stages A0'
//         phase 1: quadrant (A0,B1)  reads B1 (4)                      stages B0'
//         phase 2: quadrant (A1,B1)  reads A1 (8)                      stages B1'
//         phase 3: quadrant (A1,B0)  reads B0 again (4)                stages A1'
//     vmcnt(4) leaves the two youngest half-tiles in flight; every half-tile has two phases to land before the
//     barrier in front of its first reader (RAW: own loads counted, then the barrier), and it is written into the
//     buffer whose last reader finished a whole K-tile earlier (WAR).  The loads never drain to 0 in the loop.
//   * row maps put epilogue partners in ONE lane: SiLU — A0 = gate rows, A1 = up rows of the same 128 columns; RoPE —
//     A0 = first halves, A1 = second halves of the same heads.
//   * epilogue: results are staged in LDS (the operand buffers are dead) per 128-token half and written as 16-byte
//     pieces of contiguous rows.
#include <cstdlib>
#include "kernels.h"
#include "device_utils.h"
#include "../common.h"

namespace nvr { namespace NVR_DT_NS {

enum { GEPI_F16 = 0, GEPI_RESID = 1, GEPI_SILU = 2, GEPI_ROPE = 3, GEPI_LMHEAD = 4 };   // RESID: y is the residual stream: y <- fp16(y + fp16(x·Wᵀ)) (qwen3.rs:382,389)
// LMHEAD (r06): the LM head over >= 128 rows (ParallelLMHead::compute_logits, embed_head.rs:292-306, + the greedy branch of Sampler::forward,
// sampler.rs:109-112): W = the [V, hidden] head, nothing is written as fp16 — every tile leaves (max, lowest vocabulary index) of its 256
// columns per token row (pval / pidx [column tile][T], merged by argmax_partials) and, on demand, the f32 logits straight from the accumulators;
// the vocabulary need not be a multiple of 256 (the last column tile clamps its row loads and masks its columns)

struct G256Epi {
    const int64_t *pos; const int32_t *slots; const float *cos_t, *sin_t;
    half_t *kc, *vc;
    int32_t H, KVH, D;
    int32_t kv_cache_only;             // GEPI_ROPE: k and v rows go to the caches only (the attention that follows reads them there)
    float *logits; float *pval; int32_t *pidx;   // GEPI_LMHEAD: f32 logits [T][N] (nullable), arg-max partials [tiles_x][T]
};

constexpr int G_BK = 64, G_HT = 128, G_HALF = G_HT * G_BK * 2;          // 16 KiB per half-tile
constexpr int G_BUF = 4 * G_HALF;                                         // A0 A1 B0 B1

// W row behind local row rho = r0 + 64 i (r0 = 0..63, i = 0, 1) of A half hA of workgroup column bx, split into the part that depends
// on the thread (r0), the part that is uniform in the workgroup (hA, i) and the tile's column term (bx x 256 rows; SiLU: bx x 128):
//   plain / residual: bx*256 + hA*128 + rho          SiLU (N == I): hA*N + bx*128 + rho
//   RoPE: head = bx*(256/D) + rho/hd2, row = head*D + hA*hd2 + rho%hd2  (hd2 = D/2 divides 64: rho/hd2 = r0/hd2 + i*(64/hd2))
template <int EPI, int RD>
__device__ __forceinline__ int g_w_thr(int r0) {
    if (EPI == GEPI_ROPE) { constexpr int hd2 = RD / 2 > 0 ? RD / 2 : 1; return (r0 / hd2) * RD + r0 % hd2; }
    return r0;
}
template <int EPI, int RD>
__device__ __forceinline__ int g_w_uni(int hA, int i, int N) {
    if (EPI == GEPI_SILU) return hA * N + i * 64;
    if (EPI == GEPI_ROPE) return i * 128 + hA * (RD / 2);
    return hA * 128 + i * 64;
}

// RD: head_dim of the RoPE variant as a compile-time constant (64 or 128; 0 for the other epilogues): its row map and pair geometry
// fold to shifts, and the scalar registers they needed in the K loop are what pushed that variant into SGPR -> VGPR spills
template <int EPI, int RD = 0>
__global__ __launch_bounds__(512) void gemm256_kernel(const half_t *__restrict__ x, int64_t ldx, const half_t *__restrict__ W,
                                                      int T, int K, int N, int NW, half_t *__restrict__ y, G256Epi epi,
                                                      int tiles_x, int tiles_y, int CG) {
    extern __shared__ __attribute__((aligned(16))) char smem[];           // 2 x G_BUF
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 15, q = lane >> 4;
    const int wn = wave >> 2, wm = wave & 3;
    const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);

    // Persistent workgroup.  The K-tile stream runs ACROSS its tiles: K-tile g of the stream lives in buffer g & 1, and the
    // last K-tile of a tile already stages K-tile 0 of the next tile, so those loads fly during the epilogue.
    // Tile order is XCD-aware (workgroup w runs on XCD w % 8, each XCD has its own 4 MiB L2): the 8 XCDs form CG column
    // groups x 8/CG row groups; an XCD owns the columns of its group (its W slice stays L2-resident) and every
    // (8/CG)-th row block, and its workgroups walk that set column-fastest, so the workgroups that share an x row block
    // run on ONE XCD at the same time: x is fetched from HBM CG times instead of once per XCD that happens to see it.
    const int tiles_total = tiles_x * tiles_y;
    int xcd_step = 0, u = 0, nbx = 0, bx0 = 0, rg = 0, RG = 1, n_u = 0;
    if (CG > 0) {
        const int c = blockIdx.x & 7;
        RG = 8 / CG; nbx = tiles_x / CG; bx0 = (c % CG) * nbx; rg = c / CG;
        n_u = nbx * ((tiles_y - rg + RG - 1) / RG);                       // tiles of this XCD
        u = blockIdx.x >> 3; xcd_step = gridDim.x >> 3;
    }
    // LM head: W (the [V, hidden] head: 311 MB on Qwen3-0.6B) is the stream and x (T rows) sits in every L2, so the tiles_y row blocks of ONE
    // column tile run at the same time on the SAME XCD (workgroups w, w + 8, ...: index uu -> XCD uu & 7, slot uu >> 3; tiles_y consecutive
    // slots share a column tile): its 512 KiB of W rows are fetched from HBM once and found in that XCD's L2 by the others.  The index space is
    // padded to whole groups of 8 column tiles; the holes (only in the last group) end a workgroup's walk.
    auto lm_bx = [&](int uu) { return ((uu >> 3) / tiles_y) * 8 + (uu & 7); };
    auto tile_of = [&](int uu) {
        if (EPI == GEPI_LMHEAD) return ((uu >> 3) % tiles_y) * tiles_x + lm_bx(uu);
        return CG > 0 ? (rg + RG * (uu / nbx)) * tiles_x + bx0 + uu % nbx : uu;
    };
    auto have = [&](int uu) {
        if (EPI == GEPI_LMHEAD) return uu < ((tiles_x + 7) / 8) * 8 * tiles_y && lm_bx(uu) < tiles_x;
        return CG > 0 ? uu < n_u : uu < tiles_total;
    };
    if (CG <= 0) { u = blockIdx.x; xcd_step = gridDim.x; }
    if (!have(u)) return;
    // staging sources: thread copies pieces idx = i*512 + tid (i = 0,1) of each half-tile: row = idx/8 = r0 + 64 i, LDS slot idx%8.
    // Address = tile base (scalar) + uniform row term of (half, i) (scalar) + ONE thread constant per operand: two 64-bit registers
    // instead of eight pointers that had to be rebuilt, with the RoPE row map's divisions, inside the last K-tile of every tile
    // (r03: that rebuild spilled ~80 registers to scratch in the RoPE variant — global-memory round trips in the K loop).
    const int r0 = tid >> 3, cch = ((tid & 7) ^ (r0 & 7)) * 8;            // row inside a 64-row group, swizzled 16-byte chunk (elements)
    const int64_t a_thr = (int64_t)g_w_thr<EPI, RD>(r0) * K + cch, b_thr = (int64_t)r0 * ldx + cch;
    const half_t *wb = W, *xb = x;                                         // of the tile being STAGED (the next one during a tile's last K-tile)
    int rows_left = 0x3fffffff;                                            // T - 1 - m0 when the tile's row block is ragged
    int wrows_left = 0x3fffffff;                                           // LM head: N - 1 - bx * 256 when the tile's vocabulary block is ragged
    auto set_sources = [&](int tile) {
        const int bx = tile % tiles_x, m0 = (tile / tiles_x) * 256;
        wb = W + (int64_t)bx * ((EPI == GEPI_SILU) ? 128 : 256) * K;
        xb = x + (int64_t)m0 * ldx;
        rows_left = m0 + 256 > T ? T - 1 - m0 : 0x3fffffff;
        if (EPI == GEPI_LMHEAD) wrows_left = bx * 256 + 256 > N ? N - 1 - bx * 256 : 0x3fffffff;
    };
    // half-tile ids: 0 = A0, 1 = A1, 2 = B0, 3 = B1; kt = K-tile index inside the (current or next) tile, g = stream index
    auto stage = [&](int ht, int kt, int g) {
        char *dst = smem + (g & 1) * G_BUF + ht * G_HALF;
        const int k0 = kt * G_BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const half_t *src;
            if (ht < 2) {
                src = wb + (int64_t)g_w_uni<EPI, RD>(ht & 1, i, N) * K + a_thr + k0;
                if (EPI == GEPI_LMHEAD && wrows_left != 0x3fffffff && (ht & 1) * 128 + i * 64 + r0 > wrows_left) src = wb + (int64_t)wrows_left * K + cch + k0;   // clamp
            } else {
                const int rowu = (ht & 1) * 128 + i * 64;
                src = xb + (int64_t)rowu * ldx + b_thr + k0;
                if (rows_left != 0x3fffffff && rowu + r0 > rows_left) src = xb + (int64_t)rows_left * ldx + cch + k0;   // ragged last row block: clamp
            }
            __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void *)(dst + (i * 512 + wave * 64) * 16), 16, 0, 0);
        }
    };

    float4_t acc[2][4][2][2];                                             // [A half][n-tile][B half][m-tile]
    half8_t af[4][2], bf[2][2];                                           // [tile][k-step]
    auto read_a = [&](const char *buf, int hA) {
        const char *base = buf + hA * G_HALF;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = wn * 64 + i * 16 + r;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                af[i][ks] = *reinterpret_cast<const half8_t *>(base + (row * 8 + ((ks * 4 + q) ^ (row & 7))) * 16);
        }
    };
    auto read_b = [&](const char *buf, int hB) {
        const char *base = buf + (2 + hB) * G_HALF;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = wm * 32 + j * 16 + r;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                bf[j][ks] = *reinterpret_cast<const half8_t *>(base + (row * 8 + ((ks * 4 + q) ^ (row & 7))) * 16);
        }
    };
    auto mma = [&](float4_t (&c)[4][2][2], int hB) {                      // c = acc[hA]
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    c[i][hB][j] = mfma16(af[i][ks], bf[j][ks], c[i][hB][j]);
    };

    // The two wave groups (waves 0-3 and 4-7: one wave of each per SIMD) run ONE barrier interval apart, and a phase is
    // two intervals — R {ds_read, stage one half-tile, s_waitcnt vmcnt, barrier} and M {16 MFMAs, barrier} — so that on every
    // SIMD the fragment reads and the LDS-DMA issue of one wave (~100 clocks a piece) overlap the MFMAs of the other.  Group 1
    // takes one extra barrier before each tile's K loop, group 0 one after it.  Hazards with the stagger: a half-tile staged in
    // R(p) is first read in R(p+3); every wave counts its own loads at the end of R(p+2) (vmcnt(4): all but the two youngest
    // half-tiles, those of R(p+1) and R(p+2); the epilogue's stores count too, which only makes the wait stricter), i.e. group
    // 0 in interval 2p+4 and group 1 in 2p+5, and the earliest reader (group 0, R(p+3)) runs in interval 2p+6.  A half-tile is
    // restaged no earlier than the interval after its last ds_read by either group has returned (group 1's R(3) reads B0 while
    // group 0's R(0) of the next K-tile stages A0; B0 is restaged two intervals later).
    // RoPE: positions (low dwords of the int64 entries) and cache slots of a tile's 256 token rows travel to LDS behind the K-tile ring
    // by LDS-DMA, 4 bytes per lane (waves 0-3: 64 rows each), into one of two 2 KiB slots by tile parity.  They are requested one tile
    // AHEAD, at the start of the previous tile's epilogue (whose first barrier drains the wave's requests anyway): asked for at the
    // start of their own tile they sat in front of that tile's first staging waits (s_waitcnt vmcnt is in issue order) and stalled
    // every tile's first K-tile by a memory round trip.
    auto request_rows = [&](int tile_, int slot) {
        if (EPI == GEPI_ROPE && grp == 0) {
            int m = (tile_ / tiles_x) * 256 + wave * 64 + lane; if (m > T - 1) m = T - 1;
            char *dst = smem + 2 * G_BUF + slot * 2048 + wave * 256;
            __builtin_amdgcn_global_load_lds(epi.pos + m, (__attribute__((address_space(3))) void *)dst, 4, 0, 0);
            if (epi.slots) __builtin_amdgcn_global_load_lds(epi.slots + m, (__attribute__((address_space(3))) void *)(dst + 1024), 4, 0, 0);
        }
    };
    const int KT = K / G_BK;
    int tile = tile_of(u);
    set_sources(tile);
    request_rows(tile, 0);
    stage(0, 0, 0); stage(2, 0, 0); stage(3, 0, 0); stage(1, 0, 0);       // A0, B0, B1, A1 of the first K-tile
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();

#define G256_R_END(N_)                                                                                                \
    asm volatile("s_waitcnt vmcnt(" #N_ ")" ::: "memory");                                                            \
    __builtin_amdgcn_s_barrier();
#define G256_M(C_, HB_)                                                                                               \
    __builtin_amdgcn_s_setprio(1);                                                                                    \
    mma(C_, HB_);                                                                                                     \
    __builtin_amdgcn_s_setprio(0);                                                                                    \
    __builtin_amdgcn_s_barrier();
#define G256_S(HT_) if (more) stage(HT_, skt, g + 1);
#define G256_W(NOMORE_) if (more) { G256_R_END(4) } else { G256_R_END(NOMORE_) }

    constexpr int OUTC = (EPI == GEPI_SILU) ? 128 : 256;                  // output columns of a tile
    const int64_t ldy = (EPI == GEPI_ROPE) ? (int64_t)(epi.H + 2 * epi.KVH) * RD : (int64_t)N;
    constexpr int hd2 = (EPI == GEPI_ROPE) ? RD / 2 : 1;

    int tile_no = 0;
    for (int g = 0;; ) {

        if (grp == 1) __builtin_amdgcn_s_barrier();                       // stagger
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[a][i][b][j] = (float4_t){0.f, 0.f, 0.f, 0.f};
        const bool have_next = have(u + xcd_step);
        const int next_tile = have_next ? tile_of(u + xcd_step) : 0;
        for (int kt = 0; kt < KT; ++kt, ++g) {
            const char *buf = smem + (g & 1) * G_BUF;
            const bool last = kt == KT - 1;
            const bool more = !last || have_next;
            if (last && more) set_sources(next_tile);                     // from here on the staging feeds the next tile
            const int skt = last ? 0 : kt + 1;
            // phase 0: (A0, B0)
            read_b(buf, 0); read_a(buf, 0);
            G256_S(0)
            G256_W(2)
            G256_M(acc[0], 0)
            // phase 1: (A0, B1)
            read_b(buf, 1);
            G256_S(2)
            G256_W(0)
            G256_M(acc[0], 1)
            // phase 2: (A1, B1)
            read_a(buf, 1);
            G256_S(3)
            G256_W(0)
            G256_M(acc[1], 1)
            // phase 3: (A1, B0)
            read_b(buf, 0);
            G256_S(1)
            G256_W(0)
            G256_M(acc[1], 0)
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();                       // re-align the groups


        // ---- epilogue of `tile`: the buffer of the K-tile just consumed is dead: per 128-token half hB, stage
        // [token][column] there (16-byte chunks XOR-swizzled by the row) and write 16-byte pieces of contiguous rows,
        // while the next tile's first K-tile lands in the other buffer.
        {   // (scope: the epilogue works on shadow copies of the thread coordinates, see etid)
        char *scratch = smem + ((g - 1) & 1) * G_BUF;
        const int bx = tile % tiles_x, m0 = (tile / tiles_x) * 256;
        // The thread coordinates of the epilogue come from a value the compiler cannot see through: otherwise it hoists the epilogue's
        // address arithmetic (LDS offsets of the 16 puts and the read-back pieces, row / column terms of the stores: ~70 registers of
        // loop invariants) out of the persistent tile loop, where they cannot all stay in registers next to the K loop's 176 — they
        // were spilled to scratch at kernel start and reloaded in every epilogue, one global-memory round trip after the other
        // (RoPE variant, r03: 72-89 spilled registers).  Recomputing them per tile is ~100 VALU instructions.
        int etid = tid;
        asm volatile("" : "+v"(etid));
        const int tid = etid, wave = tid >> 6, lane = tid & 63, r = lane & 15, q = lane >> 4, wn = wave >> 2, wm = wave & 3;
        (void)wave; (void)lane;
        if (EPI == GEPI_LMHEAD) {
            // C layout of an MFMA: lane (r, q) holds 4 consecutive vocabulary entries (W rows q*4 .. q*4+3 of the 16-row tile) of token r.
            // Per token row: (max, lowest index) over this lane's 32 entries in ascending column order (strict >: the first maximum stays),
            // then over the 4 lanes of the token (larger value, else lower index), then over the two wave columns through LDS.
            float *sv = reinterpret_cast<float *>(scratch); int *si = reinterpret_cast<int *>(scratch + 1024);
            const int n_base = bx * 256 + wn * 64 + q * 4;
            float bestv[2][2]; int besti[2][2];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int ml = b * 128 + wm * 32 + j * 16 + r, m = m0 + ml;
                    float bv = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int c0 = n_base + a * 128 + i * 16;
                            const float4_t v = acc[a][i][b][j];
                            if (epi.logits && m < T && c0 < N) *reinterpret_cast<float4_t *>(epi.logits + (int64_t)m * N + c0) = v;
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (c0 + e < N && v[e] > bv) { bv = v[e]; bi = c0 + e; }
                        }
#pragma unroll
                    for (int o = 16; o < 64; o <<= 1) {
                        const float v2 = __shfl_xor(bv, o, 64); const int i2 = __shfl_xor(bi, o, 64);
                        if (v2 > bv || (v2 == bv && i2 < bi)) { bv = v2; bi = i2; }
                    }
                    bestv[b][j] = bv; besti[b][j] = bi;
                    if (wn == 1 && q == 0) { sv[ml] = bv; si[ml] = bi; }
                }
            __syncthreads();
            if (wn == 0 && q == 0) {
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int ml = b * 128 + wm * 32 + j * 16 + r, m = m0 + ml;
                        float bv = bestv[b][j]; int bi = besti[b][j];
                        const float v2 = sv[ml]; const int i2 = si[ml];
                        if (v2 > bv || (v2 == bv && i2 < bi)) { bv = v2; bi = i2; }
                        if (m < T) { epi.pval[(int64_t)bx * T + m] = bv; epi.pidx[(int64_t)bx * T + m] = bi; }
                    }
            }
            __syncthreads();                                              // the next tile's second K-tile is staged into this buffer
        }
        auto put = [&](int ml, int col, half4_t h) {                      // 4 consecutive columns (col % 4 == 0) of token row ml
            *reinterpret_cast<half4_t *>(scratch + ml * (OUTC * 2) + ((((col >> 3) ^ (ml & 15)) << 4) | ((col & 4) << 1))) = h;
        };
        // RoPE (r03): the accumulators are staged as plain fp16 like GEPI_F16 and the rotation happens in the STORE phase, on pairs of
        // 16-byte pieces (x1 = 8 columns of a head's first half, x2 = the same columns of its second half) read back from the LDS
        // image: there the accumulators of the half are dead, so the cos / sin values of a thread's four pairs (16 x 16 B) are
        // requested as ONE batch behind the staging barrier.  Rotating at the accumulators needed them one by one in front of each
        // use (each waited for with vmcnt(0) while LDS-DMA was pending) inside 256 registers with spills: 23 k of a tile's 93 k
        // cycles (profiles/r03_prefill_gemm_epilogue.txt).  Same arithmetic on the same fp16-rounded GEMM outputs: bit-identical.
        // The tile's 256 positions and cache slots were brought into LDS by LDS-DMA when the tile started (below the tile loop's
        // head), so the requests for the cos / sin rows leave right after a half's conversion and fly across its staging barrier.
        if (have_next) request_rows(next_tile, (tile_no + 1) & 1);
        const int *lds_pos = reinterpret_cast<const int *>(smem + 2 * G_BUF + (tile_no & 1) * 2048), *lds_slot = lds_pos + 256;
        constexpr int hp = (EPI == GEPI_ROPE) ? RD / 16 : 1;               // 16-byte pieces per half head
        const int pc = tid & 15, head_l = pc / hp, pcc = pc % hp;         // this thread's pair inside a row (16 pairs per 256 columns)
        const int ch1 = head_l * 2 * hp + pcc, ch2 = ch1 + hp;
        const int rhead = (EPI == GEPI_ROPE) ? bx * (256 / (RD > 0 ? RD : 256)) + head_l : 0;
        const bool rot = (EPI == GEPI_ROPE) && rhead < epi.H + epi.KVH;   // q and k heads rotate, v heads pass through
        int rslot[4];
        float4_t cs[4][2], sn[4][2];
        // plain / residual / RoPE epilogues: all 128 accumulator registers become 64 registers of packed fp16 at once (the values the
        // halves stage below), so that the second half's results do not sit in f32 under the first half's store phase: the RoPE
        // store phase holds 64 registers of cos / sin next to them, and a spill here is a global-memory round trip (scratch)
        half4_t pk[2][4][2][2];
        if (EPI != GEPI_SILU && EPI != GEPI_LMHEAD) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const float4_t v = acc[a][i][b][j];
                            pk[a][i][b][j] = (half4_t){(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                        }
        }
#pragma unroll
        for (int hB = 0; hB < (EPI == GEPI_LMHEAD ? 0 : 2); ++hB) {
            // GEPI_RESID: the residual pieces this thread will add to are requested first; their latency runs under the conversion
            // of the accumulators, the LDS staging and the barrier
            constexpr int RPC = (EPI == GEPI_RESID) ? (128 * (OUTC / 8)) / 512 : 1;
            half8_t hres[RPC];
            if (EPI == GEPI_RESID) {
#pragma unroll
                for (int kk = 0; kk < RPC; ++kk) {
                    const int pidx = tid + kk * 512, row = pidx / (OUTC / 8), ch = pidx % (OUTC / 8);
                    const int m = m0 + hB * 128 + row, col = bx * OUTC + ch * 8;
                    hres[kk] = (m < T && col < ldy) ? *reinterpret_cast<const half8_t *>(y + (int64_t)m * ldy + col) : (half8_t)(half_t)0;
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int ml = wm * 32 + j * 16 + r;                      // token row inside the half
                if (EPI == GEPI_F16 || EPI == GEPI_RESID || EPI == GEPI_ROPE) {
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int rho = wn * 64 + i * 16 + q * 4;     // local W row; RoPE: A0 / A1 hold the first / second halves of the heads
                            const int colo = (EPI == GEPI_ROPE) ? (rho / hd2) * RD + a * hd2 + rho % hd2 : a * 128 + rho;
                            put(ml, colo, pk[a][i][hB][j]);
                        }
                } else if (EPI == GEPI_SILU) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        half4_t h;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float gf = (float)to_half_rn(acc[0][i][hB][j][e]), uf = (float)to_half_rn(acc[1][i][hB][j][e]);
                            const float sg = sigmoid_fast(gf);
                            h[e] = to_half_rn(__fmul_rn(__fmul_rn(gf, sg), uf));
                        }
                        put(ml, wn * 64 + i * 16 + q * 4, h);
                    }
                }
            }
            if (EPI == GEPI_ROPE) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int row = hB * 128 + (tid >> 4) + kk * 32;
                    const int pos = lds_pos[row];
                    rslot[kk] = epi.slots ? lds_slot[row] : -1;
                    if (rot) {
#pragma unroll
                        for (int hh = 0; hh < 2; ++hh) {
                            cs[kk][hh] = *reinterpret_cast<const float4_t *>(epi.cos_t + (int64_t)pos * hd2 + pcc * 8 + hh * 4);
                            sn[kk][hh] = *reinterpret_cast<const float4_t *>(epi.sin_t + (int64_t)pos * hd2 + pcc * 8 + hh * 4);
                        }
                    }
                }
            }
            __syncthreads();
            constexpr int CPR = OUTC / 8;                                 // 16-byte pieces per row
            if (EPI == GEPI_RESID) {
                // h <- fp16(h + fp16(acc)): the rounding points of add_rmsnorm's add
#pragma unroll
                for (int kk = 0; kk < (128 * CPR) / 512; ++kk) {
                    const int pidx = tid + kk * 512, row = pidx / CPR, ch = pidx % CPR;
                    const int m = m0 + hB * 128 + row, col = bx * OUTC + ch * 8;
                    if (m >= T || col >= ldy) continue;
                    const half8_t v8 = *reinterpret_cast<const half8_t *>(scratch + row * (OUTC * 2) + ((ch ^ (row & 15)) << 4));
                    half8_t o8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o8[e] = to_half_rn((float)hres[kk][e] + (float)v8[e]);
                    *reinterpret_cast<half8_t *>(y + (int64_t)m * ldy + col) = o8;
                }
                __syncthreads();
                continue;
            }
            if (EPI == GEPI_ROPE) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int row = (tid >> 4) + kk * 32, m = m0 + hB * 128 + row;
                    const half8_t x1 = *reinterpret_cast<const half8_t *>(scratch + row * (OUTC * 2) + ((ch1 ^ (row & 15)) << 4));
                    const half8_t x2 = *reinterpret_cast<const half8_t *>(scratch + row * (OUTC * 2) + ((ch2 ^ (row & 15)) << 4));
                    half8_t o1 = x1, o2 = x2;
                    if (rot) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float a = (float)x1[e], b = (float)x2[e], cc = cs[kk][e >> 2][e & 3], ss = sn[kk][e >> 2][e & 3];
                            o1[e] = to_half_rn(mul_sub_unfused(a, cc, b, ss));
                            o2[e] = to_half_rn(mul_add_unfused(b, cc, a, ss));
                        }
                    }
                    if (m >= T) continue;
                    half_t *yrow = y + (int64_t)m * ldy + bx * OUTC;
                    if (rhead < epi.H || !epi.kv_cache_only) {
                        *reinterpret_cast<half8_t *>(yrow + ch1 * 8) = o1;
                        *reinterpret_cast<half8_t *>(yrow + ch2 * 8) = o2;
                    }
                    if (rhead >= epi.H && rslot[kk] >= 0) {
                        const bool is_k = rhead < epi.H + epi.KVH;
                        const int kvh = is_k ? rhead - epi.H : rhead - epi.H - epi.KVH;
                        half_t *crow = (is_k ? epi.kc : epi.vc) + ((int64_t)rslot[kk] * epi.KVH + kvh) * RD;
                        *reinterpret_cast<half8_t *>(crow + pcc * 8) = o1;
                        *reinterpret_cast<half8_t *>(crow + (hp + pcc) * 8) = o2;
                    }
                }
                __syncthreads();
                continue;
            }
#pragma unroll
            for (int kk = 0; kk < (128 * CPR) / 512; ++kk) {
                const int pidx = tid + kk * 512;
                const int row = pidx / CPR, ch = pidx % CPR;
                const int m = m0 + hB * 128 + row, col = bx * OUTC + ch * 8;
                if (m >= T || col >= ldy) continue;
                const half8_t v8 = *reinterpret_cast<const half8_t *>(scratch + row * (OUTC * 2) + ((ch ^ (row & 15)) << 4));
                *reinterpret_cast<half8_t *>(y + (int64_t)m * ldy + col) = v8;
            }
            __syncthreads();
        }
        }   // epilogue scope
        ++tile_no;
        if (!have_next) break;
        tile = next_tile; u += xcd_step;
    }
#undef G256_R_END
#undef G256_M
#undef G256_S
#undef G256_W
}

static int g256_prepare() {                                               // 128 KiB of dynamic LDS: opt-in once
    static bool done = false;
    if (done) return 0;
    const void *fns[] = {reinterpret_cast<const void *>(&gemm256_kernel<GEPI_F16>), reinterpret_cast<const void *>(&gemm256_kernel<GEPI_SILU>),
                         reinterpret_cast<const void *>(&gemm256_kernel<GEPI_ROPE, 128>), reinterpret_cast<const void *>(&gemm256_kernel<GEPI_ROPE, 64>),
                         reinterpret_cast<const void *>(&gemm256_kernel<GEPI_RESID>), reinterpret_cast<const void *>(&gemm256_kernel<GEPI_LMHEAD>)};
    for (const void *f : fns) {
        hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * G_BUF + 4096);
        if (e != hipSuccess) return nvr::fail(NVR_ERR_HIP, "gemm256: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    done = true;
    return 0;
}
static int g256_grid(int tiles) { return tiles < 256 ? tiles : 256; }   // one persistent workgroup per CU (128 KiB of LDS each)
// column groups CG for the XCD-aware tile order (0 = plain order).  Measured at T = 32768 (scratch/gemm_prefill_bench.py,
// +-5 % run to run): the plain GEMMs and SiLU do not care (CG 0/1/2/4 within noise; CG = 8 loses 8 % at N = 6144), the
// RoPE + KV-store epilogue gains 12 % with CG = 1 (an XCD writes whole output rows and cache rows at a time).
static int g256_cg(int tiles_x, int tiles_y, bool rope) {
    int cg = rope ? 1 : 0;
    if (cg && (tiles_x % cg || tiles_y < 8 / cg || tiles_x * tiles_y < 256)) cg = 0;
    return cg;
}
static int g256_check(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return nvr::fail(NVR_ERR_HIP, "%s launch failed: %s", what, hipGetErrorString(e));
    return 0;
}
static constexpr bool g256_enabled() { return true; }

bool gemm256_ok(int64_t T, int64_t K, int64_t N, int64_t ldx) {
    return g256_enabled() && T >= 256 && K % G_BK == 0 && K >= 2 * G_BK && N % 256 == 0 && ldx % 8 == 0;
}
bool gemm256_silu_ok(int64_t T, int64_t K, int64_t I, int64_t ldx) {
    return g256_enabled() && T >= 256 && K % G_BK == 0 && K >= 2 * G_BK && I % 128 == 0 && ldx % 8 == 0;
}
bool gemm256_rope_ok(int64_t T, int64_t K, int64_t H, int64_t KVH, int64_t D, int64_t ldx) {
    return g256_enabled() && T >= 256 && K % G_BK == 0 && K >= 2 * G_BK && (D == 128 || D == 64) && ((H + 2 * KVH) * D) % 256 == 0 && ldx % 8 == 0;
}

int gemm256(const half_bits *x, int64_t ldx, const half_bits *W, int64_t T, int64_t K, int64_t N, half_bits *y, hipStream_t s) {
    if (!gemm256_ok(T, K, N, ldx)) return nvr::fail(NVR_ERR_UNSUPPORTED, "gemm256: T=%ld K=%ld N=%ld", (long)T, (long)K, (long)N);
    if (int rc = g256_prepare()) return rc;
    const int tx = (int)(N / 256), tt = tx * (int)((T + 255) / 256);
    gemm256_kernel<GEPI_F16><<<dim3((unsigned)g256_grid(tt)), dim3(512), 2 * G_BUF, s>>>((const half_t *)x, ldx, (const half_t *)W, (int)T, (int)K,
                                                                                        (int)N, (int)N, (half_t *)y, G256Epi{}, tx, tt / tx, g256_cg(tx, tt / tx, false));
    return g256_check("gemm256");
}
// h[T,N] <- fp16(h + fp16(x·Wᵀ)): the row-parallel GEMM with the residual add of qwen3.rs:382,389 in its epilogue (prefill steps on one
// rank: the following norm then reads ONE tensor instead of h and the projection)
int gemm256_resid(const half_bits *x, int64_t ldx, const half_bits *W, int64_t T, int64_t K, int64_t N, half_bits *h, hipStream_t s) {
    if (!gemm256_ok(T, K, N, ldx)) return nvr::fail(NVR_ERR_UNSUPPORTED, "gemm256_resid: T=%ld K=%ld N=%ld", (long)T, (long)K, (long)N);
    if (int rc = g256_prepare()) return rc;
    const int tx = (int)(N / 256), tt = tx * (int)((T + 255) / 256);
    gemm256_kernel<GEPI_RESID><<<dim3((unsigned)g256_grid(tt)), dim3(512), 2 * G_BUF, s>>>((const half_t *)x, ldx, (const half_t *)W, (int)T, (int)K,
                                                                                          (int)N, (int)N, (half_t *)h, G256Epi{}, tx, tt / tx, g256_cg(tx, tt / tx, false));
    return g256_check("gemm256_resid");
}
int gemm256_silu_mul(const half_bits *x, int64_t ldx, const half_bits *W, int64_t T, int64_t K, int64_t I, half_bits *out, hipStream_t s) {
    if (!gemm256_silu_ok(T, K, I, ldx)) return nvr::fail(NVR_ERR_UNSUPPORTED, "gemm256_silu_mul: T=%ld K=%ld I=%ld", (long)T, (long)K, (long)I);
    if (int rc = g256_prepare()) return rc;
    const int tx = (int)(I / 128), tt = tx * (int)((T + 255) / 256);
    gemm256_kernel<GEPI_SILU><<<dim3((unsigned)g256_grid(tt)), dim3(512), 2 * G_BUF, s>>>((const half_t *)x, ldx, (const half_t *)W, (int)T, (int)K,
                                                                                         (int)I, (int)(2 * I), (half_t *)out, G256Epi{}, tx, tt / tx, g256_cg(tx, tt / tx, false));
    return g256_check("gemm256_silu_mul");
}
int gemm256_qkv_rope_store(const half_bits *x, int64_t ldx, const half_bits *W, int64_t T, int64_t K, int64_t H, int64_t KVH, int64_t D,
                           const int64_t *positions, const int32_t *slots, const float *cos_t, const float *sin_t, half_bits *qkv,
                           half_bits *k_cache, half_bits *v_cache, hipStream_t s, bool kv_cache_only) {
    if (!gemm256_rope_ok(T, K, H, KVH, D, ldx)) return nvr::fail(NVR_ERR_UNSUPPORTED, "gemm256_qkv_rope_store: T=%ld K=%ld D=%ld", (long)T, (long)K, (long)D);
    if (int rc = g256_prepare()) return rc;
    const int64_t N = (H + 2 * KVH) * D;
    G256Epi e{};
    e.pos = positions; e.slots = slots; e.cos_t = cos_t; e.sin_t = sin_t; e.kc = (half_t *)k_cache; e.vc = (half_t *)v_cache;
    e.H = (int32_t)H; e.KVH = (int32_t)KVH; e.D = (int32_t)D;
    e.kv_cache_only = (kv_cache_only && slots && k_cache && v_cache) ? 1 : 0;
    const int tx = (int)(N / 256), tt = tx * (int)((T + 255) / 256);
    if (D == 128)
        gemm256_kernel<GEPI_ROPE, 128><<<dim3((unsigned)g256_grid(tt)), dim3(512), 2 * G_BUF + 4096, s>>>((const half_t *)x, ldx, (const half_t *)W, (int)T, (int)K,
                                                                                                  (int)N, (int)N, (half_t *)qkv, e, tx, tt / tx, g256_cg(tx, tt / tx, true));
    else
        gemm256_kernel<GEPI_ROPE, 64><<<dim3((unsigned)g256_grid(tt)), dim3(512), 2 * G_BUF + 4096, s>>>((const half_t *)x, ldx, (const half_t *)W, (int)T, (int)K,
                                                                                                 (int)N, (int)N, (half_t *)qkv, e, tx, tt / tx, g256_cg(tx, tt / tx, true));
    return g256_check("gemm256_qkv_rope_store");
}

// LM head over >= G256_LM_MIN_T (192) rows (large decode batches — BASELINE configs[4]: 512 sequences — and many-sequence prefills): f32 logits on demand,
// one (max, lowest index) partial per 256-column tile and row; N any multiple of 16
constexpr int64_t G256_LM_MIN_T = 192;     // (measured: 128x128 tiles 78 us at 128 rows, 136 us at 255; these tiles 102 us up to 256 rows)
bool gemm256_lm_head_ok(int64_t T, int64_t K, int64_t N, int64_t ldx) {
    return g256_enabled() && T >= G256_LM_MIN_T && T <= 65536 && K % G_BK == 0 && K >= 2 * G_BK && N % 16 == 0 && N >= 256 && ldx % 8 == 0 &&
           (N + 255) / 256 <= LM_HEAD_MAX_PARTS && N < (1ll << 31) && ((N + 255) / 256) * ((T + 255) / 256) < (1ll << 30);
}
int gemm256_lm_head(const half_bits *x, int64_t ldx, const half_bits *W, int64_t T, int64_t K, int64_t N, float *logits, float *part_val,
                    int32_t *part_idx, int32_t *nparts, hipStream_t s) {
    if (!gemm256_lm_head_ok(T, K, N, ldx)) return nvr::fail(NVR_ERR_UNSUPPORTED, "gemm256_lm_head: T=%ld K=%ld N=%ld", (long)T, (long)K, (long)N);
    if (int rc = g256_prepare()) return rc;
    G256Epi e{};
    e.logits = logits; e.pval = part_val; e.pidx = part_idx;
    const int tx = (int)((N + 255) / 256), tt = tx * (int)((T + 255) / 256);
    *nparts = tx;
    const int padded = (tx + 7) / 8 * 8 * (tt / tx);                        // index space of the XCD-paired tile order (kernel: lm_bx)
    gemm256_kernel<GEPI_LMHEAD><<<dim3((unsigned)g256_grid(padded)), dim3(512), 2 * G_BUF, s>>>((const half_t *)x, ldx, (const half_t *)W, (int)T, (int)K,
                                                                                           (int)N, (int)N, nullptr, e, tx, tt / tx, 0);
    return g256_check("gemm256_lm_head");
}

}}  // namespace nvr::k / nvr::kb (NVR_DT_NS)
