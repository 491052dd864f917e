// linear_stream.hip — decode GEMMs over LARGE weights (hidden >= 2048-class models: Qwen3-8B shapes): y[T,N] = x[T,K]·W[N,K]^T
// for T <= 32 with the plain fp16, gate_up -> SiluAndMul and qkv -> RoPE + KV-store epilogues of linear.hip.
// reference call sites: QKVParallelLinear::forward src/layers/linear.rs:354-356, RowParallelLinear :228-239,
// MergedColumnParallelLinear :437-439 + SiluAndMul activation.rs:46-63, RoPE rotary_embedding.rs:23-48, store_kv_cache
// attention.rs:150-174.
//
// Why a second kernel: linear_skinny_kernel feeds the MFMA B operand (the activations) with fragment-shaped loads from L2,
// twice the bytes of the weight stream; that is free while a GEMM is one HBM round trip long (Qwen3-0.6B: 4-13 MB of weights,
// 5-7 us) but caps it at 2.2-3.3 TB/s once the weights are 30-200 MB (profiles/r01_gemm_ablation.txt, Qwen3-8B shapes).
// Here, as in lm_head.hip, the activation block lives in LDS (XOR-swizzled 16-byte chunks, conflict-free ds_read_b128), in K
// chunks of KC columns (32 x 2048 fp16 = 128 KiB), and the workgroups are persistent: workgroup w owns the 16-row weight tiles
// w, w + nwg, ... (at most TMAX of them, their accumulators stay in registers across the K chunks); inside a chunk the 8 waves
// split the k-steps of each tile (wave j takes k-steps j, j+8, ...: neighbouring waves read neighbouring 64-byte pieces of the
// same weight rows, U pieces in flight per wave = 32 KiB per CU) and the partial sums meet in LDS once per tile at the end.
// Roofline: HBM, algorithmic bytes 2·N·K.  Rounding points as in linear.hip.
#include <algorithm>
#include <type_traits>
#include <cstdlib>
#include "kernels.h"
#include "device_utils.h"
#include "../common.h"

namespace nvr { namespace NVR_DT_NS {

enum { SEPI_F16 = 0, SEPI_SILU = 2, SEPI_ROPE = 3 };

struct StreamEpi {
    const int64_t *pos; const int32_t *slots; const float *cos_t, *sin_t;
    half_t *kc, *vc;
    int32_t H, KVH, D;
    int32_t tiled;                       // W is the retile_weight copy [N/16][K/32][16][32] (SEPI_ROPE: mode 1, tiles in s_w_row order)
};

// W row behind local row r (0..15) of part nt (SiLU: 0 = gate, 1 = up) of tile t
template <int EPI>
__device__ __forceinline__ int s_w_row(int t, int nt, int r, int N, const StreamEpi &e) {
    if (EPI == SEPI_SILU) return nt * N + t * 16 + r;                        // N == I
    if (EPI == SEPI_ROPE) {
        const int tph = e.D / 16, head = t / tph, c = t % tph;
        if (head < e.H + e.KVH) return head * e.D + (r < 8 ? c * 8 + r : e.D / 2 + c * 8 + (r - 8));
        return head * e.D + c * 16 + r;
    }
    return t * 16 + r;
}

// PRE (r05; NTT = 1, at most PRE tiles per workgroup, KC <= 8 k-steps per wave): the launch was a chain of dependent round trips, not a stream — per
// K chunk the image fill (two rounds of 8 loads per thread) and then, per tile, two rounds of U weight pieces per wave: 12 round trips for a workgroup
// with two tiles of Qwen3-8B's qkv (50 MB in 21.3 us = 2.4 TB/s, scratch/prof_8b.sh).  The weight pieces depend on nothing: with PRE every wave
// requests ALL its pieces of the chunk (PRE tiles x 8 k-steps, 64 registers) in front of the fill, and the fill asks for its 16 pieces per thread
// at once — one round trip per chunk.  The MFMAs of an accumulator run over the k-steps in the order they always had: the same bits.
//
// IMG (r06; at most IMG tiles per workgroup, KC = 1024): the activation image leaves the critical path.  r05's ablation of the Qwen3-8B qkv launch (19.2 us)
// put ~6 us on the two image fills — a stage of their own per chunk: an L2 round trip under load, 128 KiB of LDS writes and a barrier, with the weight
// pieces already landed and waiting (profiles/r05_priced_levers.txt 8., 14.).  Here the image is double-buffered in LDS (2 x 32 rows x 1024 columns =
// 128 KiB) and filled by LDS-DMA (global_load_lds: no registers, no ds_write): chunk c + 1's image is requested right behind the barrier that opens
// chunk c and flies under that chunk's MFMAs and the weight round trip behind them; a chunk boundary is ONE barrier.  The weight pieces keep r05's
// shape (every wave requests all its pieces of a chunk at once); where two register sets fit (one weight part per tile: 2 x IMG x 4 pieces = 64
// registers) chunk c + 2's pieces are requested as soon as chunk c's MFMAs have read theirs, so two chunks of weights (128 KiB per CU) are in flight
// all the time, counted s_waitcnt vmcnt(pieces of one chunk) leaving the younger one in flight (vmcnt retires in issue order: image c was requested
// before the weights of chunk c + 1).  gate_up (two parts per tile, three tiles) has one register set: issue, wait, barrier, MFMAs per chunk, the image
// of the next chunk under all of it.  A wave's k-steps are those it always had (k = 32 wave mod 256, ascending), the cross-wave order too: the same bits.
// NC = K / 1024 is a template argument and the chunk sequence straight-line code: given these loads inside a LOOP next to the LDS-DMA requests, hipcc's
// s_waitcnt pass drains vmcnt(0) in front of every MFMA block and every new request (one chunk in flight, the next image exposed), and weight pieces
// requested by inline asm to hide them from that pass get COPIED at the loop's phi nodes while still in flight (caught by tools/check_kernel_isa.py).
template <int NTT, int MT, int WAVES, int U, int EPI, int TMAX, int PRE = 0, int IMG = 0, int NC = 0>
__global__ __launch_bounds__(WAVES * 64) void linear_stream_kernel(const half_t *__restrict__ x, int64_t ldx,
                                                                   const half_t *__restrict__ W, int T, int K, int N, int KC,
                                                                   int ntiles, half_t *__restrict__ y, StreamEpi epi) {
    extern __shared__ __attribute__((aligned(16))) char smem[];             // x chunk image; later the reduction scratch
    constexpr int ROWS = MT * 16;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 15, q = lane >> 4;
    const int cpr = KC / 8;                                                  // 16-byte chunks per row of the image

    // SEPI_ROPE: the epilogue of a tile was position -> cos / sin row -> rotate and slot -> cache store: two dependent round trips per tile behind the
    // stream (r05: ~4 us per tile of the 19 us launch).  The token's position and cache slot do not depend on the tile: the epilogue waves (wave j
    // owns tokens 16 j + r) request them HERE (inline asm: a compiler-visible load would be sunk to its use; the covering s_waitcnt sits in front of
    // their first use, at the start of the epilogue).  The cos / sin pieces of ALL the workgroup's tiles are then requested together in front of
    // the first tile's reduction.
    int64_t pos_pre = 0; int slot_pre = -1;
    if (EPI == SEPI_ROPE && wave < MT) {
        const int m = wave * 16 + r;
        const int64_t *psrc = epi.pos + (m < T ? m : T - 1);
        asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(pos_pre) : "v"(psrc) : "memory");
        if (epi.slots) { const int32_t *ssrc = epi.slots + (m < T ? m : T - 1); asm volatile("global_load_dword %0, %1, off" : "=v"(slot_pre) : "v"(ssrc) : "memory"); }
    }

    constexpr int TT = IMG > 0 ? IMG : TMAX;                                 // tiles (accumulator sets) a workgroup can hold
    float4_t acc[TT][NTT][MT];
#pragma unroll
    for (int i = 0; i < TT; ++i)
#pragma unroll
        for (int nt = 0; nt < NTT; ++nt)
#pragma unroll
            for (int j = 0; j < MT; ++j) acc[i][nt][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

    if constexpr (IMG > 0) {
        static_assert(IMG <= TMAX && IMG <= 3 && PRE == 0 && NC >= 1, "IMG: tiles held by a workgroup, NC chunks of 1024 columns");
        constexpr int KPW2 = (MT <= 2 && NTT == 1) ? 4 : 2;                  // k-steps per wave and chunk (gate_up, two weight parts per tile: 2, so that TWO sets
                                                                             // of its 3 x 2 x 2 pieces fit the registers and no chunk boundary drains the stream)
        constexpr int KC2 = KPW2 * WAVES * 32, cpr2 = KC2 / 8;               // 1024 columns (33..64 rows, MT = 4: 512, so that two 64-row images fit the LDS); the
                                                                             // host passes KC = KC2, K = NC * KC2
        const int kmul = epi.tiled ? 16 : 1;
        constexpr int img_bytes = ROWS * KC2 * 2;
        int ntl = 0;                                                         // live tiles of this workgroup (uniform)
#pragma unroll
        for (int i = 0; i < IMG; ++i) ntl += (int)(blockIdx.x + i * gridDim.x) < ntiles ? 1 : 0;
        // this thread's byte offset inside a 16-row weight tile: row r (through the epilogue's row map: s_w_row(tile, nt, r) - s_w_row(tile, nt, 0), the same
        // for every tile but, row-major RoPE weights, different for rotary and value heads), 16-byte piece q, this wave's first k-step
        const int rot_r = (EPI == SEPI_ROPE) ? (r < 8 ? r : epi.D / 2 + (r - 8)) : r;
        const unsigned voff = epi.tiled ? (unsigned)((r * 32 + q * 8 + wave * 512) * 2) : (unsigned)(((int64_t)rot_r * K + q * 8 + wave * 32) * 2);
        const unsigned voff_v = (unsigned)(((int64_t)r * K + q * 8 + wave * 32) * 2);
        // Image pieces by LDS-DMA, 16 bytes per lane, lane-linear in LDS: slot p = row * cpr2 + s holds chunk s ^ (row & 7) of its row.  A wave-instruction
        // covers 64 consecutive slots of ONE row (cpr2 = 64 or 128 slots per row), so the row — and with it the clamp to T - 1 and the row's address — is
        // uniform: scalar base (x + row * ldx + chunk) + ONE per-lane byte offset (two when a row takes two instructions: the swizzle term row & 7 then
        // alternates with the round), instead of eight 64-bit per-thread pointers kept (and, next to gate_up's 96 accumulators, spilled) across the chunks.
        // As inline asm: all vector-memory traffic of this loop is counted by hand (see body); M0 (the LDS destination) is saved and restored.
        constexpr int RPI = (WAVES * 64) / cpr2;                             // image rows per round of the workgroup's waves (8 or 4)
        constexpr int IMG_IT = ROWS / RPI;                                   // rounds = requests per wave and image (8)
        static_assert(ROWS % RPI == 0 && (cpr2 == 64 || cpr2 == 128) && RPI % 4 == 0, "a wave-instruction stays inside one image row");
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);                // (uniform by construction; said so, for the scalar bases below)
        const int row_w = cpr2 == 64 ? wave_u : wave_u >> 1, slot_w = cpr2 == 64 ? lane : (wave_u & 1) * 64 + lane;   // row inside a round, first slot's index in the row
        // (row & 7) of round `it` = (it * RPI + row_w) & 7: RPI = 8 -> row_w & 7 for every round; RPI = 4 -> alternates between row_w and row_w + 4
        const unsigned ioff0 = (unsigned)((slot_w ^ (row_w & 7)) * 16), ioff1 = (unsigned)((slot_w ^ ((row_w + 4) & 7)) * 16);
        const unsigned lds_img = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
        auto issue_img = [&](int c, int buf) {
#pragma unroll
            for (int it = 0; it < IMG_IT; ++it) {
                const int row = it * RPI + row_w;
                const half_t *base = x + (int64_t)(row < T ? row : T - 1) * ldx + c * KC2;
                const unsigned m0v = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_img + buf * img_bytes + (it * WAVES * 64 + wave_u * 64) * 16));
                const unsigned vo = (RPI == 4 && (it & 1)) ? ioff1 : ioff0;
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(vo), "s"(base), "s"(m0v) : "memory");
            }
        };
        // One straight-line body per number of live tiles NTL (1 .. IMG), picked once: no loop and no branch between a request and its wait.  The weight
        // pieces are requested by inline asm and waited for by hand-counted s_waitcnt: shown to the compiler next to the LDS-DMA requests, its own
        // s_waitcnt pass treats vmcnt as out of order (global_load_lds touches two address spaces) and drains vmcnt(0) in front of every MFMA block — one
        // chunk in flight, the next image exposed.  An inline-asm load is only safe while nothing copies its destination before it has landed: with loops
        // or uniform branches in between hipcc puts such copies at the phi nodes (seen; tools/check_kernel_isa.py reads the ISA of every instantiation:
        // no scratch, every destination untouched until the wait that covers it).
        auto body = [&](auto ntl_c) {
            constexpr int NTL = decltype(ntl_c)::value;
            constexpr bool DB = NTL * NTT * KPW2 * 2 <= 16 || (MT <= 2 && NTL * NTT * KPW2 * 2 <= 24);   // two register sets of weight pieces (<= 64 VGPRs; 96 beside
                                                                                                         // the 48 accumulators of gate_up at <= 32 rows)
            constexpr int NW = NTL * NTT * KPW2;                             // vector-memory requests of one chunk of weight pieces, per wave
            // address of a piece = a UNIFORM base (tile, part, chunk, k-step: scalar registers) + the thread's byte offset inside a 16-row tile (one VGPR
            // for the whole kernel; two for row-major qkv weights, whose rotary and value heads order their rows differently): as 64-bit per-thread
            // pointers the NC x NTL x NTT x 4 addresses were hoisted and spilled (gate_up: 96 pointers per chunk)
            auto issue_w = [&](half8_t (&a)[NTL][NTT][KPW2], int c) {
#pragma unroll
                for (int i = 0; i < NTL; ++i) {
                    const int tile = blockIdx.x + i * gridDim.x;
#pragma unroll
                    for (int nt = 0; nt < NTT; ++nt) {
                        const half_t *base = epi.tiled ? W + ((int64_t)(EPI == SEPI_SILU ? nt * (N / 16) + tile : tile) * (K / 32) + (c * KC2) / 32) * 512
                                                       : W + (int64_t)s_w_row<EPI>(tile, nt, 0, N, epi) * K + c * KC2;
                        const unsigned vo = (EPI == SEPI_ROPE && !epi.tiled && tile / (epi.D / 16) >= epi.H + epi.KVH) ? voff_v : voff;
#pragma unroll
                        for (int u = 0; u < KPW2; ++u) {
                            const half_t *bu = base + (int64_t)(u * WAVES * 32) * kmul;
                            asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(a[i][nt][u]) : "v"(vo), "s"(bu) : "memory");
                        }
                    }
                }
            };
            auto pin = [&](half8_t (&a)[NTL][NTT][KPW2]) {                   // uses of the pieces stay behind the wait in front of this (volatile asms keep their order)
#pragma unroll
                for (int i = 0; i < NTL; ++i)
#pragma unroll
                    for (int nt = 0; nt < NTT; ++nt)
#pragma unroll
                        for (int u = 0; u < KPW2; ++u) asm volatile("" : "+v"(a[i][nt][u]));
            };
            auto mfmas = [&](const half8_t (&a)[NTL][NTT][KPW2], int buf) {
                const char *img = smem + buf * img_bytes;
#pragma unroll
                for (int i = 0; i < NTL; ++i)
#pragma unroll
                    for (int u = 0; u < KPW2; ++u) {
                        const int ch = ((wave * 32 + u * WAVES * 32) >> 3) + q;
#pragma unroll
                        for (int j = 0; j < MT; ++j) {
                            const int row = j * 16 + r;
                            const half8_t b = *reinterpret_cast<const half8_t *>(img + (row * cpr2 + (ch ^ (r & 7))) * 16);
#pragma unroll
                            for (int nt = 0; nt < NTT; ++nt) acc[i][nt][j] = mfma16(a[i][nt][u], b, acc[i][nt][j]);
                        }
                    }
            };
            issue_img(0, 0);
            if constexpr (DB) {
                // issue order: img 0, w 0, w 1 | chunk c: [wait: all but the youngest NW = w c+1] barrier, img c+1, MFMAs c, w c+2 — so at the top of chunk
                // c + 1 the requests in flight are, oldest first, w c+1, img c+1, w c+2: vmcnt(NW) leaves exactly w c+2
                half8_t a0[NTL][NTT][KPW2], a1[NTL][NTT][KPW2];
                issue_w(a0, 0);
                if constexpr (NC > 1) issue_w(a1, 1);
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    auto &a = (c & 1) ? a1 : a0;
                    if (c + 1 < NC) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NW) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    pin(a);
                    __builtin_amdgcn_s_barrier();                            // image c complete; nobody reads buffer (c + 1) & 1 any more
                    if (c + 1 < NC) issue_img(c + 1, (c + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);                       // (the counted waits rest on the ISSUE ORDER image c + 1, then weights c + 2)
                    mfmas(a, c & 1);
                    __builtin_amdgcn_sched_barrier(0);
                    if (c + 2 < NC) issue_w(a, c + 2);
                }
            } else {
                half8_t a[NTL][NTT][KPW2];
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    issue_w(a, c);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // image c (older) and this chunk's pieces
                    pin(a);
                    __builtin_amdgcn_s_barrier();
                    if (c + 1 < NC) issue_img(c + 1, (c + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);
                    mfmas(a, c & 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        if (ntl <= 1) body(std::integral_constant<int, 1>{});
        else if (IMG >= 2 && ntl == 2) body(std::integral_constant<int, (IMG >= 2 ? 2 : 1)>{});
        else body(std::integral_constant<int, IMG>{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // (nothing is in flight here; written down for every path)
    }
    for (int kc0 = 0; IMG == 0 && kc0 < K; kc0 += KC) {
        if constexpr (PRE > 0) {
            static_assert(NTT == 1 && PRE <= TMAX, "PRE: one weight part per tile");
            constexpr int KPW = 8;                                           // k-steps per wave and chunk (host: KC <= KPW * WAVES * 32)
            half8_t a[PRE][KPW];
            const int kmul = epi.tiled ? 16 : 1;
#pragma unroll
            for (int i = 0; i < PRE; ++i) {
                const int tile = blockIdx.x + i * gridDim.x;
                const bool live = tile < ntiles;
                const int tl = live ? tile : 0;
                const half_t *wr = epi.tiled ? W + ((int64_t)tl * (K / 32) + kc0 / 32) * 512 + r * 32 + q * 8
                                             : W + (int64_t)s_w_row<EPI>(tl, 0, r, N, epi) * K + kc0 + q * 8;
#pragma unroll
                for (int u = 0; u < KPW; ++u) {
                    const int kk = wave * 32 + u * WAVES * 32;
                    a[i][u] = live && kk < KC ? __builtin_nontemporal_load(reinterpret_cast<const half8_t *>(wr + (int64_t)kk * kmul)) : (half8_t)(half_t)0;
                }
            }
            __syncthreads();                                                 // the previous chunk's readers are done
            fill_x_image<ROWS, WAVES * 64, 16>(smem, x, ldx, kc0, cpr, T, tid, (int)((blockIdx.x >> 3) * (cpr + 8)) % (ROWS * cpr));
            __syncthreads();
#pragma unroll
            for (int i = 0; i < PRE; ++i) {
                if ((int)(blockIdx.x + i * gridDim.x) >= ntiles) break;
#pragma unroll
                for (int u = 0; u < KPW; ++u) {
                    const int kk = wave * 32 + u * WAVES * 32;
                    if (kk < KC) {
                        const int ch = (kk >> 3) + q;
#pragma unroll
                        for (int j = 0; j < MT; ++j) {
                            const int row = j * 16 + r;
                            const half8_t b = *reinterpret_cast<const half8_t *>(smem + ((int64_t)row * cpr + (ch ^ (r & 7))) * 16);
                            acc[i][0][j] = mfma16(a[i][u], b, acc[i][0][j]);
                        }
                    }
                }
            }
            continue;
        }
        __syncthreads();                                                     // the previous chunk's readers are done
        fill_x_image<ROWS, WAVES * 64, 8>(smem, x, ldx, kc0, cpr, T, tid);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TT; ++i) {
            const int tile = blockIdx.x + i * gridDim.x;
            if (tile >= ntiles) break;
            const half_t *wr[NTT];
#pragma unroll
            for (int nt = 0; nt < NTT; ++nt)
                wr[nt] = epi.tiled ? W + ((int64_t)(EPI == SEPI_SILU ? nt * (N / 16) + tile : tile) * (K / 32) + kc0 / 32) * 512 + r * 32 + q * 8
                                   : W + (int64_t)s_w_row<EPI>(tile, nt, r, N, epi) * K + kc0 + q * 8;
            const int kmul = epi.tiled ? 16 : 1;                             // a k-step of 32 halfs is one 512-half tile further in the tiled copy
            for (int kb = wave * 32; kb < KC; kb += WAVES * 32 * U) {
                half8_t a[U][NTT];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int kk = kb + u * WAVES * 32;
#pragma unroll
                    for (int nt = 0; nt < NTT; ++nt)
                        a[u][nt] = kk < KC ? __builtin_nontemporal_load(reinterpret_cast<const half8_t *>(wr[nt] + (int64_t)kk * kmul)) : (half8_t)(half_t)0;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int kk = kb + u * WAVES * 32;
                    if (kk < KC) {
                        const int ch = (kk >> 3) + q;
#pragma unroll
                        for (int j = 0; j < MT; ++j) {
                            const int row = j * 16 + r;
                            const half8_t b = *reinterpret_cast<const half8_t *>(smem + ((int64_t)row * cpr + (ch ^ (r & 7))) * 16);
#pragma unroll
                            for (int nt = 0; nt < NTT; ++nt)
                                acc[i][nt][j] = mfma16(a[u][nt], b, acc[i][nt][j]);
                        }
                    }
                }
            }
        }
    }

    // per tile: the waves' partial sums meet in LDS (fixed order), then the epilogue.  C layout: row (n) = q*4 + e, col (token) = r
    float4_t *part = reinterpret_cast<float4_t *>(smem);                     // [WAVES][NTT*MT][64]
    auto reduce = [&](int slot) {
        float4_t s = part[(0 * NTT * MT + slot) * 64 + lane];
#pragma unroll
        for (int w2 = 1; w2 < WAVES; ++w2) s += part[(w2 * NTT * MT + slot) * 64 + lane];
        return s;
    };
    constexpr int ET = IMG > 0 ? IMG : PRE > 0 ? PRE : TMAX;                 // tiles a workgroup can hold
    float4_t cs_pre[EPI == SEPI_ROPE ? ET : 1], sn_pre[EPI == SEPI_ROPE ? ET : 1];
    // (landed long ago — they are older than the image fills that have been waited for — but the compiler cannot know that of an asm load: the wait is
    //  written down here, where nothing is in flight and it costs nothing, and the values are tied to it.  By EVERY wave, outside the branch of the
    //  requesting waves: every control-flow path from a request to a use of its register passes this wait, which tools/check_kernel_isa.py verifies
    //  on the ISA without having to know that `wave < MT` below is the condition the requests were issued under — ADVICE r05)
    if (EPI == SEPI_ROPE) asm volatile("s_waitcnt vmcnt(0)" : "+v"(pos_pre), "+v"(slot_pre) : : "memory");
    if (EPI == SEPI_ROPE && wave < MT) {
        const int tph = epi.D / 16, half_d = epi.D / 2;
#pragma unroll
        for (int i = 0; i < ET; ++i) {
            const int tile = blockIdx.x + i * gridDim.x;
            const int tl = tile < ntiles ? tile : 0;
            const int jj = (tl % tph) * 8 + (q & 1) * 4;
            cs_pre[i] = *reinterpret_cast<const float4_t *>(epi.cos_t + pos_pre * half_d + jj);
            sn_pre[i] = *reinterpret_cast<const float4_t *>(epi.sin_t + pos_pre * half_d + jj);
        }
    }
#pragma unroll
    for (int i = 0; i < TT; ++i) {
        const int tile = blockIdx.x + i * gridDim.x;
        if (tile >= ntiles) break;
        __syncthreads();                                                     // image / previous tile's scratch no longer read
#pragma unroll
        for (int nt = 0; nt < NTT; ++nt)
#pragma unroll
            for (int j = 0; j < MT; ++j) part[(wave * NTT * MT + nt * MT + j) * 64 + lane] = acc[i][nt][j];
        __syncthreads();
        for (int j = wave; j < MT; j += WAVES) {
            const int m = j * 16 + r;
            if (EPI == SEPI_F16) {
                const float4_t s = reduce(j);
                const int n = tile * 16 + q * 4;
                if (m < T) {
                    const half4_t h = {(half_t)s[0], (half_t)s[1], (half_t)s[2], (half_t)s[3]};
                    *reinterpret_cast<half4_t *>(y + (int64_t)m * N + n) = h;
                }
            } else if (EPI == SEPI_SILU) {
                // act = fp16(silu(fp16 gate) * fp16 up), activation.rs:46-63
                const float4_t g4 = reduce(j), u4 = reduce(MT + j);
                const int n = tile * 16 + q * 4;
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
            } else {                                                         // SEPI_ROPE
                const int tph = epi.D / 16, head = tile / tph, c = tile % tph, half_d = epi.D / 2;
                const int64_t ldq = (int64_t)(epi.H + 2 * epi.KVH) * epi.D;
                const float4_t s = reduce(j);
                float v[4], pv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = (float)to_half_rn(s[e]); pv[e] = __shfl_xor(v[e], 32, 64); }
                half4_t h;
                int col;                                                     // first of the lane's 4 consecutive head columns
                if (head < epi.H + epi.KVH) {
                    const int jj = c * 8 + (q & 1) * 4;                      // index inside the half dimension
                    const float4_t cs = cs_pre[i < ET ? i : 0], sn = sn_pre[i < ET ? i : 0];   // (requested in front of the tile loop)
#pragma unroll
                    for (int e = 0; e < 4; ++e)                              // rotary_embedding.rs:36-44
                        h[e] = (q < 2) ? to_half_rn(mul_sub_unfused(v[e], cs[e], pv[e], sn[e]))
                                       : to_half_rn(mul_add_unfused(v[e], cs[e], pv[e], sn[e]));
                    col = (q < 2) ? jj : half_d + jj;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) h[e] = to_half_rn(s[e]);
                    col = c * 16 + q * 4;
                }
                if (m < T) {
                    *reinterpret_cast<half4_t *>(y + (int64_t)m * ldq + head * epi.D + col) = h;
                    const int slot = epi.slots ? slot_pre : -1;
                    if (slot >= 0 && head >= epi.H) {
                        const bool is_k = head < epi.H + epi.KVH;
                        const int kvh = is_k ? head - epi.H : head - epi.H - epi.KVH;
                        half_t *dst = (is_k ? epi.kc : epi.vc) + ((int64_t)slot * epi.KVH + kvh) * epi.D + col;
                        *reinterpret_cast<half4_t *>(dst) = h;
                    }
                }
            }
        }
    }
}

// ---- host side -------------------------------------------------------------------------------------------------------
constexpr int S_WAVES = 8, S_TMAX = 4;
static constexpr bool stream_enabled() { return true; }
static int stream_kc(int64_t T, int64_t K) {                                 // K chunk held in LDS (<= 128 KiB image)
    const int64_t cap = T <= 16 ? 4096 : 2048;
    return (int)(K <= cap ? K : cap);
}
// worth it and expressible: large weights, whole chunks, more tiles than workgroups (with one tile per workgroup the
// activation-chunk fills weigh as much as the weight stream and the skinny kernel is as fast: scratch/stream_bench.py, Qwen3-8B
// shapes: gate_up 48 vs 63 us, qkv 23.6 vs 28.5 us, but o_proj 17.0 vs 15.5 us), at most TMAX tiles per workgroup
static bool stream_shape_ok(int64_t T, int64_t K, int64_t tiles, int64_t weight_bytes, int64_t ldx, int parts) {
    // 33..64 rows (r06): only the double-buffered-image instantiations exist (64-row images of 512 columns), i.e. hidden 2048 / 4096 and at most 2 (one weight
    // part per tile) or 3 (gate_up) tiles per workgroup — before, such batches of an 8B-class model fell to the skinny kernel over two 32-row blocks
    // (Qwen3-8B bs 64: 0.36 of its step roofline between 0.58 at bs 32 and 0.60 at bs 128)
    if (T > 32) return stream_enabled() && T <= 64 && ldx % 8 == 0 && (K == 2048 || K == 4096) && weight_bytes >= (24ll << 20) && tiles >= 320 && tiles <= (parts == 1 ? 2 : 3) * 256;
    if (!stream_enabled() || T < 1 || T > 32 || ldx % 8 || K < 2048 || weight_bytes < (24ll << 20)) return false;
    const int kc = stream_kc(T, K);
    return K % kc == 0 && kc % (S_WAVES * 32) == 0 && tiles >= 320 && tiles <= 256 * S_TMAX;
}
bool linear_stream_ok(int64_t T, int64_t K, int64_t N, int64_t ldx) { return N % 16 == 0 && stream_shape_ok(T, K, N / 16, N * K * 2, ldx, 1); }
bool linear_stream_silu_ok(int64_t T, int64_t K, int64_t I, int64_t ldx) { return I % 16 == 0 && stream_shape_ok(T, K, I / 16, 2 * I * K * 2, ldx, 2); }
bool linear_stream_rope_ok(int64_t T, int64_t K, int64_t H, int64_t KVH, int64_t D, int64_t ldx) {
    return D % 16 == 0 && stream_shape_ok(T, K, (H + 2 * KVH) * D / 16, (H + 2 * KVH) * D * K * 2, ldx, 1);
}

int linear_stream_prepare();
template <int NTT, int MT, int U, int EPI>
static int stream_launch(const half_t *x, int64_t ldx, const half_t *W, int T, int K, int N, int ntiles, half_t *y, const StreamEpi &e,
                         hipStream_t s) {
    static bool prepared = false;                                            // 128-160 KiB of dynamic LDS: opt-in (see linear_stream_prepare)
    if (!prepared) { if (int rc = linear_stream_prepare()) return rc; prepared = true; }
    const int kc = stream_kc(T, K);
    size_t lds = (size_t)MT * 16 * kc * 2;
    const size_t scratch = (size_t)S_WAVES * NTT * MT * 64 * 16;
    if (lds < scratch) lds = scratch;
    const int nwg = ntiles < 256 ? ntiles : 256;
    {   // r06: double-buffered LDS-DMA image, 1024-column chunks (IMG): up to 2 tiles per workgroup with one weight part each (qkv, plain), 3 with two (gate_up)
        constexpr int IMGT = NTT == 1 ? 2 : 3;
        if ((K == 2048 || K == 4096) && ntiles <= IMGT * nwg) {               // (NC = 2: hidden 2048; NC = 4: hidden 4096 — Qwen3-8B; other widths keep the r05 kernels)
            constexpr int KC2 = (MT <= 2 && NTT == 1) ? 1024 : 512;           // chunk width (kernel: KPW2 * WAVES * 32)
            const size_t lds2 = std::max<size_t>((size_t)2 * MT * 16 * KC2 * 2, scratch);
            if (K == 4096) linear_stream_kernel<NTT, MT, S_WAVES, U, EPI, S_TMAX, 0, IMGT, 4096 / KC2><<<dim3((unsigned)nwg), dim3(S_WAVES * 64), lds2, s>>>(x, ldx, W, T, K, N, KC2, ntiles, y, e);
            else linear_stream_kernel<NTT, MT, S_WAVES, U, EPI, S_TMAX, 0, IMGT, 2048 / KC2><<<dim3((unsigned)nwg), dim3(S_WAVES * 64), lds2, s>>>(x, ldx, W, T, K, N, KC2, ntiles, y, e);
            hipError_t er = hipGetLastError();
            if (er != hipSuccess) return nvr::fail(NVR_ERR_HIP, "linear_stream launch failed: %s", hipGetErrorString(er));
            return 0;
        }
    }
    if constexpr (MT > 2) return nvr::fail(NVR_ERR_UNSUPPORTED, "linear_stream: %d rows need hidden 2048 / 4096 (K = %d)", T, K);
    else {
    if constexpr (NTT == 1) {
        if (kc <= 8 * S_WAVES * 32 && ntiles <= 2 * nwg) {                   // all weight pieces of a chunk requested in front of the fill (PRE)
            linear_stream_kernel<NTT, MT, S_WAVES, U, EPI, S_TMAX, 2><<<dim3((unsigned)nwg), dim3(S_WAVES * 64), lds, s>>>(x, ldx, W, T, K, N, kc, ntiles, y, e);
            hipError_t er = hipGetLastError();
            if (er != hipSuccess) return nvr::fail(NVR_ERR_HIP, "linear_stream launch failed: %s", hipGetErrorString(er));
            return 0;
        }
    }
    linear_stream_kernel<NTT, MT, S_WAVES, U, EPI, S_TMAX><<<dim3((unsigned)nwg), dim3(S_WAVES * 64), lds, s>>>(x, ldx, W, T, K, N, kc, ntiles, y, e);
    hipError_t er = hipGetLastError();
    if (er != hipSuccess) return nvr::fail(NVR_ERR_HIP, "linear_stream launch failed: %s", hipGetErrorString(er));
    return 0;
    }
}

// opt every instance in to > 64 KiB of dynamic LDS up front (runner init: never inside a stream capture)
int linear_stream_prepare() {
    const void *fns[] = {
        reinterpret_cast<const void *>(&linear_stream_kernel<1, 1, S_WAVES, 4, SEPI_F16, S_TMAX>), reinterpret_cast<const void *>(&linear_stream_kernel<1, 2, S_WAVES, 4, SEPI_F16, S_TMAX>),
        reinterpret_cast<const void *>(&linear_stream_kernel<2, 1, S_WAVES, 2, SEPI_SILU, S_TMAX>), reinterpret_cast<const void *>(&linear_stream_kernel<2, 2, S_WAVES, 2, SEPI_SILU, S_TMAX>),
        reinterpret_cast<const void *>(&linear_stream_kernel<1, 1, S_WAVES, 4, SEPI_ROPE, S_TMAX>), reinterpret_cast<const void *>(&linear_stream_kernel<1, 2, S_WAVES, 4, SEPI_ROPE, S_TMAX>),
        reinterpret_cast<const void *>(&linear_stream_kernel<1, 1, S_WAVES, 4, SEPI_F16, S_TMAX, 2>), reinterpret_cast<const void *>(&linear_stream_kernel<1, 2, S_WAVES, 4, SEPI_F16, S_TMAX, 2>),
        reinterpret_cast<const void *>(&linear_stream_kernel<1, 1, S_WAVES, 4, SEPI_ROPE, S_TMAX, 2>), reinterpret_cast<const void *>(&linear_stream_kernel<1, 2, S_WAVES, 4, SEPI_ROPE, S_TMAX, 2>),
#define NVR_LS_IMG(NC_) \
        reinterpret_cast<const void *>(&linear_stream_kernel<1, 1, S_WAVES, 4, SEPI_F16, S_TMAX, 0, 2, NC_>), reinterpret_cast<const void *>(&linear_stream_kernel<1, 2, S_WAVES, 4, SEPI_F16, S_TMAX, 0, 2, NC_>), \
        reinterpret_cast<const void *>(&linear_stream_kernel<1, 1, S_WAVES, 4, SEPI_ROPE, S_TMAX, 0, 2, NC_>), reinterpret_cast<const void *>(&linear_stream_kernel<1, 2, S_WAVES, 4, SEPI_ROPE, S_TMAX, 0, 2, NC_>), \
        reinterpret_cast<const void *>(&linear_stream_kernel<2, 1, S_WAVES, 2, SEPI_SILU, S_TMAX, 0, 3, 2 * NC_>), reinterpret_cast<const void *>(&linear_stream_kernel<2, 2, S_WAVES, 2, SEPI_SILU, S_TMAX, 0, 3, 2 * NC_>)
        NVR_LS_IMG(2), NVR_LS_IMG(4),
#define NVR_LS_IMG4(NC_) \
        reinterpret_cast<const void *>(&linear_stream_kernel<1, 4, S_WAVES, 4, SEPI_F16, S_TMAX, 0, 2, NC_>), reinterpret_cast<const void *>(&linear_stream_kernel<1, 4, S_WAVES, 4, SEPI_ROPE, S_TMAX, 0, 2, NC_>), \
        reinterpret_cast<const void *>(&linear_stream_kernel<2, 4, S_WAVES, 2, SEPI_SILU, S_TMAX, 0, 3, NC_>)
        NVR_LS_IMG4(4), NVR_LS_IMG4(8)};
#undef NVR_LS_IMG4
#undef NVR_LS_IMG
    for (const void *f : fns) {
        hipError_t er = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (er != hipSuccess) return nvr::fail(NVR_ERR_HIP, "linear_stream: hipFuncSetAttribute: %s", hipGetErrorString(er));
    }
    return 0;
}

int linear_stream(const half_bits *x, int64_t ldx, const half_bits *W, int64_t T, int64_t K, int64_t N, half_bits *y, hipStream_t s, const half_bits *Wt) {
    if (!linear_stream_ok(T, K, N, ldx)) return nvr::fail(NVR_ERR_UNSUPPORTED, "linear_stream: T=%ld K=%ld N=%ld", (long)T, (long)K, (long)N);
    StreamEpi e{};
    if (Wt) { W = Wt; e.tiled = 1; }
    if (T <= 16) return stream_launch<1, 1, 4, SEPI_F16>((const half_t *)x, ldx, (const half_t *)W, (int)T, (int)K, (int)N, (int)(N / 16), (half_t *)y, e, s);
    if (T > 32) return stream_launch<1, 4, 4, SEPI_F16>((const half_t *)x, ldx, (const half_t *)W, (int)T, (int)K, (int)N, (int)(N / 16), (half_t *)y, e, s);
    return stream_launch<1, 2, 4, SEPI_F16>((const half_t *)x, ldx, (const half_t *)W, (int)T, (int)K, (int)N, (int)(N / 16), (half_t *)y, e, s);
}
int linear_stream_silu_mul(const half_bits *x, int64_t ldx, const half_bits *W, int64_t T, int64_t K, int64_t I, half_bits *out, hipStream_t s,
                           const half_bits *Wt) {
    if (!linear_stream_silu_ok(T, K, I, ldx)) return nvr::fail(NVR_ERR_UNSUPPORTED, "linear_stream_silu_mul: T=%ld K=%ld I=%ld", (long)T, (long)K, (long)I);
    StreamEpi e{};
    if (Wt) { W = Wt; e.tiled = 1; }
    if (T <= 16) return stream_launch<2, 1, 2, SEPI_SILU>((const half_t *)x, ldx, (const half_t *)W, (int)T, (int)K, (int)I, (int)(I / 16), (half_t *)out, e, s);
    if (T > 32) return stream_launch<2, 4, 2, SEPI_SILU>((const half_t *)x, ldx, (const half_t *)W, (int)T, (int)K, (int)I, (int)(I / 16), (half_t *)out, e, s);
    return stream_launch<2, 2, 2, SEPI_SILU>((const half_t *)x, ldx, (const half_t *)W, (int)T, (int)K, (int)I, (int)(I / 16), (half_t *)out, e, s);
}
int linear_stream_qkv_rope_store(const half_bits *x, int64_t ldx, const half_bits *W, int64_t T, int64_t K, int64_t H, int64_t KVH, int64_t D,
                                 const int64_t *positions, const int32_t *slots, const float *cos_t, const float *sin_t, half_bits *qkv,
                                 half_bits *k_cache, half_bits *v_cache, hipStream_t s, const half_bits *Wt) {
    if (!linear_stream_rope_ok(T, K, H, KVH, D, ldx)) return nvr::fail(NVR_ERR_UNSUPPORTED, "linear_stream_qkv_rope_store: T=%ld K=%ld D=%ld", (long)T, (long)K, (long)D);
    StreamEpi e{};
    e.pos = positions; e.slots = slots; e.cos_t = cos_t; e.sin_t = sin_t; e.kc = (half_t *)k_cache; e.vc = (half_t *)v_cache;
    e.H = (int32_t)H; e.KVH = (int32_t)KVH; e.D = (int32_t)D;
    if (Wt) { W = Wt; e.tiled = 1; }
    const int N = (int)((H + 2 * KVH) * D);
    if (T <= 16) return stream_launch<1, 1, 4, SEPI_ROPE>((const half_t *)x, ldx, (const half_t *)W, (int)T, (int)K, N, N / 16, (half_t *)qkv, e, s);
    if (T > 32) return stream_launch<1, 4, 4, SEPI_ROPE>((const half_t *)x, ldx, (const half_t *)W, (int)T, (int)K, N, N / 16, (half_t *)qkv, e, s);
    return stream_launch<1, 2, 4, SEPI_ROPE>((const half_t *)x, ldx, (const half_t *)W, (int)T, (int)K, N, N / 16, (half_t *)qkv, e, s);
}

}}  // namespace nvr::k / nvr::kb (NVR_DT_NS)
