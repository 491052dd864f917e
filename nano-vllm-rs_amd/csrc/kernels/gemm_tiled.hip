// gemm_tiled.hip — LDS-tiled MFMA GEMM for the prefill regime (T >= 128 tokens): y[T,N] = x[T,K]·W[N,K]^T
// with the same three epilogues as the decode kernel (plain fp16, gate_up -> SiluAndMul, qkv -> RoPE + KV store).
// reference call sites as in linear.hip (linear.rs:354-356,228-239,437-439; activation.rs:46-63;
// rotary_embedding.rs:23-48; attention.rs:150-174).  Bound: MFMA (dense contraction, SURVEY §8d:
// 880.8 MFLOP per prefill token for Qwen3-0.6B).
//
// Structure (cdna_hip_programming.md §5 "standard MFMA GEMM main loop", the 128x128 two-barrier form):
//   workgroup = 4 waves = 128 (n) x 128 (m) output tile, BK = 64; wave (wn, wm) owns 64 x 64 = 4x4 MFMA tiles
//   of v_mfma_f32_16x16x32_f16 (A = W rows, B = x rows, both K-contiguous in memory);
//   both operand tiles are staged global -> LDS by global_load_lds_dwordx4 (16 B per lane, no VGPR round
//   trip) into two buffers, the load of K-tile t+1 overlapping the MFMAs on tile t;
//   LDS image = [row][8 chunks of 16 B] with chunk' = chunk ^ (row & 7): the swizzle is applied to the per-lane
//   SOURCE address (the LDS destination of glds is lane-linear) and again on the ds_read_b128 address, so a
//   16-lane fragment read touches 8 distinct 16-B slots (2-way instead of 16-way conflicts).
// Row mapping of the A operand lets one lane hold what its epilogue needs: for SiluAndMul a wave's 4 n-tiles are
// [gate c, gate c+16, up c, up c+16]; for RoPE an n-tile is 8 columns of the first half of a head and their 8
// partners of the second half (the partner of lane l sits in lane l^32).
#include <cstdlib>
#include "kernels.h"
#include "device_utils.h"
#include "../common.h"

namespace nvr { namespace NVR_DT_NS {

enum { TEPI_F16 = 0, TEPI_SILU = 2, TEPI_ROPE = 3, TEPI_LMHEAD = 4, TEPI_SLAB = 5 };

struct TileEpi {
    const int64_t *pos; const int32_t *slots; const float *cos_t, *sin_t;
    half_t *kc, *vc;
    int32_t H, KVH, D;
    // TEPI_LMHEAD: f32 logits [T, N] (nullable) and the greedy arg-max of every row over this workgroup's 128 columns:
    // pval / pidx [column tile][T] (maximum, lowest index), merged by argmax_partials
    float *logits; float *pval; int32_t *pidx;
    // TEPI_SLAB: blockIdx.z owns k in [z*kslice, (z+1)*kslice) and writes its f32 partial tile to slabs[z][T][N]
    float *slabs; int64_t slab_stride; int32_t kslice;
};

constexpr int BM = 128, BN = 128, BK = 64;
// ring depths of the mid-batch instances (one workgroup per CU): K-tiles in flight = depth - 1.  r04 read the launches of these GEMMs as "3.4 us + the
// workgroup's operand bytes at 50-56 GB/s per CU"; that rate is bytes in flight over the L2 round trip (3 K-tiles x 24 KiB per CU in the 4-deep ring)
#ifndef NVR_RING64
#define NVR_RING64 4
#endif
#ifndef NVR_RING32
#define NVR_RING32 4
#endif
#ifndef NVR_RING96
#define NVR_RING96 4
#endif
constexpr int RING64 = NVR_RING64, RING32 = NVR_RING32, RING96 = NVR_RING96;

// s_waitcnt vmcnt(younger x PER + EXTRA): K-tile kt has landed once at most the PER requests per thread of each of the `younger` K-tiles behind it (and EXTRA
// requests issued after them) are outstanding
template <int PER, int EXTRA = 0>
__device__ __forceinline__ void wait_ring(int younger) {
    static_assert(6 * PER + EXTRA <= 63, "vmcnt is a 6-bit count");
    switch (younger) {
    case 0: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(EXTRA) : "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER + EXTRA) : "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER + EXTRA) : "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PER + EXTRA) : "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * PER + EXTRA) : "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * PER + EXTRA) : "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * PER + EXTRA) : "memory"); break;
    }
}

// W row of local row r (0..15) of the workgroup's 16-row n-tile `t` (0..7)
template <int EPI>
__device__ __forceinline__ int tile_w_row(int bx, int t, int r, int N, const TileEpi &e) {
    if (EPI == TEPI_SILU) {                       // N == I; the workgroup owns 64 output columns
        const int wn = t >> 2, nt = t & 3;
        return (nt >> 1) * N + bx * 64 + wn * 32 + (nt & 1) * 16 + r;
    }
    if (EPI == TEPI_ROPE) {
        const int g = bx * 8 + t, tph = e.D / 16, head = g / tph, c = g % tph;
        if (head < e.H + e.KVH) return head * e.D + (r < 8 ? c * 8 + r : e.D / 2 + c * 8 + (r - 8));
        return head * e.D + c * 16 + r;
    }
    return bx * BN + t * 16 + r;
}

// NS = 2: two LDS buffers (64 KiB, two workgroups per CU), the load of K-tile t+1 overlaps the MFMAs on tile t and is waited
// for in full at the end of the step — right when tiles outnumber the CUs.  NS = 4: a ring of four buffers (128 KiB, one
// workgroup per CU) with three K-tiles in flight and counted s_waitcnt vmcnt: when there are fewer tiles than CUs (decode
// batches of 129..~1000 rows, short prefills) a workgroup is alone on its CU and every K-step of the NS = 2 form costs one
// full memory latency (21 us for a K = 1024 tile); the ring hides it.
// MT: 16-token m-tiles per wave.  4 = the 128-token tile; 2 / 1 = a 64- / 32-token tile for batches of 65..~1000 rows, where 128 x 128 tiles
// leave half the CUs without a workgroup (T = 512: 128 / 96 / 128 workgroups for qkv / gate_up / the split-k GEMMs; 23 / 16 / 12 us):
// twice the workgroups, each with the same 128 W rows and half the tokens.
template <int EPI, int NS, int MT = 4>
__global__ __launch_bounds__(256) void gemm_tiled_kernel(const half_t *__restrict__ x, int64_t ldx,
                                                         const half_t *__restrict__ W, int T, int K, int N, int NW,
                                                         half_t *__restrict__ y, TileEpi epi) {
    // one LDS array (cdna guide §5 item 4a): [NS buffers][A 16 KiB | B 16 or 8 KiB]
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BMt = 32 * MT;                                          // tokens per workgroup
    constexpr int BUF = (BN + BMt) * BK * 2;                              // bytes of one LDS buffer
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 15, q = lane >> 4;
    const int wn = wave >> 1, wm = wave & 1;
    const int m0 = blockIdx.y * BMt;

    // staging: thread t copies 16-byte piece idx = i*256 + t (row = idx/8, LDS chunk c' = idx%8) of each operand tile
    const half_t *asrc[4], *bsrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = i * 256 + threadIdx.x, row = idx >> 3, c = (idx & 7) ^ (row & 7);
        int wr = tile_w_row<EPI>(blockIdx.x, row >> 4, row & 15, N, epi);
        if (wr > NW - 1) wr = NW - 1;
        int xr = m0 + row; if (xr > T - 1) xr = T - 1;
        asrc[i] = W + (int64_t)wr * K + c * 8;
        bsrc[i] = x + (int64_t)xr * ldx + c * 8;                          // (rows >= BMt: unused)
        if (EPI == TEPI_SLAB) { asrc[i] += (int64_t)blockIdx.z * epi.kslice; bsrc[i] += (int64_t)blockIdx.z * epi.kslice; }
    }
    auto stage = [&](int buf, int k0) {
        char *a_dst = smem + buf * BUF, *b_dst = a_dst + BN * BK * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = (i * 256 + wave * 64) * 16;               // wave-uniform LDS base; hardware adds lane*16
            __builtin_amdgcn_global_load_lds(asrc[i] + k0, (__attribute__((address_space(3))) void *)(a_dst + piece), 16, 0, 0);
            if (i < MT) __builtin_amdgcn_global_load_lds(bsrc[i] + k0, (__attribute__((address_space(3))) void *)(b_dst + piece), 16, 0, 0);
        }
    };

    float4_t acc[4][MT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

    // TEPI_ROPE on the 32- / 64-token tiles (decode batches of 65..~1000 rows: one wave of workgroups, every microsecond of the tile is a
    // microsecond of the launch): the epilogue's inputs arrive UNDER the K loop instead of as three dependent round trips behind it
    // (position -> cos / sin row -> rotate; cache slot -> store: 15.9 us against 10.95 for the same GEMM with a plain epilogue, T = 512,
    // profiles/r04_mid_batch_gemm.txt).  Positions and slots are requested here, in front of the first operand tile; the cos / sin
    // pieces in K-step 0, once the positions are in (they are older than the tile that step has just waited for).
    constexpr bool PRE = EPI == TEPI_ROPE && MT <= 2;
    constexpr int NWO = (BMt * 16) / 256;                                 // write-out rounds of the epilogue: round i stores row i*16 + tid/16
    int64_t pre_pos[MT];
    int pre_slot[NWO];
    float4_t pre_cs[MT][4], pre_sn[MT][4];
    const int rope_head = PRE ? (int)(blockIdx.x * 8 + wn * 4) / (epi.D / 16) : 0;    // the head of this wave's four n-tiles (16 | D)
    const bool rope_rot = PRE && rope_head < epi.H + epi.KVH;
    if (PRE) {
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            // (as inline asm: a compiler-visible load whose first use sits in the K loop makes hipcc drain vmcnt(0) there, operand tiles in
            //  flight included; the value is complete once K-step 0 has waited for its tile, which is younger — see the pin there)
            const int m = m0 + wm * (16 * MT) + j * 16 + r;
            const int64_t *src = epi.pos + (m < T ? m : T - 1);
            asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(pre_pos[j]) : "v"(src) : "memory");
        }
#pragma unroll
        for (int i = 0; i < NWO; ++i) { const int m = m0 + i * 16 + (int)(threadIdx.x >> 4); pre_slot[i] = (epi.slots && m < T) ? epi.slots[m] : -1; }
    }

    const int KT = (EPI == TEPI_SLAB ? epi.kslice : K) / BK;
    if (NS == 2) {
        stage(0, 0);
        __syncthreads();                                                  // includes s_waitcnt vmcnt(0)
    } else {
#pragma unroll
        for (int st = 0; st < NS - 1; ++st) if (st < KT) stage(st, st * BK);
    }
    for (int kt = 0; kt < KT; ++kt) {
        int cur = kt & 1;
        if (NS == 2) {
            if (kt + 1 < KT) stage(cur ^ 1, (kt + 1) * BK);
        } else {
            // K-tile kt has landed once at most the 4 + MT loads per thread of each younger K-tile in flight are outstanding
            const int younger = min(NS - 2, KT - 1 - kt);
            // (K-step 1 of a rotating TEPI_ROPE wave: the 8 * MT cos / sin requests of K-step 0 are younger than the two tiles that may stay in
            //  flight and may stay in flight with them; later steps find them landed)
            if (PRE && kt == 1 && rope_rot) wait_ring<4 + MT, (PRE ? 8 * MT : 0)>(younger);
            else wait_ring<4 + MT>(younger);
            __builtin_amdgcn_s_barrier();                                 // every thread's pieces are in; buffer (kt-1) % NS is free
            cur = kt % NS;
            if (kt + NS - 1 < KT) stage((kt + NS - 1) % NS, (kt + NS - 1) * BK);
        }
        if (PRE && kt == 0 && rope_rot) {
            const int tph = epi.D / 16, half_d = epi.D / 2;
#pragma unroll
            for (int j = 0; j < MT; ++j) asm volatile("" : "+v"(pre_pos[j]));   // uses stay behind this step's counted wait (volatile asm keeps its order)
#pragma unroll
            for (int j = 0; j < MT; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int jj = ((int)(blockIdx.x * 8 + wn * 4 + i) % tph) * 8 + (q & 1) * 4;
                    pre_cs[j][i] = *reinterpret_cast<const float4_t *>(epi.cos_t + pre_pos[j] * half_d + jj);
                    pre_sn[j][i] = *reinterpret_cast<const float4_t *>(epi.sin_t + pre_pos[j] * half_d + jj);
                }
        }
        const char *a_lds = smem + cur * BUF, *b_lds = a_lds + BN * BK * 2;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8_t a[4], b[MT];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int arow = wn * 64 + i * 16 + r;
                a[i] = *reinterpret_cast<const half8_t *>(a_lds + (arow * 8 + ((ks * 4 + q) ^ (arow & 7))) * 16);
            }
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                const int brow = wm * (16 * MT) + j * 16 + r;
                b[j] = *reinterpret_cast<const half8_t *>(b_lds + (brow * 8 + ((ks * 4 + q) ^ (brow & 7))) * 16);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j)
                    acc[i][j] = mfma16(a[i], b[j], acc[i][j]);
        }
        if (NS == 2) __syncthreads();
    }
    if (NS != 2) __syncthreads();                                         // the epilogue reuses the operand buffers

    if (EPI == TEPI_SLAB) {                                                // f32 partial sums, C layout: 4 consecutive columns of token r
        float *sl = epi.slabs + (int64_t)blockIdx.z * epi.slab_stride;
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int m = m0 + wm * (16 * MT) + j * 16 + r;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = blockIdx.x * BN + wn * 64 + i * 16 + q * 4;
                if (m < T && n < N) *reinterpret_cast<float4_t *>(sl + (int64_t)m * N + n) = acc[i][j];
            }
        }
        return;
    }
    if (EPI == TEPI_LMHEAD) {
        // C layout: row (n) = q*4 + reg, col (token) = r.  f32 logits go out as they are (embed_head.rs:292-306, A-21); the
        // row maxima are taken over the f32 accumulators in increasing column order (ties keep the lowest index, A-12)
        float *sv = reinterpret_cast<float *>(smem);                      // [2 (wn)][BMt tokens]
        int *si = reinterpret_cast<int *>(smem + 2 * BMt * 4);
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int ml = wm * (16 * MT) + j * 16 + r, m = m0 + ml;
            float bv = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = blockIdx.x * BN + wn * 64 + i * 16 + q * 4;
                if (epi.logits && m < T && n < N) *reinterpret_cast<float4_t *>(epi.logits + (int64_t)m * N + n) = acc[i][j];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = acc[i][j][e];
                    if (n + e < N && (v > bv || (v == bv && n + e < bi))) { bv = v; bi = n + e; }
                }
            }
#pragma unroll
            for (int o = 16; o <= 32; o <<= 1) {                          // the four q groups of a token
                const float v = __shfl_xor(bv, o, 64); const int i2 = __shfl_xor(bi, o, 64);
                if (v > bv || (v == bv && i2 < bi)) { bv = v; bi = i2; }
            }
            if (q == 0) { sv[wn * BMt + ml] = bv; si[wn * BMt + ml] = bi; }
        }
        __syncthreads();
        if (threadIdx.x < BMt && m0 + threadIdx.x < T) {
            float bv = sv[threadIdx.x]; int bi = si[threadIdx.x];
            const float v = sv[BMt + threadIdx.x]; const int i2 = si[BMt + threadIdx.x];
            if (v > bv || (v == bv && i2 < bi)) { bv = v; bi = i2; }
            epi.pval[(int64_t)blockIdx.x * T + m0 + threadIdx.x] = bv;
            epi.pidx[(int64_t)blockIdx.x * T + m0 + threadIdx.x] = bi;
        }
        return;
    }
    // epilogue: C layout row (n) = q*4 + reg, col (token) = r.  Each lane produces 4 consecutive output columns of one
    // token; they are staged in LDS (the operand buffers are free now; row stride 272 B keeps ds_read_b128 aligned)
    // and written out as full 16-byte pieces of contiguous rows (a lane-per-row 8-byte store pattern is issue-bound).
    constexpr int OST = 272;                                              // bytes per staged row (128 fp16 + pad)
    constexpr int OUTC = (EPI == TEPI_SILU) ? 64 : 128;                   // output columns of this workgroup
    char *ot = smem;
    auto stash = [&](int ml, int lc, half4_t hv) { *reinterpret_cast<half4_t *>(ot + ml * OST + lc * 2) = hv; };
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int ml = wm * (16 * MT) + j * 16 + r;
        const int m = m0 + ml;
        const int mc = m < T ? m : T - 1;
        if (EPI == TEPI_F16) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                half4_t h = {(half_t)acc[i][j][0], (half_t)acc[i][j][1], (half_t)acc[i][j][2], (half_t)acc[i][j][3]};
                stash(ml, wn * 64 + i * 16 + q * 4, h);
            }
        } else if (EPI == TEPI_SILU) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {                                 // tiles i (gate) and i+2 (up), same 16 columns
                half4_t h;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float gf = (float)to_half_rn(acc[i][j][e]), uf = (float)to_half_rn(acc[i + 2][j][e]);
                    const float sg = sigmoid_fast(gf);
                    h[e] = to_half_rn(__fmul_rn(__fmul_rn(gf, sg), uf));
                }
                stash(ml, wn * 32 + i * 16 + q * 4, h);
            }
        } else {                                                          // TEPI_ROPE
            const int tph = epi.D / 16, half_d = epi.D / 2;
            const int64_t p = PRE ? pre_pos[j] : epi.pos[mc];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int g = blockIdx.x * 8 + wn * 4 + i, head = g / tph, c = g % tph;
                if (head >= epi.H + 2 * epi.KVH) continue;
                float v[4], pv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = (float)to_half_rn(acc[i][j][e]); pv[e] = __shfl_xor(v[e], 32, 64); }
                half4_t h; int col;
                if (head < epi.H + epi.KVH) {
                    const int jj = c * 8 + (q & 1) * 4;
                    const float4_t cs = PRE ? pre_cs[j][i] : *reinterpret_cast<const float4_t *>(epi.cos_t + p * half_d + jj);
                    const float4_t sn = PRE ? pre_sn[j][i] : *reinterpret_cast<const float4_t *>(epi.sin_t + p * half_d + jj);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        h[e] = (q < 2) ? to_half_rn(mul_sub_unfused(v[e], cs[e], pv[e], sn[e]))
                                       : to_half_rn(mul_add_unfused(v[e], cs[e], pv[e], sn[e]));
                    col = (q < 2) ? jj : half_d + jj;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) h[e] = to_half_rn(acc[i][j][e]);
                    col = c * 16 + q * 4;
                }
                stash(ml, head * epi.D + col - blockIdx.x * 128, h);       // the 8 n-tiles cover 128 consecutive columns
            }
        }
    }
    __syncthreads();
    // cooperative write-out: 16-byte pieces, consecutive threads -> consecutive pieces of a row
    const int64_t ldy = (EPI == TEPI_ROPE) ? (int64_t)(epi.H + 2 * epi.KVH) * epi.D : (int64_t)N;
    const int ncols = (EPI == TEPI_ROPE) ? (int)ldy : N;
    constexpr int CPR = OUTC / 8;                                         // pieces per row
#pragma unroll
    for (int i = 0; i < (BMt * CPR) / 256; ++i) {
        const int pidx = i * 256 + threadIdx.x, row = pidx / CPR, ch = pidx % CPR;
        const int m = m0 + row, col = blockIdx.x * OUTC + ch * 8;
        if (m >= T || col >= ncols) continue;
        const half8_t v8 = *reinterpret_cast<const half8_t *>(ot + row * OST + ch * 16);
        *reinterpret_cast<half8_t *>(y + (int64_t)m * ldy + col) = v8;
        if (EPI == TEPI_ROPE) {
            const int head = col / epi.D;
            if (head >= epi.H) {
                const int slot = PRE ? pre_slot[i] : (epi.slots ? epi.slots[m] : -1);
                if (slot >= 0) {
                    const bool is_k = head < epi.H + epi.KVH;
                    const int kvh = is_k ? head - epi.H : head - epi.H - epi.KVH;
                    *reinterpret_cast<half8_t *>((is_k ? epi.kc : epi.vc) + ((int64_t)slot * epi.KVH + kvh) * epi.D + (col - head * epi.D)) = v8;
                }
            }
        }
    }
}

// gate_up + SiluAndMul on 96 W rows (48 gate + 48 up = 48 output columns) x 128 tokens (r05).  At T = 512 the 128 x 128
// tiles above give the gate_up GEMM of Qwen3-0.6B 48 x 4 = 192 workgroups — 64 CUs idle — and a launch of these GEMMs is paced by the operand
// bytes a CU takes in through its L2 -> LDS path (profiles/r04_mid_batch_gemm.txt: ~52 GB/s per CU): 64 x 4 = 256 workgroups of (96 + 128) rows
// x 2 KB = 458 KB each instead of 192 of 524 KB.  Four waves side by side over the tokens (wave w: tokens 32 w .. 32 w + 31, all six n-tiles:
// 12 MFMA tiles per wave and k-step), the four-buffer ring with counted waits of gemm_tiled_kernel<.., 4, ..>, the same K order per output as
// every other tiling of this file (same bits as the 128-row tiles and as the unfused GEMM + SiluAndMul).
constexpr int S96_A = 96 * BK * 2, S96_BUF = S96_A + BM * BK * 2;        // bytes of the A image / of one ring buffer
__global__ __launch_bounds__(256) void gemm_tiled_silu96_kernel(const half_t *__restrict__ x, int64_t ldx, const half_t *__restrict__ W, int T, int K, int I,
                                                                half_t *__restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NS = RING96;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.y * BM;
    const half_t *asrc[3], *bsrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = i * 256 + threadIdx.x, row = idx >> 3, c = (idx & 7) ^ (row & 7);
        if (i < 3) {                                                      // A: local row = 16 t + rr, t = 0..2 gate tiles, 3..5 up tiles of the same columns
            const int t = row >> 4, rr = row & 15;
            const int wr = (t < 3 ? 0 : I) + (int)blockIdx.x * 48 + (t % 3) * 16 + rr;
            asrc[i] = W + (int64_t)wr * K + c * 8;
        }
        int xr = m0 + row; if (xr > T - 1) xr = T - 1;
        bsrc[i] = x + (int64_t)xr * ldx + c * 8;
    }
    auto stage = [&](int buf, int k0) {
        char *a_dst = smem + buf * S96_BUF, *b_dst = a_dst + S96_A;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = (i * 256 + wave * 64) * 16;               // wave-uniform LDS base; hardware adds lane*16
            if (i < 3) __builtin_amdgcn_global_load_lds(asrc[i] + k0, (__attribute__((address_space(3))) void *)(a_dst + piece), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(bsrc[i] + k0, (__attribute__((address_space(3))) void *)(b_dst + piece), 16, 0, 0);
        }
    };
    float4_t acc[6][2];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};
    const int KT = K / BK;
#pragma unroll
    for (int st = 0; st < NS - 1; ++st) if (st < KT) stage(st, st * BK);
    for (int kt = 0; kt < KT; ++kt) {
        // K-tile kt has landed once at most the 7 loads per thread of each younger K-tile in flight are outstanding
        const int younger = min(NS - 2, KT - 1 - kt);
        wait_ring<7>(younger);
        __builtin_amdgcn_s_barrier();                                     // every thread's pieces are in; buffer (kt-1) % NS is free
        const int cur = kt % NS;
        if (kt + NS - 1 < KT) stage((kt + NS - 1) % NS, (kt + NS - 1) * BK);
        const char *a_lds = smem + cur * S96_BUF, *b_lds = a_lds + S96_A;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8_t a[6], b[2];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int arow = i * 16 + r;
                a[i] = *reinterpret_cast<const half8_t *>(a_lds + (arow * 8 + ((ks * 4 + q) ^ (arow & 7))) * 16);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int brow = wave * 32 + j * 16 + r;
                b[j] = *reinterpret_cast<const half8_t *>(b_lds + (brow * 8 + ((ks * 4 + q) ^ (brow & 7))) * 16);
            }
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma16(a[i], b[j], acc[i][j]);
        }
    }
    __syncthreads();                                                      // the epilogue reuses the operand buffers
    // act = fp16(silu(fp16 g) * fp16 u) (activation.rs:46-63; the rounding points of the unfused graph), staged in LDS and written as 16-byte pieces
    constexpr int OST = 112;                                              // bytes per staged row (48 fp16 + pad)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ml = wave * 32 + j * 16 + r;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            half4_t h;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gf = (float)to_half_rn(acc[c][j][e]), uf = (float)to_half_rn(acc[c + 3][j][e]);
                const float sg = sigmoid_fast(gf);
                h[e] = to_half_rn(__fmul_rn(__fmul_rn(gf, sg), uf));
            }
            *reinterpret_cast<half4_t *>(smem + ml * OST + (c * 16 + q * 4) * 2) = h;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < (BM * 6) / 256; ++i) {                            // 6 pieces per row
        const int pidx = i * 256 + threadIdx.x, row = pidx / 6, ch = pidx % 6;
        const int m = m0 + row;
        if (m >= T) continue;
        *reinterpret_cast<half8_t *>(y + (int64_t)m * I + blockIdx.x * 48 + ch * 8) = *reinterpret_cast<const half8_t *>(smem + row * OST + ch * 16);
    }
}

static int tiled_check(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return nvr::fail(NVR_ERR_HIP, "%s launch failed: %s", what, hipGetErrorString(e));
    return 0;
}

// ring (NS = 4) when the grid leaves CUs idle anyway
static bool tiled_ring(unsigned tiles) { return tiles < 256; }
// 64- or 32-token tiles when 128-token tiles would leave CUs without a workgroup: the largest tile that reaches ~192 workgroups — unless that count
// needs a second, mostly empty round: with the 4-buffer ring a 64-token workgroup is alone on its CU (96 KB of LDS), so 257..511 of them run as one full
// round + a tail while half as many tiles of twice the height finish in one (scratch/route_scan.py: Qwen3-0.6B bs 384 -> 385 got FASTER, 3.87 -> 3.67
// ms/step; r06: bs 321..384 -3..5 %, bs 513 -4 %).  Returns the m-tiles per wave (4, 2, 1).
static int tiled_mt(int64_t T, int64_t nx_nz) {
    int mt = 1;
    if (T > 64 && nx_nz * ((T + 127) / 128) >= 192) mt = 4;       // (192 workgroups of 128 tokens beat 384 of 64: 15.6 vs 19.3 us, gate_up at T = 512)
    else if (nx_nz * ((T + 63) / 64) >= 192) mt = 2;              // (33..64 rows never take a 128-token tile: half of it would be masked rows)
    if (mt == 2 && T > 64) {
        const int64_t wgs = nx_nz * ((T + 63) / 64), taller = nx_nz * ((T + 127) / 128);
        if (wgs > 256 && wgs < 512 && taller >= 128) mt = 4;      // (< 512: NVR_TILED_LAUNCH gives them the ring, one workgroup per CU)
    }
    return mt;
}
constexpr size_t kStageBytes = 2 * BM * BK * 2;                          // MT = 4
constexpr size_t kStageBytes64 = (BN + 64) * BK * 2;                     // MT = 2
constexpr size_t kStageBytes32 = (BN + 32) * BK * 2;                     // MT = 1
// grid: x = column tiles, y is filled in here (token tiles of 128, 64 or 32), z = k slices; the 4-buffer ring whenever a
// workgroup is alone on its CU (the smaller tiles fit two workgroups per CU even with the ring)
#define NVR_TILED_LAUNCH(EPI_, grid, T_, ...)                                                                      \
    do {                                                                                                           \
        const int mt_ = tiled_mt((T_), (int64_t)(grid).x * (grid).z);                                              \
        (grid).y = (unsigned)(((T_) + 32 * mt_ - 1) / (32 * mt_));                                                 \
        const unsigned wgs_ = (grid).x * (grid).y * (grid).z;                                                      \
        if (mt_ == 1) {                                                                                            \
            if (tiled_ring(wgs_ / 2)) gemm_tiled_kernel<EPI_, RING32, 1><<<grid, dim3(256), RING32 * kStageBytes32, s>>>(__VA_ARGS__); \
            else gemm_tiled_kernel<EPI_, 2, 1><<<grid, dim3(256), 2 * kStageBytes32, s>>>(__VA_ARGS__);           \
        } else if (mt_ == 2) {                                                                                     \
            if (tiled_ring(wgs_ / 2)) gemm_tiled_kernel<EPI_, RING64, 2><<<grid, dim3(256), RING64 * kStageBytes64, s>>>(__VA_ARGS__); \
            else gemm_tiled_kernel<EPI_, 2, 2><<<grid, dim3(256), 2 * kStageBytes64, s>>>(__VA_ARGS__);           \
        } else {                                                                                                   \
            if (tiled_ring(wgs_)) gemm_tiled_kernel<EPI_, 4><<<grid, dim3(256), 4 * kStageBytes, s>>>(__VA_ARGS__); \
            else gemm_tiled_kernel<EPI_, 2><<<grid, dim3(256), 2 * kStageBytes, s>>>(__VA_ARGS__);                 \
        }                                                                                                          \
    } while (0)
// > 64 KiB of dynamic LDS needs an opt-in per kernel instance; called at runner creation (never inside a stream capture)
int gemm_tiled_prepare() {
    static bool done = false;
    if (done) return 0;
#define NVR_TILED_ATTR(EPI_)                                                                                          \
    { hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_tiled_kernel<EPI_, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                         (int)(4 * kStageBytes));                                                    \
      if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_tiled_kernel<EPI_, RING64, 2>),   \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)(RING64 * kStageBytes64)); \
      if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_tiled_kernel<EPI_, RING32, 1>),   \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)(RING32 * kStageBytes32)); \
      if (e != hipSuccess) return nvr::fail(NVR_ERR_HIP, "gemm_tiled: LDS opt-in failed: %s", hipGetErrorString(e)); }
    NVR_TILED_ATTR(TEPI_F16) NVR_TILED_ATTR(TEPI_SILU) NVR_TILED_ATTR(TEPI_ROPE) NVR_TILED_ATTR(TEPI_LMHEAD) NVR_TILED_ATTR(TEPI_SLAB)
#undef NVR_TILED_ATTR
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_tiled_silu96_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, RING96 * S96_BUF) != hipSuccess)
        return nvr::fail(NVR_ERR_HIP, "gemm_tiled: LDS opt-in failed (96-row SiLU tiles)");
    done = true;
    return 0;
}

static constexpr bool tiled_enabled() { return true; }
// LM head for more than 32 rows (large decode batches, many-sequence prefills): weights streamed once per 128-row block
bool gemm_tiled_lm_head_ok(int64_t T, int64_t K, int64_t N, int64_t ldx) {
    return tiled_enabled() && T > 32 && T <= 65536 && K % BK == 0 && N % 16 == 0 && ldx % 8 == 0 && (N + BN - 1) / BN <= LM_HEAD_MAX_PARTS &&
           N < (1ll << 31);
}
int gemm_tiled_lm_head(const half_bits *x, int64_t ldx, const half_bits *W, int64_t T, int64_t K, int64_t N, float *logits,
                       float *part_val, int32_t *part_idx, int32_t *nparts, hipStream_t s) {
    if (!gemm_tiled_lm_head_ok(T, K, N, ldx)) return nvr::fail(NVR_ERR_UNSUPPORTED, "gemm_tiled_lm_head: T=%ld K=%ld N=%ld", (long)T, (long)K, (long)N);
    TileEpi e{};
    e.logits = logits; e.pval = part_val; e.pidx = part_idx;
    dim3 grid((unsigned)((N + BN - 1) / BN), (unsigned)((T + BM - 1) / BM));
    *nparts = (int32_t)grid.x;
    if (int rc = gemm_tiled_prepare()) return rc;
    NVR_TILED_LAUNCH(TEPI_LMHEAD, grid, T, (const half_t *)x, ldx, (const half_t *)W, (int)T, (int)K, (int)N, (int)N, nullptr, e);
    return tiled_check("gemm_tiled_lm_head");
}

// split-k form for the narrow row-parallel GEMMs (o_proj / down_proj) at 65..1024 rows: slabs[z][T][N] f32, S slices of K
bool gemm_tiled_splitk_ok(int64_t T, int64_t K, int64_t N, int64_t S, int64_t ldx) {
    return tiled_enabled() && T > 32 && S >= 1 && K % (S * BK) == 0 && N % 16 == 0 && ldx % 8 == 0;
}
int gemm_tiled_splitk(const half_bits *x, int64_t ldx, const half_bits *W, int64_t T, int64_t K, int64_t N, int64_t S, float *slabs, hipStream_t s) {
    if (!gemm_tiled_splitk_ok(T, K, N, S, ldx)) return nvr::fail(NVR_ERR_UNSUPPORTED, "gemm_tiled_splitk: T=%ld K=%ld N=%ld S=%ld", (long)T, (long)K, (long)N, (long)S);
    if (int rc = gemm_tiled_prepare()) return rc;
    TileEpi e{};
    e.slabs = slabs; e.slab_stride = T * N; e.kslice = (int32_t)(K / S);
    dim3 grid((unsigned)((N + BN - 1) / BN), (unsigned)((T + BM - 1) / BM), (unsigned)S);
    NVR_TILED_LAUNCH(TEPI_SLAB, grid, T, (const half_t *)x, ldx, (const half_t *)W, (int)T, (int)K, (int)N, (int)N, nullptr, e);
    return tiled_check("gemm_tiled_splitk");
}

bool gemm_tiled_ok(int64_t T, int64_t K, int64_t N, int64_t ldx) { return tiled_enabled() && T > 32 && K % BK == 0 && N % 16 == 0 && ldx % 8 == 0; }

int gemm_tiled(const half_bits *x, int64_t ldx, const half_bits *W, int64_t T, int64_t K, int64_t N, half_bits *y, hipStream_t s) {
    if (!gemm_tiled_ok(T, K, N, ldx)) return nvr::fail(NVR_ERR_UNSUPPORTED, "gemm_tiled: T=%ld K=%ld N=%ld", (long)T, (long)K, (long)N);
    dim3 grid((unsigned)((N + BN - 1) / BN), (unsigned)((T + BM - 1) / BM));
    if (int rc = gemm_tiled_prepare()) return rc;
    NVR_TILED_LAUNCH(TEPI_F16, grid, T, (const half_t *)x, ldx, (const half_t *)W, (int)T, (int)K, (int)N, (int)N, (half_t *)y, TileEpi{});
    return tiled_check("gemm_tiled");
}
int gemm_tiled_silu_mul(const half_bits *x, int64_t ldx, const half_bits *W, int64_t T, int64_t K, int64_t I, half_bits *out, hipStream_t s) {
    if (!gemm_tiled_ok(T, K, I, ldx) || I % 64) return nvr::fail(NVR_ERR_UNSUPPORTED, "gemm_tiled_silu_mul: T=%ld K=%ld I=%ld", (long)T, (long)K, (long)I);
    dim3 grid((unsigned)(I / 64), (unsigned)((T + BM - 1) / BM));
    if (int rc = gemm_tiled_prepare()) return rc;
    {   // 96-row tiles when they put a workgroup on (nearly) every CU where the 128-row tiles leave a quarter of them idle (T = 512, I = 3072: 256 against 192;
        // T = 321..384: 192 against 144)
        const int64_t ty = (T + BM - 1) / BM, w128 = (I / 64) * ty, w96 = (I / 48) * ty;
        if (I % 48 == 0 && tiled_mt(T, I / 64) == 4 && tiled_ring((unsigned)w128) && w96 <= 256 && w96 > w128 && w128 * 8 <= w96 * 7) {
            gemm_tiled_silu96_kernel<<<dim3((unsigned)(I / 48), (unsigned)ty), dim3(256), RING96 * S96_BUF, s>>>((const half_t *)x, ldx, (const half_t *)W, (int)T, (int)K, (int)I, (half_t *)out);
            return tiled_check("gemm_tiled_silu_mul (96-row tiles)");
        }
    }
    NVR_TILED_LAUNCH(TEPI_SILU, grid, T, (const half_t *)x, ldx, (const half_t *)W, (int)T, (int)K, (int)I, (int)(2 * I), (half_t *)out, TileEpi{});
    return tiled_check("gemm_tiled_silu_mul");
}
int gemm_tiled_qkv_rope_store(const half_bits *x, int64_t ldx, const half_bits *W, int64_t T, int64_t K, int64_t H, int64_t KVH, int64_t D,
                              const int64_t *positions, const int32_t *slots, const float *cos_t, const float *sin_t, half_bits *qkv,
                              half_bits *k_cache, half_bits *v_cache, hipStream_t s) {
    const int64_t N = (H + 2 * KVH) * D;
    if (!gemm_tiled_ok(T, K, N, ldx) || D % 16 || 128 % D) return nvr::fail(NVR_ERR_UNSUPPORTED, "gemm_tiled_qkv_rope_store: T=%ld K=%ld D=%ld", (long)T, (long)K, (long)D);
    TileEpi e{};
    e.pos = positions; e.slots = slots; e.cos_t = cos_t; e.sin_t = sin_t; e.kc = (half_t *)k_cache; e.vc = (half_t *)v_cache;
    e.H = (int32_t)H; e.KVH = (int32_t)KVH; e.D = (int32_t)D;
    dim3 grid((unsigned)((N / 16 + 7) / 8), (unsigned)((T + BM - 1) / BM));
    if (int rc = gemm_tiled_prepare()) return rc;
    NVR_TILED_LAUNCH(TEPI_ROPE, grid, T, (const half_t *)x, ldx, (const half_t *)W, (int)T, (int)K, (int)N, (int)N, (half_t *)qkv, e);
    return tiled_check("gemm_tiled_qkv_rope_store");
}

}}  // namespace nvr::k / nvr::kb (NVR_DT_NS)
