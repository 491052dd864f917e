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
template <int NTT, int MT, int WAVES, int U, int EPI, int TMAX, int PRE = 0>
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

    float4_t acc[TMAX][NTT][MT];
#pragma unroll
    for (int i = 0; i < TMAX; ++i)
#pragma unroll
        for (int nt = 0; nt < NTT; ++nt)
#pragma unroll
            for (int j = 0; j < MT; ++j) acc[i][nt][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

    for (int kc0 = 0; kc0 < K; kc0 += KC) {
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
        for (int i = 0; i < TMAX; ++i) {
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
    constexpr int ET = PRE > 0 ? PRE : TMAX;                                 // tiles a workgroup can hold
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
    for (int i = 0; i < TMAX; ++i) {
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
static bool stream_shape_ok(int64_t T, int64_t K, int64_t tiles, int64_t weight_bytes, int64_t ldx) {
    if (!stream_enabled() || T < 1 || T > 32 || ldx % 8 || K < 2048 || weight_bytes < (24ll << 20)) return false;
    const int kc = stream_kc(T, K);
    return K % kc == 0 && kc % (S_WAVES * 32) == 0 && tiles >= 320 && tiles <= 256 * S_TMAX;
}
bool linear_stream_ok(int64_t T, int64_t K, int64_t N, int64_t ldx) { return N % 16 == 0 && stream_shape_ok(T, K, N / 16, N * K * 2, ldx); }
bool linear_stream_silu_ok(int64_t T, int64_t K, int64_t I, int64_t ldx) { return I % 16 == 0 && stream_shape_ok(T, K, I / 16, 2 * I * K * 2, ldx); }
bool linear_stream_rope_ok(int64_t T, int64_t K, int64_t H, int64_t KVH, int64_t D, int64_t ldx) {
    return D % 16 == 0 && stream_shape_ok(T, K, (H + 2 * KVH) * D / 16, (H + 2 * KVH) * D * K * 2, ldx);
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

// opt every instance in to > 64 KiB of dynamic LDS up front (runner init: never inside a stream capture)
int linear_stream_prepare() {
    const void *fns[] = {
        reinterpret_cast<const void *>(&linear_stream_kernel<1, 1, S_WAVES, 4, SEPI_F16, S_TMAX>), reinterpret_cast<const void *>(&linear_stream_kernel<1, 2, S_WAVES, 4, SEPI_F16, S_TMAX>),
        reinterpret_cast<const void *>(&linear_stream_kernel<2, 1, S_WAVES, 2, SEPI_SILU, S_TMAX>), reinterpret_cast<const void *>(&linear_stream_kernel<2, 2, S_WAVES, 2, SEPI_SILU, S_TMAX>),
        reinterpret_cast<const void *>(&linear_stream_kernel<1, 1, S_WAVES, 4, SEPI_ROPE, S_TMAX>), reinterpret_cast<const void *>(&linear_stream_kernel<1, 2, S_WAVES, 4, SEPI_ROPE, S_TMAX>),
        reinterpret_cast<const void *>(&linear_stream_kernel<1, 1, S_WAVES, 4, SEPI_F16, S_TMAX, 2>), reinterpret_cast<const void *>(&linear_stream_kernel<1, 2, S_WAVES, 4, SEPI_F16, S_TMAX, 2>),
        reinterpret_cast<const void *>(&linear_stream_kernel<1, 1, S_WAVES, 4, SEPI_ROPE, S_TMAX, 2>), reinterpret_cast<const void *>(&linear_stream_kernel<1, 2, S_WAVES, 4, SEPI_ROPE, S_TMAX, 2>)};
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
    return stream_launch<1, 2, 4, SEPI_F16>((const half_t *)x, ldx, (const half_t *)W, (int)T, (int)K, (int)N, (int)(N / 16), (half_t *)y, e, s);
}
int linear_stream_silu_mul(const half_bits *x, int64_t ldx, const half_bits *W, int64_t T, int64_t K, int64_t I, half_bits *out, hipStream_t s,
                           const half_bits *Wt) {
    if (!linear_stream_silu_ok(T, K, I, ldx)) return nvr::fail(NVR_ERR_UNSUPPORTED, "linear_stream_silu_mul: T=%ld K=%ld I=%ld", (long)T, (long)K, (long)I);
    StreamEpi e{};
    if (Wt) { W = Wt; e.tiled = 1; }
    if (T <= 16) return stream_launch<2, 1, 2, SEPI_SILU>((const half_t *)x, ldx, (const half_t *)W, (int)T, (int)K, (int)I, (int)(I / 16), (half_t *)out, e, s);
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
    return stream_launch<1, 2, 4, SEPI_ROPE>((const half_t *)x, ldx, (const half_t *)W, (int)T, (int)K, N, N / 16, (half_t *)qkv, e, s);
}

}}  // namespace nvr::k / nvr::kb (NVR_DT_NS)
