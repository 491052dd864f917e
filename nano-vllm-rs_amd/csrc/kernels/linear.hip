// linear.hip — y[T,N] = x[T,K] · W[N,K]^T on MFMA (K3/K10/K12/K14/K16), with the two epilogue fusions of
// the decode path: gate_up -> SiluAndMul (K12+K13) and qkv -> RoPE + KV store (K3+K4+K5+K6).
// reference call sites: QKVParallelLinear::forward src/layers/linear.rs:354-356 (+split_qkv :331-340),
// RowParallelLinear :228-239, MergedColumnParallelLinear :437-439 (+SiluAndMul activation.rs:46-63),
// ParallelLMHead::compute_logits src/layers/embed_head.rs:292-306; all `candle_nn::Linear::forward`,
// x·Wᵀ with W [out, in]; RoPE src/layers/rotary_embedding.rs:23-48; store_kv_cache attention.rs:150-174.
//
// Weight-streaming kernel for the decode regime (T <= 64: every weight byte is read once from HBM
// and the kernel is HBM/latency-bound).  Roofline: algorithmic bytes = 2·N·K (weights)
// + 2·T·K (x) + out; MFMA is used only because T=32 tokens x 8 k per 16-byte weight load exceeds
// the VALU rate at HBM speed (cdna_hip_programming.md §5 "GEMV / M <= 16" row: weights straight to
// VGPRs, no LDS round trip).
//
// Tiling: v_mfma_f32_16x16x32_f16 with A = W tile (16 output columns n x 32 k: each lane loads 16
// contiguous bytes of one W row, 4 lanes cover 64 B of the row) and B = x tile (16 tokens x 32 k,
// L2 resident).  A workgroup = WAVES waves owns NT*16 output columns for MT*16 tokens; its waves
// interleave 32-wide k steps (together they read 64·WAVES contiguous bytes per W row per step, all
// k-steps of a small GEMM in flight at once) and reduce their f32 partials through LDS.
// Grid = (N/(16 NT), ceil(T/(16 MT))): for large T the same kernel re-streams W from L2 per 16·MT-token
// slab (correct, not yet the tiled prefill GEMM).
//
// Rounding points mirror the oracle (oracle/model_oracle.py): the f32 accumulator is rounded to fp16
// where the unfused graph stores an fp16 tensor (qkv, gate_up), then the epilogue math runs in f32
// without FMA contraction and is rounded again.
#include <cstdlib>
#include "kernels.h"
#include "device_utils.h"
#include "../common.h"

namespace nvr { namespace NVR_DT_NS {

enum { EPI_F16 = 0, EPI_F32 = 1, EPI_SILU = 2, EPI_ROPE = 3, EPI_SLAB = 4 };

struct LinEpi {
    // EPI_SILU: N is the intermediate size I; W holds gate rows [0,I) and up rows [I,2I)
    // EPI_ROPE: qkv output [T, (H+2KVH)*D] + caches
    const int64_t *pos; const int32_t *slots; const float *cos_t, *sin_t;
    half_t *kc, *vc;
    int32_t H, KVH, D;
    // EPI_SLAB: blockIdx.z owns k in [z*kslice, (z+1)*kslice) and writes its f32 partial tile to slab z
    int32_t kslice; int64_t slab_stride;
    // W is the TILED copy [N/16][K/32][16][32] (retile_weight): one wave-instruction reads 1 KiB contiguous
    int32_t tiled;
};

// W row handled by local row r (0..15) of n-tile i of workgroup bx
template <int EPI>
__device__ __forceinline__ int w_row(int bx, int i, int NT, int r, int N, const LinEpi &e) {
    if (EPI == EPI_SILU) return (i == 0 ? 0 : N) + bx * 16 + r;             // N == I here
    if (EPI == EPI_ROPE) {
        const int tph = e.D / 16, head = bx / tph, c = bx % tph;            // tiles per head
        if (head < e.H + e.KVH) return head * e.D + (r < 8 ? c * 8 + r : e.D / 2 + c * 8 + (r - 8));
        return head * e.D + c * 16 + r;
    }
    int n = (bx * NT + i) * 16 + r;
    return n < N ? n : N - 1;
}

template <int NT, int MT, int WAVES, int EPI>
__global__ __launch_bounds__(WAVES * 64) void linear_skinny_kernel(const half_t *__restrict__ x, int64_t ldx,
                                                                   const half_t *__restrict__ W, int T, int K, int N,
                                                                   void *__restrict__ y, LinEpi epi) {
    constexpr int U = 4;                                    // k-steps in flight per wave
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.y * (16 * MT);

    const half_t *wrow[NT];
    const half_t *xrow[MT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        if (epi.tiled) {
            // tile index in the tiled copy: SiLU — gate tile bx, up tile I/16 + bx; RoPE — tile bx of the permuted row order;
            // else tile bx*NT + i (clamped like w_row clamps its rows)
            int64_t tile = EPI == EPI_SILU ? (i == 0 ? (int64_t)blockIdx.x : (int64_t)(N / 16) + blockIdx.x) : (int64_t)blockIdx.x * NT + i;
            if (EPI != EPI_SILU && EPI != EPI_ROPE && tile > N / 16 - 1) tile = N / 16 - 1;
            wrow[i] = W + tile * (int64_t)(K / 32) * 512 + r * 32 + q * 8;
        } else wrow[i] = W + (int64_t)w_row<EPI>(blockIdx.x, i, NT, r, N, epi) * K + q * 8;
    }
    const int kmul = epi.tiled ? 16 : 1;                            // k -> element offset: (k / 32) * 512 in the tiled copy
#pragma unroll
    for (int i = 0; i < MT; ++i) { int m = m0 + i * 16 + r; if (m > T - 1) m = T - 1; xrow[i] = x + (int64_t)m * ldx + q * 8; }

    float4_t acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

    // EPI_ROPE: the epilogue of token tile j runs on wave j; its position / slot / cos / sin loads are issued here, ahead of the
    // weight stream, instead of as a chain of dependent global loads (pos -> cos,sin) after the reduction barrier
    float4_t rope_cs = (float4_t){0.f, 0.f, 0.f, 0.f}, rope_sn = rope_cs;
    int rope_slot = -1;
    if (EPI == EPI_ROPE && wave < MT) {
        const int m = m0 + wave * 16 + r, mc = m < T ? m : T - 1;
        const int tph = epi.D / 16, head = blockIdx.x / tph, c = blockIdx.x % tph, half_d = epi.D / 2;
        if (head < epi.H + epi.KVH) {
            const int jj = c * 8 + (q & 1) * 4;
            const int64_t p = epi.pos[mc];
            rope_cs = *reinterpret_cast<const float4_t *>(epi.cos_t + p * half_d + jj);
            rope_sn = *reinterpret_cast<const float4_t *>(epi.sin_t + p * half_d + jj);
        }
        if (head >= epi.H && epi.slots && m < T) rope_slot = epi.slots[m];
    }
    constexpr int KS = 32 * WAVES;
    constexpr bool SLABBED = EPI == EPI_SLAB;
    const int kbeg = SLABBED ? blockIdx.z * epi.kslice : 0;
    if (SLABBED) K = min(K, kbeg + epi.kslice);
    for (int k = kbeg + wave * 32; k < K; k += KS * U) {
        half8_t a[U][NT], b[U][MT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kk = k + u * KS;
            if (kk < K) {
#pragma unroll
                for (int i = 0; i < NT; ++i) a[u][i] = __builtin_nontemporal_load(reinterpret_cast<const half8_t *>(wrow[i] + (int64_t)kk * kmul));
#pragma unroll
                for (int j = 0; j < MT; ++j) b[u][j] = *reinterpret_cast<const half8_t *>(xrow[j] + kk);
            } else {
#pragma unroll
                for (int i = 0; i < NT; ++i) a[u][i] = (half8_t)(half_t)0;
#pragma unroll
                for (int j = 0; j < MT; ++j) b[u][j] = (half8_t)(half_t)0;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j)
                    acc[i][j] = mfma16(a[u][i], b[u][j], acc[i][j]);
    }

    // cross-wave (split-k) reduction through LDS: part[wave][tile][lane] as float4
    __shared__ float4_t part[WAVES][NT * MT][64];
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
    if (EPI == EPI_F16 || EPI == EPI_F32 || EPI == EPI_SLAB) {
        for (int tile = wave; tile < NT * MT; tile += WAVES) {
            const float4_t s = reduce(tile);
            const int i = tile / MT, j = tile % MT;
            const int n = (blockIdx.x * NT + i) * 16 + q * 4, m = m0 + j * 16 + r;
            if (m < T && n < N) {
                if (EPI == EPI_F32) {
                    *reinterpret_cast<float4_t *>(reinterpret_cast<float *>(y) + (int64_t)m * N + n) = s;
                } else if (EPI == EPI_SLAB) {
                    *reinterpret_cast<float4_t *>(reinterpret_cast<float *>(y) + blockIdx.z * epi.slab_stride + (int64_t)m * N + n) = s;
                } else {
                    half4_t h = {(half_t)s[0], (half_t)s[1], (half_t)s[2], (half_t)s[3]};
                    *reinterpret_cast<half4_t *>(reinterpret_cast<half_t *>(y) + (int64_t)m * N + n) = h;
                }
            }
        }
    } else if (EPI == EPI_SILU) {
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
                *reinterpret_cast<half4_t *>(reinterpret_cast<half_t *>(y) + (int64_t)m * N + n) = h;
            }
        }
    } else {   // EPI_ROPE: NT == 1
        const int tph = epi.D / 16, head = blockIdx.x / tph, c = blockIdx.x % tph, half_d = epi.D / 2;
        const int64_t ldq = (int64_t)(epi.H + 2 * epi.KVH) * epi.D;
        for (int j = wave; j < MT; j += WAVES) {
            const float4_t s = reduce(j);
            const int m = m0 + j * 16 + r;
            const int mc = m < T ? m : T - 1;
            float v[4], pv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = (float)to_half_rn(s[e]); pv[e] = __shfl_xor(v[e], 32, 64); }
            half4_t h;
            int col;                                           // first of the lane's 4 consecutive head columns
            if (head < epi.H + epi.KVH) {
                const int jj = c * 8 + (q & 1) * 4;            // index inside the half dimension
                float4_t cs = rope_cs, sn = rope_sn;
                if (MT > WAVES) {                               // never instantiated: every token tile has its own wave
                    cs = *reinterpret_cast<const float4_t *>(epi.cos_t + epi.pos[mc] * half_d + jj);
                    sn = *reinterpret_cast<const float4_t *>(epi.sin_t + epi.pos[mc] * half_d + jj);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // rotary_embedding.rs:36-44: out1 = x1*c - x2*s ; out2 = x2*c + x1*s
                    h[e] = (q < 2) ? to_half_rn(mul_sub_unfused(v[e], cs[e], pv[e], sn[e]))
                                   : to_half_rn(mul_add_unfused(v[e], cs[e], pv[e], sn[e]));
                }
                col = (q < 2) ? jj : half_d + jj;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = to_half_rn(s[e]);
                col = c * 16 + q * 4;
            }
            if (m < T) {
                *reinterpret_cast<half4_t *>(reinterpret_cast<half_t *>(y) + (int64_t)m * ldq + head * epi.D + col) = h;
                const int slot = MT > WAVES ? (epi.slots ? epi.slots[m] : -1) : rope_slot;
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

template <int NT, int MT, int WAVES, int EPI>
static void launch(const half_t *x, int64_t ldx, const half_t *W, int T, int K, int N, void *y, const LinEpi &e,
                   unsigned gx, hipStream_t s, unsigned gz = 1) {
    dim3 grid(gx, (unsigned)((T + 16 * MT - 1) / (16 * MT)), gz);
    linear_skinny_kernel<NT, MT, WAVES, EPI><<<grid, dim3(WAVES * 64), 0, s>>>(x, ldx, W, T, K, N, y, e);
}

// 256x256 kernel or 128x128 kernel?  Fitted to the sweep (us, profiles/r01_gemm_ablation.txt): a wave of <= 256 tiles costs
// 17*K/1024 + 20 on the 256x256 kernel (long single-tile latency); the 128x128 kernel costs 15*K/1024 while its tiles fit one per
// CU and 21*K/1024 per wave of 512 (two workgroups per CU) beyond that, + 5 per launch.  N = W rows
static inline bool prefer_256(int64_t T, int64_t K, int64_t N) {
    const int64_t t256 = ((N + 255) / 256) * ((T + 255) / 256), t128 = ((N + 127) / 128) * ((T + 127) / 128);
    const double kk = (double)K / 1024.0;
    const double c256 = (double)((t256 + 255) / 256) * (17.0 * kk + 20.0);
    const double c128 = (t128 <= 256 ? 15.0 * kk : (double)((t128 + 511) / 512) * 21.0 * kk) + 5.0;
    return c256 < c128;
}
// one rule for the plain and the fused launchers (N = output features), so that a fused GEMM and its unfused twin run the same kernel
bool gemm256_preferred(int64_t T, int64_t K, int64_t N, int64_t ldx) { return gemm256_ok(T, K, N, ldx) && prefer_256(T, K, N); }
int64_t stream_row_limit() { return 64; }
// split-k GEMMs of the N = hidden projections: LDS tiles beyond the streaming kernels' rows — and from 33 rows on over large weights, where the
// streaming split-k kernel walks the weights once per 32-row block (Qwen3-8B bs 64: o + down 92 us per layer against 46 us at bs 128 on the tiles;
// up to 32 rows the streaming kernel stays ahead: Qwen3-8B bs 32 / 16 5.00 / 4.20 ms per step against 5.10 / 4.38 on 32-token tiles, 8 slices)
bool splitk_prefers_tiles(int64_t T, int64_t K, int64_t N) { return T > stream_row_limit() || (T > 32 && N * K * 2 >= (24ll << 20)); }
// k-slices of the N = hidden GEMMs of a decode-sized step (linear_splitk): reach ~256 workgroups with slices of >= 256 columns that are
// multiples of 64
int decode_splitk_slices(int64_t T, int64_t K, int64_t N) {
    int64_t S = 1;
    const int64_t tiles = (N / 16) * ((T + 31) / 32);
    while (S < 4 && tiles * S < 256 && K % (32 * S * 2) == 0 && K / (S * 2) >= 128) S *= 2;
    return (int)S;
}
static inline bool prefer_stream(int64_t T, int64_t N) {
    // up to 64 rows the weight-streaming kernel; beyond, the LDS-tiled kernel with 32- / 64-token tiles and the 4-buffer ring (r02,
    // ctx 256: bs 66 2.20 -> 1.83 ms/step, bs 96 2.48 -> 2.15, bs 128 2.82 -> 2.23).
    return T <= stream_row_limit() || T * N <= 384 * 1024;
}
static inline int waves_for(int64_t K) { return K >= 2048 ? 16 : (K >= 1024 ? 8 : 4); }

static int launch_check(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return nvr::fail(NVR_ERR_HIP, "%s launch failed: %s", what, hipGetErrorString(e));
    return 0;
}

int linear(const half_bits *x, int64_t ldx, const half_bits *W, int64_t T, int64_t K, int64_t N, void *y,
           bool y_f32, hipStream_t s, const half_bits *Wt) {
    if (K % 32 || N % 16 || ldx % 8)
        return nvr::fail(NVR_ERR_UNSUPPORTED, "linear: K=%ld must be a multiple of 32, N=%ld of 16, ldx=%ld of 8",
                         (long)K, (long)N, (long)ldx);
    if (T == 0) return 0;
    // Routing by measured crossovers (scratch/gemm_route_sweep.py, profiles/r01_gemm_ablation.txt): prefer_256 above; the 128x128 kernel's
    // own single-tile latency (21-31 us) loses to the weight-streaming kernel run over 32-token blocks while T or T*N is small
    if (!y_f32 && gemm256_ok(T, K, N, ldx) && prefer_256(T, K, N)) return gemm256(x, ldx, W, T, K, N, (half_bits *)y, s);   // prefill regime
    if (!y_f32 && gemm_tiled_ok(T, K, N, ldx) && !prefer_stream(T, N)) return gemm_tiled(x, ldx, W, T, K, N, (half_bits *)y, s);
    if (!y_f32 && linear_stream_ok(T, K, N, ldx)) return linear_stream(x, ldx, W, T, K, N, (half_bits *)y, s, Wt);   // large weights
    const half_t *xx = (const half_t *)x, *ww = (const half_t *)(Wt ? Wt : W);
    const int t = (int)T, k = (int)K, n = (int)N;
    LinEpi e{};
    e.tiled = Wt != nullptr;
    const int64_t mtiles = (T + 31) / 32;
    // wide GEMMs (LM head, prefill slabs): 64 columns per workgroup amortise the x fragments; narrow ones keep
    // 16 columns per workgroup (more workgroups) and split k over more waves (more bytes in flight per CU)
    const bool wide = (N / 64) * (T <= 16 ? 1 : mtiles) >= 1024;
    const int wv = waves_for(K);
#define NVR_LIN(NT_, MT_, WV_)                                                                      \
    do {                                                                                            \
        const unsigned gx = (unsigned)((N + 16 * NT_ - 1) / (16 * NT_));                            \
        if (y_f32) launch<NT_, MT_, WV_, EPI_F32>(xx, ldx, ww, t, k, n, y, e, gx, s);               \
        else launch<NT_, MT_, WV_, EPI_F16>(xx, ldx, ww, t, k, n, y, e, gx, s);                     \
    } while (0)
    // narrow outputs (N/16 < 128 column tiles: the row-parallel GEMMs of a tensor-parallel rank) with 17..32 tokens: 16-token
    // workgroups double the workgroup count (the W tile of a column block is read by both token blocks from the same XCD's L2):
    // 6.1 vs 8.2 us at K = 2048, N = 1024 (scratch/gemm_chain.py)
    const bool mt1 = T <= 16 || (!wide && T <= 32 && N / 16 < 128);
    if (mt1) {
        if (wide) NVR_LIN(4, 1, 4);
        else if (wv == 16) NVR_LIN(1, 1, 16);
        else if (wv == 8) NVR_LIN(1, 1, 8);
        else NVR_LIN(1, 1, 4);
    } else {
        if (wide) NVR_LIN(4, 2, 4);
        else if (wv == 16) NVR_LIN(1, 2, 16);
        else if (wv == 8) NVR_LIN(1, 2, 8);
        else NVR_LIN(1, 2, 4);
    }
#undef NVR_LIN
    return launch_check("linear");
}

// split-k over S workgroups per tile: slabs[z][T][N] f32 partial sums (summed by add_rmsnorm_slabs); the narrow
// row-parallel GEMMs of the decode step (o_proj, down_proj: N = hidden) reach all 256 CUs this way
int linear_splitk(const half_bits *x, int64_t ldx, const half_bits *W, int64_t T, int64_t K, int64_t N, int64_t S,
                  float *slabs, hipStream_t s, const half_bits *Wt) {
    if (K % (32 * S) || N % 16 || ldx % 8 || S < 1)
        return nvr::fail(NVR_ERR_UNSUPPORTED, "linear_splitk: K=%ld must be a multiple of 32*S (S=%ld), N=%ld of 16", (long)K, (long)S, (long)N);
    if (T == 0) return 0;
    if (splitk_prefers_tiles(T, K, N) && gemm_tiled_splitk_ok(T, K, N, S, ldx)) return gemm_tiled_splitk(x, ldx, W, T, K, N, S, slabs, s);   // LDS tiles
    LinEpi e{};
    e.kslice = (int32_t)(K / S); e.slab_stride = T * N;
    if (Wt) { W = Wt; e.tiled = 1; }
    const unsigned gx = (unsigned)(N / 16);
    if (N * K * 2 >= (24ll << 20) && N % 64 == 0) {
        // large weights (Qwen3-8B o/down: N = 4096): 64 output columns per workgroup so that an x fragment feeds four weight
        // tiles (x costs half the weight bytes instead of twice), 8 waves; with S = 4: 10.8 / 29.5 us vs 15.5 / 45 us for the
        // 16-column kernel at K = 4096 / 12288 (scratch/splitk_8b.py)
        if (T <= 16) launch<4, 1, 8, EPI_SLAB>((const half_t *)x, ldx, (const half_t *)W, (int)T, (int)K, (int)N, slabs, e, (unsigned)(N / 64), s, (unsigned)S);
        else launch<4, 2, 8, EPI_SLAB>((const half_t *)x, ldx, (const half_t *)W, (int)T, (int)K, (int)N, slabs, e, (unsigned)(N / 64), s, (unsigned)S);
        return launch_check("linear_splitk");
    }
    if (T <= 16) launch<1, 1, 4, EPI_SLAB>((const half_t *)x, ldx, (const half_t *)W, (int)T, (int)K, (int)N, slabs, e, gx, s, (unsigned)S);
    else launch<1, 2, 4, EPI_SLAB>((const half_t *)x, ldx, (const half_t *)W, (int)T, (int)K, (int)N, slabs, e, gx, s, (unsigned)S);
    return launch_check("linear_splitk");
}

// gate_up GEMM + SiluAndMul: W [2I, K] (gate rows then up rows), out [T, I]
int linear_silu_mul(const half_bits *x, int64_t ldx, const half_bits *W, int64_t T, int64_t K, int64_t I,
                    half_bits *out, hipStream_t s, const half_bits *Wt) {
    if (K % 32 || I % 16 || ldx % 8)
        return nvr::fail(NVR_ERR_UNSUPPORTED, "linear_silu_mul: K=%ld must be a multiple of 32, I=%ld of 16", (long)K, (long)I);
    if (T == 0) return 0;
    if (gemm256_silu_ok(T, K, I, ldx) && prefer_256(T, K, 2 * I)) return gemm256_silu_mul(x, ldx, W, T, K, I, out, s);   // see linear()
    if (gemm_tiled_ok(T, K, I, ldx) && I % 64 == 0 && !prefer_stream(T, 2 * I)) return gemm_tiled_silu_mul(x, ldx, W, T, K, I, out, s);
    if (linear_stream_silu_ok(T, K, I, ldx)) return linear_stream_silu_mul(x, ldx, W, T, K, I, out, s, Wt);              // large weights
    const half_t *xx = (const half_t *)x, *ww = (const half_t *)(Wt ? Wt : W);
    LinEpi e{};
    e.tiled = Wt != nullptr;
    const unsigned gx = (unsigned)(I / 16);
    const int wv = waves_for(K);
    // few column tiles (the shard of a tensor-parallel rank): 16-token workgroups double the workgroup count (the second token
    // block finds its W tile in the XCD's L2) and long rows are split over 16 waves — same rule as linear()
    const bool narrow = I / 16 < 128 && T <= 32;
    if (narrow && wv == 16) {
        launch<2, 1, 16, EPI_SILU>(xx, ldx, ww, (int)T, (int)K, (int)I, out, e, gx, s);
    } else if (T <= 16 || narrow) {
        if (wv >= 8) launch<2, 1, 8, EPI_SILU>(xx, ldx, ww, (int)T, (int)K, (int)I, out, e, gx, s);
        else launch<2, 1, 4, EPI_SILU>(xx, ldx, ww, (int)T, (int)K, (int)I, out, e, gx, s);
    } else {
        if (wv >= 8) launch<2, 2, 8, EPI_SILU>(xx, ldx, ww, (int)T, (int)K, (int)I, out, e, gx, s);
        else launch<2, 2, 4, EPI_SILU>(xx, ldx, ww, (int)T, (int)K, (int)I, out, e, gx, s);
    }
    return launch_check("linear_silu_mul");
}

// qkv GEMM + RoPE on q,k heads + store of k,v rows into the paged caches
int linear_qkv_rope_store(const half_bits *x, int64_t ldx, const half_bits *W, int64_t T, int64_t K, int64_t H,
                          int64_t KVH, int64_t D, const int64_t *positions, const int32_t *slots, const float *cos_t,
                          const float *sin_t, half_bits *qkv, half_bits *k_cache, half_bits *v_cache, hipStream_t s, const half_bits *Wt,
                          bool kv_cache_only) {
    if (K % 32 || D % 16 || ldx % 8)
        return nvr::fail(NVR_ERR_UNSUPPORTED, "linear_qkv_rope_store: K=%ld must be a multiple of 32, D=%ld of 16", (long)K, (long)D);
    if (T == 0) return 0;
    if (gemm256_rope_ok(T, K, H, KVH, D, ldx) && prefer_256(T, K, (H + 2 * KVH) * D))                                  // see linear()
        return gemm256_qkv_rope_store(x, ldx, W, T, K, H, KVH, D, positions, slots, cos_t, sin_t, qkv, k_cache, v_cache, s, kv_cache_only);
    if (gemm_tiled_ok(T, K, (H + 2 * KVH) * D, ldx) && 128 % D == 0 && !prefer_stream(T, (H + 2 * KVH) * D))
        return gemm_tiled_qkv_rope_store(x, ldx, W, T, K, H, KVH, D, positions, slots, cos_t, sin_t, qkv, k_cache, v_cache, s);
    if (linear_stream_rope_ok(T, K, H, KVH, D, ldx))                                                                  // large weights
        return linear_stream_qkv_rope_store(x, ldx, W, T, K, H, KVH, D, positions, slots, cos_t, sin_t, qkv, k_cache, v_cache, s, Wt);
    const half_t *xx = (const half_t *)x, *ww = (const half_t *)(Wt ? Wt : W);
    LinEpi e{};
    e.tiled = Wt != nullptr;
    e.pos = positions; e.slots = slots; e.cos_t = cos_t; e.sin_t = sin_t; e.kc = (half_t *)k_cache; e.vc = (half_t *)v_cache;
    e.H = (int32_t)H; e.KVH = (int32_t)KVH; e.D = (int32_t)D;
    const int N = (int)((H + 2 * KVH) * D);
    const unsigned gx = (unsigned)(N / 16);
    const int wv = waves_for(K);
    const bool narrow = N / 16 < 128 && T <= 32;                 // see linear_silu_mul
    if (narrow && wv == 16) {
        launch<1, 1, 16, EPI_ROPE>(xx, ldx, ww, (int)T, (int)K, N, qkv, e, gx, s);
    } else if (T <= 16 || narrow) {
        if (wv >= 8) launch<1, 1, 8, EPI_ROPE>(xx, ldx, ww, (int)T, (int)K, N, qkv, e, gx, s);
        else launch<1, 1, 4, EPI_ROPE>(xx, ldx, ww, (int)T, (int)K, N, qkv, e, gx, s);
    } else {
        if (wv >= 8) launch<1, 2, 8, EPI_ROPE>(xx, ldx, ww, (int)T, (int)K, N, qkv, e, gx, s);
        else launch<1, 2, 4, EPI_ROPE>(xx, ldx, ww, (int)T, (int)K, N, qkv, e, gx, s);
    }
    return launch_check("linear_qkv_rope_store");
}

}}  // namespace nvr::k / nvr::kb (NVR_DT_NS)
