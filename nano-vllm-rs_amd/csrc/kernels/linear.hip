// linear.hip — y[T,N] = x[T,K] · W[N,K]^T on MFMA (K3/K10/K12/K14/K16).
// reference call sites: QKVParallelLinear::forward src/layers/linear.rs:354-356, RowParallelLinear
// :228-239, MergedColumnParallelLinear :437-439, ParallelLMHead::compute_logits
// src/layers/embed_head.rs:292-306 (all `candle_nn::Linear::forward`, x·Wᵀ with W [out, in]).
//
// Weight-streaming kernel for the decode regime (T <= 64: every weight byte is read once from HBM
// and the kernel is HBM-bound).  Roofline: algorithmic bytes = 2·N·K (weights) +
// 2·T·K (x) + out; MFMA is used only because T=32 tokens x 8 k per 16-byte weight load exceeds
// the VALU rate at HBM speed (cdna_hip_programming.md §5 "GEMV / M <= 16" row: weights straight to
// VGPRs, no LDS round trip).
//
// Tiling: v_mfma_f32_16x16x32_f16 with A = W tile (16 output columns n x 32 k: each lane loads 16
// contiguous bytes of one W row, 4 lanes cover 64 B of the row) and B = x tile (16 tokens x 32 k,
// L2 resident).  A workgroup = 4 waves owns NT*16 output columns for MT*16 tokens; the 4 waves
// interleave 32-wide k steps (so together they read 256 contiguous bytes per W row per step) and
// reduce their f32 partials through LDS.  Grid = (N/(16 NT), ceil(T/(16 MT))): for large T the
// same kernel re-streams W from L2 per 16·MT-token slab (correct, not yet the tiled prefill GEMM).
#include "kernels.h"
#include "device_utils.h"
#include "../common.h"

namespace nvr { namespace k {

template <int NT, int MT, bool F32OUT>
__global__ __launch_bounds__(256) void linear_skinny_kernel(const half_t *__restrict__ x, int64_t ldx,
                                                            const half_t *__restrict__ W, int T, int K, int N,
                                                            void *__restrict__ y) {
    constexpr int U = 4;                                    // k-steps in flight per wave
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 15, q = lane >> 4;
    const int n0 = blockIdx.x * (16 * NT), m0 = blockIdx.y * (16 * MT);

    const half_t *wrow[NT];
    const half_t *xrow[MT];
#pragma unroll
    for (int i = 0; i < NT; ++i) { int n = n0 + i * 16 + r; if (n > N - 1) n = N - 1; wrow[i] = W + (int64_t)n * K + q * 8; }
#pragma unroll
    for (int i = 0; i < MT; ++i) { int m = m0 + i * 16 + r; if (m > T - 1) m = T - 1; xrow[i] = x + (int64_t)m * ldx + q * 8; }

    float4_t acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

    for (int k = wave * 32; k < K; k += 128 * U) {
        half8_t a[U][NT], b[U][MT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kk = k + u * 128;
            if (kk < K) {
#pragma unroll
                for (int i = 0; i < NT; ++i) a[u][i] = *reinterpret_cast<const half8_t *>(wrow[i] + kk);
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
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u][i], b[u][j], acc[i][j], 0, 0, 0);
    }

    // cross-wave (split-k) reduction through LDS: part[wave][tile][lane] as float4
    __shared__ float4_t part[4][NT * MT][64];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) part[wave][i * MT + j][lane] = acc[i][j];
    __syncthreads();
    for (int tile = wave; tile < NT * MT; tile += 4) {
        float4_t s = part[0][tile][lane];
#pragma unroll
        for (int w2 = 1; w2 < 4; ++w2) { float4_t p = part[w2][tile][lane]; s += p; }
        const int i = tile / MT, j = tile % MT;
        // C layout of 16x16 MFMA: row (n) = q*4 + reg, col (token) = r
        const int n = n0 + i * 16 + q * 4, m = m0 + j * 16 + r;
        if (m < T && n < N) {
            if (F32OUT) {
                *reinterpret_cast<float4_t *>(reinterpret_cast<float *>(y) + (int64_t)m * N + n) = s;
            } else {
                half4_t h = {(half_t)s[0], (half_t)s[1], (half_t)s[2], (half_t)s[3]};
                *reinterpret_cast<half4_t *>(reinterpret_cast<half_t *>(y) + (int64_t)m * N + n) = h;
            }
        }
    }
}

template <int NT, int MT>
static void launch(const half_t *x, int64_t ldx, const half_t *W, int T, int K, int N, void *y, bool f32,
                   hipStream_t s) {
    dim3 grid((unsigned)((N + 16 * NT - 1) / (16 * NT)), (unsigned)((T + 16 * MT - 1) / (16 * MT)));
    if (f32) linear_skinny_kernel<NT, MT, true><<<grid, dim3(256), 0, s>>>(x, ldx, W, T, K, N, y);
    else linear_skinny_kernel<NT, MT, false><<<grid, dim3(256), 0, s>>>(x, ldx, W, T, K, N, y);
}

int linear(const half_bits *x, int64_t ldx, const half_bits *W, int64_t T, int64_t K, int64_t N, void *y,
           bool y_f32, hipStream_t s) {
    if (K % 32 || N % 16 || ldx % 8)
        return nvr::fail(NVR_ERR_UNSUPPORTED, "linear: K=%ld must be a multiple of 32, N=%ld of 16, ldx=%ld of 8",
                         (long)K, (long)N, (long)ldx);
    if (T == 0) return 0;
    const half_t *xx = (const half_t *)x, *ww = (const half_t *)W;
    const int64_t mtiles = (T + 31) / 32;
    if (T <= 16) {
        if (N / 64 >= 1024) launch<4, 1>(xx, ldx, ww, (int)T, (int)K, (int)N, y, y_f32, s);
        else if (N / 32 >= 512) launch<2, 1>(xx, ldx, ww, (int)T, (int)K, (int)N, y, y_f32, s);
        else launch<1, 1>(xx, ldx, ww, (int)T, (int)K, (int)N, y, y_f32, s);
    } else {
        if ((N / 64) * mtiles >= 1024) launch<4, 2>(xx, ldx, ww, (int)T, (int)K, (int)N, y, y_f32, s);
        else if ((N / 32) * mtiles >= 512) launch<2, 2>(xx, ldx, ww, (int)T, (int)K, (int)N, y, y_f32, s);
        else launch<1, 2>(xx, ldx, ww, (int)T, (int)K, (int)N, y, y_f32, s);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return nvr::fail(NVR_ERR_HIP, "linear launch failed: %s", hipGetErrorString(e));
    return 0;
}

}}  // namespace nvr::k
