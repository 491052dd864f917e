// mlp_engine.hip — the MLP pair of a decode layer (gate_up + SiluAndMul -> down_proj) as ONE persistent launch on a loader / consumer engine.
// reference: Qwen3MLP::forward src/models/qwen3.rs:305-314 (gate_up_proj :307, SiluAndMul :310, down_proj :313) with
// MergedColumnParallelLinear src/layers/linear.rs:437-439 and RowParallelLinear :228-239; activation.rs:46-63.
//
// What it replaces: the launches linear_silu_mul (192 workgroups, 12.6 MB of weights) and linear_splitk (256 workgroups, 6.3 MB) of the
// six-launch decode chain, i.e. one dependent kernel boundary (~2 us) plus the first-byte latency of the second weight stream.  Built on the
// guide's weight-streaming engine (cdna_hip_programming.md §5.6, MI355X_MICROARCH price rows prefetch-credit / engine-vs-launches /
// ldsdma-fill / handoff-flag / fanin): per CU ONE workgroup = 1 LDS-DMA loader wave + 3 MFMA consumer waves; the loader streams the
// workgroup's gate / up weight tiles (nt) into LDS in MFMA lane order, publishes them k-step group by group through an LDS word, and runs
// AHEAD OF THE SEAM: the workgroup's down_proj weight tile (its k-slice) is in LDS before the activations it multiplies exist.
//
//   phase 1  workgroup w < I/16: act[:, 16w .. 16w+16) = fp16(silu(fp16 g) * fp16 u), g / u = x · W_gate / W_up tile w; the consumers split k
//            three ways (x fragments straight from L2 into registers), reduce through LDS in wave order, store the tile WRITE-THROUGH (sc1) and
//            add to the arrival counter of the tile's down_proj k-slice after their own vmcnt(0)           (hand-off form: counter, R1)
//   seam     workgroup (column tile j = w / 4, k-slice s = w % 4): one lane polls counter[s] (relaxed, s_sleep, bounded on the 100 MHz wall
//            clock: a timeout sets a word and the launch ends, the GPU never hangs), ONE agent acquire, barrier
//   phase 2  slab[s][:, 16j .. 16j+16) = act[:, k-slice s] · W_down tile j (weights already in LDS), f32, summed by the add + RMSNorm launch
//            that follows — the same slabs linear_splitk writes, so the rest of the chain is unchanged.
// Rounding points are those of the two launches (gate, up -> fp16 -> SiLU·up -> fp16; down f32 slabs); the k summation order differs (3
// consumer waves instead of 8 / 4 wave slices): results agree with the launches to f32 summation order, tolerance as for every GEMM route.
// Placement-independent: nothing assumes which CU or XCD a workgroup runs on; it needs every workgroup RESIDENT (grid <= CU count, one
// workgroup per CU by its LDS footprint) — the launcher refuses other shapes, the poll is bounded.
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>
#include "kernels.h"
#include "device_utils.h"
#include "../common.h"

namespace nvr { namespace NVR_DT_NS {

struct MlpArgs {
    const half_t *x; int64_t ldx;            // [T, Hd] normalised input
    const half_t *wgu, *wd;                  // tiled copies: gate_up [2I/16][Hd/32][16][32], down [Hd/16][I/32][16][32]
    int32_t T, Hd, I;
    half_t *act;                             // [T, I]
    float *slabs;                            // [4][T][Hd]
    unsigned *sync;                          // [4] timeout word, [8 ...) arrival flags (tile, token tile); zeroed before every launch
    unsigned timeout_ticks;                  // 100 MHz ticks a workgroup waits at the seam
};

constexpr int MLP_S = 4;                     // k-slices of down_proj (= the slabs of linear_splitk at this shape)

// one 1-KiB weight k-step (16 rows x 32 k of the tiled copy) into LDS in MFMA lane order: lane l = q*16 + r takes the 16 bytes of row r, k 8q..8q+7
__device__ __forceinline__ void dma_kstep(const half_t *tile_kstep, unsigned lds_dst, int lane) {
    const half_t *src = tile_kstep + (lane & 15) * 32 + (lane >> 4) * 8;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
}

#ifdef NVR_MLP_STAMPS   // diagnostic build (tools/build_variant.sh): 100 MHz wall-clock stamps of wave 1 of every workgroup -> profiles/r04_mlp_engine.txt
__device__ unsigned long long mlp_stamp_buf[512 * 8];
#define MLP_STAMP(i) do { if (threadIdx.x == 64) mlp_stamp_buf[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MLP_STAMP(i) do { } while (0)
#endif
template <int MT, int KB1, int KB2>
__global__ __launch_bounds__(256, 1) void mlp_engine_kernel(MlpArgs a) {
    constexpr int NI1 = (KB1 + 2) / 3, NI2 = (KB2 + 2) / 3;          // k-steps per consumer wave
    constexpr int GU_BYTES = KB1 * 2048, D_BYTES = KB2 * 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];       // [gate | up k-steps][down k-steps][partials 3 x 4 KiB][ready word]
    char *lds_gu = smem, *lds_d = smem + GU_BYTES, *lds_part = lds_d + D_BYTES;
    volatile unsigned *ready = reinterpret_cast<volatile unsigned *>(lds_part + 3 * 4096);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 15, q = lane >> 4;
    const int w = blockIdx.x;
    const bool has_tile = w < a.I / 16, has_down = w < (a.Hd / 16) * MLP_S;
    const int j2 = w / MLP_S, sl = w % MLP_S;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem;
    MLP_STAMP(0);
    if (threadIdx.x == 0) *ready = 0;
    __syncthreads();

    float4_t accg[MT], accu[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) { accg[j] = (float4_t){0.f, 0.f, 0.f, 0.f}; accu[j] = accg[j]; }

    if (wave == 0) {
        // ---- loader: gate / up k-steps, published four at a time with twelve in flight behind them; then the down_proj k-slice
        if (has_tile) {
            const half_t *gt = a.wgu + (int64_t)w * KB1 * 512, *ut = a.wgu + ((int64_t)(a.I / 16) + w) * KB1 * 512;
#pragma unroll 1
            for (int kb = 0; kb < KB1; ++kb) {
                dma_kstep(gt + kb * 512, lds0 + kb * 2048, lane);
                dma_kstep(ut + kb * 512, lds0 + kb * 2048 + 1024, lane);
                if ((kb & 3) == 3 && kb >= 15) {                     // 24 requests (12 k-steps) may fly behind the ones published
                    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
                    if (lane == 0) *ready = (unsigned)(kb + 1 - 12);
                }
            }
        }
        if (has_down) {
            const half_t *dt = a.wd + ((int64_t)j2 * (a.I / 32) + (int64_t)sl * KB2) * 512;
#pragma unroll 1
            for (int kb = 0; kb < KB2; ++kb) dma_kstep(dt + kb * 512, lds0 + GU_BYTES + kb * 1024, lane);
        }
        if (has_tile) {
            // every gate / up k-step has landed when at most the down_proj requests are outstanding
            if (has_down) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KB2 < 63 ? KB2 : 63) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) *ready = (unsigned)KB1;
        }
    } else if (has_tile) {
        // ---- consumers, phase 1: k-steps c, c+3, ...; the x fragments of all of them are requested up front (L2-resident, 16 B per lane)
        const int c = wave - 1;
        half8_t xf[NI1][MT];
#pragma unroll
        for (int i = 0; i < NI1; ++i) {
            const int kb = c + 3 * i;
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                int m = j * 16 + r; if (m > a.T - 1) m = a.T - 1;
                xf[i][j] = kb < KB1 ? *reinterpret_cast<const half8_t *>(a.x + (int64_t)m * a.ldx + kb * 32 + q * 8) : (half8_t)(half_t)0;
            }
        }
        unsigned have = 0;
#pragma unroll
        for (int i = 0; i < NI1; ++i) {
            const int kb = c + 3 * i;
            if (kb < KB1) {
                while (have <= (unsigned)kb) { have = *ready; if (have <= (unsigned)kb) __builtin_amdgcn_s_sleep(1); }
                const half8_t gf = *reinterpret_cast<const half8_t *>(lds_gu + kb * 2048 + lane * 16);
                const half8_t uf = *reinterpret_cast<const half8_t *>(lds_gu + kb * 2048 + 1024 + lane * 16);
#pragma unroll
                for (int j = 0; j < MT; ++j) { accg[j] = mfma16(gf, xf[i][j], accg[j]); accu[j] = mfma16(uf, xf[i][j], accu[j]); }
            }
        }
        float4_t *part = reinterpret_cast<float4_t *>(lds_part + c * 4096);
#pragma unroll
        for (int j = 0; j < MT; ++j) { part[j * 64 + lane] = accg[j]; part[(2 + j) * 64 + lane] = accu[j]; }
    }
    MLP_STAMP(1);
    __syncthreads();                                                  // (1) partials of the three consumers are in LDS
    MLP_STAMP(2);
    if (has_tile && wave >= 1 && wave <= MT) {
        // act tile of token tile j = wave - 1: C layout row (n) = q*4 + reg, column (token) = r
        const int j = wave - 1;
        const float4_t *p0 = reinterpret_cast<const float4_t *>(lds_part), *p1 = p0 + 256, *p2 = p0 + 512;
        const float4_t g4 = (p0[j * 64 + lane] + p1[j * 64 + lane]) + p2[j * 64 + lane];
        const float4_t u4 = (p0[(2 + j) * 64 + lane] + p1[(2 + j) * 64 + lane]) + p2[(2 + j) * 64 + lane];
        const int m = j * 16 + r, n = w * 16 + q * 4;
        union { half4_t v; unsigned long long u; } hv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gf = (float)to_half_rn(g4[e]), uf = (float)to_half_rn(u4[e]);
            hv.v[e] = to_half_rn(__fmul_rn(__fmul_rn(gf, sigmoid_fast(gf)), uf));
        }
        if (m < a.T)                                                  // write-through: the consumers of this tile sit on other CUs (guide G16 R1)
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(a.act + (int64_t)m * a.I + n), hv.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's stores have left
        // arrival: ONE write-through flag word per (tile, token tile) — no atomics (first build: an arrival counter per k-slice; 96 atomic adds
        // on one word while 64 workgroups polled it: the last arrival became visible 4-6 us after it was issued, profiles/r04_mlp_engine.txt)
        if (lane == 0) __hip_atomic_store(a.sync + 8 + w * 2 + j, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (wave == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the down_proj k-slice is in LDS
    MLP_STAMP(3);
    // ---- seam: every tile of my k-slice has arrived (MT arrivals per tile), or the wait ran out
    if (has_down && wave == 1) {
        // the flags of my k-slice are consecutive words: the whole wave sweeps them with one or two loads per poll
        const int tiles = a.I / MLP_S / 16, nflag = tiles * 2;       // (tile, token tile) words; token tile 1 is unused when MT == 1
        const unsigned *fl = a.sync + 8 + sl * nflag;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (;;) {
            bool ok = true;
            for (int i = lane; i < nflag; i += 64)
                if (MT == 2 || (i & 1) == 0) ok &= __hip_atomic_load(fl + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
            if (__all(ok)) break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > a.timeout_ticks) {
                if (lane == 0) __hip_atomic_store(a.sync + MLP_S, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    MLP_STAMP(4);
    __syncthreads();                                                  // (2)
    MLP_STAMP(5);
    float4_t accd[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) accd[j] = (float4_t){0.f, 0.f, 0.f, 0.f};
    if (has_down && wave >= 1) {
        const int c = wave - 1;
        half8_t af[NI2][MT];
#pragma unroll
        for (int i = 0; i < NI2; ++i) {
            const int kb = c + 3 * i;
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                int m = j * 16 + r; if (m > a.T - 1) m = a.T - 1;
                af[i][j] = kb < KB2 ? *reinterpret_cast<const half8_t *>(a.act + (int64_t)m * a.I + (sl * KB2 + kb) * 32 + q * 8) : (half8_t)(half_t)0;
            }
        }
#pragma unroll
        for (int i = 0; i < NI2; ++i) {
            const int kb = c + 3 * i;
            if (kb < KB2) {
                const half8_t wf = *reinterpret_cast<const half8_t *>(lds_d + kb * 1024 + lane * 16);
#pragma unroll
                for (int j = 0; j < MT; ++j) accd[j] = mfma16(wf, af[i][j], accd[j]);
            }
        }
        float4_t *part = reinterpret_cast<float4_t *>(lds_part + c * 4096);
#pragma unroll
        for (int j = 0; j < MT; ++j) part[j * 64 + lane] = accd[j];
    }
    MLP_STAMP(6);
    __syncthreads();                                                  // (3)
    if (has_down && wave >= 1 && wave <= MT) {
        const int j = wave - 1;
        const float4_t *p0 = reinterpret_cast<const float4_t *>(lds_part), *p1 = p0 + 256, *p2 = p0 + 512;
        const float4_t s4 = (p0[j * 64 + lane] + p1[j * 64 + lane]) + p2[j * 64 + lane];
        const int m = j * 16 + r, n = j2 * 16 + q * 4;
        if (m < a.T) *reinterpret_cast<float4_t *>(a.slabs + ((int64_t)sl * a.T + m) * a.Hd + n) = s4;
    }
    MLP_STAMP(7);
}

static size_t mlp_lds_bytes(int64_t Hd, int64_t I) { return (size_t)(Hd / 32) * 2048 + (size_t)(I / (32 * MLP_S)) * 1024 + 3 * 4096 + 16; }

// shapes the engine is instantiated for (k-steps of gate_up, k-steps of a down_proj k-slice)
#define NVR_MLP_SHAPES(X) X(32, 24) X(8, 4) X(16, 8)

bool mlp_engine_ok(int64_t T, int64_t Hd, int64_t I, int ncu) {
    if (T < 1 || T > 32 || Hd % 32 || I % (32 * MLP_S) || I % 16) return false;
    const int64_t grid = std::max(I / 16, (Hd / 16) * MLP_S);
    if (grid > ncu || mlp_lds_bytes(Hd, I) > 160 * 1024 || I / (32 * MLP_S) > 60 || I / 16 > 512) return false;   // every workgroup resident; the loader's counted wait fits vmcnt
    bool inst = false;
#define X(A, B) inst |= (Hd / 32 == A && I / (32 * MLP_S) == B);
    NVR_MLP_SHAPES(X)
#undef X
    return inst;
}
size_t mlp_engine_sync_bytes() { return 32 + 512 * 2 * 4; }     // [4 spare | timeout | 3 pad] + one flag word per (gate/up tile <= 512, token tile)

int mlp_engine(const half_bits *x, int64_t ldx, const half_bits *gate_up_t, const half_bits *down_t, int64_t T, int64_t Hd, int64_t I,
               half_bits *act, float *slabs, unsigned *sync, hipStream_t s) {
    int dev = 0, ncu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) ncu = 0;
    if (!mlp_engine_ok(T, Hd, I, ncu) || ldx % 8)
        return nvr::fail(NVR_ERR_UNSUPPORTED, "mlp_engine: T=%ld Hd=%ld I=%ld on %d CUs", (long)T, (long)Hd, (long)I, ncu);
    MlpArgs a{};
    a.x = (const half_t *)x; a.ldx = ldx; a.wgu = (const half_t *)gate_up_t; a.wd = (const half_t *)down_t;
    a.T = (int32_t)T; a.Hd = (int32_t)Hd; a.I = (int32_t)I; a.act = (half_t *)act; a.slabs = slabs; a.sync = sync;
    a.timeout_ticks = 20000000u;                                     // 200 ms on the 100 MHz wall clock
    const unsigned grid = (unsigned)std::max(I / 16, (Hd / 16) * MLP_S);
    const size_t lds = mlp_lds_bytes(Hd, I);
    hipError_t e = hipMemsetAsync(sync, 0, mlp_engine_sync_bytes(), s);   // counters and the timeout word: re-initialised by every launch (a memset node under capture)
    if (e != hipSuccess) return nvr::fail(NVR_ERR_HIP, "mlp_engine: hipMemsetAsync: %s", hipGetErrorString(e));
#define X(A, B)                                                                                                                    \
    if (Hd / 32 == A && I / (32 * MLP_S) == B) {                                                                                   \
        static bool ready1 = false, ready2 = false;                                                                               \
        if (T <= 16) {                                                                                                             \
            if (!ready1) { hipFuncSetAttribute(reinterpret_cast<const void *>(&mlp_engine_kernel<1, A, B>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); ready1 = true; } \
            mlp_engine_kernel<1, A, B><<<grid, 256, lds, s>>>(a);                                                                  \
        } else {                                                                                                                   \
            if (!ready2) { hipFuncSetAttribute(reinterpret_cast<const void *>(&mlp_engine_kernel<2, A, B>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); ready2 = true; } \
            mlp_engine_kernel<2, A, B><<<grid, 256, lds, s>>>(a);                                                                  \
        }                                                                                                                          \
    }
    NVR_MLP_SHAPES(X)
#undef X
    e = hipGetLastError();
    if (e != hipSuccess) return nvr::fail(NVR_ERR_HIP, "mlp_engine launch failed: %s", hipGetErrorString(e));
#ifdef NVR_MLP_STAMPS
    {
        static int calls = 0;
        if (++calls == 6) {
            hipStreamSynchronize(s);
            std::vector<unsigned long long> hb(grid * 8);
            hipMemcpyFromSymbol(hb.data(), HIP_SYMBOL(mlp_stamp_buf), grid * 8 * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
            unsigned long long t0 = ~0ull, t1 = 0; double d[7] = {0, 0, 0, 0, 0, 0, 0}, start = 0;
            for (unsigned i = 0; i < grid; ++i) { t0 = std::min(t0, hb[i * 8]); t1 = std::max(t1, hb[i * 8 + 7]); }
            const unsigned np = (unsigned)(I / 16);
            for (unsigned i = 0; i < grid; ++i) { start += (double)(hb[i * 8] - t0); for (int j = 0; j < 7; ++j) d[j] += (double)(hb[i * 8 + j + 1] - hb[i * 8 + j]); }
            {
                std::vector<double> p1, sm, endp1;
                for (unsigned i = 0; i < grid; ++i) {
                    if (i < np) { p1.push_back((double)(hb[i * 8 + 1] - hb[i * 8]) / 100.0); endp1.push_back((double)(hb[i * 8 + 3] - t0) / 100.0); }
                    sm.push_back((double)(hb[i * 8 + 4] - hb[i * 8 + 3]) / 100.0);
                }
                std::sort(p1.begin(), p1.end()); std::sort(sm.begin(), sm.end()); std::sort(endp1.begin(), endp1.end());
                std::fprintf(stderr, "[mlp stamps] phase-1 k loop of the %zu producing workgroups: min %.2f median %.2f p90 %.2f max %.2f us; their act tiles signalled at (from launch start) median %.2f p90 %.2f max %.2f us; "
                             "seam wait: min %.2f median %.2f max %.2f us\n", p1.size(), p1.front(), p1[p1.size() / 2], p1[p1.size() * 9 / 10], p1.back(),
                             endp1[endp1.size() / 2], endp1[endp1.size() * 9 / 10], endp1.back(), sm.front(), sm[sm.size() / 2], sm.back());
            }
            std::fprintf(stderr, "[mlp stamps] %u workgroups (%u with a gate/up tile): kernel span %.2f us; mean start skew %.2f us; per workgroup (wave 1, us): phase-1 k loop %.2f, barrier %.2f, act tile + signal %.2f, "
                         "seam poll + acquire %.2f, barrier %.2f, phase-2 k loop %.2f, barrier + slab store %.2f\n", grid, np, (double)(t1 - t0) / 100.0, start / grid / 100.0,
                         d[0] / grid / 100.0, d[1] / grid / 100.0, d[2] / grid / 100.0, d[3] / grid / 100.0, d[4] / grid / 100.0, d[5] / grid / 100.0, d[6] / grid / 100.0);
        }
    }
#endif
    return 0;
}

}}  // namespace nvr::k / nvr::kb (NVR_DT_NS)
