// Sustained matrix-pipe rate at the power / clock limit: every SIMD of the chip runs an MFMA-only loop (2 waves per SIMD, operands in
// registers, random-ish nonzero data) for ~3 ms: PFLOP/s and the clock the chip holds, 16x16x32 against 32x32x16 (f16).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4_ __attribute__((ext_vector_type(4)));
typedef float float16_ __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int M32>
__global__ __launch_bounds__(512) void k_mfma(float *out, long long *cyc, int iters) {
    half8 a[4], b[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) { a[i][e] = (_Float16)(0.37f + 0.001f * ((threadIdx.x * 7 + e * 13 + i * 3) % 97)); b[i][e] = (_Float16)(-0.21f + 0.002f * ((threadIdx.x * 5 + e * 11 + i) % 89)); }
    float4_ acc[8] = {}; float16_ acc32[4] = {};
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (M32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc32[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u & 3], b[(u + 1) & 3], acc32[u & 3], 0, 0, 0);
        } else {
#pragma unroll
            for (int u = 0; u < 16; ++u) acc[u & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u & 3], b[(u + 1) & 3], acc[u & 7], 0, 0, 0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int u = 0; u < 8; ++u) s += acc[u][0]; for (int u = 0; u < 4; ++u) s += acc32[u][0];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    float *out; long long *cyc; CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&cyc, 64)); long long h;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) for (int m32 = 0; m32 < 2; ++m32) {
        const int iters = 60000;           // x 16 MFMAs of 16 cycles (or 8 of 32): ~256 cycles per iteration per wave
        for (int r = 0; r < 2; ++r) {
            CK(hipEventRecord(e0));
            if (m32) k_mfma<1><<<256, 512>>>(out, cyc, iters); else k_mfma<0><<<256, 512>>>(out, cyc, iters);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
        const double flop = 256.0 * 8 * iters * (m32 ? 8 * 32768.0 : 16 * 16384.0);
        printf("%s: %.2f ms, %.3f PFLOP/s, clock %.2f GHz (wave 0 of workgroup 0: %lld cycles)\n", m32 ? "32x32x16" : "16x16x32", ms, flop / (ms * 1e-3) / 1e15, h / (ms * 1e6), h);
    }
    return 0;
}
