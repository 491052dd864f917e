// Issue-overlap microbenchmark (gfx950): how many independent VALU instructions fit behind one MFMA without costing time
//   in-wave:    one wave per SIMD runs a loop of { 1 MFMA ; n VALU }  (all independent accumulators)
//   cross-wave: two waves per SIMD, wave A runs MFMAs only, wave B VALU only: time of both against each alone
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4_ __attribute__((ext_vector_type(4)));
typedef float float16_ __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NV, int MF32, int EXP>   // NV VALU per MFMA; MF32: 32x32x16 instead of 16x16x32; EXP: the VALU ops are v_exp_f32
__global__ __launch_bounds__(256) void k_inwave(float *out, long long *cyc, int iters) {
    half8 a, b; for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f); b[e] = (_Float16)1.0f; }
    float4_ acc[4] = {}; float16_ acc32[2] = {};
    float v[8]; for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.5f + i;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MF32) acc32[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc32[u & 1], 0, 0, 0);
            else acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[u], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                if (EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(v[(u * NV + j) & 7]));
                else asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[(u * NV + j) & 7]));
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int i = 0; i < 8; ++i) s += v[i];
    for (int u = 0; u < 4; ++u) s += acc[u][0]; s += acc32[0][0] + acc32[1][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// 512 threads = 8 waves = 2 per SIMD; waves 0-3 role A, waves 4-7 role B
template <int ROLE_A, int ROLE_B>      // 0 idle (exit), 1 MFMA 16x16x32 loop, 2 VALU fma loop, 3 exp loop
__global__ __launch_bounds__(512) void k_cross(float *out, long long *cyc, int iters) {
    const int wave = threadIdx.x >> 6; const int role = wave < 4 ? ROLE_A : ROLE_B;
    half8 a, b; for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f); b[e] = (_Float16)1.0f; }
    float4_ acc[4] = {}; float v[8]; for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.5f + i;
    long long t0 = __builtin_readcyclecounter();
    if (role == 1) { for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[u & 3], 0, 0, 0); } }
    else if (role == 2) { for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[u & 7])); } }
    else if (role == 3) { for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) asm volatile("v_exp_f32 %0, %0" : "+v"(v[u & 7])); } }
    long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int i = 0; i < 8; ++i) s += v[i]; for (int u = 0; u < 4; ++u) s += acc[u][0];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[1 + wave] = t1 - t0;
}
int main() {
    float *out; long long *cyc; CK(hipMalloc(&out, 1 << 22)); CK(hipMalloc(&cyc, 256)); long long h[16];
    const int iters = 20000;
#define RUN_IN(NV, M32, EX) { k_inwave<NV, M32, EX><<<256, 256>>>(out, cyc, iters); CK(hipDeviceSynchronize()); CK(hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost)); \
    printf("in-wave  %s + %d %s per MFMA: %6.1f cycles per MFMA\n", M32 ? "32x32x16" : "16x16x32", NV, EX ? "v_exp" : "v_fma", (double)h[0] / (iters * 4.0)); }
    RUN_IN(0, 0, 0) RUN_IN(1, 0, 0) RUN_IN(2, 0, 0) RUN_IN(3, 0, 0) RUN_IN(4, 0, 0) RUN_IN(5, 0, 0) RUN_IN(6, 0, 0) RUN_IN(8, 0, 0)
    RUN_IN(1, 0, 1) RUN_IN(2, 0, 1)
    RUN_IN(0, 1, 0) RUN_IN(2, 1, 0) RUN_IN(4, 1, 0) RUN_IN(6, 1, 0) RUN_IN(7, 1, 0) RUN_IN(8, 1, 0) RUN_IN(10, 1, 0) RUN_IN(2, 1, 1) RUN_IN(4, 1, 1)
#define RUN_X(A, B, what) { CK(hipMemset(cyc, 0, 256)); k_cross<A, B><<<256, 512>>>(out, cyc, iters); CK(hipDeviceSynchronize()); CK(hipMemcpy(h, cyc, 9 * 8, hipMemcpyDeviceToHost)); \
    printf("cross    %-44s wave0 %7.0f k cycles, wave4 %7.0f k cycles\n", what, h[1] / 1e3, h[5] / 1e3); }
    RUN_X(1, 0, "A: 8 MFMA/iter alone") RUN_X(0, 2, "B: 32 v_fma/iter alone") RUN_X(1, 2, "A MFMA + B v_fma together")
    RUN_X(0, 3, "B: 8 v_exp/iter alone") RUN_X(1, 3, "A MFMA + B v_exp together") RUN_X(1, 1, "A MFMA + B MFMA") RUN_X(2, 2, "A v_fma + B v_fma")
    return 0;
}
