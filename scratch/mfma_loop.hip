#include <hip/hip_runtime.h>
#include <chrono>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4_ __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(float *out, int iters) {
    half8 a[4], b[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) { a[i][e] = (_Float16)(0.37f + 0.001f * ((threadIdx.x * 7 + e * 13 + i * 3) % 97)); b[i][e] = (_Float16)(-0.21f + 0.002f * ((threadIdx.x * 5 + e * 11 + i) % 89)); }
    float4_ acc[8] = {};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u & 3], b[(u + 1) & 3], acc[u & 7], 0, 0, 0);
    float s = 0; for (int u = 0; u < 8; ++u) s += acc[u][0];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main() { float *o; hipMalloc(&o, 256 * 512 * 4); auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 6.0) { k<<<256, 512>>>(o, 60000); hipDeviceSynchronize(); } return 0; }
