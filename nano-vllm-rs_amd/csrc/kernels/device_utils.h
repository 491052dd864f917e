// device_utils.h — shared device helpers for the gfx950 kernels (wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nvr {

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float16_t __attribute__((ext_vector_type(16)));

constexpr int kWave = 64;

// f32 -> fp16, round-to-nearest-even, as a standalone conversion.  The empty asm makes the f32 value
// opaque so that hipcc (-ffp-contract=fast) cannot fold the producing multiply/add into a single-rounding
// v_fma_mixlo_f16: the oracle rounds twice (f32 op, then fp16), and bit-exact parity needs the same.
__device__ __forceinline__ half_t to_half_rn(float f) {
    asm volatile("" : "+v"(f));
    return (half_t)f;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// synthetic weight generator — bit-identical twin of oracle/nvr_oracle.c:weight_value
__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ float weight_value(uint64_t key, uint64_t idx, float scale) {
    uint64_t r = splitmix64(key ^ idx);
    int32_t s = (int32_t)(r & 0xFFFF) + (int32_t)((r >> 16) & 0xFFFF) + (int32_t)((r >> 32) & 0xFFFF) +
                (int32_t)((r >> 48) & 0xFFFF) - 131070;
    return (float)s * scale;
}

}  // namespace nvr
