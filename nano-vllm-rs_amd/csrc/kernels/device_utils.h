// device_utils.h — shared device helpers for the gfx950 kernels (wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nvr {

// The 16-bit storage / MFMA operand type of a kernel file is chosen when the file is COMPILED: every kernel source is built twice,
// once for fp16 (namespace nvr::k) and once with -DNVR_BF16 for bfloat16 (namespace nvr::kb; Config.dtype = "bfloat16",
// reference src/config.rs:51,113-116).  half_t is that type in both builds; the few instructions that name the type go through the
// wrappers below (same MFMA rate, same LDS images and swizzles: both are 16-bit).
#ifdef NVR_BF16
typedef __bf16 half_t;
typedef __bf16 half2_t __attribute__((ext_vector_type(2)));
typedef __bf16 half4_t __attribute__((ext_vector_type(4)));
typedef __bf16 half8_t __attribute__((ext_vector_type(8)));
#define NVR_DT_NS kb
#else
typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
#define NVR_DT_NS k
#endif
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef float float16_t __attribute__((ext_vector_type(16)));

constexpr int kWave = 64;

// D = A(16 x 32) . B(32 x 16) + C on the matrix cores, f32 accumulate (v_mfma_f32_16x16x32_f16 / _bf16)
__device__ __forceinline__ float4_t mfma16(half8_t a, half8_t b, float4_t c) {
#ifdef NVR_BF16
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#endif
}
// c + a.x * b.x + a.y * b.y in f32 (v_dot2_f32_f16 / v_dot2c_f32_bf16)
__device__ __forceinline__ float dot2(half2_t a, half2_t b, float c) {
#ifdef NVR_BF16
    return __builtin_amdgcn_fdot2_f32_bf16(a, b, c, false);
#else
    return __builtin_amdgcn_fdot2(a, b, c, false);
#endif
}
// sigmoid of SiluAndMul (activation.rs:46-63) for every kernel that applies it (the fused and the unfused forms must agree bit for bit):
// v_exp_f32 + v_rcp_f32 (1 ulp each in f32, far below the fp16 / bf16 rounding that follows) instead of the ~10-instruction IEEE division
// — the epilogue of the 256^2 gate_up GEMM is not overlapped with matrix work: -3..5 % on that kernel.
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// Activation block x[T, cols] (row stride ldx, columns col0 .. col0 + 8*cpr) -> LDS image of ROWS rows x cpr 16-byte chunks, chunk ch of
// row r at slot ch ^ (r & 7) (conflict-free ds_read_b128 of MFMA B fragments); rows >= T repeat row T-1.  FB loads are requested per
// thread before the first LDS write: written as a plain load / store loop hipcc keeps ONE load in flight per thread (load, s_waitcnt
// vmcnt(0), ds_write, branch), i.e. one L2 round trip per 8 KiB of the image (r03: 16 round trips per 128 KiB chunk of the streaming
// GEMMs, 8 in front of the LM head).
template <int ROWS, int THREADS, int FB>
__device__ __forceinline__ void fill_x_image(char *smem, const half_t *__restrict__ x, int64_t ldx, int col0, int cpr, int T, int tid, int rot = 0) {
    const int total = ROWS * cpr;
    for (int c0 = tid; c0 < total; c0 += THREADS * FB) {
        half8_t v[FB];
        int slot[FB];
#pragma unroll
        for (int f = 0; f < FB; ++f) {
            const int c = c0 + f * THREADS;
            int cc = c < total ? c : total - 1;                                   // (clamped: the load is unconditional, only the write is guarded)
            cc += rot; if (cc >= total) cc -= total;                              // rot < total: where this workgroup starts its walk over the block
            const int row = cc / cpr, ch = cc - row * cpr;
            const int m = row < T ? row : T - 1;
            v[f] = *reinterpret_cast<const half8_t *>(x + (int64_t)m * ldx + col0 + ch * 8);
            slot[f] = c < total ? row * cpr + (ch ^ (row & 7)) : -1;
        }
#pragma unroll
        for (int f = 0; f < FB; ++f)
            if (slot[f] >= 0) *reinterpret_cast<half8_t *>(smem + (int64_t)slot[f] * 16) = v[f];
    }
}
// ds_read_b64_tr_b16: the hardware-transposed LDS read of 4 rows x 16 columns of 16-bit elements (cdna guide T10)
__device__ __forceinline__ half4_t lds_read_tr16(const char *lds_addr) {
#ifdef NVR_BF16
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) half4_t *)lds_addr);
#else
    typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
    const fp16x4_t r = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t *)lds_addr);
    return (half4_t){(half_t)r[0], (half_t)r[1], (half_t)r[2], (half_t)r[3]};
#endif
}

// f32 -> fp16 (bf16 build: -> bfloat16), round-to-nearest-even, as a standalone conversion.  The empty asm makes the f32 value
// opaque so that hipcc (-ffp-contract=fast) cannot fold the producing multiply/add into a single-rounding
// v_fma_mixlo_f16: the oracle rounds twice (f32 op, then fp16), and bit-exact parity needs the same.
// a*b - c*d and a*b + c*d with every product and the sum rounded on its own (RotaryEmbedding::apply_rotary_emb_single, rotary_embedding.rs:128-158, as the
// oracle evaluates it): HIP's __fmul_rn / __fadd_rn / __fsub_rn are plain * + - and, under the device default -ffp-contract=fast, may or may not become an fma
// depending on the kernel they are inlined into (r05: two instantiations of one attention loop disagreed by 1 ulp) — the RoPE rotation exists in five kernels
// whose results are promised to agree bit for bit, so contraction is switched off where it is written down.
__device__ __forceinline__ float mul_sub_unfused(float a, float b, float c, float d) {
#pragma clang fp contract(off)
    const float x = a * b, y = c * d;
    return x - y;
}
__device__ __forceinline__ float mul_add_unfused(float a, float b, float c, float d) {
#pragma clang fp contract(off)
    const float x = a * b, y = c * d;
    return x + y;
}
__device__ __forceinline__ half_t to_half_rn(float f) {
    asm volatile("" : "+v"(f));
    return (half_t)f;
}

// sum over the LPR (= 16 or 8) consecutive lanes that hold one K row, result in every lane of the group: DPP row
// rotations / quad permutes fused into v_add_f32 (no LDS traffic; __shfl_xor compiles to ds_bpermute_b32)
template <int CTRL>
__device__ __forceinline__ float dpp_add(float x) {
    const int y = __builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false);
    return x + __int_as_float(y);
}
template <int LPR>
__device__ __forceinline__ float row_sum(float x) {
    if (LPR == 16) { x = dpp_add<0x128>(x); x = dpp_add<0x124>(x); }     // row_ror:8, row_ror:4
    else x = dpp_add<0x141>(x);                                            // row_half_mirror (8 lanes: i <-> 7-i)
    x = dpp_add<0x4E>(x);                                                  // quad_perm [2,3,0,1]
    x = dpp_add<0xB1>(x);                                                  // quad_perm [1,0,3,2]
    return x;
}

// lane-wise butterflies across the 16-lane rows of a wave on the gfx950 lane-swap instructions (no LDS traffic):
// v_permlane16_swap exchanges the odd rows of one register with the even rows of another, v_permlane32_swap the
// upper half of one with the lower half of the other; fed the same value twice, the two results are x and its
// xor-16 (xor-32) partner.
__device__ __forceinline__ float xor16_partner_sum(float x) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_partner_sum(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor16_partner_max(float x) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor32_partner_max(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
// all-lane sum / max of a wave: xor-32 and xor-16 partners on the lane-swap instructions, then the 16-lane row on DPP —
// the same pairing order as a 32,16,8,4,2,1 xor butterfly (bit-identical sums), without its six ds_bpermute round trips
__device__ __forceinline__ float wave_sum(float v) { return row_sum<16>(xor16_partner_sum(xor32_partner_sum(v))); }
template <int CTRL>
__device__ __forceinline__ float dpp_max(float x) {
    return fmaxf(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false)));
}
__device__ __forceinline__ float wave_max(float v) {
    v = xor16_partner_max(xor32_partner_max(v));
    v = dpp_max<0x128>(v); v = dpp_max<0x124>(v); v = dpp_max<0x4E>(v); v = dpp_max<0xB1>(v);
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
