// handoff.hip — what does a producer -> consumer hand-off between workgroups of ONE persistent launch cost on gfx950?
// (decides whether a persistent decode-chain kernel can beat one launch per GEMM: a kernel boundary costs ~2.0 us.)
// 256 workgroups (one per CU); one round = the three hand-offs of a decoder layer's GEMM chain with 1 KiB output tiles:
// A (NA workgroups) -> B (256-NA workgroups, each needs all of A's block) -> C (NA, each needs all of B's) -> everybody needs C's.
// Every consumer reduces the block it read into the tile it publishes next (true dependency).  Variants:
//   0 flag : sc1 stores, drain, one flag word per producer (value = round); consumer polls the flags with sc1 loads, then
//            reads the block with sc1 loads
//   1 ll   : every 8 bytes carry {4 data bytes, round}: consumer polls the data itself (2x the bytes, one round trip)
//   2 inv  : like 0, but after the flags: buffer_inv sc1, then plain (L2-cached) loads
//   3 ll16 : 16-byte units {12 data bytes, round} (relies on 16-byte stores landing whole)
// build: hipcc -O3 --offload-arch=gfx950 scratch/handoff.hip -o scratch/handoff ; run: scratch/handoff [rounds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int THREADS = 512;
constexpr unsigned SPIN_MAX = 1u << 15;

template <int VAR>
__device__ __forceinline__ void produce(unsigned *data, unsigned *flags, int me, unsigned carry, unsigned r, int tid) {
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(data, 0, 1 << 22, 0x00020000);
    if (VAR == 0 || VAR == 2) {
        if (tid < 64) {
            u4 v = {carry + tid, carry ^ r, (unsigned)tid, 1u};
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, me * 1024 + tid * 16, 0, 16);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (tid == 0) __hip_atomic_store(flags + me, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else if (VAR == 1) {
        if (tid < 256) {
            u2 v = {carry + tid, r};
            __builtin_amdgcn_raw_buffer_store_b64(v, rs, me * 2048 + tid * 8, 0, 16);
        }
    } else {
        if (tid < 86) {                                        // 86 units x 12 B >= 1 KiB
            u4 v = {carry + tid, carry ^ r, (unsigned)tid, r};
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, me * 1376 + tid * 16, 0, 16);
        }
    }
}

// MAXU: 16-byte loads per thread, all in flight at once (like the B-operand fragments of a GEMM wave); a unit that has not
// arrived yet makes the thread re-issue its whole batch
template <int VAR, int MAXU>
__device__ __forceinline__ unsigned consume(unsigned *data, unsigned *flags, int P, unsigned r, int tid, unsigned *err) {
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(data, 0, 1 << 22, 0x00020000);
    unsigned acc = 0;
    if (VAR == 0 || VAR == 2) {
        if (tid < 64) {
            const auto fr = __builtin_amdgcn_make_buffer_rsrc(flags, 0, 1024, 0x00020000);
            unsigned spins = 0;
            for (;;) {
                u4 f = __builtin_amdgcn_raw_buffer_load_b128(fr, tid * 16, 0, 16);
                asm volatile("" ::: "memory");
                bool ok = true;
#pragma unroll
                for (int e = 0; e < 4; ++e) ok = ok && (tid * 4 + e >= P || f[e] == r);
                if (__all(ok)) break;
                if (++spins > SPIN_MAX) { if (tid == 0) *err = 1; break; }
            }
            if (VAR == 2) asm volatile("buffer_inv sc1" ::: "memory");
        }
        __syncthreads();
        const int units = P * 64;                              // 16-byte units
        u4 v[MAXU];
#pragma unroll
        for (int i = 0; i < MAXU; ++i) {
            const int u = tid + i * THREADS;
            if (u < units) v[i] = VAR == 2 ? *reinterpret_cast<const u4 *>(reinterpret_cast<const char *>(data) + u * 16)
                                           : __builtin_amdgcn_raw_buffer_load_b128(rs, u * 16, 0, 16);
            else v[i] = (u4){0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < MAXU; ++i) acc += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    } else {
        // LL: 8-byte units {data, round} two per load (VAR 1: payload 8 of 16 bytes) or 16-byte units {3 data, round} (VAR 3)
        const int units = VAR == 1 ? P * 128 : P * 86;         // 16-byte loads
        unsigned spins = 0;
        for (;;) {
            u4 v[MAXU];
#pragma unroll
            for (int i = 0; i < MAXU; ++i) {
                const int u = tid + i * THREADS;
                const int off = VAR == 1 ? u * 16 : (u / 86) * 1376 + (u % 86) * 16;
                if (u < units) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);
                else v[i] = (u4){0, r, 0, r};
            }
            asm volatile("" ::: "memory");
            bool ok = true;
            unsigned a2 = 0;
#pragma unroll
            for (int i = 0; i < MAXU; ++i) {
                if (VAR == 1) { ok = ok && v[i][1] == r && v[i][3] == r; a2 += v[i][0] + v[i][2]; }
                else { ok = ok && v[i][3] == r; a2 += v[i][0] + v[i][1] + v[i][2]; }
            }
            if (ok) { acc = a2; break; }
            if (++spins > SPIN_MAX) { *err = 2; break; }
        }
    }
    return acc;
}

__device__ __forceinline__ unsigned wg_sum(unsigned acc, unsigned *red, int tid) {
#pragma unroll
    for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o, 64);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    unsigned tot = 0;
    for (int w = 0; w < THREADS / 64; ++w) tot += red[w];
    return tot;
}

// one round = the decode chain's three hand-offs: A (NA tiles: o_proj) -> B (256-NA workgroups: gate_up) -> C (NA: down) -> all (qkv)
template <int VAR>
__global__ __launch_bounds__(THREADS) void handoff_kernel(unsigned *bufA, unsigned *bufB, unsigned *bufC, unsigned *flags, int NA,
                                                          int R, unsigned *out, unsigned *err) {
    const int wg = blockIdx.x, tid = threadIdx.x, NB = gridDim.x - NA;
    __shared__ unsigned red[THREADS / 64];
    unsigned carry = wg;
    for (unsigned r = 1; r <= (unsigned)R; ++r) {
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;   // somebody timed out: everybody leaves
        if (wg < NA) {
            produce<VAR>(bufA, flags, wg, carry, r, tid);
            carry = carry * 1664525u + wg_sum(consume<VAR, (VAR == 1 ? 2 : 1) * 24>(bufB, flags + 256, NB, r, tid, err), red, tid);
            produce<VAR>(bufC, flags + 512, wg, carry, r, tid);
        } else {
            carry = carry * 1664525u + wg_sum(consume<VAR, (VAR == 1 ? 2 : 1) * 16>(bufA, flags, NA, r, tid, err), red, tid);
            produce<VAR>(bufB, flags + 256, wg - NA, carry, r, tid);
        }
        carry = carry * 1664525u + wg_sum(consume<VAR, (VAR == 1 ? 2 : 1) * 16>(bufC, flags + 512, NA, r, tid, err), red, tid);
    }
    if (tid == 0) out[wg] = carry;
}

int main(int argc, char **argv) {
    int R = argc > 1 ? atoi(argv[1]) : 300;
    unsigned *bA, *bB, *bC, *flags, *out, *err;
    CK(hipMalloc(&bA, 1 << 22)); CK(hipMalloc(&bB, 1 << 22)); CK(hipMalloc(&bC, 1 << 22));
    CK(hipMalloc(&flags, 4096)); CK(hipMalloc(&out, 4096)); CK(hipMalloc(&err, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[] = {"flag + sc1 read", "LL 8-byte", "flag + buffer_inv + cached read", "LL 16-byte"};
    for (int NA : {64, 128}) {
        printf("round = A(%d tiles) -> B(%d workgroups read %d KiB) -> C(%d read %d KiB) -> all 256 read %d KiB\n", NA, 256 - NA, NA, NA,
               256 - NA, NA);
        for (int var = 0; var < 4; ++var) {
            float best = 1e9f;
            unsigned herr = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemset(bA, 0, 1 << 22)); CK(hipMemset(bB, 0, 1 << 22)); CK(hipMemset(bC, 0, 1 << 22));
                CK(hipMemset(flags, 0, 4096)); CK(hipMemset(err, 0, 4));
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, 0));
                switch (var) {
                    case 0: handoff_kernel<0><<<256, THREADS>>>(bA, bB, bC, flags, NA, R, out, err); break;
                    case 1: handoff_kernel<1><<<256, THREADS>>>(bA, bB, bC, flags, NA, R, out, err); break;
                    case 2: handoff_kernel<2><<<256, THREADS>>>(bA, bB, bC, flags, NA, R, out, err); break;
                    default: handoff_kernel<3><<<256, THREADS>>>(bA, bB, bC, flags, NA, R, out, err); break;
                }
                CK(hipEventRecord(e1, 0));
                CK(hipDeviceSynchronize());
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
                CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
            }
            printf("   %-34s %7.3f us / round (3 hand-offs)%s\n", names[var], best * 1e3f / R, herr ? "   (SPIN TIMEOUT)" : "");
            fflush(stdout);
        }
    }
    return 0;
}
