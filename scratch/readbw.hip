// Read-only HBM streaming ceiling for a decode-attention-sized launch: 28 buffers of BYTES each, cycled (HBM-cold),
// one launch per buffer; variants over workgroup count / waves / loads in flight / non-temporal.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ void rd(const f4 *__restrict__ p, size_t n16, float *out) {
    const size_t per = n16 / gridDim.x;                       // contiguous slice per workgroup
    const f4 *b = p + per * blockIdx.x;
    f4 acc = {0, 0, 0, 0};
    for (size_t i = threadIdx.x; i + (U - 1) * blockDim.x < per; i += (size_t)U * blockDim.x) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(b + i + u * blockDim.x) : b[i + u * blockDim.x];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1.f;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main(int argc, char **argv) {
    const size_t BYTES = argc > 1 ? strtoull(argv[1], 0, 10) : 136839168ull;
    const int L = 28;
    char *pool; float *out; CK(hipMalloc(&pool, BYTES * L)); CK(hipMalloc(&out, 4));
    CK(hipMemset(pool, 1, BYTES * L));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char *name, auto kern, int wgs, int threads) {
        float best = 1e9;
        for (int rnd = 0; rnd < 4; ++rnd) {
            for (int l = 0; l < L; ++l) kern<<<wgs, threads, 0, s>>>((const f4 *)(pool + BYTES * l), BYTES / 16, out);
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < 6; ++r)
                for (int l = 0; l < L; ++l) kern<<<wgs, threads, 0, s>>>((const f4 *)(pool + BYTES * l), BYTES / 16, out);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double us = best * 1e3 / (6 * L);
        printf("%-28s wgs=%5d thr=%4d  %7.2f us  %7.1f GB/s\n", name, wgs, threads, us, BYTES / us / 1e3);
    };
    for (int thr : {256, 512, 1024}) {
        for (int wgs : {256, 512, 1024, 2048}) {
            if ((size_t)wgs * thr > 256 * 2048) continue;
            run("U4 nt", rd<4, true>, wgs, thr);
            run("U8 nt", rd<8, true>, wgs, thr);
            run("U8", rd<8, false>, wgs, thr);
        }
    }
    run("U16 nt", rd<16, true>, 256, 1024);
    run("U16 nt", rd<16, true>, 512, 512);
    run("U2 nt", rd<2, true>, 512, 1024);
    return 0;
}
