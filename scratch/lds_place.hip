// microbenchmark: cost of placing one workgroup per CU as a function of its LDS allocation and thread count (empty kernel body),
// as N back-to-back launches inside one captured graph (what a decode step's kernel boundary costs at least)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
extern __shared__ char dyn[];
__global__ void k_empty(int *out, int touch) {
    if (touch && threadIdx.x == 0) { dyn[0] = 1; if (dyn[0] == 7) out[blockIdx.x] = 1; }
}
int main() {
    int *out; CK(hipMalloc(&out, 4096 * 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    const int reps = 200;
    for (int threads : {256, 512}) for (int grid : {256, 512}) for (int lds : {0, 16, 48, 64, 96, 128, 160}) {
        CK(hipFuncSetAttribute((const void *)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < reps; ++i) k_empty<<<grid, threads, lds * 1024, s>>>(out, lds > 0);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        float best = 1e9;
        for (int r = 0; r < 5; ++r) { CK(hipEventRecord(a, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
        printf("threads %3d grid %3d lds %3d KiB: %.2f us per launch\n", threads, grid, lds, best * 1000 / reps);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
