// comm_local.hip — the device half of the in-process communicator (comm.h: LocalGroup): out = fp16(sum_r in_r), f32 accumulation
// in rank order.  Test / bring-up backend only; the product exchange is RCCL (comm.cpp).
#include "kernels.h"
#include "device_utils.h"
#include "../common.h"

namespace nvr { namespace NVR_DT_NS {

struct LocalPtrs { const half_t *p[8]; };

__global__ void local_sum_kernel(LocalPtrs in, int n, half_t *__restrict__ out, size_t count) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        float acc = (float)in.p[0][i];
        for (int r = 1; r < n; ++r) acc += (float)in.p[r][i];
        out[i] = to_half_rn(acc);
    }
}

int local_sum_16(const void *const *ptrs, int n, void *out, size_t count, hipStream_t s) {
    if (n < 1 || n > 8) return fail(NVR_ERR_INVALID_ARG, "local_sum_16: %d ranks (1..8)", n);
    if (count == 0) return NVR_OK;
    LocalPtrs lp{};
    for (int r = 0; r < n; ++r) lp.p[r] = (const half_t *)ptrs[r];
    const unsigned blocks = (unsigned)((count + 255) / 256 < 1024 ? (count + 255) / 256 : 1024);
    local_sum_kernel<<<dim3(blocks), dim3(256), 0, s>>>(lp, n, (half_t *)out, count);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(NVR_ERR_HIP, "local_sum launch failed: %s", hipGetErrorString(e));
    return NVR_OK;
}

}}  // namespace nvr::NVR_DT_NS
