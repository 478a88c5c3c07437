// Total-variation gradient, in place: drop-in for
// total_variation_cuda.total_variation_add_grad
// (reference: app/utils/base/cuda/total_variation_kernel.cu:13-35,68-98).
//
// Pure HBM streaming (read param + read/modify/write grad).  The fastest axis
// (k) is walked by consecutive lanes so the +-1 neighbours are served from the
// same cache lines; the +-sz_k and +-sz_k*sz_j neighbours are coalesced row reads.
#include "esr_common.h"

namespace {

__device__ __forceinline__ float clamp1(float v) { return fminf(fmaxf(v, -1.f), 1.f); }

template <bool DENSE>
__global__ void __launch_bounds__(256) tv_add_grad_kernel(const float *__restrict__ param,
                                                          float *__restrict__ grad, float wy, float wz,
                                                          int64_t sz_i, int64_t sz_j, int64_t sz_k,
                                                          int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t plane = sz_k * sz_j;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < n; idx += stride) {
        const float g0 = grad[idx];
        if (!DENSE && g0 == 0.f) continue;
        const int64_t k = idx % sz_k, j = idx / sz_k % sz_j, i = idx / plane % sz_i;
        const float p = param[idx];
        float g = 0.f;
        // same accumulation order as the reference; note wz on BOTH the k and i axes
        g += (k == 0)        ? 0.f : wz * clamp1(p - param[idx - 1]);
        g += (k == sz_k - 1) ? 0.f : wz * clamp1(p - param[idx + 1]);
        g += (j == 0)        ? 0.f : wy * clamp1(p - param[idx - sz_k]);
        g += (j == sz_j - 1) ? 0.f : wy * clamp1(p - param[idx + sz_k]);
        g += (i == 0)        ? 0.f : wz * clamp1(p - param[idx - plane]);
        g += (i == sz_i - 1) ? 0.f : wz * clamp1(p - param[idx + plane]);
        grad[idx] = g0 + g;
    }
}

}  // namespace

ESR_API int esr_tv_add_grad(const float *param, float *grad, float wx, float wy, float wz,
                            int64_t sz_i, int64_t sz_j, int64_t sz_k, int64_t n, int dense_mode,
                            void *stream)
{
    (void)wx;  // unused by the reference kernel as well
    if (n < 0 || sz_i < 1 || sz_j < 1 || sz_k < 1) return ESR_EINVAL;
    if (n == 0) return 0;
    if (!param || !grad) return ESR_EINVAL;
    wy /= 6;
    wz /= 6;
    const int grid = esr_grid_for(n, 256, 256 * 16);
    if (dense_mode)
        tv_add_grad_kernel<true><<<grid, 256, 0, esr_stream(stream)>>>(param, grad, wy, wz, sz_i, sz_j, sz_k, n);
    else
        tv_add_grad_kernel<false><<<grid, 256, 0, esr_stream(stream)>>>(param, grad, wy, wz, sz_i, sz_j, sz_k, n);
    ESR_CHECK_LAUNCH();
    return 0;
}
