// Dense whole-grid operators of the coarse stage and their adjoints.
//
// Reference algorithm (paths under the reference tree):
//   app/utils/base/module.py:145-177     Gaussian3DConv: nn.Conv3d(1,1,k, padding=k//2, padding_mode="replicate")
//                                        applied to the SDF grid on every forward (voxurfc.py:202)
//   app/coarse/model/voxurfc.py:597-616  neus_sdf_gradient: central differences /2 /voxel_size, zero on the
//                                        boundary layer, sampled trilinearly per sample (voxurfc.py:205-206)
//
// MI355X notes.  Both are pure streaming stencils over a grid of at most a few million cells (coarse.yaml:
// 96^3): one lane per output cell with z (the contiguous axis) on consecutive lanes, so the +-r
// neighbours along z come out of the same cache lines and the x / y neighbours are coalesced row reads out
// of L2.  The adjoints are written as GATHERS (no atomics): the adjoint of the replicate-padded correlation
// at a boundary cell collects every (output cell, tap) pair that was clamped onto it; interior cells
// reduce to the flipped kernel.  The gradient grid is produced channels-last [X,Y,Z,3] so that the
// per-sample trilinear gather of the 3-vector reads one 12-byte record per corner.
#include "esr_common.h"

namespace {

constexpr int MAX_K = 7;       // kernel size up to 7^3

struct ConvParams {
    const float *in;
    float *out;
    int gx, gy, gz, k;
    float w[MAX_K * MAX_K * MAX_K];
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// out[o] = sum_k w[k] * in[clamp(o + k - r)]      (cross-correlation, as nn.Conv3d)
__global__ void __launch_bounds__(256) gauss3d_fwd_kernel(ConvParams P)
{
    __shared__ float w[MAX_K * MAX_K * MAX_K];
    const int k = P.k, r = k / 2, k3 = k * k * k;
    for (int i = threadIdx.x; i < k3; i += blockDim.x) w[i] = P.w[i];
    __syncthreads();
    const int64_t n = (int64_t)P.gx * P.gy * P.gz;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
        const int z = (int)(idx % P.gz), y = (int)(idx / P.gz % P.gy), x = (int)(idx / ((int64_t)P.gz * P.gy));
        float acc = 0.f;
        for (int a = 0; a < k; ++a) {
            const int xx = clampi(x + a - r, 0, P.gx - 1);
            for (int b = 0; b < k; ++b) {
                const int yy = clampi(y + b - r, 0, P.gy - 1);
                const float *row = P.in + ((int64_t)xx * P.gy + yy) * P.gz;
                for (int c = 0; c < k; ++c) acc += w[(a * k + b) * k + c] * row[clampi(z + c - r, 0, P.gz - 1)];
            }
        }
        P.out[idx] = acc;
    }
}

// tap indices t in [0, 2r] of one axis with clamp(o + t - r, 0, n-1) == i: [lo, hi] (empty: lo > hi)
__device__ __forceinline__ void tap_range(int i, int o, int n, int r, int &lo, int &hi)
{
    if (n == 1) { lo = 0; hi = 2 * r; return; }
    if (i == 0) { lo = 0; hi = min(2 * r, r - o); return; }                    // everything clamped up to 0
    if (i == n - 1) { lo = max(0, n - 1 - o + r); hi = 2 * r; return; }        // everything clamped down to n-1
    lo = hi = i - o + r;
    if (lo < 0 || lo > 2 * r) { lo = 1; hi = 0; }
}

// gin[i] += sum over (o, tap) with clamp(o + tap - r) == i of w[tap] * gout[o]
__global__ void __launch_bounds__(256) gauss3d_bwd_kernel(ConvParams P)
{
    __shared__ float w[MAX_K * MAX_K * MAX_K];
    const int k = P.k, r = k / 2, k3 = k * k * k;
    for (int i = threadIdx.x; i < k3; i += blockDim.x) w[i] = P.w[i];
    __syncthreads();
    const int64_t n = (int64_t)P.gx * P.gy * P.gz;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
        const int z = (int)(idx % P.gz), y = (int)(idx / P.gz % P.gy), x = (int)(idx / ((int64_t)P.gz * P.gy));
        float acc = 0.f;
        for (int ox = max(0, x - r); ox <= min(P.gx - 1, x + r); ++ox) {
            int alo, ahi;
            tap_range(x, ox, P.gx, r, alo, ahi);
            if (alo > ahi) continue;
            for (int oy = max(0, y - r); oy <= min(P.gy - 1, y + r); ++oy) {
                int blo, bhi;
                tap_range(y, oy, P.gy, r, blo, bhi);
                if (blo > bhi) continue;
                const float *row = P.in + ((int64_t)ox * P.gy + oy) * P.gz;
                for (int oz = max(0, z - r); oz <= min(P.gz - 1, z + r); ++oz) {
                    int clo, chi;
                    tap_range(z, oz, P.gz, r, clo, chi);
                    if (clo > chi) continue;
                    float ws = 0.f;
                    for (int a = alo; a <= ahi; ++a)
                        for (int b = blo; b <= bhi; ++b)
                            for (int c = clo; c <= chi; ++c) ws += w[(a * k + b) * k + c];
                    acc += ws * row[oz];
                }
            }
        }
        P.out[idx] += acc;
    }
}

// grad[x,y,z,:] = central differences / 2 / voxel_size, zero on the boundary layer of each axis
__global__ void __launch_bounds__(256) central_grad_fwd_kernel(const float *__restrict__ s, int gx, int gy, int gz,
                                                               float voxel, float *__restrict__ out)
{
#pragma clang fp contract(off)
    const int64_t n = (int64_t)gx * gy * gz, sy = gz, sx = (int64_t)gy * gz;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
        const int z = (int)(idx % gz), y = (int)(idx / gz % gy), x = (int)(idx / sx);
        const float dx = (x > 0 && x < gx - 1) ? (s[idx + sx] - s[idx - sx]) / 2.f / voxel : 0.f;
        const float dy = (y > 0 && y < gy - 1) ? (s[idx + sy] - s[idx - sy]) / 2.f / voxel : 0.f;
        const float dz = (z > 0 && z < gz - 1) ? (s[idx + 1] - s[idx - 1]) / 2.f / voxel : 0.f;
        out[3 * idx] = dx; out[3 * idx + 1] = dy; out[3 * idx + 2] = dz;
    }
}

// gsdf[i] += (g_a[i - e_a] [i - e_a interior] - g_a[i + e_a] [i + e_a interior]) / 2 / voxel, summed over the axes
__global__ void __launch_bounds__(256) central_grad_bwd_kernel(const float *__restrict__ g, int gx, int gy, int gz,
                                                               float voxel, float *__restrict__ gsdf)
{
    const int64_t n = (int64_t)gx * gy * gz, sy = gz, sx = (int64_t)gy * gz;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
        const int z = (int)(idx % gz), y = (int)(idx / gz % gy), x = (int)(idx / sx);
        float acc = 0.f;
        if (x - 1 >= 1 && x - 1 <= gx - 2) acc += g[3 * (idx - sx)];
        if (x + 1 >= 1 && x + 1 <= gx - 2) acc -= g[3 * (idx + sx)];
        if (y - 1 >= 1 && y - 1 <= gy - 2) acc += g[3 * (idx - sy) + 1];
        if (y + 1 >= 1 && y + 1 <= gy - 2) acc -= g[3 * (idx + sy) + 1];
        if (z - 1 >= 1 && z - 1 <= gz - 2) acc += g[3 * (idx - 1) + 2];
        if (z + 1 >= 1 && z + 1 <= gz - 2) acc -= g[3 * (idx + 1) + 2];
        if (acc != 0.f) gsdf[idx] += acc / 2.f / voxel;
    }
}

int conv_params(ConvParams &P, const float *in, const float *weights, int ksize, int gx, int gy, int gz, float *out)
{
    if (ksize < 1 || ksize > MAX_K || (ksize & 1) == 0 || gx < 1 || gy < 1 || gz < 1) return ESR_EINVAL;
    if (!in || !weights || !out) return ESR_EINVAL;
    P.in = in; P.out = out; P.gx = gx; P.gy = gy; P.gz = gz; P.k = ksize;
    for (int i = 0; i < ksize * ksize * ksize; ++i) P.w[i] = weights[i];
    return 0;
}

}  // namespace

ESR_API int esr_gauss3d_fwd(const float *in, const float *weights_host, int ksize, int32_t gx, int32_t gy,
                            int32_t gz, float *out, void *stream)
{
    ConvParams P;
    const int rc = conv_params(P, in, weights_host, ksize, gx, gy, gz, out);
    if (rc) return rc;
    gauss3d_fwd_kernel<<<esr_grid_for((int64_t)gx * gy * gz, 256, 256 * 16), 256, 0, esr_stream(stream)>>>(P);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_gauss3d_bwd(const float *gout, const float *weights_host, int ksize, int32_t gx, int32_t gy,
                            int32_t gz, float *gin, void *stream)
{
    ConvParams P;
    const int rc = conv_params(P, gout, weights_host, ksize, gx, gy, gz, gin);
    if (rc) return rc;
    gauss3d_bwd_kernel<<<esr_grid_for((int64_t)gx * gy * gz, 256, 256 * 16), 256, 0, esr_stream(stream)>>>(P);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_central_grad_fwd(const float *sdf, int32_t gx, int32_t gy, int32_t gz, float voxel_size,
                                 float *grad, void *stream)
{
    if (gx < 1 || gy < 1 || gz < 1 || !sdf || !grad) return ESR_EINVAL;
    central_grad_fwd_kernel<<<esr_grid_for((int64_t)gx * gy * gz, 256, 256 * 16), 256, 0, esr_stream(stream)>>>(
        sdf, gx, gy, gz, voxel_size, grad);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_central_grad_bwd(const float *ggrad, int32_t gx, int32_t gy, int32_t gz, float voxel_size,
                                 float *gsdf, void *stream)
{
    if (gx < 1 || gy < 1 || gz < 1 || !ggrad || !gsdf) return ESR_EINVAL;
    central_grad_bwd_kernel<<<esr_grid_for((int64_t)gx * gy * gz, 256, 256 * 16), 256, 0, esr_stream(stream)>>>(
        ggrad, gx, gy, gz, voxel_size, gsdf);
    ESR_CHECK_LAUNCH();
    return 0;
}
