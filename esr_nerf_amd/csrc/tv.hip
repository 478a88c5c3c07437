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

// (x, y, z) of linear cell i in a [.., gy, gz] grid.  gfx950 has no integer divide: a 64-bit division by a run-time value is
// ~150 instructions, a 32-bit one ~25 -- every grid of the reference's configurations has < 2^31 cells.
__device__ __forceinline__ void cell_xyz(int64_t i, int gy, int gz, int &x, int &y, int &z)
{
    if (i < 0x7fffffffLL) {
        const unsigned u = (unsigned)i, q = u / (unsigned)gz, q2 = q / (unsigned)gy;
        z = (int)(u - q * (unsigned)gz); y = (int)(q - q2 * (unsigned)gy); x = (int)q2;
    } else {
        z = (int)(i % gz); y = (int)(i / gz % gy); x = (int)(i / ((int64_t)gz * gy));
    }
}

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
        int i, j, k;
        cell_xyz(idx, (int)sz_j, (int)sz_k, i, j, k);
        i = (int)((unsigned)i % (unsigned)sz_i);           // (several channels stacked along the slowest axis)
        // The six neighbours are loaded UNCONDITIONALLY (a face cell re-reads itself): a load behind a condition is an
        // exec-masked region with its own wait -- six serial round trips per cell (esr_common.h: esr_ld_or0).
        const float p = param[idx];
        const float n0 = param[idx - (k == 0 ? 0 : 1)], n1 = param[idx + (k == sz_k - 1 ? 0 : 1)];
        const float n2 = param[idx - (j == 0 ? 0 : sz_k)], n3 = param[idx + (j == sz_j - 1 ? 0 : sz_k)];
        const float n4 = param[idx - (i == 0 ? 0 : plane)], n5 = param[idx + (i == sz_i - 1 ? 0 : plane)];
        float g = 0.f;
        // same accumulation order as the reference; note wz on BOTH the k and i axes
        g += (k == 0)        ? 0.f : wz * clamp1(p - n0);
        g += (k == sz_k - 1) ? 0.f : wz * clamp1(p - n1);
        g += (j == 0)        ? 0.f : wy * clamp1(p - n2);
        g += (j == sz_j - 1) ? 0.f : wy * clamp1(p - n3);
        g += (i == 0)        ? 0.f : wz * clamp1(p - n4);
        g += (i == sz_i - 1) ? 0.f : wz * clamp1(p - n5);
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

// ---------------------------------------------------------------------------------------------------------------
// Smoothed-gradient TV term of the fine / lts trainers, forward AND backward:
//   gradient = neus_sdf_gradient()                       (app/fine/model/voxurff.py:723-742, interior central differences)
//   err      = tv_smooth_conv(gradient).detach() - gradient        (voxurff.py:611-613; 3x3x3 replicate-padded conv,
//                                                                    app/utils/base/module.py:180-211)
//   loss     = mean(err[nonempty_mask x 3]^2) * smooth_grad_tv      (voxurff.py:614-617), every tv_every-th iteration
// The reference runs this as ~15 dense torch kernels over three-channel copies of the grid plus their autograd
// twins.  Here: pass 1 writes the gradient field, pass 2 the masked error field and the loss (block reduction + one
// atomic), pass 3 gathers d loss / d sdf through the central differences (the conv branch is detached, so the
// adjoint is the difference stencil only) -- three streaming stencil passes, z on consecutive lanes.
namespace {

struct TvGrid { int gx, gy, gz; float voxel_size; };

__device__ __forceinline__ float central(const float *__restrict__ s, const TvGrid G, int x, int y, int z, int c)
{
#pragma clang fp contract(off)
    const int64_t i = ((int64_t)x * G.gy + y) * G.gz + z;
    const int64_t st = c == 0 ? (int64_t)G.gy * G.gz : (c == 1 ? G.gz : 1);
    const int p = c == 0 ? x : (c == 1 ? y : z), n = c == 0 ? G.gx : (c == 1 ? G.gy : G.gz);
    const bool interior = p >= 1 && p <= n - 2;
    const int64_t so = interior ? st : 0;                       // (unconditional loads: see tv_add_grad_kernel)
    const float v = __fdiv_rn(__fdiv_rn(s[i + so] - s[i - so], 2.0f), G.voxel_size);
    return interior ? v : 0.f;
}

__global__ void __launch_bounds__(256) tv_gradient_kernel(const float *__restrict__ sdf, TvGrid G, float *__restrict__ g)
{
    const int64_t n = (int64_t)G.gx * G.gy * G.gz;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int x, y, z;
        cell_xyz(i, G.gy, G.gz, x, y, z);
#pragma unroll
        for (int c = 0; c < 3; ++c) g[c * n + i] = central(sdf, G, x, y, z, c);
    }
}

struct Conv27 { float w[27]; float bias; };

__global__ void __launch_bounds__(256) tv_error_kernel(const float *__restrict__ g, const uint8_t *__restrict__ mask,
                                                       TvGrid G, Conv27 K, float *__restrict__ err,
                                                       float inv_count, float *__restrict__ loss)
{
    const int64_t n = (int64_t)G.gx * G.gy * G.gz;
    float part = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int x, y, z;
        cell_xyz(i, G.gy, G.gz, x, y, z);
        const bool m = mask[i] != 0;
        // The 27 taps are loaded UNCONDITIONALLY (clamped addresses are always valid) and the mask selects the VALUE: behind
        // `if (m)` every load was an exec-masked region with a wait of its own -- 73 waits for 84 loads, 72 us for 4.2 M cells
        // (esr_common.h: esr_ld_or0).  Row offsets of the nine (dx, dy) columns once per cell, shared by the three components.
        const int xs[3] = {max(x - 1, 0), x, min(x + 1, G.gx - 1)}, ys[3] = {max(y - 1, 0), y, min(y + 1, G.gy - 1)};
        const int zs[3] = {max(z - 1, 0), z, min(z + 1, G.gz - 1)};
        int row[9];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) row[a * 3 + b] = (xs[a] * G.gy + ys[b]) * G.gz;       // (< 2^31 cells: cell_xyz)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float *__restrict__ gc = g + c * n;
            // all 27 loads of a component are issued before the first is consumed (left to itself the compiler reuses ONE
            // register: load, wait, multiply-add, 81 memory round trips per cell)
            float t[27];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b)
#pragma unroll
                    for (int d = 0; d < 3; ++d) t[a * 9 + b * 3 + d] = gc[row[a * 3 + b] + zs[d]];
            const float centre = gc[i];
            asm volatile("" ::: "memory");
            float s = K.bias;
#pragma unroll
            for (int k = 0; k < 27; ++k) s += K.w[k] * t[k];
            const float e = m ? s - centre : 0.f;
            part += e * e;
            err[c * n + i] = e;                      // zero outside the mask: pass 3 needs no mask
        }
    }
    // block sum -> one atomic
    __shared__ float red[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss, ((red[0] + red[1]) + (red[2] + red[3])) * inv_count);
}

// d loss / d gradient_c = -2 * err_c * scale / count (conv branch detached); gradient_c[p] = (s[p+e_c] - s[p-e_c]) / 2 / vs
// for interior p, so  d loss / d s[q] = (dg_c[q - e_c] - dg_c[q + e_c]) / 2 / vs  summed over c, interior sources only.
__global__ void __launch_bounds__(256) tv_error_bwd_kernel(const float *__restrict__ err, TvGrid G, float coeff,
                                                           const float *__restrict__ grad_out,
                                                           float *__restrict__ grad_sdf)
{
    const int64_t n = (int64_t)G.gx * G.gy * G.gz;
    if (grad_out) coeff *= grad_out[0];
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int x, y, z;
        cell_xyz(i, G.gy, G.gz, x, y, z);
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int64_t st = c == 0 ? (int64_t)G.gy * G.gz : (c == 1 ? G.gz : 1);
            const int p = c == 0 ? x : (c == 1 ? y : z), m = c == 0 ? G.gx : (c == 1 ? G.gy : G.gz);
            const bool lo = p - 1 >= 1 && p - 1 <= m - 2, hi = p + 1 >= 1 && p + 1 <= m - 2;
            const float el = err[c * n + i - (lo ? st : 0)], eh = err[c * n + i + (hi ? st : 0)];   // (unconditional loads)
            if (lo) acc += el;
            if (hi) acc -= eh;
        }
        grad_sdf[i] += coeff * acc;
    }
}

}  // namespace

ESR_API int esr_smooth_grad_tv_fwd(const float *sdf, const uint8_t *mask, const float *conv_w27, float conv_bias,
                                   int32_t gx, int32_t gy, int32_t gz, float voxel_size, int64_t masked_cells,
                                   float weight, float *work6, float *loss, void *stream)
{
    if (gx < 1 || gy < 1 || gz < 1 || masked_cells < 0) return ESR_EINVAL;
    if (!sdf || !mask || !conv_w27 || !work6 || !loss) return ESR_EINVAL;
    const int64_t n = (int64_t)gx * gy * gz;
    TvGrid G = {gx, gy, gz, voxel_size};
    Conv27 K;
    for (int i = 0; i < 27; ++i) K.w[i] = conv_w27[i];          // host array
    K.bias = conv_bias;
    const int grid = esr_grid_for(n, 256, 256 * 16);
    hipStream_t s = esr_stream(stream);
    tv_gradient_kernel<<<grid, 256, 0, s>>>(sdf, G, work6);
    ESR_CHECK_LAUNCH();
    // mean over the 3 * masked_cells selected elements, times the term's weight (an empty mask gives NaN in torch;
    // here it contributes nothing)
    const float inv = masked_cells > 0 ? weight / (3.0f * (float)masked_cells) : 0.f;
    tv_error_kernel<<<grid, 256, 0, s>>>(work6, mask, G, K, work6 + 3 * n, inv, loss);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_smooth_grad_tv_bwd(const float *work6, int32_t gx, int32_t gy, int32_t gz, float voxel_size,
                                   int64_t masked_cells, float weight, const float *grad_out, float *grad_sdf,
                                   void *stream)
{
    if (gx < 1 || gy < 1 || gz < 1 || masked_cells < 0) return ESR_EINVAL;
    if (!work6 || !grad_sdf) return ESR_EINVAL;
    if (masked_cells == 0) return 0;
    const int64_t n = (int64_t)gx * gy * gz;
    TvGrid G = {gx, gy, gz, voxel_size};
    const float coeff = -2.0f * weight / (3.0f * (float)masked_cells) / 2.0f / voxel_size;
    tv_error_bwd_kernel<<<esr_grid_for(n, 256, 256 * 16), 256, 0, esr_stream(stream)>>>(work6 + 3 * n, G, coeff, grad_out,
                                                                                         grad_sdf);
    ESR_CHECK_LAUNCH();
    return 0;
}
