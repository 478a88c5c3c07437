// The rest of the reference's pybind surface: the ops render_utils_cuda / total_variation_cuda EXPORT but the reference's own
// Python never calls (render_utils.cpp:171-173,175-181, total_variation.cpp:31).  They are not on the
// accelerated path -- elementwise launches, one thread per ray / point / cell, nothing to fuse -- and exist so that the shim
// (esr_nerf_amd/render_utils.py) answers every name of the two modules with the reference's arithmetic instead of an error.
//
// Arithmetic notes (fp32 instantiation of the reference's templates, restated from the kernels; oracle/esr_oracle.c has the
// same statements in plain C):
//   * the ray helpers ARE the sampler's (esr_common.h): separately rounded operations, correctly rounded divisions and sqrt;
//   * sample_bg_pts_on_rays mixes float and double exactly where the reference's literals (`1.`) do;
//   * raw2alpha_backward's min(e, 1e10) promotes the product chain to double, the result is rounded to fp32 once.
#include "esr_common.h"

namespace {

__global__ void __launch_bounds__(256) infer_t_minmax_kernel(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                             const float *__restrict__ xyz_min, const float *__restrict__ xyz_max,
                                                             float near_, float far_, int64_t n_rays,
                                                             float *__restrict__ t_min, float *__restrict__ t_max)
{
    const float bmin[3] = {xyz_min[0], xyz_min[1], xyz_min[2]}, bmax[3] = {xyz_max[0], xyz_max[1], xyz_max[2]};
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n_rays; r += (int64_t)gridDim.x * blockDim.x) {
        const float o[3] = {rays_o[3 * r], rays_o[3 * r + 1], rays_o[3 * r + 2]};
        const float d[3] = {rays_d[3 * r], rays_d[3 * r + 1], rays_d[3 * r + 2]};
        esr_ray_trange(o, d, bmin, bmax, near_, far_, t_min[r], t_max[r]);
    }
}

__global__ void __launch_bounds__(256) infer_n_samples_kernel(const float *__restrict__ rays_d, const float *__restrict__ t_min,
                                                              const float *__restrict__ t_max, float stepdist, int64_t n_rays,
                                                              int64_t *__restrict__ n_samples)
{
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n_rays; r += (int64_t)gridDim.x * blockDim.x) {
        const float d[3] = {rays_d[3 * r], rays_d[3 * r + 1], rays_d[3 * r + 2]};
        n_samples[r] = esr_ray_nsteps(t_min[r], t_max[r], esr_ray_norm(d), stepdist);
    }
}

__global__ void __launch_bounds__(256) infer_ray_start_dir_kernel(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                                  const float *__restrict__ t_min, int64_t n_rays,
                                                                  float *__restrict__ rays_start, float *__restrict__ rays_dir)
{
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n_rays; r += (int64_t)gridDim.x * blockDim.x) {
        const float o[3] = {rays_o[3 * r], rays_o[3 * r + 1], rays_o[3 * r + 2]};
        const float d[3] = {rays_d[3 * r], rays_d[3 * r + 1], rays_d[3 * r + 2]};
        float start[3], dir[3];
        esr_ray_start_dir(o, d, t_min[r], esr_ray_norm(d), start, dir);
#pragma unroll
        for (int a = 0; a < 3; ++a) { rays_start[3 * r + a] = start[a]; rays_dir[3 * r + a] = dir[a]; }
    }
}

// N_samples points per ray, uniform in the NDC parameter: dist = step / (N - 1)   (render_utils_kernel.cu:243-269)
__global__ void __launch_bounds__(256) sample_ndc_kernel(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                         const float *__restrict__ xyz_min, const float *__restrict__ xyz_max,
                                                         int n_samples, int64_t total, float *__restrict__ rays_pts,
                                                         uint8_t *__restrict__ mask_outbbox)
{
#pragma clang fp contract(off)
    const float bmin[3] = {xyz_min[0], xyz_min[1], xyz_min[2]}, bmax[3] = {xyz_max[0], xyz_max[1], xyz_max[2]};
    const float denom = (float)(n_samples - 1);
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = idx / n_samples;
        const int step = (int)(idx - r * n_samples);
        const float dist = __fdiv_rn((float)step, denom);
        float p[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            p[a] = rays_o[3 * r + a] + rays_d[3 * r + a] * dist;
            rays_pts[3 * idx + a] = p[a];
        }
        mask_outbbox[idx] = esr_out_of_box(p, bmin, bmax) ? 1 : 0;
    }
}

// Inverted-sphere background points (render_utils_kernel.cu:301-340): t_outer = t_max - 1 + 1 / (1 - step / N), the point is
// pulled back inside the unit cube's shell by o2i = R^2 / t^2 (1 - bg_preserve) + R / t bg_preserve
__global__ void __launch_bounds__(256) sample_bg_kernel(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                        const float *__restrict__ t_max, float bg_preserve, int n_samples,
                                                        int64_t total, float *__restrict__ rays_pts)
{
#pragma clang fp contract(off)
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = idx / n_samples;
        const int step = (int)(idx - r * n_samples);
        const float frac = __fdiv_rn((float)step, (float)n_samples);
        const float t_o = (float)(((double)t_max[r] - 1.0) + 1.0 / (1.0 - (double)frac));
        float q[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) q[a] = rays_o[3 * r + a] + rays_d[3 * r + a] * t_o;
        const float t_outer = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
        const float m = fmaxf(fabsf(q[0]), fmaxf(fabsf(q[1]), fabsf(q[2])));
        const float R = __fdiv_rn(t_outer, m);
        const float a1 = __fdiv_rn(R * R, t_outer * t_outer), a2 = __fdiv_rn(R, t_outer) * bg_preserve;
        const float o2i = (float)((double)a1 * (1.0 - (double)bg_preserve) + (double)a2);
#pragma unroll
        for (int a = 0; a < 3; ++a) rays_pts[3 * idx + a] = q[a] * o2i;
    }
}

// nearest-voxel lookup in a bool volume; points outside it read false (render_utils_kernel.cu:374-392)
__global__ void __launch_bounds__(256) maskcache_lookup_kernel(const uint8_t *__restrict__ world, const float *__restrict__ xyz,
                                                               const float *__restrict__ scale, const float *__restrict__ shift,
                                                               int sz_i, int sz_j, int sz_k, int64_t n_pts,
                                                               uint8_t *__restrict__ out)
{
#pragma clang fp contract(off)
    const float sc[3] = {scale[0], scale[1], scale[2]}, sh[3] = {shift[0], shift[1], shift[2]};
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n_pts; p += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)roundf(xyz[3 * p] * sc[0] + sh[0]);
        const int j = (int)roundf(xyz[3 * p + 1] * sc[1] + sh[1]);
        const int k = (int)roundf(xyz[3 * p + 2] * sc[2] + sh[2]);
        const bool in = 0 <= i && i < sz_i && 0 <= j && j < sz_j && 0 <= k && k < sz_k;
        out[p] = in ? world[((int64_t)i * sz_j + j) * sz_k + k] : 0;
    }
}

// alpha = 1 - (1 + e^(density + shift))^(-interval); e is kept for the backward (render_utils_kernel.cu:431-460).
// interval_t != NULL: one interval per point (the _nonuni forms).
__global__ void __launch_bounds__(256) raw2alpha_kernel(const float *__restrict__ density, float shift, float interval,
                                                        const float *__restrict__ interval_t, int64_t n_pts,
                                                        float *__restrict__ exp_d, float *__restrict__ alpha)
{
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n_pts; p += (int64_t)gridDim.x * blockDim.x) {
        const float e = expf(density[p] + shift);                       // can be inf
        const float iv = interval_t ? interval_t[p] : interval;
        exp_d[p] = e;
        alpha[p] = 1.f - powf(1.f + e, -iv);
    }
}

// grad = min(e, 1e10) (1 + e)^(-interval - 1) interval grad_back (render_utils_kernel.cu:504-530)
__global__ void __launch_bounds__(256) raw2alpha_bwd_kernel(const float *__restrict__ exp_d, const float *__restrict__ grad_back,
                                                            float interval, const float *__restrict__ interval_t,
                                                            int64_t n_pts, float *__restrict__ grad)
{
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n_pts; p += (int64_t)gridDim.x * blockDim.x) {
        const float e = exp_d[p];
        const float iv = interval_t ? interval_t[p] : interval;
        const double lim = (double)e < 1e10 ? (double)e : 1e10;
        grad[p] = (float)(lim * (double)powf(1.f + e, -iv - 1.f) * (double)iv * (double)grad_back[p]);
    }
}

__device__ __forceinline__ float clamp1m(float v) { return fminf(fmaxf(v, -1.f), 1.f); }

// total_variation_add_grad_new (total_variation_kernel.cu:38-66): every term times mask[cell] mask[neighbour]; unlike the live
// kernel this one uses wx on the fastest axis
template <bool DENSE>
__global__ void __launch_bounds__(256) tv_add_grad_masked_kernel(const float *__restrict__ param, float *__restrict__ grad,
                                                                 const float *__restrict__ mask, float wx, float wy, float wz,
                                                                 int64_t sz_i, int64_t sz_j, int64_t sz_k, int64_t n)
{
#pragma clang fp contract(off)
    const int64_t plane = sz_k * sz_j;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
        const float g0 = grad[idx];
        if (!DENSE && g0 == 0.f) continue;
        const int64_t k = idx % sz_k, j = idx / sz_k % sz_j, i = idx / sz_k / sz_j % sz_i;
        // (neighbours loaded unconditionally: a face cell re-reads itself -- esr_common.h: esr_ld_or0)
        const int64_t nb[6] = {idx - (k == 0 ? 0 : 1), idx + (k == sz_k - 1 ? 0 : 1), idx - (j == 0 ? 0 : sz_k),
                               idx + (j == sz_j - 1 ? 0 : sz_k), idx - (i == 0 ? 0 : plane), idx + (i == sz_i - 1 ? 0 : plane)};
        const bool edge[6] = {k == 0, k == sz_k - 1, j == 0, j == sz_j - 1, i == 0, i == sz_i - 1};
        const float w[6] = {wx, wx, wy, wy, wz, wz};
        const float p = param[idx], m = mask[idx];
        float pn[6], mn[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) { pn[q] = param[nb[q]]; mn[q] = mask[nb[q]]; }
        float g = 0.f;
#pragma unroll
        for (int q = 0; q < 6; ++q) g += edge[q] ? 0.f : w[q] * clamp1m(p - pn[q]) * m * mn[q];
        grad[idx] = g0 + g;
    }
}

}  // namespace

ESR_API int esr_infer_t_minmax(const float *rays_o, const float *rays_d, const float *xyz_min, const float *xyz_max,
                               float near_, float far_, int64_t n_rays, float *t_min, float *t_max, void *stream)
{
    if (n_rays < 0) return ESR_EINVAL;
    if (n_rays == 0) return 0;
    if (!rays_o || !rays_d || !xyz_min || !xyz_max || !t_min || !t_max) return ESR_EINVAL;
    infer_t_minmax_kernel<<<esr_grid_for(n_rays, 256, 256 * 8), 256, 0, esr_stream(stream)>>>(
        rays_o, rays_d, xyz_min, xyz_max, near_, far_, n_rays, t_min, t_max);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_infer_n_samples(const float *rays_d, const float *t_min, const float *t_max, float stepdist, int64_t n_rays,
                                int64_t *n_samples, void *stream)
{
    if (n_rays < 0) return ESR_EINVAL;
    if (n_rays == 0) return 0;
    if (!rays_d || !t_min || !t_max || !n_samples) return ESR_EINVAL;
    infer_n_samples_kernel<<<esr_grid_for(n_rays, 256, 256 * 8), 256, 0, esr_stream(stream)>>>(rays_d, t_min, t_max, stepdist,
                                                                                                 n_rays, n_samples);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_infer_ray_start_dir(const float *rays_o, const float *rays_d, const float *t_min, int64_t n_rays,
                                    float *rays_start, float *rays_dir, void *stream)
{
    if (n_rays < 0) return ESR_EINVAL;
    if (n_rays == 0) return 0;
    if (!rays_o || !rays_d || !t_min || !rays_start || !rays_dir) return ESR_EINVAL;
    infer_ray_start_dir_kernel<<<esr_grid_for(n_rays, 256, 256 * 8), 256, 0, esr_stream(stream)>>>(rays_o, rays_d, t_min, n_rays,
                                                                                                     rays_start, rays_dir);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_sample_ndc_pts(const float *rays_o, const float *rays_d, const float *xyz_min, const float *xyz_max,
                               int32_t n_samples, int64_t n_rays, float *rays_pts, uint8_t *mask_outbbox, void *stream)
{
    if (n_rays < 0 || n_samples < 0) return ESR_EINVAL;
    const int64_t total = n_rays * n_samples;
    if (total == 0) return 0;
    if (!rays_o || !rays_d || !xyz_min || !xyz_max || !rays_pts || !mask_outbbox) return ESR_EINVAL;
    sample_ndc_kernel<<<esr_grid_for(total, 256, 256 * 16), 256, 0, esr_stream(stream)>>>(rays_o, rays_d, xyz_min, xyz_max,
                                                                                            n_samples, total, rays_pts, mask_outbbox);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_sample_bg_pts(const float *rays_o, const float *rays_d, const float *t_max, float bg_preserve, int32_t n_samples,
                              int64_t n_rays, float *rays_pts, void *stream)
{
    if (n_rays < 0 || n_samples < 0) return ESR_EINVAL;
    const int64_t total = n_rays * n_samples;
    if (total == 0) return 0;
    if (!rays_o || !rays_d || !t_max || !rays_pts) return ESR_EINVAL;
    sample_bg_kernel<<<esr_grid_for(total, 256, 256 * 16), 256, 0, esr_stream(stream)>>>(rays_o, rays_d, t_max, bg_preserve,
                                                                                           n_samples, total, rays_pts);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_maskcache_lookup(const uint8_t *world, const float *xyz, const float *xyz2ijk_scale, const float *xyz2ijk_shift,
                                 int32_t sz_i, int32_t sz_j, int32_t sz_k, int64_t n_pts, uint8_t *out, void *stream)
{
    if (n_pts < 0 || sz_i < 1 || sz_j < 1 || sz_k < 1) return ESR_EINVAL;
    if (n_pts == 0) return 0;
    if (!world || !xyz || !xyz2ijk_scale || !xyz2ijk_shift || !out) return ESR_EINVAL;
    maskcache_lookup_kernel<<<esr_grid_for(n_pts, 256, 256 * 16), 256, 0, esr_stream(stream)>>>(
        world, xyz, xyz2ijk_scale, xyz2ijk_shift, sz_i, sz_j, sz_k, n_pts, out);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_raw2alpha(const float *density, float shift, float interval, const float *interval_per_point, int64_t n_pts,
                          float *exp_d, float *alpha, void *stream)
{
    if (n_pts < 0) return ESR_EINVAL;
    if (n_pts == 0) return 0;
    if (!density || !exp_d || !alpha) return ESR_EINVAL;
    raw2alpha_kernel<<<esr_grid_for(n_pts, 256, 256 * 16), 256, 0, esr_stream(stream)>>>(density, shift, interval,
                                                                                           interval_per_point, n_pts, exp_d, alpha);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_raw2alpha_bwd(const float *exp_d, const float *grad_back, float interval, const float *interval_per_point,
                              int64_t n_pts, float *grad, void *stream)
{
    if (n_pts < 0) return ESR_EINVAL;
    if (n_pts == 0) return 0;
    if (!exp_d || !grad_back || !grad) return ESR_EINVAL;
    raw2alpha_bwd_kernel<<<esr_grid_for(n_pts, 256, 256 * 16), 256, 0, esr_stream(stream)>>>(exp_d, grad_back, interval,
                                                                                               interval_per_point, n_pts, grad);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_tv_add_grad_masked(const float *param, float *grad, const float *mask, float wx, float wy, float wz,
                                   int64_t sz_i, int64_t sz_j, int64_t sz_k, int64_t n, int dense_mode, void *stream)
{
    if (n < 0 || sz_i < 1 || sz_j < 1 || sz_k < 1) return ESR_EINVAL;
    if (n == 0) return 0;
    if (!param || !grad || !mask) return ESR_EINVAL;
    wx /= 6; wy /= 6; wz /= 6;
    const int grid = esr_grid_for(n, 256, 256 * 16);
    if (dense_mode)
        tv_add_grad_masked_kernel<true><<<grid, 256, 0, esr_stream(stream)>>>(param, grad, mask, wx, wy, wz, sz_i, sz_j, sz_k, n);
    else
        tv_add_grad_masked_kernel<false><<<grid, 256, 0, esr_stream(stream)>>>(param, grad, mask, wx, wy, wz, sz_i, sz_j, sz_k, n);
    ESR_CHECK_LAUNCH();
    return 0;
}
