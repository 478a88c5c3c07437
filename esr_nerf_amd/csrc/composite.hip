// alpha -> weights compositing: drop-in for render_utils_cuda.alpha2weight /
// alpha2weight_backward (reference: app/utils/base/cuda/render_utils_kernel.cu:577-707).
//
// MI355X design: the reference runs ONE THREAD per ray with a serial,
// uncoalesced loop.  Here one 64-lane wavefront owns a ray: each chunk of 64
// samples is loaded/stored coalesced, and the transmittance recurrence -- kept in
// the reference's exact serial order and mixed float/double arithmetic so the
// early-stop index is bit-identical -- runs on wave-uniform values broadcast
// with v_readlane.
#include "esr_common.h"

namespace {

__global__ void __launch_bounds__(256) a2w_init_kernel(int64_t n_pts, int64_t n_rays,
                                                       float *__restrict__ weight, float *__restrict__ T,
                                                       float *__restrict__ last,
                                                       int64_t *__restrict__ i_start,
                                                       int64_t *__restrict__ i_end)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    for (int64_t i = t0; i < n_pts; i += stride) { weight[i] = 0.f; T[i] = 1.f; }
    for (int64_t r = t0; r < n_rays; r += stride) { last[r] = 1.f; i_start[r] = 0; i_end[r] = 0; }
}

__global__ void __launch_bounds__(256) a2w_bounds_kernel(const int64_t *__restrict__ ray_id,
                                                         int64_t n_pts, int64_t *__restrict__ i_start,
                                                         int64_t *__restrict__ i_end)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_pts; i += stride) {
        if (i > 0 && ray_id[i] != ray_id[i - 1]) {
            i_start[ray_id[i]] = i;
            i_end[ray_id[i - 1]] = i;
        }
        if (i == n_pts - 1) i_end[ray_id[i]] = n_pts;
    }
}

__global__ void __launch_bounds__(256) a2w_fwd_kernel(const float *__restrict__ alpha, int64_t n_rays,
                                                      float *__restrict__ weight, float *__restrict__ T,
                                                      float *__restrict__ last,
                                                      const int64_t *__restrict__ i_start,
                                                      int64_t *__restrict__ i_end)
{
    const int lane = esr_lane();
    const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave; r < n_rays; r += nwaves) {
        const int64_t b = i_start[r], e = i_end[r];
        float tc = 1.f;
        int64_t stop = e;
        bool stopped = false;
        for (int64_t c0 = b; c0 < e && !stopped; c0 += 64) {
            const int cnt = (int)((e - c0 < 64) ? (e - c0) : 64);
            const float a = (lane < cnt) ? alpha[c0 + lane] : 0.f;
            float myT = 1.f;
            bool done = false;
            for (int i = 0; i < cnt; ++i) {
                const float ai = esr_readlane(a, i);
                if (lane == i) { myT = tc; done = true; }
                tc = (float)((double)tc * (1.0 - (double)ai));
                if ((double)tc < 1e-3) { stop = c0 + i + 1; stopped = true; break; }
            }
            if (done) {
                T[c0 + lane] = myT;
                weight[c0 + lane] = myT * a;
            }
        }
        if (lane == 0) {
            i_end[r] = stop;
            last[r] = tc;
        }
    }
}

__global__ void __launch_bounds__(256) a2w_bwd_kernel(
    const float *__restrict__ alpha, const float *__restrict__ weight, const float *__restrict__ T,
    const float *__restrict__ last, const int64_t *__restrict__ i_start,
    const int64_t *__restrict__ i_end, int64_t n_rays, const float *__restrict__ gw,
    const float *__restrict__ gl, float *__restrict__ grad)
{
    const int lane = esr_lane();
    const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave; r < n_rays; r += nwaves) {
        const int64_t b = i_start[r], e = i_end[r];
        float back = gl[r] * last[r];
        // chunks from the far end towards the ray origin
        for (int64_t hi = e; hi > b; hi -= 64) {
            const int64_t c0 = (hi - 64 > b) ? hi - 64 : b;
            const int cnt = (int)(hi - c0);
            const bool ok = lane < cnt;
            const float g_w = ok ? gw[c0 + lane] : 0.f;
            const float x = ok ? g_w * weight[c0 + lane] : 0.f;
            float myback = 0.f;
            for (int i = cnt - 1; i >= 0; --i) {
                if (lane == i) myback = back;
                back += esr_readlane(x, i);
            }
            if (ok) {
                const float a = alpha[c0 + lane];
                const double den = (double)(1.0f - a) + 1e-10;
                grad[c0 + lane] = (float)((double)(g_w * T[c0 + lane]) - (double)myback / den);
            }
            if (c0 == b) break;
        }
    }
}

// ---- the reference's DOUBLE instantiation (render_utils_kernel.cu:639,692): `float T_cum` / `float back_cum` stay float
// locals, everything else is double.  One thread per ray, the reference's own loop (nothing on the path calls it).
__global__ void __launch_bounds__(256) a2w_init_f64_kernel(int64_t n_pts, int64_t n_rays, double *__restrict__ weight,
                                                           double *__restrict__ T, double *__restrict__ last,
                                                           int64_t *__restrict__ i_start, int64_t *__restrict__ i_end)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    for (int64_t i = t0; i < n_pts; i += stride) { weight[i] = 0.0; T[i] = 1.0; }
    for (int64_t r = t0; r < n_rays; r += stride) { last[r] = 1.0; i_start[r] = 0; i_end[r] = 0; }
}

__global__ void __launch_bounds__(256) a2w_fwd_f64_kernel(const double *__restrict__ alpha, int64_t n_rays,
                                                          double *__restrict__ weight, double *__restrict__ T,
                                                          double *__restrict__ last, const int64_t *__restrict__ i_start,
                                                          int64_t *__restrict__ i_end)
{
#pragma clang fp contract(off)
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n_rays; r += (int64_t)gridDim.x * blockDim.x) {
        int64_t i = i_start[r];
        const int64_t e = i_end[r];
        float tc = 1.f;
        while (i < e) {
            T[i] = (double)tc;
            weight[i] = (double)tc * alpha[i];
            tc = (float)((double)tc * (1.0 - alpha[i]));
            ++i;
            if ((double)tc < 1e-3) break;
        }
        i_end[r] = i;
        last[r] = (double)tc;
    }
}

__global__ void __launch_bounds__(256) a2w_bwd_f64_kernel(
    const double *__restrict__ alpha, const double *__restrict__ weight, const double *__restrict__ T,
    const double *__restrict__ last, const int64_t *__restrict__ i_start, const int64_t *__restrict__ i_end, int64_t n_rays,
    const double *__restrict__ gw, const double *__restrict__ gl, double *__restrict__ grad)
{
#pragma clang fp contract(off)
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n_rays; r += (int64_t)gridDim.x * blockDim.x) {
        float back = (float)(gl[r] * last[r]);
        for (int64_t i = i_end[r] - 1; i >= i_start[r]; --i) {
            grad[i] = gw[i] * T[i] - (double)back / ((1.0 - alpha[i]) + 1e-10);
            back = (float)((double)back + gw[i] * weight[i]);
        }
    }
}

__global__ void __launch_bounds__(256) zero_f64_kernel(double *__restrict__ p, int64_t n)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0.0;
}

__global__ void __launch_bounds__(256) zero_f32_kernel(float *__restrict__ p, int64_t n)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        p[i] = 0.f;
}

}  // namespace

ESR_API int esr_alpha2weight_fwd(const float *alpha, const int64_t *ray_id, int64_t n_pts,
                                 int64_t n_rays, float *weight, float *T, float *alphainv_last,
                                 int64_t *i_start, int64_t *i_end, void *stream)
{
    if (n_pts < 0 || n_rays < 0) return ESR_EINVAL;
    if (n_rays > 0 && (!alphainv_last || !i_start || !i_end)) return ESR_EINVAL;
    if (n_pts > 0 && (!alpha || !ray_id || !weight || !T)) return ESR_EINVAL;
    hipStream_t s = esr_stream(stream);
    if (n_pts + n_rays == 0) return 0;
    a2w_init_kernel<<<esr_grid_for(n_pts > n_rays ? n_pts : n_rays, 256), 256, 0, s>>>(
        n_pts, n_rays, weight, T, alphainv_last, i_start, i_end);
    ESR_CHECK_LAUNCH();
    if (n_pts == 0) return 0;
    a2w_bounds_kernel<<<esr_grid_for(n_pts, 256), 256, 0, s>>>(ray_id, n_pts, i_start, i_end);
    ESR_CHECK_LAUNCH();
    a2w_fwd_kernel<<<esr_grid_for(n_rays * 64, 256), 256, 0, s>>>(alpha, n_rays, weight, T,
                                                                  alphainv_last, i_start, i_end);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_alpha2weight_bwd(const float *alpha, const float *weight, const float *T,
                                 const float *alphainv_last, const int64_t *i_start,
                                 const int64_t *i_end, int64_t n_pts, int64_t n_rays,
                                 const float *grad_weights, const float *grad_last, float *grad,
                                 void *stream)
{
    if (n_pts < 0 || n_rays < 0) return ESR_EINVAL;
    if (n_pts == 0) return 0;
    if (!alpha || !weight || !T || !alphainv_last || !i_start || !i_end || !grad_weights ||
        !grad_last || !grad)
        return ESR_EINVAL;
    hipStream_t s = esr_stream(stream);
    zero_f32_kernel<<<esr_grid_for(n_pts, 256), 256, 0, s>>>(grad, n_pts);
    ESR_CHECK_LAUNCH();
    if (n_rays == 0) return 0;
    a2w_bwd_kernel<<<esr_grid_for(n_rays * 64, 256), 256, 0, s>>>(
        alpha, weight, T, alphainv_last, i_start, i_end, n_rays, grad_weights, grad_last, grad);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_alpha2weight_fwd_f64(const double *alpha, const int64_t *ray_id, int64_t n_pts, int64_t n_rays, double *weight,
                                     double *T, double *alphainv_last, int64_t *i_start, int64_t *i_end, void *stream)
{
    if (n_pts < 0 || n_rays < 0) return ESR_EINVAL;
    if (n_rays > 0 && (!alphainv_last || !i_start || !i_end)) return ESR_EINVAL;
    if (n_pts > 0 && (!alpha || !ray_id || !weight || !T)) return ESR_EINVAL;
    hipStream_t s = esr_stream(stream);
    if (n_pts + n_rays == 0) return 0;
    a2w_init_f64_kernel<<<esr_grid_for(n_pts > n_rays ? n_pts : n_rays, 256), 256, 0, s>>>(n_pts, n_rays, weight, T,
                                                                                             alphainv_last, i_start, i_end);
    ESR_CHECK_LAUNCH();
    if (n_pts == 0) return 0;
    a2w_bounds_kernel<<<esr_grid_for(n_pts, 256), 256, 0, s>>>(ray_id, n_pts, i_start, i_end);
    ESR_CHECK_LAUNCH();
    a2w_fwd_f64_kernel<<<esr_grid_for(n_rays, 256), 256, 0, s>>>(alpha, n_rays, weight, T, alphainv_last, i_start, i_end);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_alpha2weight_bwd_f64(const double *alpha, const double *weight, const double *T, const double *alphainv_last,
                                     const int64_t *i_start, const int64_t *i_end, int64_t n_pts, int64_t n_rays,
                                     const double *grad_weights, const double *grad_last, double *grad, void *stream)
{
    if (n_pts < 0 || n_rays < 0) return ESR_EINVAL;
    if (n_pts == 0) return 0;
    if (!alpha || !weight || !T || !alphainv_last || !i_start || !i_end || !grad_weights || !grad_last || !grad)
        return ESR_EINVAL;
    hipStream_t s = esr_stream(stream);
    zero_f64_kernel<<<esr_grid_for(n_pts, 256), 256, 0, s>>>(grad, n_pts);
    ESR_CHECK_LAUNCH();
    if (n_rays == 0) return 0;
    a2w_bwd_f64_kernel<<<esr_grid_for(n_rays, 256), 256, 0, s>>>(alpha, weight, T, alphainv_last, i_start, i_end, n_rays,
                                                                 grad_weights, grad_last, grad);
    ESR_CHECK_LAUNCH();
    return 0;
}
