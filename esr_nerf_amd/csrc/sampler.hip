// Ray sampler: drop-in for render_utils_cuda.sample_pts_on_rays
// (reference: app/utils/base/cuda/render_utils_kernel.cu:12-79,144-242).
//
// MI355X design: the reference issues 6 launches + 2 torch cumsums.  Here phase 1
// is one launch (per-ray t-range + step count, then an in-kernel single-workgroup
// scan is a second tiny launch), phase 2 is one launch in which every thread
// finds its ray by binary search in the L2-resident inclusive scan -- no
// scatter-ones + cumsum pass over `total` elements, and the 3 coordinate stores
// of 64 consecutive samples are contiguous (768 B per wave instruction).
#include "esr_common.h"

namespace {

__global__ void __launch_bounds__(256) ray_count_kernel(
    const float *__restrict__ rays_o, const float *__restrict__ rays_d,
    const float *__restrict__ xyz_min, const float *__restrict__ xyz_max, float near_, float far_,
    float stepdist, int64_t n_rays, float *__restrict__ t_min, float *__restrict__ t_max,
    int64_t *__restrict__ n_steps)
{
    const float bmin[3] = {xyz_min[0], xyz_min[1], xyz_min[2]};
    const float bmax[3] = {xyz_max[0], xyz_max[1], xyz_max[2]};
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n_rays;
         r += (int64_t)gridDim.x * blockDim.x) {
        float o[3] = {rays_o[3 * r], rays_o[3 * r + 1], rays_o[3 * r + 2]};
        float d[3] = {rays_d[3 * r], rays_d[3 * r + 1], rays_d[3 * r + 2]};
        float lo, hi;
        esr_ray_trange(o, d, bmin, bmax, near_, far_, lo, hi);
        t_min[r] = lo;
        t_max[r] = hi;
        n_steps[r] = esr_ray_nsteps(lo, hi, esr_ray_norm(d), stepdist);
    }
}

// Inclusive scan of an int64 array by ONE workgroup of 1024 threads.
__global__ void __launch_bounds__(1024) scan_i64_kernel(const int64_t *__restrict__ in, int64_t n,
                                                        int64_t *__restrict__ out,
                                                        int64_t *__restrict__ total)
{
    __shared__ int64_t part[1024];
    const int tid = threadIdx.x;
    const int64_t per = (n + 1023) / 1024;
    const int64_t b = tid * per, e = (b + per < n) ? b + per : n;
    int64_t s = 0;
    for (int64_t i = b; i < e; ++i) s += in[i];
    part[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int64_t v = (tid >= off) ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int64_t run = part[tid] - s;   // exclusive prefix of this thread's chunk
    for (int64_t i = b; i < e; ++i) {
        run += in[i];
        out[i] = run;
    }
    if (tid == 1023) *total = part[1023];
}

__global__ void __launch_bounds__(256) sample_fill_kernel(
    const float *__restrict__ rays_o, const float *__restrict__ rays_d,
    const float *__restrict__ xyz_min, const float *__restrict__ xyz_max,
    const float *__restrict__ t_min, const int64_t *__restrict__ cumsum, float stepdist,
    int64_t n_rays, int64_t total, float *__restrict__ ray_pts, uint8_t *__restrict__ mask_outbbox,
    int64_t *__restrict__ ray_id, int64_t *__restrict__ step_id)
{
    const float bmin[3] = {xyz_min[0], xyz_min[1], xyz_min[2]};
    const float bmax[3] = {xyz_max[0], xyz_max[1], xyz_max[2]};
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        // first ray whose inclusive scan exceeds i
        int64_t lo = 0, hi = n_rays - 1;
        while (lo < hi) {
            int64_t mid = (lo + hi) >> 1;
            if (cumsum[mid] > i) hi = mid; else lo = mid + 1;
        }
        const int64_t r = lo;
        const int64_t base = r ? cumsum[r - 1] : 0;
        const int step = (int)(i - base);
        float o[3] = {rays_o[3 * r], rays_o[3 * r + 1], rays_o[3 * r + 2]};
        float d[3] = {rays_d[3 * r], rays_d[3 * r + 1], rays_d[3 * r + 2]};
        float start[3], dir[3], p[3];
        esr_ray_start_dir(o, d, t_min[r], esr_ray_norm(d), start, dir);
        esr_ray_point(start, dir, stepdist, step, p);
        ray_pts[3 * i] = p[0];
        ray_pts[3 * i + 1] = p[1];
        ray_pts[3 * i + 2] = p[2];
        mask_outbbox[i] = esr_out_of_box(p, bmin, bmax) ? 1 : 0;
        ray_id[i] = r;
        step_id[i] = step;
    }
}

// ---- the reference's DOUBLE instantiation of the same op (AT_DISPATCH_FLOATING_TYPES, render_utils_kernel.cu:93-101,113-120,
// 130-138,229): its kernels keep `float` LOCALS, so with double tensors the t-range, the ray length and every sample point are
// computed in double and ROUNDED TO FLOAT before they are stored as doubles; only rays_start / rays_dir carry double precision.
// Restated literally (one thread per ray / sample, no tuning: nothing on the path calls it); oracle/esr_oracle.c has the twin.
__device__ __forceinline__ void ray_trange_f64(const double *o, const double *d, const double *bmin, const double *bmax,
                                               float near_, float far_, float &tmin, float &tmax)
{
    float lo = 0.f, hi = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float v = (float)((d[a] == 0.0) ? 1e-6 : d[a]);
        const float ta = (float)((bmax[a] - o[a]) / (double)v);
        const float tb = (float)((bmin[a] - o[a]) / (double)v);
        const float mn = fminf(ta, tb), mx = fmaxf(ta, tb);
        if (a == 0) { lo = mn; hi = mx; }
        else        { lo = fmaxf(lo, mn); hi = fminf(hi, mx); }
    }
    tmin = fmaxf(fminf(lo, far_), near_);
    tmax = fmaxf(fminf(hi, far_), near_);
}

__device__ __forceinline__ float ray_norm_f64(const double *d)
{
#pragma clang fp contract(off)
    return (float)sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
}

__global__ void __launch_bounds__(256) ray_count_f64_kernel(
    const double *__restrict__ rays_o, const double *__restrict__ rays_d, const double *__restrict__ xyz_min,
    const double *__restrict__ xyz_max, float near_, float far_, float stepdist, int64_t n_rays,
    double *__restrict__ t_min, double *__restrict__ t_max, int64_t *__restrict__ n_steps)
{
#pragma clang fp contract(off)
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n_rays; r += (int64_t)gridDim.x * blockDim.x) {
        float lo, hi;
        ray_trange_f64(rays_o + 3 * r, rays_d + 3 * r, xyz_min, xyz_max, near_, far_, lo, hi);
        t_min[r] = (double)lo;
        t_max[r] = (double)hi;
        const double len = ((double)hi - (double)lo) * (double)ray_norm_f64(rays_d + 3 * r) / (double)stepdist;
        const double c = ceil(len);
        n_steps[r] = (int64_t)(c > 1.0 ? c : 1.0);
    }
}

__global__ void __launch_bounds__(256) sample_fill_f64_kernel(
    const double *__restrict__ rays_o, const double *__restrict__ rays_d, const double *__restrict__ xyz_min,
    const double *__restrict__ xyz_max, const double *__restrict__ t_min, const int64_t *__restrict__ cumsum, float stepdist,
    int64_t n_rays, int64_t total, double *__restrict__ ray_pts, uint8_t *__restrict__ mask_outbbox,
    int64_t *__restrict__ ray_id, int64_t *__restrict__ step_id)
{
#pragma clang fp contract(off)
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t lo = 0, hi = n_rays - 1;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (cumsum[mid] > i) hi = mid; else lo = mid + 1;
        }
        const int64_t r = lo;
        const int step = (int)(i - (r ? cumsum[r - 1] : 0));
        const float nrm = ray_norm_f64(rays_d + 3 * r);
        const float dist = stepdist * (float)step;
        bool out = false;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double start = rays_o[3 * r + a] + rays_d[3 * r + a] * t_min[r];
            const double dir = rays_d[3 * r + a] / (double)nrm;
            const float p = (float)(start + dir * (double)dist);
            ray_pts[3 * i + a] = (double)p;
            out |= (xyz_min[a] > (double)p) | (xyz_max[a] < (double)p);
        }
        mask_outbbox[i] = out ? 1 : 0;
        ray_id[i] = r;
        step_id[i] = step;
    }
}

// out[index[i], :] += src[i, :] for a sorted index: a wave walks 64 consecutive
// rows, does a segmented inclusive scan with shuffles and issues ONE atomic per
// (segment, channel) instead of one per row.
__global__ void __launch_bounds__(256) segment_sum_kernel(const float *__restrict__ src,
                                                          const int64_t *__restrict__ index, int64_t n,
                                                          int c, float *__restrict__ out)
{
    const int lane = esr_lane();
    const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t base = wave * 64; base < n; base += nwaves * 64) {
        const int64_t i = base + lane;
        const bool ok = i < n;
        const int seg = ok ? (int)index[i] : -1;      // segment ids are ray ids (< 2^31)
        const int seg_next = __shfl_down(seg, 1);
        const bool tail = ok && (lane == 63 || seg_next != seg);
        for (int ch = 0; ch < c; ++ch) {
            float v = ok ? src[i * c + ch] : 0.f;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                float u = __shfl_up(v, off);
                int su = __shfl_up(seg, off);
                if (lane >= off && su == seg) v += u;
            }
            if (tail) atomicAdd(&out[(int64_t)seg * c + ch], v);
        }
    }
}

}  // namespace

ESR_API int esr_abi_version(void) { return 25; }
ESR_API const char *esr_build_info(void) { return "libesr_hip gfx950 " __DATE__ " " __TIME__; }

ESR_API int esr_sample_count(const float *rays_o, const float *rays_d, const float *xyz_min,
                             const float *xyz_max, float near_, float far_, float stepdist,
                             int64_t n_rays, float *t_min, float *t_max, int64_t *n_steps,
                             int64_t *cumsum, int64_t *total, void *stream)
{
    if (n_rays < 0 || !total) return ESR_EINVAL;
    hipStream_t s = esr_stream(stream);
    if (n_rays == 0) return (int)hipMemsetAsync(total, 0, sizeof(int64_t), s);
    if (!rays_o || !rays_d || !xyz_min || !xyz_max || !t_min || !t_max || !n_steps || !cumsum)
        return ESR_EINVAL;
    ray_count_kernel<<<esr_grid_for(n_rays, 256), 256, 0, s>>>(rays_o, rays_d, xyz_min, xyz_max, near_,
                                                               far_, stepdist, n_rays, t_min, t_max,
                                                               n_steps);
    ESR_CHECK_LAUNCH();
    scan_i64_kernel<<<1, 1024, 0, s>>>(n_steps, n_rays, cumsum, total);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_sample_fill(const float *rays_o, const float *rays_d, const float *xyz_min,
                            const float *xyz_max, const float *t_min, const int64_t *cumsum,
                            float stepdist, int64_t n_rays, int64_t total, float *ray_pts,
                            uint8_t *mask_outbbox, int64_t *ray_id, int64_t *step_id, void *stream)
{
    if (n_rays < 0 || total < 0) return ESR_EINVAL;
    if (total == 0 || n_rays == 0) return 0;
    if (!rays_o || !rays_d || !xyz_min || !xyz_max || !t_min || !cumsum || !ray_pts ||
        !mask_outbbox || !ray_id || !step_id)
        return ESR_EINVAL;
    sample_fill_kernel<<<esr_grid_for(total, 256, 256 * 16), 256, 0, esr_stream(stream)>>>(
        rays_o, rays_d, xyz_min, xyz_max, t_min, cumsum, stepdist, n_rays, total, ray_pts,
        mask_outbbox, ray_id, step_id);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_sample_count_f64(const double *rays_o, const double *rays_d, const double *xyz_min, const double *xyz_max,
                                 float near_, float far_, float stepdist, int64_t n_rays, double *t_min, double *t_max,
                                 int64_t *n_steps, int64_t *cumsum, int64_t *total, void *stream)
{
    if (n_rays < 0 || !total) return ESR_EINVAL;
    hipStream_t s = esr_stream(stream);
    if (n_rays == 0) return (int)hipMemsetAsync(total, 0, sizeof(int64_t), s);
    if (!rays_o || !rays_d || !xyz_min || !xyz_max || !t_min || !t_max || !n_steps || !cumsum) return ESR_EINVAL;
    ray_count_f64_kernel<<<esr_grid_for(n_rays, 256), 256, 0, s>>>(rays_o, rays_d, xyz_min, xyz_max, near_, far_, stepdist,
                                                                   n_rays, t_min, t_max, n_steps);
    ESR_CHECK_LAUNCH();
    scan_i64_kernel<<<1, 1024, 0, s>>>(n_steps, n_rays, cumsum, total);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_sample_fill_f64(const double *rays_o, const double *rays_d, const double *xyz_min, const double *xyz_max,
                                const double *t_min, const int64_t *cumsum, float stepdist, int64_t n_rays, int64_t total,
                                double *ray_pts, uint8_t *mask_outbbox, int64_t *ray_id, int64_t *step_id, void *stream)
{
    if (n_rays < 0 || total < 0) return ESR_EINVAL;
    if (total == 0 || n_rays == 0) return 0;
    if (!rays_o || !rays_d || !xyz_min || !xyz_max || !t_min || !cumsum || !ray_pts || !mask_outbbox || !ray_id || !step_id)
        return ESR_EINVAL;
    sample_fill_f64_kernel<<<esr_grid_for(total, 256, 256 * 16), 256, 0, esr_stream(stream)>>>(
        rays_o, rays_d, xyz_min, xyz_max, t_min, cumsum, stepdist, n_rays, total, ray_pts, mask_outbbox, ray_id, step_id);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_segment_sum(const float *src, const int64_t *index, int64_t n, int64_t c, float *out,
                            int64_t n_seg, void *stream)
{
    if (n < 0 || c < 1 || n_seg < 0) return ESR_EINVAL;
    if (n == 0) return 0;
    if (!src || !index || !out) return ESR_EINVAL;
    segment_sum_kernel<<<esr_grid_for((n + 63) / 64 * 64, 256), 256, 0, esr_stream(stream)>>>(
        src, index, n, (int)c, out);
    ESR_CHECK_LAUNCH();
    return 0;
}
