/*
 * esr_oracle.c -- TEST INFRASTRUCTURE ONLY (not product code).
 *
 * Plain-C, single-threaded restatement of the arithmetic of the reference's
 * three live native ops (+ total-variation add-grad + sorted segment sum), used
 * as the CPU checker for the HIP kernels.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may link or call this library.
 *
 * Algorithms restated (reference file:line, paths relative to /root/reference):
 *   ray/AABB t-range ........ app/utils/base/cuda/render_utils_kernel.cu:12-35
 *   per-ray step count ...... app/utils/base/cuda/render_utils_kernel.cu:38-55
 *   ray start / unit dir .... app/utils/base/cuda/render_utils_kernel.cu:58-79
 *   ray_id / step_id ........ app/utils/base/cuda/render_utils_kernel.cu:144-164,211-219
 *   sample points + mask .... app/utils/base/cuda/render_utils_kernel.cu:167-194
 *   alpha -> weight (fwd) ... app/utils/base/cuda/render_utils_kernel.cu:577-651
 *   alpha -> weight (bwd) ... app/utils/base/cuda/render_utils_kernel.cu:654-707
 *   TV add-grad ............. app/utils/base/cuda/total_variation_kernel.cu:13-35,68-98
 *   segment_coo(sum) ........ torch_scatter (third party, not vendored, unpinned);
 *                             call sites app/fine/model/voxurff.py:260-272
 *   exported but never called by the reference's Python (bottom of this file):
 *   NDC / background points . app/utils/base/cuda/render_utils_kernel.cu:243-269,301-340
 *   mask-cache lookup ....... app/utils/base/cuda/render_utils_kernel.cu:366-392
 *   raw2alpha (+ backward) .. app/utils/base/cuda/render_utils_kernel.cu:431-460,504-530
 *   masked TV add-grad ...... app/utils/base/cuda/total_variation_kernel.cu:38-66,101-131
 *
 * Pinning: the reference has no tests or golden vectors.  This
 * file is pinned by plugging it into the *imported* reference Python models in
 * the build container (oracle/gen_golden.py) and by an independent numpy
 * restatement in tests/test_oracle_native.py.
 *
 * Floating point: compile with -ffp-contract=off -fno-fast-math so every
 * operation is a separately rounded IEEE binary32 op (the HIP kernels for the
 * sampler are built the same way, which is what makes the integer outputs
 * bit-identical).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ESR_API __attribute__((visibility("default")))

/* ---- sampler ------------------------------------------------------------ */

static inline float fsel_dir(float d) { return (d == 0.0f) ? (float)1e-6 : d; }

/* t_min/t_max of the ray against the box, clamped into [near, far]. */
static void ray_t_range(const float *o, const float *d, const float *bmin,
                        const float *bmax, float near_, float far_, float *tmin,
                        float *tmax)
{
    float lo = 0.f, hi = 0.f;
    for (int a = 0; a < 3; ++a) {
        float v = fsel_dir(d[a]);
        float ta = (bmax[a] - o[a]) / v;
        float tb = (bmin[a] - o[a]) / v;
        float mn = fminf(ta, tb), mx = fmaxf(ta, tb);
        if (a == 0) { lo = mn; hi = mx; }
        else        { lo = fmaxf(lo, mn); hi = fminf(hi, mx); }
    }
    *tmin = fmaxf(fminf(lo, far_), near_);
    *tmax = fmaxf(fminf(hi, far_), near_);
}

static inline float ray_norm(const float *d)
{
    /* left-to-right sum of squares, then sqrtf: three separately rounded products */
    float s = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    return sqrtf(s);
}

/*
 * Phase 1: per-ray quantities.  n_steps is int64 as in the reference; the
 * float ceil() is promoted to double only for the max(.,1.) and the store.
 */
ESR_API void esr_oracle_sample_count(const float *rays_o, const float *rays_d,
                                     const float *xyz_min, const float *xyz_max,
                                     float near_, float far_, float stepdist,
                                     int64_t n_rays, float *t_min, float *t_max,
                                     int64_t *n_steps)
{
    for (int64_t r = 0; r < n_rays; ++r) {
        const float *o = rays_o + 3 * r, *d = rays_d + 3 * r;
        ray_t_range(o, d, xyz_min, xyz_max, near_, far_, &t_min[r], &t_max[r]);
        float len = (t_max[r] - t_min[r]) * ray_norm(d) / stepdist;
        double c = (double)ceilf(len);
        n_steps[r] = (int64_t)(c > 1.0 ? c : 1.0);
    }
}

/*
 * Phase 2: expand rays into samples.  Outputs have sum(n_steps) rows.
 * mask_outbbox is 1 where the point lies outside [xyz_min, xyz_max].
 */
ESR_API void esr_oracle_sample_fill(const float *rays_o, const float *rays_d,
                                    const float *xyz_min, const float *xyz_max,
                                    const float *t_min, const int64_t *n_steps,
                                    float stepdist, int64_t n_rays, float *ray_pts,
                                    uint8_t *mask_outbbox, int64_t *ray_id,
                                    int64_t *step_id)
{
    int64_t at = 0;
    for (int64_t r = 0; r < n_rays; ++r) {
        const float *o = rays_o + 3 * r, *d = rays_d + 3 * r;
        float nrm = ray_norm(d);
        float start[3], dir[3];
        for (int a = 0; a < 3; ++a) {
            start[a] = o[a] + d[a] * t_min[r];
            dir[a] = d[a] / nrm;
        }
        for (int64_t s = 0; s < n_steps[r]; ++s, ++at) {
            /* the reference narrows ids to int before the float multiply */
            float dist = stepdist * (float)(int)s;
            float p[3];
            int out = 0;
            for (int a = 0; a < 3; ++a) {
                p[a] = start[a] + dir[a] * dist;
                ray_pts[3 * at + a] = p[a];
                out |= (xyz_min[a] > p[a]) | (xyz_max[a] < p[a]);
            }
            mask_outbbox[at] = (uint8_t)out;
            ray_id[at] = r;
            step_id[at] = s;
        }
    }
}

/* ---- compositing: alpha -> weights -------------------------------------- */

/*
 * Forward.  weight/T/alphainv_last/i_start/i_end are fully written here
 * (pre-initialised to 0/1/1/0/0 like the reference wrapper).  ray_id must be
 * sorted.  The transmittance accumulator is a float; the early-stop test is
 * applied after the update and the stop index is one past the stopping sample.
 */
ESR_API void esr_oracle_alpha2weight(const float *alpha, const int64_t *ray_id,
                                     int64_t n_pts, int64_t n_rays, float *weight,
                                     float *T, float *alphainv_last, int64_t *i_start,
                                     int64_t *i_end)
{
    for (int64_t i = 0; i < n_pts; ++i) { weight[i] = 0.f; T[i] = 1.f; }
    for (int64_t r = 0; r < n_rays; ++r) { alphainv_last[r] = 1.f; i_start[r] = 0; i_end[r] = 0; }
    if (n_pts == 0) return;
    for (int64_t i = 1; i < n_pts; ++i) {
        if (ray_id[i] != ray_id[i - 1]) {
            i_start[ray_id[i]] = i;
            i_end[ray_id[i - 1]] = i;
        }
    }
    i_end[ray_id[n_pts - 1]] = n_pts;
    for (int64_t r = 0; r < n_rays; ++r) {
        int64_t i = i_start[r];
        const int64_t e = i_end[r];
        float tc = 1.f;
        while (i < e) {
            T[i] = tc;
            weight[i] = tc * alpha[i];
            /* (1. - alpha) is evaluated in double, the product is rounded to float */
            tc = (float)((double)tc * (1.0 - (double)alpha[i]));
            ++i;
            if (tc < 1e-3) break;
        }
        i_end[r] = i;
        alphainv_last[r] = tc;
    }
}

/* Backward: reverse scan over [i_start, i_end) of every ray. */
ESR_API void esr_oracle_alpha2weight_backward(
    const float *alpha, const float *weight, const float *T,
    const float *alphainv_last, const int64_t *i_start, const int64_t *i_end,
    int64_t n_pts, int64_t n_rays, const float *grad_weights, const float *grad_last,
    float *grad)
{
    for (int64_t i = 0; i < n_pts; ++i) grad[i] = 0.f;
    for (int64_t r = 0; r < n_rays; ++r) {
        float back = grad_last[r] * alphainv_last[r];
        for (int64_t i = i_end[r] - 1; i >= i_start[r]; --i) {
            /* 1 - alpha + 1e-10 is a double expression in the reference */
            double den = (double)(1.0f - alpha[i]) + 1e-10;
            grad[i] = (float)((double)(grad_weights[i] * T[i]) - (double)back / den);
            back += grad_weights[i] * weight[i];
        }
    }
}

/* ---- total-variation gradient (in place) -------------------------------- */

static inline float clamp1(float v) { return fminf(fmaxf(v, -1.f), 1.f); }

/*
 * grad += sum over the 6 face neighbours of w/6 * clamp(p - p_nbr, -1, 1).
 * Quirk kept from the reference: the slowest (i) and fastest (k) axes both use
 * wz, wx is unused.  Sparse mode only touches cells whose grad is non-zero.
 * param/grad are [C?,] sz_i, sz_j, sz_k contiguous; n = total element count.
 */
ESR_API void esr_oracle_tv_add_grad(const float *param, float *grad, float wx, float wy,
                                    float wz, int64_t sz_i, int64_t sz_j, int64_t sz_k,
                                    int64_t n, int dense_mode)
{
    (void)wx;
    wy /= 6; wz /= 6;
    /* every cell reads only param[] and its own grad[], so in-place is safe */
    for (int64_t idx = 0; idx < n; ++idx) {
        if (!dense_mode && grad[idx] == 0.f) continue;
        const int64_t k = idx % sz_k, j = idx / sz_k % sz_j, i = idx / sz_k / sz_j % sz_i;
        const float p = param[idx];
        float g = 0.f;
        g += (k == 0)        ? 0.f : wz * clamp1(p - param[idx - 1]);
        g += (k == sz_k - 1) ? 0.f : wz * clamp1(p - param[idx + 1]);
        g += (j == 0)        ? 0.f : wy * clamp1(p - param[idx - sz_k]);
        g += (j == sz_j - 1) ? 0.f : wy * clamp1(p - param[idx + sz_k]);
        g += (i == 0)        ? 0.f : wz * clamp1(p - param[idx - sz_k * sz_j]);
        g += (i == sz_i - 1) ? 0.f : wz * clamp1(p - param[idx + sz_k * sz_j]);
        grad[idx] += g;
    }
}

/* ---- sorted segment sum (torch_scatter.segment_coo, reduce="sum") ------- */

/* out[index[i], :] += src[i, :]; out is caller-initialised, index sorted. */
ESR_API void esr_oracle_segment_sum(const float *src, const int64_t *index, int64_t n,
                                    int64_t c, float *out)
{
    for (int64_t i = 0; i < n; ++i)
        for (int64_t a = 0; a < c; ++a) out[index[i] * c + a] += src[i * c + a];
}

/* ---- the exported-but-dead ops of the two pybind modules ----------------
 * fp32 instantiations, statement by statement; float / double mixing exactly where the reference's literals put it.
 * No reference caller, test or golden vector exists for these: pinned by independent torch restatements of the formulas
 * (tests/test_oracle_native.py), incl. the "original pytorch implementation" the background sampler's source quotes. */

ESR_API void esr_oracle_infer_n_samples(const float *rays_d, const float *t_min, const float *t_max, float stepdist,
                                        int64_t n_rays, int64_t *n_samples)
{
    for (int64_t r = 0; r < n_rays; ++r) {
        float len = (t_max[r] - t_min[r]) * ray_norm(rays_d + 3 * r) / stepdist;
        double c = (double)ceilf(len);
        n_samples[r] = (int64_t)(c > 1.0 ? c : 1.0);
    }
}

ESR_API void esr_oracle_infer_ray_start_dir(const float *rays_o, const float *rays_d, const float *t_min, int64_t n_rays,
                                            float *rays_start, float *rays_dir)
{
    for (int64_t r = 0; r < n_rays; ++r) {
        const float *o = rays_o + 3 * r, *d = rays_d + 3 * r;
        const float nrm = ray_norm(d);
        for (int a = 0; a < 3; ++a) {
            rays_start[3 * r + a] = o[a] + d[a] * t_min[r];
            rays_dir[3 * r + a] = d[a] / nrm;
        }
    }
}

ESR_API void esr_oracle_sample_ndc_pts(const float *rays_o, const float *rays_d, const float *xyz_min, const float *xyz_max,
                                       int n_samples, int64_t n_rays, float *rays_pts, uint8_t *mask_outbbox)
{
    for (int64_t r = 0; r < n_rays; ++r)
        for (int s = 0; s < n_samples; ++s) {
            const int64_t idx = r * n_samples + s;
            const float dist = ((float)s) / (float)(n_samples - 1);
            int out = 0;
            for (int a = 0; a < 3; ++a) {
                const float p = rays_o[3 * r + a] + rays_d[3 * r + a] * dist;
                rays_pts[3 * idx + a] = p;
                out |= (xyz_min[a] > p) | (xyz_max[a] < p);
            }
            mask_outbbox[idx] = (uint8_t)out;
        }
}

ESR_API void esr_oracle_sample_bg_pts(const float *rays_o, const float *rays_d, const float *t_max, float bg_preserve,
                                      int n_samples, int64_t n_rays, float *rays_pts)
{
    for (int64_t r = 0; r < n_rays; ++r)
        for (int s = 0; s < n_samples; ++s) {
            const int64_t idx = r * n_samples + s;
            const float frac = ((float)s) / (float)n_samples;
            const float t_o = (float)(((double)t_max[r] - 1.) + 1. / (1. - (double)frac));
            float q[3];
            for (int a = 0; a < 3; ++a) q[a] = rays_o[3 * r + a] + rays_d[3 * r + a] * t_o;
            const float t_outer = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
            const float m = fmaxf(fabsf(q[0]), fmaxf(fabsf(q[1]), fabsf(q[2])));
            const float R = t_outer / m;
            const float a1 = R * R / (t_outer * t_outer), a2 = R / t_outer * bg_preserve;
            const float o2i = (float)((double)a1 * (1. - (double)bg_preserve) + (double)a2);
            for (int a = 0; a < 3; ++a) rays_pts[3 * idx + a] = q[a] * o2i;
        }
}

ESR_API void esr_oracle_maskcache_lookup(const uint8_t *world, const float *xyz, const float *scale, const float *shift,
                                         int sz_i, int sz_j, int sz_k, int64_t n_pts, uint8_t *out)
{
    for (int64_t p = 0; p < n_pts; ++p) {
        const int i = (int)roundf(xyz[3 * p] * scale[0] + shift[0]);
        const int j = (int)roundf(xyz[3 * p + 1] * scale[1] + shift[1]);
        const int k = (int)roundf(xyz[3 * p + 2] * scale[2] + shift[2]);
        const int in = 0 <= i && i < sz_i && 0 <= j && j < sz_j && 0 <= k && k < sz_k;
        out[p] = in ? world[((int64_t)i * sz_j + j) * sz_k + k] : 0;      /* (the reference's output starts as zeros) */
    }
}

/* interval_t != NULL: one interval per point (raw2alpha_nonuni) */
ESR_API void esr_oracle_raw2alpha(const float *density, float shift, float interval, const float *interval_t, int64_t n_pts,
                                  float *exp_d, float *alpha)
{
    for (int64_t p = 0; p < n_pts; ++p) {
        const float e = expf(density[p] + shift);
        const float iv = interval_t ? interval_t[p] : interval;
        exp_d[p] = e;
        alpha[p] = 1.f - powf(1.f + e, -iv);
    }
}

ESR_API void esr_oracle_raw2alpha_bwd(const float *exp_d, const float *grad_back, float interval, const float *interval_t,
                                      int64_t n_pts, float *grad)
{
    for (int64_t p = 0; p < n_pts; ++p) {
        const float e = exp_d[p];
        const float iv = interval_t ? interval_t[p] : interval;
        const double lim = (double)e < 1e10 ? (double)e : 1e10;          /* min(float, 1e10): the double overload */
        grad[p] = (float)(lim * (double)powf(1.f + e, -iv - 1.f) * (double)iv * (double)grad_back[p]);
    }
}

/* total_variation_add_grad_new: wx on the fastest axis (the live kernel uses wz there), every term x mask[cell] mask[nbr] */
ESR_API void esr_oracle_tv_add_grad_masked(const float *param, float *grad, const float *mask, float wx, float wy, float wz,
                                           int64_t sz_i, int64_t sz_j, int64_t sz_k, int64_t n, int dense_mode)
{
    wx /= 6; wy /= 6; wz /= 6;
    for (int64_t idx = 0; idx < n; ++idx) {
        if (!dense_mode && grad[idx] == 0.f) continue;
        const int64_t k = idx % sz_k, j = idx / sz_k % sz_j, i = idx / sz_k / sz_j % sz_i;
        const float p = param[idx], m = mask[idx];
        float g = 0.f;
        g += (k == 0)        ? 0.f : wx * clamp1(p - param[idx - 1]) * m * mask[idx - 1];
        g += (k == sz_k - 1) ? 0.f : wx * clamp1(p - param[idx + 1]) * m * mask[idx + 1];
        g += (j == 0)        ? 0.f : wy * clamp1(p - param[idx - sz_k]) * m * mask[idx - sz_k];
        g += (j == sz_j - 1) ? 0.f : wy * clamp1(p - param[idx + sz_k]) * m * mask[idx + sz_k];
        g += (i == 0)        ? 0.f : wz * clamp1(p - param[idx - sz_k * sz_j]) * m * mask[idx - sz_k * sz_j];
        g += (i == sz_i - 1) ? 0.f : wz * clamp1(p - param[idx + sz_k * sz_j]) * m * mask[idx + sz_k * sz_j];
        grad[idx] += g;
    }
}

/* ---- the DOUBLE instantiations of the three live ops (AT_DISPATCH_FLOATING_TYPES; render_utils_kernel.cu:93-101,113-120,
 * 130-138,229,639,692).  The reference's kernels keep `float` LOCALS whatever scalar_t is: with double tensors the t-range,
 * the ray length, every sample point, the running transmittance and the backward's running sum are rounded to float on the
 * way; restated literally.  Pinned by an independent numpy statement (tests/test_oracle_native.py). */

static void ray_t_range_f64(const double *o, const double *d, const double *bmin, const double *bmax, float near_, float far_,
                            float *tmin, float *tmax)
{
    float lo = 0.f, hi = 0.f;
    for (int a = 0; a < 3; ++a) {
        const float v = (float)((d[a] == 0.0) ? 1e-6 : d[a]);
        const float ta = (float)((bmax[a] - o[a]) / (double)v);
        const float tb = (float)((bmin[a] - o[a]) / (double)v);
        const float mn = fminf(ta, tb), mx = fmaxf(ta, tb);
        if (a == 0) { lo = mn; hi = mx; }
        else        { lo = fmaxf(lo, mn); hi = fminf(hi, mx); }
    }
    *tmin = fmaxf(fminf(lo, far_), near_);
    *tmax = fmaxf(fminf(hi, far_), near_);
}

static inline float ray_norm_f64(const double *d) { return (float)sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]); }

ESR_API void esr_oracle_sample_count_f64(const double *rays_o, const double *rays_d, const double *xyz_min, const double *xyz_max,
                                         float near_, float far_, float stepdist, int64_t n_rays, double *t_min, double *t_max,
                                         int64_t *n_steps)
{
    for (int64_t r = 0; r < n_rays; ++r) {
        float lo, hi;
        ray_t_range_f64(rays_o + 3 * r, rays_d + 3 * r, xyz_min, xyz_max, near_, far_, &lo, &hi);
        t_min[r] = (double)lo;
        t_max[r] = (double)hi;
        const double len = ((double)hi - (double)lo) * (double)ray_norm_f64(rays_d + 3 * r) / (double)stepdist;
        const double c = ceil(len);
        n_steps[r] = (int64_t)(c > 1.0 ? c : 1.0);
    }
}

ESR_API void esr_oracle_sample_fill_f64(const double *rays_o, const double *rays_d, const double *xyz_min, const double *xyz_max,
                                        const double *t_min, const int64_t *n_steps, float stepdist, int64_t n_rays,
                                        double *ray_pts, uint8_t *mask_outbbox, int64_t *ray_id, int64_t *step_id)
{
    int64_t at = 0;
    for (int64_t r = 0; r < n_rays; ++r) {
        const double *o = rays_o + 3 * r, *d = rays_d + 3 * r;
        const float nrm = ray_norm_f64(d);
        for (int64_t s = 0; s < n_steps[r]; ++s, ++at) {
            const float dist = stepdist * (float)(int)s;
            int out = 0;
            for (int a = 0; a < 3; ++a) {
                const double start = o[a] + d[a] * t_min[r];
                const double dir = d[a] / (double)nrm;
                const float p = (float)(start + dir * (double)dist);
                ray_pts[3 * at + a] = (double)p;
                out |= (xyz_min[a] > (double)p) | (xyz_max[a] < (double)p);
            }
            mask_outbbox[at] = (uint8_t)out;
            ray_id[at] = r;
            step_id[at] = s;
        }
    }
}

ESR_API void esr_oracle_alpha2weight_f64(const double *alpha, const int64_t *ray_id, int64_t n_pts, int64_t n_rays,
                                         double *weight, double *T, double *alphainv_last, int64_t *i_start, int64_t *i_end)
{
    for (int64_t i = 0; i < n_pts; ++i) { weight[i] = 0.0; T[i] = 1.0; }
    for (int64_t r = 0; r < n_rays; ++r) { alphainv_last[r] = 1.0; i_start[r] = 0; i_end[r] = 0; }
    if (n_pts == 0) return;
    for (int64_t i = 1; i < n_pts; ++i)
        if (ray_id[i] != ray_id[i - 1]) { i_start[ray_id[i]] = i; i_end[ray_id[i - 1]] = i; }
    i_end[ray_id[n_pts - 1]] = n_pts;
    for (int64_t r = 0; r < n_rays; ++r) {
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
        alphainv_last[r] = (double)tc;
    }
}

ESR_API void esr_oracle_alpha2weight_backward_f64(const double *alpha, const double *weight, const double *T,
                                                  const double *alphainv_last, const int64_t *i_start, const int64_t *i_end,
                                                  int64_t n_pts, int64_t n_rays, const double *grad_weights,
                                                  const double *grad_last, double *grad)
{
    for (int64_t i = 0; i < n_pts; ++i) grad[i] = 0.0;
    for (int64_t r = 0; r < n_rays; ++r) {
        float back = (float)(grad_last[r] * alphainv_last[r]);
        for (int64_t i = i_end[r] - 1; i >= i_start[r]; --i) {
            grad[i] = grad_weights[i] * T[i] - (double)back / ((1.0 - alpha[i]) + 1e-10);
            back = (float)((double)back + grad_weights[i] * weight[i]);
        }
    }
}

/* ---- what-if: the sampler with the contractions nvcc's default -fmad=true would most plausibly make -----------------
 * The reference JIT-builds its kernels with torch's default nvcc flags (app/utils/base/functions.py:14-31), so a*b+c patterns
 * of the real binary are fused; which ones is the compiler's choice and cannot be observed here (no nvcc).  This variant fuses
 * every multiply-add of the sampler's statements (sum of squares, start = o + d t_min, point = start + dir dist) and exists only
 * to MEASURE how far such a build can be from the separately rounded statement above (tests/test_oracle_native.py,
 * DESIGN.md section 3); nothing compares the HIP kernels with it. */
ESR_API void esr_oracle_sample_count_fma(const float *rays_o, const float *rays_d, const float *xyz_min, const float *xyz_max,
                                         float near_, float far_, float stepdist, int64_t n_rays, float *t_min, float *t_max,
                                         int64_t *n_steps)
{
    for (int64_t r = 0; r < n_rays; ++r) {
        const float *o = rays_o + 3 * r, *d = rays_d + 3 * r;
        ray_t_range(o, d, xyz_min, xyz_max, near_, far_, &t_min[r], &t_max[r]);
        const float nrm = sqrtf(fmaf(d[2], d[2], fmaf(d[1], d[1], d[0] * d[0])));
        float len = (t_max[r] - t_min[r]) * nrm / stepdist;
        double c = (double)ceilf(len);
        n_steps[r] = (int64_t)(c > 1.0 ? c : 1.0);
    }
}

ESR_API void esr_oracle_sample_fill_fma(const float *rays_o, const float *rays_d, const float *xyz_min, const float *xyz_max,
                                        const float *t_min, const int64_t *n_steps, float stepdist, int64_t n_rays,
                                        float *ray_pts, uint8_t *mask_outbbox)
{
    int64_t at = 0;
    for (int64_t r = 0; r < n_rays; ++r) {
        const float *o = rays_o + 3 * r, *d = rays_d + 3 * r;
        const float nrm = sqrtf(fmaf(d[2], d[2], fmaf(d[1], d[1], d[0] * d[0])));
        for (int64_t s = 0; s < n_steps[r]; ++s, ++at) {
            const float dist = stepdist * (float)(int)s;
            int out = 0;
            for (int a = 0; a < 3; ++a) {
                const float start = fmaf(d[a], t_min[r], o[a]);
                const float p = fmaf(d[a] / nrm, dist, start);
                ray_pts[3 * at + a] = p;
                out |= (xyz_min[a] > p) | (xyz_max[a] < p);
            }
            mask_outbbox[at] = (uint8_t)out;
        }
    }
}
