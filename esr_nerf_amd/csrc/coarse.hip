// Per-sample kernels of the coarse renderer (VoxurfC.forward_training).
//
// Reference algorithm (paths under the reference tree):
//   app/coarse/model/voxurfc.py:221-250   feature assembly: rays_xyz, PE (freqs 2^0..2^4), view PE (freq 2^0),
//                                         normal = gradient / (|gradient| + 1e-5), 12-channel colour grids
//   app/coarse/model/voxurfc.py:240-250   rgb = sigmoid(emo_rgbnet(on rays)) + sigmoid(off_rgbnet(all))
//   app/coarse/model/voxurfc.py:252-271   weights * rgb segment sum, white_bg = 1 - sum of weights
//
// The coarse stage is the small configuration (512 rays x 64 samples): one lane per sample,
// channels-last grids (one 48-byte / 12-byte record per trilinear corner), direct float atomics in the
// backward (no LDS window -- at a few thousand samples per step the scatter is launch-latency bound).
// Feature tile XC [tiles,72,32]:
//   0-11 off colour | 12-23 emo colour (on-tiles, else 0) | 24-26 normal | 27-29 xyz | 30-44 sin | 45-59 cos
//   | 60-62 viewdir | 63-65 sin | 66-68 cos | 69-71 zero
#include "esr_common.h"

namespace {

constexpr int XC = 72, DXR = 64;
constexpr int R_COL = 0, R_ALT = 12, R_NRM = 24, R_XYZ = 27, R_SIN = 30, R_COS = 45, R_VD = 60, R_VS = 63, R_VC = 66;

struct CoarseParams {
    esr_scene_t sc;
    const float *rays_o, *rays_d, *viewdirs;
    const int32_t *rec_ray, *rec_step;
    int tiles_on, tiles_all;
    const float *grad_grid, *off_color, *emo_color;     // [X,Y,Z,3], [X,Y,Z,12], [X,Y,Z,12]
    float *X, *gnorm;                                    // [tiles,72,32], [tiles,32]
    const float *dX_off, *dX_emo;                        // [tiles,64,32]
    float *g_grad_grid, *g_off_color, *g_emo_color;
};

template <int C>
__device__ __forceinline__ void tri_fetch_c(const float *__restrict__ g, const int dims[3], const float idx[3],
                                            float out[C])
{
    Tri t = esr_tri_setup(idx);
#pragma unroll
    for (int c = 0; c < C; ++c) out[c] = 0.f;
#pragma unroll
    for (int cx = 0; cx < 2; ++cx)
#pragma unroll
        for (int cy = 0; cy < 2; ++cy)
#pragma unroll
            for (int cz = 0; cz < 2; ++cz) {
                const int x = t.i0[0] + cx, y = t.i0[1] + cy, z = t.i0[2] + cz;
                const bool inb = (x >= 0) & (x < dims[0]) & (y >= 0) & (y < dims[1]) & (z >= 0) & (z < dims[2]);
                const float w = esr_corner_w(t, idx, cx, cy, cz);
                const float *v = g + (inb ? (((int64_t)x * dims[1] + y) * dims[2] + z) * C : 0);   // esr_ld_or0's reasoning
                const float wz = inb ? w : 0.f;
#pragma unroll
                for (int c = 0; c < C; ++c) out[c] += v[c] * wz;
            }
}

template <int C>
__device__ __forceinline__ void tri_scatter_c(float *__restrict__ g, const int dims[3], const float idx[3],
                                              const float v[C])
{
    Tri t = esr_tri_setup(idx);
#pragma unroll
    for (int cx = 0; cx < 2; ++cx)
#pragma unroll
        for (int cy = 0; cy < 2; ++cy)
#pragma unroll
            for (int cz = 0; cz < 2; ++cz) {
                const int x = t.i0[0] + cx, y = t.i0[1] + cy, z = t.i0[2] + cz;
                const bool inb = (x >= 0) & (x < dims[0]) & (y >= 0) & (y < dims[1]) & (z >= 0) & (z < dims[2]);
                const float w = esr_corner_w(t, idx, cx, cy, cz);
                if (inb && w != 0.f) {
                    float *p = g + (((int64_t)x * dims[1] + y) * dims[2] + z) * C;
#pragma unroll
                    for (int c = 0; c < C; ++c)
                        if (v[c] != 0.f) atomicAdd(p + c, v[c] * w);
                }
            }
}

__device__ __forceinline__ bool sample_pos(const CoarseParams &P, int j, int &ray, float p[3])
{
    ray = P.rec_ray[j];
    if (ray < 0) return false;
    const RayGeom g = esr_ray_geom(P.rays_o, P.rays_d, ray, P.sc.xyz_min, P.sc.xyz_max, P.sc.near_, 1e9f,
                                   P.sc.stepdist);
    esr_ray_point(g.start, g.dir, P.sc.stepdist, P.rec_step[j], p);
    return true;
}

__global__ void __launch_bounds__(256) coarse_feat_fwd_kernel(CoarseParams P)
{
    const esr_scene_t &sc = P.sc;
    const int gdims[3] = {sc.gx, sc.gy, sc.gz};
    const int total = P.tiles_all * 32;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < total; j += gridDim.x * blockDim.x) {
        const int t = j >> 5, s = j & 31;
        float *Xt = P.X + (size_t)t * XC * 32 + s;
        float p[3], ind[3];
        int ray;
        if (!sample_pos(P, j, ray, p)) {
            for (int r = 0; r < XC; ++r) Xt[r * 32] = 0.f;
            P.gnorm[j] = 0.f;
            continue;
        }
        esr_world_to_index(p, sc.xyz_min, sc.xyz_max, gdims, ind);
        float col[12];
        tri_fetch_c<12>(P.off_color, gdims, ind, col);
#pragma unroll
        for (int c = 0; c < 12; ++c) Xt[(R_COL + c) * 32] = col[c];
        if (t < P.tiles_on) tri_fetch_c<12>(P.emo_color, gdims, ind, col);
#pragma unroll
        for (int c = 0; c < 12; ++c) Xt[(R_ALT + c) * 32] = (t < P.tiles_on) ? col[c] : 0.f;
        float g[3];
        tri_fetch_c<3>(P.grad_grid, gdims, ind, g);
        const float nrm = sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
        P.gnorm[j] = nrm;
#pragma unroll
        for (int a = 0; a < 3; ++a) Xt[(R_NRM + a) * 32] = g[a] / (nrm + 1e-5f);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float unit;
            {
#pragma clang fp contract(off)
                unit = __fdiv_rn(p[c] - sc.xyz_min[c], sc.xyz_max[c] - sc.xyz_min[c]);
            }
            Xt[(R_XYZ + c) * 32] = unit;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const float a = unit * (float)(1 << i);
                Xt[(R_SIN + c * 5 + i) * 32] = sinf(a);
                Xt[(R_COS + c * 5 + i) * 32] = cosf(a);
            }
            const float v = P.viewdirs[3 * ray + c];
            Xt[(R_VD + c) * 32] = v;
            Xt[(R_VS + c) * 32] = sinf(v);
            Xt[(R_VC + c) * 32] = cosf(v);
        }
        Xt[69 * 32] = 0.f; Xt[70 * 32] = 0.f; Xt[71 * 32] = 0.f;
    }
}

__global__ void __launch_bounds__(256) coarse_feat_bwd_kernel(CoarseParams P)
{
    const esr_scene_t &sc = P.sc;
    const int gdims[3] = {sc.gx, sc.gy, sc.gz};
    const int total = P.tiles_all * 32;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < total; j += gridDim.x * blockDim.x) {
        const int t = j >> 5, s = j & 31;
        float p[3], ind[3];
        int ray;
        if (!sample_pos(P, j, ray, p)) continue;
        esr_world_to_index(p, sc.xyz_min, sc.xyz_max, gdims, ind);
        const bool on = t < P.tiles_on;
        const float *dO = P.dX_off + (size_t)t * DXR * 32 + s;
        const float *dE = P.dX_emo + (size_t)t * DXR * 32 + s;
        float v[12];
#pragma unroll
        for (int c = 0; c < 12; ++c) v[c] = dO[(R_COL + c) * 32];
        tri_scatter_c<12>(P.g_off_color, gdims, ind, v);
        if (on) {
#pragma unroll
            for (int c = 0; c < 12; ++c) v[c] = dE[(R_COL + c) * 32];      // the emo net's first 12 inputs
            tri_scatter_c<12>(P.g_emo_color, gdims, ind, v);
        }
        // normal = g / (|g| + eps)   ->   dg = dn / (|g| + eps) - g (dn . g) / ((|g| + eps)^2 |g|)
        const float *Xt = P.X + (size_t)t * XC * 32 + s;
        const float nrm = P.gnorm[j], den = nrm + 1e-5f;
        float dn[3], g[3], dot = 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            dn[a] = dO[(R_NRM + a) * 32] + (on ? dE[(R_NRM + a) * 32] : 0.f);
            g[a] = Xt[(R_NRM + a) * 32] * den;
            dot += dn[a] * g[a];
        }
        float dg[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) dg[a] = dn[a] / den - (nrm > 0.f ? g[a] * dot / (den * den * nrm) : 0.f);
        tri_scatter_c<3>(P.g_grad_grid, gdims, ind, dg);
    }
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// rgb = sigmoid(z_off) + [on] sigmoid(z_emo); srgb[ray] += w * rgb (segmented reduction over sorted rays)
__global__ void __launch_bounds__(256) coarse_shade_fwd_kernel(const float *__restrict__ z_off,
                                                               const float *__restrict__ z_emo,
                                                               const int32_t *__restrict__ rec_ray,
                                                               const float *__restrict__ rec_w, int tiles_on,
                                                               int tiles_all, float *__restrict__ rgb,
                                                               float *__restrict__ srgb)
{
    const int lane = esr_lane();
    const int total = tiles_all * 32;
    const int stride = gridDim.x * blockDim.x;
    for (int j0 = (blockIdx.x * blockDim.x + threadIdx.x) - lane; j0 < total; j0 += stride) {
        const int j = j0 + lane;
        const bool in = j < total;
        const int t = j >> 5, s = j & 31;
        const int ray = in ? rec_ray[j] : -1;
        const float w = ray >= 0 ? rec_w[j] : 0.f;
        float c[3] = {0.f, 0.f, 0.f};
        if (in) {
            const size_t b = (size_t)t * 4 * 32 + s;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float v = esr_sigmoid(z_off[b + k * 32]);
                if (t < tiles_on) v += esr_sigmoid(z_emo[b + k * 32]);
                if (ray < 0) v = 0.f;
                rgb[b + k * 32] = v;
                c[k] = w * v;
            }
            rgb[b + 96] = 0.f;
        }
        // segmented inclusive scan by ray id, the last lane of every segment adds the segment total
        const int nxt = __shfl_down(ray, 1);
        const bool tail = (lane == 63) || (nxt != ray);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float v = c[k];
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const float u = __shfl_up(v, off);
                const int r2 = __shfl_up(ray, off);
                if (lane >= off && r2 == ray) v += u;
            }
            if (tail && ray >= 0 && v != 0.f) atomicAdd(&srgb[3 * ray + k], v);
        }
    }
}

// dz_off / dz_emo through the sigmoids, dweight = g_srgb[ray] . rgb - g_wbg[ray]  (white_bg = 1 - sum w)
__global__ void __launch_bounds__(256) coarse_shade_bwd_kernel(const float *__restrict__ g_srgb,
                                                               const float *__restrict__ g_wbg,
                                                               const float *__restrict__ rgb,
                                                               const float *__restrict__ z_off,
                                                               const float *__restrict__ z_emo,
                                                               const int32_t *__restrict__ rec_ray,
                                                               const float *__restrict__ rec_w, int tiles_on,
                                                               int tiles_all, float *__restrict__ dz_off,
                                                               float *__restrict__ dz_emo,
                                                               float *__restrict__ dweight)
{
    const int total = tiles_all * 32;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < total; j += gridDim.x * blockDim.x) {
        const int t = j >> 5, s = j & 31;
        const size_t b = (size_t)t * 4 * 32 + s;
        const int ray = rec_ray[j];
        float dw = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float go = 0.f, ge = 0.f;
            if (ray >= 0) {
                const float g = g_srgb[3 * ray + k];
                const float d = g * rec_w[j];
                const float so = esr_sigmoid(z_off[b + k * 32]);
                go = d * so * (1.f - so);
                if (t < tiles_on) {
                    const float se = esr_sigmoid(z_emo[b + k * 32]);
                    ge = d * se * (1.f - se);
                }
                dw += g * rgb[b + k * 32];
            }
            dz_off[b + k * 32] = go;
            if (t < tiles_on) dz_emo[b + k * 32] = ge;
        }
        dz_off[b + 96] = 0.f;
        if (t < tiles_on) dz_emo[b + 96] = 0.f;
        dweight[j] = ray >= 0 ? dw - g_wbg[ray] : 0.f;
    }
}

int fill(CoarseParams &P, const esr_scene_t *scene, const float *rays_o, const float *rays_d, const float *viewdirs,
         const int32_t *rec_ray, const int32_t *rec_step, int tiles_on, int tiles_all)
{
    if (!scene || tiles_all < 0 || tiles_on < 0 || tiles_on > tiles_all) return ESR_EINVAL;
    if (tiles_all == 0) return 0;
    if (!rays_o || !rays_d || !viewdirs || !rec_ray || !rec_step) return ESR_EINVAL;
    P.sc = *scene; P.rays_o = rays_o; P.rays_d = rays_d; P.viewdirs = viewdirs; P.rec_ray = rec_ray;
    P.rec_step = rec_step; P.tiles_on = tiles_on; P.tiles_all = tiles_all;
    return 1;
}

}  // namespace

ESR_API int esr_coarse_feat_fwd(const esr_scene_t *scene, const float *rays_o, const float *rays_d,
                                const float *viewdirs, const int32_t *rec_ray, const int32_t *rec_step,
                                int32_t tiles_on, int32_t tiles_all, const float *grad_grid,
                                const float *off_color, const float *emo_color, float *X, float *gnorm,
                                void *stream)
{
    CoarseParams P = {};
    const int c = fill(P, scene, rays_o, rays_d, viewdirs, rec_ray, rec_step, tiles_on, tiles_all);
    if (c <= 0) return c;
    if (!grad_grid || !off_color || (tiles_on && !emo_color) || !X || !gnorm) return ESR_EINVAL;
    P.grad_grid = grad_grid; P.off_color = off_color; P.emo_color = emo_color; P.X = X; P.gnorm = gnorm;
    coarse_feat_fwd_kernel<<<esr_grid_for((int64_t)tiles_all * 32, 256), 256, 0, esr_stream(stream)>>>(P);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_coarse_feat_bwd(const esr_scene_t *scene, const float *rays_o, const float *rays_d,
                                const float *viewdirs, const int32_t *rec_ray, const int32_t *rec_step,
                                int32_t tiles_on, int32_t tiles_all, const float *X, const float *gnorm,
                                const float *dX_off, const float *dX_emo, float *g_grad_grid,
                                float *g_off_color, float *g_emo_color, void *stream)
{
    CoarseParams P = {};
    const int c = fill(P, scene, rays_o, rays_d, viewdirs, rec_ray, rec_step, tiles_on, tiles_all);
    if (c <= 0) return c;
    if (!X || !gnorm || !dX_off || (tiles_on && (!dX_emo || !g_emo_color)) || !g_grad_grid || !g_off_color)
        return ESR_EINVAL;
    P.X = const_cast<float *>(X); P.gnorm = const_cast<float *>(gnorm); P.dX_off = dX_off;
    P.dX_emo = dX_emo ? dX_emo : dX_off;
    P.g_grad_grid = g_grad_grid; P.g_off_color = g_off_color; P.g_emo_color = g_emo_color;
    coarse_feat_bwd_kernel<<<esr_grid_for((int64_t)tiles_all * 32, 256), 256, 0, esr_stream(stream)>>>(P);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_coarse_shade_fwd(const float *z_off, const float *z_emo, const int32_t *rec_ray,
                                 const float *rec_w, int32_t tiles_on, int32_t tiles_all, float *rgb,
                                 float *srgb_marched, void *stream)
{
    if (tiles_all < 0 || tiles_on < 0 || tiles_on > tiles_all) return ESR_EINVAL;
    if (tiles_all == 0) return 0;
    if (!z_off || (tiles_on && !z_emo) || !rec_ray || !rec_w || !rgb || !srgb_marched) return ESR_EINVAL;
    coarse_shade_fwd_kernel<<<esr_grid_for(((int64_t)tiles_all * 32 + 63) / 64 * 64, 256), 256, 0,
                              esr_stream(stream)>>>(z_off, z_emo, rec_ray, rec_w, tiles_on, tiles_all, rgb,
                                                    srgb_marched);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_coarse_shade_bwd(const float *g_srgb, const float *g_white_bg, const float *rgb,
                                 const float *z_off, const float *z_emo, const int32_t *rec_ray,
                                 const float *rec_w, int32_t tiles_on, int32_t tiles_all, float *dz_off,
                                 float *dz_emo, float *dweight, void *stream)
{
    if (tiles_all < 0 || tiles_on < 0 || tiles_on > tiles_all) return ESR_EINVAL;
    if (tiles_all == 0) return 0;
    if (!g_srgb || !g_white_bg || !rgb || !z_off || (tiles_on && (!z_emo || !dz_emo)) || !rec_ray || !rec_w ||
        !dz_off || !dweight)
        return ESR_EINVAL;
    coarse_shade_bwd_kernel<<<esr_grid_for((int64_t)tiles_all * 32, 256), 256, 0, esr_stream(stream)>>>(
        g_srgb, g_white_bg, rgb, z_off, z_emo, rec_ray, rec_w, tiles_on, tiles_all, dz_off, dz_emo, dweight);
    ESR_CHECK_LAUNCH();
    return 0;
}
