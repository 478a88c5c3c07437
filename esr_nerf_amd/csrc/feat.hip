// Per-sample feature assembly for the radiance MLPs and its backward.
//
// Reference algorithm (paths under the reference tree):
//   app/fine/model/voxurff.py:219-254   feature vector of forward_training
//   app/fine/model/voxurff.py:678-721   sample_sdfeat_grad_normal (24 clamped taps)
//   app/utils/base/module.py:24-35      DenseGrid.forward (6-channel colour grids)
//
// MI355X design: one lane per surviving sample, lanes = consecutive samples of a
// ray, so the 8-corner gathers of neighbouring lanes fall into the same 64/128-B
// lines (the stencil of two samples 0.5 voxel apart overlaps almost entirely) and
// a wave-wide atomic in the backward touches a handful of lines instead of 64.
// The output is written straight in the MLP kernels' operand layout
// X[tile][row][32 samples]: every store instruction is one or two fully used
// 128-B lines.  Colour grids are read channel-last (24 B per voxel, z-adjacent
// corners contiguous) instead of the reference's channel-first layout (one cache
// line per channel per corner).
#include "esr_common.h"

namespace {

constexpr int XROWS = 96;   // rows of an X tile
constexpr int DXROWS = 64;  // rows of a dX tile (0..42 used)
constexpr int ROW_COL = 0, ROW_SDF = 6, ROW_FEAT = 7, ROW_NRM = 31, ROW_XYZ = 43, ROW_SIN = 46,
              ROW_COS = 61, ROW_VD = 76, ROW_VSIN = 79, ROW_VCOS = 82, ROW_ALT = 88;

struct FeatParams {
    esr_scene_t sc;
    const float *rays_o, *rays_d, *viewdirs, *sdf, *off_color, *emo_color;
    const int32_t *rec_ray, *rec_step;
    const float *rec_sdf;
    int tiles_on, tiles_all;
    float *X, *gnorm;
    // backward
    const float *dX;
    float *grad_sdf, *grad_off, *grad_emo;
};

__device__ __forceinline__ void tri_fetch6(const float *__restrict__ g, const int dims[3],
                                           const float idx[3], float out[6])
{
    Tri t = esr_tri_setup(idx);
#pragma unroll
    for (int c = 0; c < 6; ++c) out[c] = 0.f;
#pragma unroll
    for (int cx = 0; cx < 2; ++cx)
#pragma unroll
        for (int cy = 0; cy < 2; ++cy)
#pragma unroll
            for (int cz = 0; cz < 2; ++cz) {
                int x = t.i0[0] + cx, y = t.i0[1] + cy, z = t.i0[2] + cz;
                bool inb = (x >= 0) & (x < dims[0]) & (y >= 0) & (y < dims[1]) & (z >= 0) & (z < dims[2]);
                float w = esr_corner_w(t, idx, cx, cy, cz);
                if (inb) {
                    const float2 *v = reinterpret_cast<const float2 *>(
                        g + (((int64_t)x * dims[1] + y) * dims[2] + z) * 6);
                    float2 a = v[0], b = v[1], c = v[2];
                    out[0] += a.x * w; out[1] += a.y * w; out[2] += b.x * w;
                    out[3] += b.y * w; out[4] += c.x * w; out[5] += c.y * w;
                }
            }
}

__device__ __forceinline__ void tri_scatter6(float *__restrict__ g, const int dims[3],
                                             const float idx[3], const float v[6])
{
    Tri t = esr_tri_setup(idx);
#pragma unroll
    for (int cx = 0; cx < 2; ++cx)
#pragma unroll
        for (int cy = 0; cy < 2; ++cy)
#pragma unroll
            for (int cz = 0; cz < 2; ++cz) {
                int x = t.i0[0] + cx, y = t.i0[1] + cy, z = t.i0[2] + cz;
                bool inb = (x >= 0) & (x < dims[0]) & (y >= 0) & (y < dims[1]) & (z >= 0) & (z < dims[2]);
                float w = esr_corner_w(t, idx, cx, cy, cz);
                if (inb && w != 0.f) {
                    float *p = g + (((int64_t)x * dims[1] + y) * dims[2] + z) * 6;
#pragma unroll
                    for (int c = 0; c < 6; ++c) atomicAdd(p + c, v[c] * w);
                }
            }
}

// Continuous index of one clamped stencil tap and the clamped coordinate along
// its axis.  Replicates ind + offset -> clamp -> /(size-1)*2-1 -> grid_sample's
// ((n+1)/2)*(size-1) so the tap lands on the same float as the reference's.
__device__ __forceinline__ float tap_index(const float ind[3], const int dims[3], int axis, float disp,
                                           float ix[3])
{
#pragma clang fp contract(off)
    float along = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float t = ind[a] + ((a == axis) ? disp : 0.f);
        t = fminf(fmaxf(t, 0.f), (float)(dims[a] - 1));
        if (a == axis) along = t;
        float n = __fdiv_rn(t, (float)(dims[a] - 1)) * 2.0f - 1.0f;
        ix[a] = __fdiv_rn(n + 1.0f, 2.0f) * (float)(dims[a] - 1);
    }
    return along;
}

__device__ __forceinline__ void sample_point(const FeatParams &P, int ray, int step, float p[3])
{
    const RayGeom g = esr_ray_geom(P.rays_o, P.rays_d, ray, P.sc.xyz_min, P.sc.xyz_max, P.sc.near_,
                                   1e9f, P.sc.stepdist);
    esr_ray_point(g.start, g.dir, P.sc.stepdist, step, p);
}

__global__ void __launch_bounds__(256) feat_fwd_kernel(FeatParams P)
{
    const esr_scene_t &sc = P.sc;
    const int gdims[3] = {sc.gx, sc.gy, sc.gz};
    const int total = P.tiles_all * 32;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < total; j += gridDim.x * blockDim.x) {
        const int t = j >> 5, s = j & 31;
        float *Xt = P.X + (size_t)t * XROWS * 32 + s;
        float *Gn = P.gnorm + (size_t)t * 4 * 32 + s;
        const int ray = P.rec_ray[j];
        if (ray < 0) {                                   // padding lane: inert zeros
            for (int r = 0; r < XROWS; ++r) Xt[r * 32] = 0.f;
            for (int k = 0; k < 4; ++k) Gn[k * 32] = 0.f;
            continue;
        }
        const bool on_tile = t < P.tiles_on;
        float p[3], ind[3], unit[3];
        sample_point(P, ray, P.rec_step[j], p);
        esr_world_to_index(p, sc.xyz_min, sc.xyz_max, gdims, ind);
        {
#pragma clang fp contract(off)
#pragma unroll
            for (int a = 0; a < 3; ++a) unit[a] = __fdiv_rn(p[a] - sc.xyz_min[a], sc.xyz_max[a] - sc.xyz_min[a]);
        }
        // colour grids
        float col[6];
        tri_fetch6(on_tile ? P.emo_color : P.off_color, gdims, ind, col);
#pragma unroll
        for (int c = 0; c < 6; ++c) Xt[(ROW_COL + c) * 32] = col[c];
        if (on_tile) {
            tri_fetch6(P.off_color, gdims, ind, col);
#pragma unroll
            for (int c = 0; c < 6; ++c) Xt[(ROW_ALT + c) * 32] = col[c];
        } else {
#pragma unroll
            for (int c = 0; c < 6; ++c) Xt[(ROW_ALT + c) * 32] = 0.f;
        }
        Xt[(ROW_ALT + 6) * 32] = 0.f;
        Xt[(ROW_ALT + 7) * 32] = 0.f;
        Xt[85 * 32] = 0.f; Xt[86 * 32] = 0.f; Xt[87 * 32] = 0.f;
        Xt[ROW_SDF * 32] = P.rec_sdf[j];
        // 24-tap SDF stencil: reference axis order is (z, y, x) = grid axes (2, 1, 0)
        float grad[3][4];
#pragma unroll
        for (int ar = 0; ar < 3; ++ar) {
            const int axis = 2 - ar;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float ixm[3], ixp[3];
                const float cm = tap_index(ind, gdims, axis, -sc.grad_feat[k], ixm);
                const float cp = tap_index(ind, gdims, axis, sc.grad_feat[k], ixp);
                const float fm = esr_tri_fetch1(P.sdf, gdims, ixm);
                const float fp = esr_tri_fetch1(P.sdf, gdims, ixp);
                Xt[(ROW_FEAT + (2 * ar) * 4 + k) * 32] = fm;
                Xt[(ROW_FEAT + (2 * ar + 1) * 4 + k) * 32] = fp;
                grad[ar][k] = (fp - fm) / (cp - cm) / sc.voxel_size;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float nrm = sqrtf(grad[0][k] * grad[0][k] + grad[1][k] * grad[1][k] + grad[2][k] * grad[2][k]);
            const float den = fmaxf(nrm, 1e-12f);
            Gn[k * 32] = nrm;
#pragma unroll
            for (int ar = 0; ar < 3; ++ar) Xt[(ROW_NRM + ar * 4 + k) * 32] = grad[ar][k] / den;
        }
        // positional encodings (coordinate-major, frequencies 1,2,4,8,16)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            Xt[(ROW_XYZ + c) * 32] = unit[c];
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const float a = unit[c] * (float)(1 << i);
                Xt[(ROW_SIN + c * 5 + i) * 32] = sinf(a);
                Xt[(ROW_COS + c * 5 + i) * 32] = cosf(a);
            }
            const float v = P.viewdirs[3 * ray + c];
            Xt[(ROW_VD + c) * 32] = v;
            Xt[(ROW_VSIN + c) * 32] = sinf(v);
            Xt[(ROW_VCOS + c) * 32] = cosf(v);
        }
    }
}

__global__ void __launch_bounds__(256) feat_bwd_kernel(FeatParams P)
{
    const esr_scene_t &sc = P.sc;
    const int gdims[3] = {sc.gx, sc.gy, sc.gz};
    const int total = P.tiles_all * 32;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < total; j += gridDim.x * blockDim.x) {
        const int ray = P.rec_ray[j];
        if (ray < 0) continue;
        const int t = j >> 5, s = j & 31;
        const float *Xt = P.X + (size_t)t * XROWS * 32 + s;
        const float *dXt = P.dX + (size_t)t * DXROWS * 32 + s;
        const float *Gn = P.gnorm + (size_t)t * 4 * 32 + s;
        const bool on_tile = t < P.tiles_on;
        float p[3], ind[3];
        sample_point(P, ray, P.rec_step[j], p);
        esr_world_to_index(p, sc.xyz_min, sc.xyz_max, gdims, ind);
        float dcol[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) dcol[c] = dXt[(ROW_COL + c) * 32];
        tri_scatter6(on_tile ? P.grad_emo : P.grad_off, gdims, ind, dcol);
        esr_tri_scatter1(P.grad_sdf, gdims, ind, dXt[ROW_SDF * 32]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float nrm = Gn[k * 32];
            float n[3], dn[3], dot = 0.f;
#pragma unroll
            for (int ar = 0; ar < 3; ++ar) {
                n[ar] = Xt[(ROW_NRM + ar * 4 + k) * 32];
                dn[ar] = dXt[(ROW_NRM + ar * 4 + k) * 32];
                dot += n[ar] * dn[ar];
            }
#pragma unroll
            for (int ar = 0; ar < 3; ++ar) {
                const int axis = 2 - ar;
                // d normal / d grad: projection for |g| > eps, plain 1/eps scaling below it
                const float dg = (nrm > 1e-12f) ? (dn[ar] - n[ar] * dot) / nrm : dn[ar] / 1e-12f;
                float ixm[3], ixp[3];
                const float cm = tap_index(ind, gdims, axis, -sc.grad_feat[k], ixm);
                const float cp = tap_index(ind, gdims, axis, sc.grad_feat[k], ixp);
                const float through = dg / (cp - cm) / sc.voxel_size;
                const float dfm = dXt[(ROW_FEAT + (2 * ar) * 4 + k) * 32] - through;
                const float dfp = dXt[(ROW_FEAT + (2 * ar + 1) * 4 + k) * 32] + through;
                esr_tri_scatter1(P.grad_sdf, gdims, ixm, dfm);
                esr_tri_scatter1(P.grad_sdf, gdims, ixp, dfp);
            }
        }
    }
}

}  // namespace

ESR_API int esr_fine_feat_fwd(const esr_scene_t *scene, const float *rays_o, const float *rays_d,
                              const float *viewdirs, const float *sdf, const float *off_color,
                              const float *emo_color, const int32_t *rec_ray, const int32_t *rec_step,
                              const float *rec_sdf, int32_t tiles_on, int32_t tiles_all, float *X,
                              float *gnorm, void *stream)
{
    if (!scene || tiles_all < 0 || tiles_on < 0 || tiles_on > tiles_all) return ESR_EINVAL;
    if (tiles_all == 0) return 0;
    if (!rays_o || !rays_d || !viewdirs || !sdf || !off_color || !emo_color || !rec_ray || !rec_step ||
        !rec_sdf || !X || !gnorm)
        return ESR_EINVAL;
    FeatParams P = {};
    P.sc = *scene; P.rays_o = rays_o; P.rays_d = rays_d; P.viewdirs = viewdirs; P.sdf = sdf;
    P.off_color = off_color; P.emo_color = emo_color; P.rec_ray = rec_ray; P.rec_step = rec_step;
    P.rec_sdf = rec_sdf; P.tiles_on = tiles_on; P.tiles_all = tiles_all; P.X = X; P.gnorm = gnorm;
    feat_fwd_kernel<<<esr_grid_for((int64_t)tiles_all * 32, 256, 256 * 16), 256, 0, esr_stream(stream)>>>(P);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_fine_feat_bwd(const esr_scene_t *scene, const float *rays_o, const float *rays_d,
                              const int32_t *rec_ray, const int32_t *rec_step, const float *X,
                              const float *gnorm, const float *dX, int32_t tiles_on, int32_t tiles_all,
                              float *grad_sdf, float *grad_off_color, float *grad_emo_color, void *stream)
{
    if (!scene || tiles_all < 0 || tiles_on < 0 || tiles_on > tiles_all) return ESR_EINVAL;
    if (tiles_all == 0) return 0;
    if (!rays_o || !rays_d || !rec_ray || !rec_step || !X || !gnorm || !dX || !grad_sdf ||
        !grad_off_color || !grad_emo_color)
        return ESR_EINVAL;
    FeatParams P = {};
    P.sc = *scene; P.rays_o = rays_o; P.rays_d = rays_d; P.rec_ray = rec_ray; P.rec_step = rec_step;
    P.tiles_on = tiles_on; P.tiles_all = tiles_all; P.X = const_cast<float *>(X);
    P.gnorm = const_cast<float *>(gnorm); P.dX = dX; P.grad_sdf = grad_sdf; P.grad_off = grad_off_color;
    P.grad_emo = grad_emo_color;
    feat_bwd_kernel<<<esr_grid_for((int64_t)tiles_all * 32, 256, 256 * 16), 256, 0, esr_stream(stream)>>>(P);
    ESR_CHECK_LAUNCH();
    return 0;
}
