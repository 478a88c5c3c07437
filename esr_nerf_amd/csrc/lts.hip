// Kernels specific to the light-transport-segment (LTS / PDRA) stages.
//
// Reference algorithm (paths under the reference tree):
//   app/fine/model/esrnerf.py:1572-1605 + app/utils/base/functions.py:142-309
//       sample_sdf_expgrad: exact spatial gradient of the trilinear SDF interpolant
//   app/utils/pbr/functions.py:10-18     diffuse_scattering (hemisphere directions)
//   app/utils/pbr/functions.py:108-173   disney_reflection
//   app/utils/pbr/module.py:86-143       SphericalGaussian environment map
//   app/fine/model/esrnerf.py:565-572,653-677   reflection weights, env term, hemisphere means
//
// MI355X design notes
//  * The reference obtains the SDF gradient with autograd(create_graph=True) through a
//    python re-implementation of grid_sample and then differentiates THAT again for the
//    normal-smoothness loss.  The interpolant is linear in the grid, so the gradient has
//    the closed form  d f/d x = (X-1)/(max_x-min_x) * sum_corners (+-) w_y w_z g_c  and its
//    backward is a plain 8-corner scatter of (+-) w_y w_z: one gather kernel, one atomic
//    scatter kernel, no second-order graph.
//  * The light-transport combine works on one workgroup per surface point: its R secondary
//    rays sit on the lanes, the hemisphere means are LDS block reductions, and the 48-lobe
//    environment-map parameter gradients are reduced per workgroup before touching HBM.
#include "esr_common.h"

namespace {

// ---- exact SDF gradient -------------------------------------------------------------
struct ExpGradParams {
    esr_scene_t sc;
    const float *rays_o, *rays_d;
    const int32_t *rec_ray, *rec_step;   // ray/step mode (pts == nullptr)
    const float *pts;                    // explicit mode [n,3]
    const float *noise;                  // optional [n,3]
    float eps;
    const float *sdf;
    int n, zero_pad;
    float *out;                          // [n,4]: sdf value, d/dx, d/dy, d/dz
    const float *g;                      // backward: [n,4] (component 0 = grad w.r.t. the value)
    float *grad_sdf;
};

__device__ __forceinline__ bool expgrad_point(const ExpGradParams &P, int i, float p[3])
{
    if (P.pts) {
        p[0] = P.pts[3 * i]; p[1] = P.pts[3 * i + 1]; p[2] = P.pts[3 * i + 2];
    } else {
        const int ray = P.rec_ray[i];
        if (ray < 0) return false;
        const RayGeom g = esr_ray_geom(P.rays_o, P.rays_d, ray, P.sc.xyz_min, P.sc.xyz_max, P.sc.near_, 1e9f,
                                       P.sc.stepdist);
        esr_ray_point(g.start, g.dir, P.sc.stepdist, P.rec_step[i], p);
    }
    if (P.noise) {
#pragma unroll
        for (int a = 0; a < 3; ++a) p[a] = p[a] + P.noise[3 * i + a] * P.eps;
    }
    return true;
}

template <bool BWD>
__global__ void __launch_bounds__(256) expgrad_kernel(ExpGradParams P)
{
    const esr_scene_t &sc = P.sc;
    const int dims[3] = {sc.gx, sc.gy, sc.gz};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P.n; i += gridDim.x * blockDim.x) {
        float p[3], idx[3];
        if (!expgrad_point(P, i, p)) {
            if (!BWD) { P.out[4 * i] = P.out[4 * i + 1] = P.out[4 * i + 2] = P.out[4 * i + 3] = 0.f; }
            continue;
        }
        esr_world_to_index(p, sc.xyz_min, sc.xyz_max, dims, idx);
        int i0[3];
        float w[3][2], scale[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float fl = floorf(idx[a]);
            i0[a] = (int)fl;
            w[a][0] = (fl + 1.f) - idx[a];
            w[a][1] = idx[a] - fl;
            scale[a] = (float)(dims[a] - 1) / (sc.xyz_max[a] - sc.xyz_min[a]);
        }
        float val = 0.f, d[3] = {0.f, 0.f, 0.f};
        float gv = 0.f, gd[3] = {0.f, 0.f, 0.f};
        if (BWD) {
            gv = P.g[4 * i];
#pragma unroll
            for (int a = 0; a < 3; ++a) gd[a] = P.g[4 * i + 1 + a] * scale[a];
        }
#pragma unroll
        for (int cx = 0; cx < 2; ++cx)
#pragma unroll
            for (int cy = 0; cy < 2; ++cy)
#pragma unroll
                for (int cz = 0; cz < 2; ++cz) {
                    // border-replicated corner (weights are taken before the clamp, as the reference does)
                    const int x = min(max(i0[0] + cx, 0), dims[0] - 1);
                    const int y = min(max(i0[1] + cy, 0), dims[1] - 1);
                    const int z = min(max(i0[2] + cz, 0), dims[2] - 1);
                    const size_t cell = ((size_t)x * dims[1] + y) * dims[2] + z;
                    const float sx = cx ? 1.f : -1.f, sy = cy ? 1.f : -1.f, sz = cz ? 1.f : -1.f;
                    const float c0 = w[0][cx] * w[1][cy] * w[2][cz];
                    const float c1 = sx * w[1][cy] * w[2][cz], c2 = w[0][cx] * sy * w[2][cz],
                                c3 = w[0][cx] * w[1][cy] * sz;
                    const bool inb = (i0[0] + cx >= 0) & (i0[0] + cx < dims[0]) & (i0[1] + cy >= 0) &
                                     (i0[1] + cy < dims[1]) & (i0[2] + cz >= 0) & (i0[2] + cz < dims[2]);
                    const bool dropped = P.zero_pad && !inb;        // F.grid_sample zero padding
                    if (!BWD) {
                        const float raw = P.sdf[cell];                 // (cell is clamped: always loadable)
                        const float gcell = dropped ? 0.f : raw;
                        val += gcell * c0; d[0] += gcell * c1; d[1] += gcell * c2; d[2] += gcell * c3;
                    } else {
                        const float t = gv * c0 + gd[0] * c1 + gd[1] * c2 + gd[2] * c3;
                        if (t != 0.f && !dropped) atomicAdd(&P.grad_sdf[cell], t);
                    }
                }
        if (!BWD) {
            P.out[4 * i] = val;
#pragma unroll
            for (int a = 0; a < 3; ++a) P.out[4 * i + 1 + a] = d[a] * scale[a];
        }
    }
}

// ---- hemisphere directions ---------------------------------------------------------------
// (o2 / d2 / v_rand: optional -- the secondary rays' origins and directions and the random view direction, which the
//  caller used to build with repeat_interleave / reshape().contiguous() / neg: four more launches)
__global__ void __launch_bounds__(256) lts_dirs_kernel(const float *__restrict__ raw,
                                                       const float *__restrict__ normal, int n_pts, int r1,
                                                       float *__restrict__ dirs, const float *__restrict__ pts,
                                                       float *__restrict__ o2, float *__restrict__ d2, float *__restrict__ v_rand)
{
    const int total = n_pts * r1;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int p = i / r1, k = i - p * r1;
        float v[3] = {raw[3 * i], raw[3 * i + 1], raw[3 * i + 2]};
        const float nrm = fmaxf(sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), 1e-12f);   // F.normalize eps
#pragma unroll
        for (int a = 0; a < 3; ++a) v[a] = v[a] / nrm;
        const float dt = v[0] * normal[3 * p] + v[1] * normal[3 * p + 1] + v[2] * normal[3 * p + 2];
        const float sgn = dt < 0.f ? -1.f : 1.f;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float d = v[a] * sgn;
            dirs[3 * i + a] = d;
            if (o2) {
                if (k < r1 - 1) {
                    const int64_t q = (int64_t)p * (r1 - 1) + k;
                    d2[3 * q + a] = d;
                    o2[3 * q + a] = pts[3 * p + a];
                } else {
                    v_rand[3 * p + a] = -d;
                }
            }
        }
    }
}

// ---- glue of the light-transport step, fused ---------------------------------------------------------------------------
// The reference runs these lines as ~60 small torch ops (gathers, cats, index_put); as that many launches they were
// host-bound here: ~15 us of enqueue each against ~5 us on the device (tools/trace_step.py: 0.46 ms per C4 step in two
// runs of 27 and 32 sub-15-us kernels).  Four kernels instead.

// compact (tile-order) slot j of a surviving sample -> its position in the reference's ray-sorted order:
// perm[pos] = j with pos = (samples of earlier rays) + (j - first slot of its ray); also the int64 ray id of every slot
__global__ void __launch_bounds__(256) lts_ref_order_kernel(const int32_t *__restrict__ rec_ray, const int32_t *__restrict__ cnt3,
                                                            const int32_t *__restrict__ off3, const int64_t *__restrict__ csum,
                                                            int n_slots, int64_t *__restrict__ perm, int64_t *__restrict__ ray64,
                                                            int32_t *__restrict__ inv)
{
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n_slots; j += gridDim.x * blockDim.x) {
        const int r = rec_ray[j];
        ray64[j] = r;
        int64_t pos = -1;
        if (r >= 0) { pos = csum[r] - cnt3[r] + (j - off3[r]); perm[pos] = j; }
        if (inv) inv[j] = (int32_t)pos;
    }
}

// perturbed positions and the scattered normal noise of esrnerf.py:807-830: for the k-th sample in reference order
//   pts_e[k] = pts_all[perm[k]] + noise_emit[k] * eps      noise_n[perm[k]] = noise_normal[k]     (noise_n pre-zeroed)
__global__ void __launch_bounds__(256) lts_perturb_kernel(const float *__restrict__ pts_all, const int64_t *__restrict__ perm,
                                                          const float *__restrict__ nn, const float *__restrict__ ne, float eps,
                                                          int m3, float *__restrict__ noise_n, float *__restrict__ pts_e)
{
#pragma clang fp contract(off)
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < m3; k += gridDim.x * blockDim.x) {
        const int64_t j = perm[k];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            noise_n[3 * j + a] = nn[3 * k + a];
            const float t = ne[3 * k + a] * eps;              // (mul, then add: as the two torch ops round)
            pts_e[3 * k + a] = pts_all[3 * j + a] + t;
        }
    }
}

// out[k][c] = src(row perm[k] (or k), column c0 + c), c < n_ch.  src is tile-major [tiles][rows][32] (rows > 0) or
// row-major with `stride` floats per row (rows == 0)
__global__ void __launch_bounds__(256) lts_gather_rows_kernel(const float *__restrict__ src, int rows, int stride, int c0, int n_ch,
                                                              const int64_t *__restrict__ perm, int n, float *__restrict__ out)
{
    const int64_t total = (int64_t)n * n_ch;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i / n_ch), c = (int)(i - (int64_t)k * n_ch);
        const int64_t j = perm ? perm[k] : k;
        out[i] = rows > 0 ? src[((j >> 5) * rows + (c0 + c)) * 32 + (j & 31)] : src[j * stride + c0 + c];
    }
}

struct GatherBatch {
    int n;
    esr_gather_job_t job[ESR_GATHER_MAX_JOBS];
};
__global__ void __launch_bounds__(256) lts_gather_rows_batch_kernel(GatherBatch B)
{
    const esr_gather_job_t &J = B.job[blockIdx.y];
    const int64_t total = (int64_t)J.n * J.n_ch;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i / J.n_ch), c = (int)(i - (int64_t)k * J.n_ch);
        const int64_t j = J.perm ? J.perm[k] : k;
        J.out[i] = J.tile_rows > 0 ? J.src[((j >> 5) * J.tile_rows + (J.col0 + c)) * 32 + (j & 31)]
                                   : J.src[j * J.row_stride + J.col0 + c];
    }
}

// everything the light-transport segment needs at its P surface points (esrnerf.py:792-806), one thread per point:
// pts2 [2P,3] (the point twice), vd2 [2P,3] (rows < P: the camera direction of the point's ray), sdf2 [2P], unit normal
// (F.normalize of the exact SDF gradient), base colour / roughness / metallic / emission heads, uncertainty mask
struct LtsGather {
    const int64_t *jp;                       // compact slot of every point
    const int64_t *ray64;                    // ray id of every compact slot
    const float *pts_all, *eg, *rec_sdf;     // [T32,3], [T32,4] (sdf | gradient), [T32]
    const float *viewdirs;                   // [N,3]
    const float *brdf_a, *emit_a;            // tile-major heads [T][8][32], [T][4][32]
    const uint8_t *umask_rays;               // [N] (bool)
    int n_pts;
    float *pts2, *vd2, *sdf2, *normal, *base, *rough, *metal, *emis;
    uint8_t *umask;
    int32_t *pt1;                            // optional: slot -> point + 1
};
__global__ void __launch_bounds__(256) lts_gather_points_kernel(LtsGather G)
{
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < G.n_pts; p += gridDim.x * blockDim.x) {
        const int64_t j = G.jp[p];
        const int64_t r = G.ray64[j];
        const int64_t t = j >> 5;
        const int s = (int)(j & 31);
        float g[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float x = G.pts_all[3 * j + a];
            G.pts2[3 * p + a] = x;
            G.pts2[3 * (G.n_pts + p) + a] = x;
            G.vd2[3 * p + a] = G.viewdirs[3 * r + a];
            g[a] = G.eg[4 * j + 1 + a];
            G.base[3 * p + a] = G.brdf_a[(t * 8 + a) * 32 + s];
            G.emis[3 * p + a] = G.emit_a[(t * 4 + a) * 32 + s];
        }
        const float nrm = fmaxf(sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]), 1e-12f);     // F.normalize eps
#pragma unroll
        for (int a = 0; a < 3; ++a) G.normal[3 * p + a] = g[a] / nrm;
        const float sd = G.rec_sdf[j];
        G.sdf2[p] = sd;
        G.sdf2[G.n_pts + p] = sd;
        G.rough[p] = G.brdf_a[(t * 8 + 3) * 32 + s];
        G.metal[p] = G.brdf_a[(t * 8 + 4) * 32 + s];
        G.umask[p] = G.umask_rays[r];
        if (G.pt1) G.pt1[j] = p + 1;
    }
}

// ---- emission edit of the re-lighting fine-tune (esrnerf.py:427-441, pbr/functions.py:214-255) --------
// mode 0: off; 2, 4: intensity scale; 3, 4: hue and saturation replaced (value kept) through the
// reference's rgb<->hsv pair.  `%` below is torch's remainder (result in [0, divisor)).
__device__ __forceinline__ float esr_rem(float a, float b) { return a - floorf(a / b) * b; }

__global__ void __launch_bounds__(256) emit_edit_kernel(float *__restrict__ emit, const int64_t *__restrict__ modes,
                                                        const float *__restrict__ inten,
                                                        const float *__restrict__ colors, int n)
{
#pragma clang fp contract(off)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int64_t m = modes[i];
        float r = emit[3 * i], g = emit[3 * i + 1], b = emit[3 * i + 2];
        if (m == 0) { r = g = b = 0.f; }
        if (m == 2 || m == 4) { const float k = inten[i]; r = r * k; g = g * k; b = b * k; }
        if (m == 3 || m == 4) {
            // rgb -> hsv: only v survives (h, s come from the edit), hsv -> rgb
            const float v = fmaxf(r, fmaxf(g, b));
            const float h = colors[2 * i], sat = colors[2 * i + 1];
            const float h6 = h * 6.f;
            const float sector = esr_rem(floorf(h6), 6.f);
            const float f = esr_rem(h6, 6.f) - sector;
            const float p = v * (1.f - sat), q = v * (1.f - f * sat), t = v * (1.f - (1.f - f) * sat);
            const int k = (int)sector;
            r = k == 0 ? v : k == 1 ? q : k == 2 ? p : k == 3 ? p : k == 4 ? t : v;
            g = k == 0 ? t : k == 1 ? v : k == 2 ? v : k == 3 ? q : k == 4 ? p : p;
            b = k == 0 ? p : k == 1 ? p : k == 2 ? t : k == 3 ? v : k == 4 ? v : q;
        }
        emit[3 * i] = r; emit[3 * i + 1] = g; emit[3 * i + 2] = b;
    }
}

// ---- light-transport combine ----------------------------------------------------------------
struct LtsParams {
    int n_pts, n_rays, n_sg, pdra;            // R = n_rays secondary rays per point
    const float *base, *rough, *metal;        // [P,3] [P] [P]
    const float *normal, *view;               // [P,3] (unit), [P,3]
    const float *dirs;                        // [P,R+1,3]; the last one gives the random view direction
    const float *off_m, *emo_m, *last2;       // [P*R,3] [P*R,3] [P*R]
    const float *mus, *lambdas, *lobes;       // [J,3] [J] [J,3]
    const float *emission;                    // [P,3]
    const uint8_t *umask;                     // [P]
    float *off_hat, *emo_hat;                 // [2P,3]
    // backward
    const float *g_off_hat, *g_emo_hat;       // [2P,3]
    float *d_off_m, *d_emo_m, *d_last2;       // [P*R,3] [P*R,3] [P*R]
    float *d_base, *d_rough, *d_metal, *d_emission;   // [P,3] [P] [P] [P,3]
    float *d_mus, *d_lambdas, *d_lobes;       // accumulated
};

constexpr float kPi = 3.14159265358979323846f;
constexpr int MAX_SG = 64;

struct Disney {
    float R[3];
    float dR_da[3];      // d R[c] / d albedo[c]
    float dR_dro[3], dR_dm[3];
};

// (fd + fs) * (wi.n) * 2 pi with the clamps of the reference; gradients w.r.t. albedo, roughness, metallic
__device__ __forceinline__ Disney disney_eval(const float a[3], float ro, float m, const float n[3],
                                              const float wi[3], const float wo[3])
{
    const float EPS = 1e-7f;
    float hv[3] = {wi[0] + wo[0], wi[1] + wo[1], wi[2] + wo[2]};
    const float hn = fmaxf(sqrtf(hv[0] * hv[0] + hv[1] * hv[1] + hv[2] * hv[2]), 1e-12f);
#pragma unroll
    for (int k = 0; k < 3; ++k) hv[k] /= hn;
    const float noh = fmaxf(n[0] * hv[0] + n[1] * hv[1] + n[2] * hv[2], 0.f);
    const float ooh = fmaxf(wo[0] * hv[0] + wo[1] * hv[1] + wo[2] * hv[2], 0.f);
    const float ion = fmaxf(wi[0] * n[0] + wi[1] * n[1] + wi[2] * n[2], 0.f);
    const float oon = fmaxf(wo[0] * n[0] + wo[1] * n[1] + wo[2] * n[2], 0.f);
    const float r2raw = ro * ro;
    const float r2 = fmaxf(r2raw, EPS);
    const float D = 1.f / (r2 * kPi) * expf(2.f / r2 * (noh - 1.f));
    const float dD_dr2 = D * (-1.f / r2 - 2.f * (noh - 1.f) / (r2 * r2));
    const float dr2_dro = (r2raw >= EPS) ? 2.f * ro : 0.f;          // clamp(min) passes the gradient at equality
    const float om = 1.f - ooh;
    const float t5 = om * om * om * om * om;
    const float k = (1.f + ro) * (1.f + ro) / 8.f, dk_dro = (1.f + ro) / 4.f;
    const float den_i = ion * (1.f - k) + k, den_o = oon * (1.f - k) + k;
    const float Vi = 0.5f / fmaxf(den_i, EPS), Vo = 0.5f / fmaxf(den_o, EPS);
    const float dVi_dk = (den_i >= EPS) ? -0.5f / (den_i * den_i) * (1.f - ion) : 0.f;
    const float dVo_dk = (den_o >= EPS) ? -0.5f / (den_o * den_o) * (1.f - oon) : 0.f;
    const float V = Vi * Vo;
    const float dV_dro = (dVi_dk * Vo + Vi * dVo_dk) * dk_dro;
    const float lam = ion * kPi * 2.f;
    Disney o;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float F0 = 0.04f * (1.f - m) + a[c] * m;
        const float Fr = F0 + (1.f - F0) * t5;
        const float fd = (1.f - m) * a[c] / kPi;
        o.R[c] = (fd + D * Fr * V) * lam;
        o.dR_da[c] = ((1.f - m) / kPi + D * V * m * (1.f - t5)) * lam;
        o.dR_dm[c] = (-a[c] / kPi + D * V * (a[c] - 0.04f) * (1.f - t5)) * lam;
        o.dR_dro[c] = Fr * (dD_dr2 * dr2_dro * V + D * dV_dro) * lam;
    }
    return o;
}

__device__ __forceinline__ float block_sum(float v, float *scratch)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[wave] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += scratch[i];
    return t;
}

// One workgroup per surface point.
template <bool BWD>
__global__ void __launch_bounds__(256) lts_combine_kernel(LtsParams L)
{
    __shared__ float sg_mu[MAX_SG * 3], sg_lobe[MAX_SG * 3], sg_lam[MAX_SG], sg_inv[MAX_SG];
    __shared__ float red[8];
    __shared__ float acc_mu[MAX_SG * 3], acc_lobe[MAX_SG * 3], acc_lam[MAX_SG];
    const int p = blockIdx.x, R = L.n_rays, J = L.n_sg;
    for (int j = threadIdx.x; j < J; j += blockDim.x) {
        const float lx = L.lobes[3 * j], ly = L.lobes[3 * j + 1], lz = L.lobes[3 * j + 2];
        const float nn = fmaxf(sqrtf(lx * lx + ly * ly + lz * lz), 1e-12f);
        sg_lobe[3 * j] = lx / nn; sg_lobe[3 * j + 1] = ly / nn; sg_lobe[3 * j + 2] = lz / nn;
        sg_inv[j] = 1.f / nn;
        sg_lam[j] = fabsf(L.lambdas[j]);
        sg_mu[3 * j] = L.mus[3 * j]; sg_mu[3 * j + 1] = L.mus[3 * j + 1]; sg_mu[3 * j + 2] = L.mus[3 * j + 2];
        if (BWD) {
            acc_mu[3 * j] = acc_mu[3 * j + 1] = acc_mu[3 * j + 2] = 0.f;
            acc_lobe[3 * j] = acc_lobe[3 * j + 1] = acc_lobe[3 * j + 2] = 0.f;
            acc_lam[j] = 0.f;
        }
    }
    __syncthreads();
    const float a[3] = {L.base[3 * p], L.base[3 * p + 1], L.base[3 * p + 2]};
    const float ro = L.rough[p], m = L.metal[p];
    const float n[3] = {L.normal[3 * p], L.normal[3 * p + 1], L.normal[3 * p + 2]};
    const float *dl = L.dirs + (size_t)p * (R + 1) * 3;
    // wout of the two copies: -viewdirs and -viewdirs_rand = +dirs[:, -1]
    const float wo0[3] = {-L.view[3 * p], -L.view[3 * p + 1], -L.view[3 * p + 2]};
    const float wo1[3] = {dl[3 * R], dl[3 * R + 1], dl[3 * R + 2]};
    const bool um = L.umask ? (L.umask[p] != 0) : false;
    const int P = L.n_pts;
    float s_off[2][3] = {{0, 0, 0}, {0, 0, 0}}, s_ref[2][3] = {{0, 0, 0}, {0, 0, 0}};
    float g_base[3] = {0, 0, 0}, g_ro = 0.f, g_m = 0.f;
    float goh[2][3], geh[2][3];
    if (BWD) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                goh[c][k] = L.g_off_hat[3 * (c * P + p) + k] / (float)R;
                // pdra: on uncertain points the reflect term is detached (esrnerf.py:672-675)
                geh[c][k] = (L.pdra && um) ? 0.f : L.g_emo_hat[3 * (c * P + p) + k] / (float)R;
            }
    }
    for (int r0 = 0; r0 < R; r0 += blockDim.x) {
        const int r = r0 + (int)threadIdx.x;
        const bool active = r < R;
        float denv_pre[3] = {0.f, 0.f, 0.f};                  // BWD: d loss / d (pre-softplus env colour) of this ray
        float wi[3] = {0.f, 0.f, 0.f};
        if (active) {
        const size_t s2 = (size_t)p * R + r;
        wi[0] = dl[3 * r]; wi[1] = dl[3 * r + 1]; wi[2] = dl[3 * r + 2];
        // environment map along the secondary ray
        float pre[3] = {0.f, 0.f, 0.f};
        for (int j = 0; j < J; ++j) {
            const float dtl = wi[0] * sg_lobe[3 * j] + wi[1] * sg_lobe[3 * j + 1] + wi[2] * sg_lobe[3 * j + 2];
            const float e = expf(sg_lam[j] * (dtl - 1.f));
            pre[0] += sg_mu[3 * j] * e; pre[1] += sg_mu[3 * j + 1] * e; pre[2] += sg_mu[3 * j + 2] * e;
        }
        const float last = L.last2[s2];
        float inc_off[3], inc_emo[3], envv[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            envv[k] = esr_softplus(pre[k]);
            inc_off[k] = L.off_m[3 * s2 + k] + envv[k] * last;
            inc_emo[k] = L.emo_m[3 * s2 + k];
        }
        const Disney d0 = disney_eval(a, ro, m, n, wi, wo0);
        const Disney d1 = disney_eval(a, ro, m, n, wi, wo1);
        if (!BWD) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                s_off[0][k] += inc_off[k] * d0.R[k]; s_off[1][k] += inc_off[k] * d1.R[k];
                s_ref[0][k] += inc_emo[k] * d0.R[k]; s_ref[1][k] += inc_emo[k] * d1.R[k];
            }
        } else {
            float d_inc_off[3];
            float dlast = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                d_inc_off[k] = goh[0][k] * d0.R[k] + goh[1][k] * d1.R[k];
                L.d_off_m[3 * s2 + k] = d_inc_off[k];
                L.d_emo_m[3 * s2 + k] = geh[0][k] * d0.R[k] + geh[1][k] * d1.R[k];
                dlast += d_inc_off[k] * envv[k];
                denv_pre[k] = d_inc_off[k] * last * (pre[k] > 20.f ? 1.f : esr_sigmoid(pre[k]));
                const float dR0 = goh[0][k] * inc_off[k] + geh[0][k] * inc_emo[k];
                const float dR1 = goh[1][k] * inc_off[k] + geh[1][k] * inc_emo[k];
                g_base[k] += dR0 * d0.dR_da[k] + dR1 * d1.dR_da[k];
                g_ro += dR0 * d0.dR_dro[k] + dR1 * d1.dR_dro[k];
                g_m += dR0 * d0.dR_dm[k] + dR1 * d1.dR_dm[k];
            }
            L.d_last2[s2] = dlast;
        }
        }   // active
        if (BWD) {
            // environment-map parameter gradients: every lane holds one ray's (denv_pre, wi); the 7 sums per
            // lobe are reduced over the wave with shuffles and added to LDS by one lane (256 threads adding to
            // the same 7 x 48 LDS words serialised: 0.66 ms for 100 workgroups)
            const bool any = __any(denv_pre[0] != 0.f || denv_pre[1] != 0.f || denv_pre[2] != 0.f);
            if (any) {
                const int lane = esr_lane();
                for (int j = 0; j < J; ++j) {
                    const float dtl = wi[0] * sg_lobe[3 * j] + wi[1] * sg_lobe[3 * j + 1] + wi[2] * sg_lobe[3 * j + 2];
                    const float e = expf(sg_lam[j] * (dtl - 1.f));
                    float v[7];
                    float de = 0.f;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        v[k] = denv_pre[k] * e;
                        de += denv_pre[k] * sg_mu[3 * j + k];
                    }
                    de *= e;
                    v[3] = de * (dtl - 1.f);
                    const float dd = de * sg_lam[j];               // d / d (wi . lobe_unit)
#pragma unroll
                    for (int k = 0; k < 3; ++k) v[4 + k] = dd * wi[k];
#pragma unroll
                    for (int q = 0; q < 7; ++q)
#pragma unroll
                        for (int off = 32; off > 0; off >>= 1) v[q] += __shfl_xor(v[q], off);
                    if (lane == 0) {
#pragma unroll
                        for (int k = 0; k < 3; ++k) atomicAdd(&acc_mu[3 * j + k], v[k]);
                        atomicAdd(&acc_lam[j], v[3]);
#pragma unroll
                        for (int k = 0; k < 3; ++k) atomicAdd(&acc_lobe[3 * j + k], v[4 + k]);
                    }
                }
            }
        }
    }
    if (!BWD) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float so = block_sum(s_off[c][k], red) / (float)R;
                const float sr = block_sum(s_ref[c][k], red) / (float)R;
                if (threadIdx.x == 0) {
                    L.off_hat[3 * (c * P + p) + k] = so;
                    // lts: emission + reflect everywhere; pdra: reflect alone on certain points
                    const float em = (L.pdra && !um) ? 0.f : L.emission[3 * p + k];
                    L.emo_hat[3 * (c * P + p) + k] = em + sr;
                }
            }
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float t = block_sum(g_base[k], red);
            if (threadIdx.x == 0) {
                L.d_base[3 * p + k] = t;
                const bool has_em = !(L.pdra && !um);
                L.d_emission[3 * p + k] =
                    has_em ? (L.g_emo_hat[3 * p + k] + L.g_emo_hat[3 * (P + p) + k]) : 0.f;
            }
        }
        const float tr = block_sum(g_ro, red), tm = block_sum(g_m, red);
        if (threadIdx.x == 0) { L.d_rough[p] = tr; L.d_metal[p] = tm; }
        __syncthreads();
        // flush the SG parameter gradients of this workgroup (through |lambda| and normalize(lobe))
        for (int j = threadIdx.x; j < J; j += blockDim.x) {
#pragma unroll
            for (int k = 0; k < 3; ++k) atomicAdd(&L.d_mus[3 * j + k], acc_mu[3 * j + k]);
            const float sgn = L.lambdas[j] > 0.f ? 1.f : (L.lambdas[j] < 0.f ? -1.f : 0.f);
            atomicAdd(&L.d_lambdas[j], acc_lam[j] * sgn);
            const float gl[3] = {acc_lobe[3 * j], acc_lobe[3 * j + 1], acc_lobe[3 * j + 2]};
            const float dotl = gl[0] * sg_lobe[3 * j] + gl[1] * sg_lobe[3 * j + 1] + gl[2] * sg_lobe[3 * j + 2];
#pragma unroll
            for (int k = 0; k < 3; ++k)
                atomicAdd(&L.d_lobes[3 * j + k], (gl[k] - sg_lobe[3 * j + k] * dotl) * sg_inv[j]);
        }
    }
}

}  // namespace

ESR_API int esr_expgrad_fwd(const esr_scene_t *scene, const float *rays_o, const float *rays_d,
                            const int32_t *rec_ray, const int32_t *rec_step, const float *pts,
                            const float *noise, float eps, const float *sdf, int32_t n, int zero_pad,
                            float *out, void *stream)
{
    if (!scene || n < 0) return ESR_EINVAL;
    if (n == 0) return 0;
    if (!sdf || !out || (!pts && (!rays_o || !rays_d || !rec_ray || !rec_step))) return ESR_EINVAL;
    ExpGradParams P = {};
    P.sc = *scene; P.rays_o = rays_o; P.rays_d = rays_d; P.rec_ray = rec_ray; P.rec_step = rec_step;
    P.pts = pts; P.noise = noise; P.eps = eps; P.sdf = sdf; P.n = n; P.out = out; P.zero_pad = zero_pad;
    expgrad_kernel<false><<<esr_grid_for(n, 256), 256, 0, esr_stream(stream)>>>(P);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_expgrad_bwd(const esr_scene_t *scene, const float *rays_o, const float *rays_d,
                            const int32_t *rec_ray, const int32_t *rec_step, const float *pts,
                            const float *noise, float eps, const float *g, int32_t n, int zero_pad,
                            float *grad_sdf, void *stream)
{
    if (!scene || n < 0) return ESR_EINVAL;
    if (n == 0) return 0;
    if (!g || !grad_sdf || (!pts && (!rays_o || !rays_d || !rec_ray || !rec_step))) return ESR_EINVAL;
    ExpGradParams P = {};
    P.sc = *scene; P.rays_o = rays_o; P.rays_d = rays_d; P.rec_ray = rec_ray; P.rec_step = rec_step;
    P.pts = pts; P.noise = noise; P.eps = eps; P.n = n; P.g = g; P.grad_sdf = grad_sdf; P.zero_pad = zero_pad;
    expgrad_kernel<true><<<esr_grid_for(n, 256), 256, 0, esr_stream(stream)>>>(P);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_lts_dirs(const float *raw, const float *normal, int32_t n_pts, int32_t rays_plus_one,
                         float *dirs, void *stream)
{
    if (n_pts < 0 || rays_plus_one < 1) return ESR_EINVAL;
    if (n_pts == 0) return 0;
    if (!raw || !normal || !dirs) return ESR_EINVAL;
    lts_dirs_kernel<<<esr_grid_for((int64_t)n_pts * rays_plus_one, 256), 256, 0, esr_stream(stream)>>>(
        raw, normal, n_pts, rays_plus_one, dirs, nullptr, nullptr, nullptr, nullptr);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_lts_dirs_rays(const float *raw, const float *normal, const float *pts, int32_t n_pts, int32_t rays_plus_one,
                              float *dirs, float *o2, float *d2, float *v_rand, void *stream)
{
    if (n_pts < 0 || rays_plus_one < 2) return ESR_EINVAL;
    if (n_pts == 0) return 0;
    if (!raw || !normal || !pts || !dirs || !o2 || !d2 || !v_rand) return ESR_EINVAL;
    lts_dirs_kernel<<<esr_grid_for((int64_t)n_pts * rays_plus_one, 256), 256, 0, esr_stream(stream)>>>(
        raw, normal, n_pts, rays_plus_one, dirs, pts, o2, d2, v_rand);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_lts_ref_order(const int32_t *rec_ray, const int32_t *cnt3, const int32_t *off3, const int64_t *cnt3_cumsum,
                              int32_t n_slots, int64_t *perm, int64_t *ray64, void *stream)
{
    if (n_slots < 0) return ESR_EINVAL;
    if (n_slots == 0) return 0;
    if (!rec_ray || !cnt3 || !off3 || !cnt3_cumsum || !perm || !ray64) return ESR_EINVAL;
    lts_ref_order_kernel<<<esr_grid_for(n_slots, 256), 256, 0, esr_stream(stream)>>>(rec_ray, cnt3, off3, cnt3_cumsum, n_slots,
                                                                                    perm, ray64, nullptr);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_lts_ref_order_inv(const int32_t *rec_ray, const int32_t *cnt3, const int32_t *off3, const int64_t *cnt3_cumsum,
                                  int32_t n_slots, int64_t *perm, int64_t *ray64, int32_t *inv, void *stream)
{
    if (n_slots < 0) return ESR_EINVAL;
    if (n_slots == 0) return 0;
    if (!rec_ray || !cnt3 || !off3 || !cnt3_cumsum || !perm || !ray64 || !inv) return ESR_EINVAL;
    lts_ref_order_kernel<<<esr_grid_for(n_slots, 256), 256, 0, esr_stream(stream)>>>(rec_ray, cnt3, off3, cnt3_cumsum, n_slots,
                                                                                    perm, ray64, inv);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_lts_perturb(const float *pts_all, const int64_t *perm, const float *noise_normal, const float *noise_emit,
                            float emit_eps, int32_t m3, float *noise_n, float *pts_e, void *stream)
{
    if (m3 < 0) return ESR_EINVAL;
    if (m3 == 0) return 0;
    if (!pts_all || !perm || !noise_normal || !noise_emit || !noise_n || !pts_e) return ESR_EINVAL;
    lts_perturb_kernel<<<esr_grid_for(m3, 256), 256, 0, esr_stream(stream)>>>(pts_all, perm, noise_normal, noise_emit, emit_eps,
                                                                              m3, noise_n, pts_e);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_lts_gather_rows(const float *src, int32_t tile_rows, int32_t row_stride, int32_t col0, int32_t n_ch,
                                const int64_t *perm, int32_t n, float *out, void *stream)
{
    if (n < 0 || n_ch < 1 || col0 < 0 || tile_rows < 0 || (tile_rows == 0 && row_stride < col0 + n_ch) ||
        (tile_rows > 0 && tile_rows < col0 + n_ch))
        return ESR_EINVAL;
    if (n == 0) return 0;
    if (!src || !out) return ESR_EINVAL;
    lts_gather_rows_kernel<<<esr_grid_for((int64_t)n * n_ch, 256), 256, 0, esr_stream(stream)>>>(src, tile_rows, row_stride, col0,
                                                                                                n_ch, perm, n, out);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_lts_gather_rows_batch(const esr_gather_job_t *jobs, int32_t n_jobs, void *stream)
{
    if (n_jobs < 0 || n_jobs > ESR_GATHER_MAX_JOBS || (n_jobs > 0 && !jobs)) return ESR_EINVAL;
    GatherBatch B = {};
    int64_t most = 0;
    for (int i = 0; i < n_jobs; ++i) {
        const esr_gather_job_t &J = jobs[i];
        if (J.n < 0 || J.n_ch < 1 || J.col0 < 0 || J.tile_rows < 0 || (J.tile_rows == 0 && J.row_stride < J.col0 + J.n_ch) ||
            (J.tile_rows > 0 && J.tile_rows < J.col0 + J.n_ch))
            return ESR_EINVAL;
        if (J.n == 0) continue;
        if (!J.src || !J.out) return ESR_EINVAL;
        B.job[B.n++] = J;
        const int64_t tot = (int64_t)J.n * J.n_ch;
        most = tot > most ? tot : most;
    }
    if (B.n == 0) return 0;
    lts_gather_rows_batch_kernel<<<dim3(esr_grid_for(most, 256, 512), B.n), 256, 0, esr_stream(stream)>>>(B);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_lts_gather_points(const esr_lts_gather_t *g, void *stream)
{
    if (!g || g->n_pts < 0) return ESR_EINVAL;
    if (g->n_pts == 0) return 0;
    if (!g->jp || !g->ray64 || !g->pts_all || !g->eg || !g->rec_sdf || !g->viewdirs || !g->brdf_a || !g->emit_a ||
        !g->umask_rays || !g->pts2 || !g->vd2 || !g->sdf2 || !g->normal || !g->base || !g->rough || !g->metal || !g->emis ||
        !g->umask)
        return ESR_EINVAL;
    LtsGather G;
    G.jp = g->jp; G.ray64 = g->ray64; G.pts_all = g->pts_all; G.eg = g->eg; G.rec_sdf = g->rec_sdf; G.viewdirs = g->viewdirs;
    G.brdf_a = g->brdf_a; G.emit_a = g->emit_a; G.umask_rays = g->umask_rays; G.n_pts = g->n_pts;
    G.pts2 = g->pts2; G.vd2 = g->vd2; G.sdf2 = g->sdf2; G.normal = g->normal; G.base = g->base; G.rough = g->rough;
    G.metal = g->metal; G.emis = g->emis; G.umask = g->umask; G.pt1 = g->pt1;
    lts_gather_points_kernel<<<esr_grid_for(g->n_pts, 256), 256, 0, esr_stream(stream)>>>(G);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_emit_edit(float *emit, const int64_t *em_modes, const float *em_intensities,
                          const float *em_colors, int32_t n, void *stream)
{
    if (n < 0) return ESR_EINVAL;
    if (n == 0) return 0;
    if (!emit || !em_modes || !em_intensities || !em_colors) return ESR_EINVAL;
    emit_edit_kernel<<<esr_grid_for(n, 256), 256, 0, esr_stream(stream)>>>(emit, em_modes, em_intensities,
                                                                          em_colors, n);
    ESR_CHECK_LAUNCH();
    return 0;
}

static int lts_check(const esr_lts_args_t *a)
{
    if (!a || a->n_pts < 0 || a->n_rays < 1 || a->n_sg < 1 || a->n_sg > MAX_SG) return ESR_EINVAL;
    if (a->n_pts == 0) return 0;
    if (!a->base || !a->rough || !a->metal || !a->normal || !a->view || !a->dirs || !a->off_m || !a->emo_m ||
        !a->last2 || !a->mus || !a->lambdas || !a->lobes || !a->emission)
        return ESR_EINVAL;
    return 1;
}

static LtsParams lts_params(const esr_lts_args_t *a)
{
    LtsParams L = {};
    L.n_pts = a->n_pts; L.n_rays = a->n_rays; L.n_sg = a->n_sg; L.pdra = a->pdra_mode;
    L.base = a->base; L.rough = a->rough; L.metal = a->metal; L.normal = a->normal; L.view = a->view;
    L.dirs = a->dirs; L.off_m = a->off_m; L.emo_m = a->emo_m; L.last2 = a->last2;
    L.mus = a->mus; L.lambdas = a->lambdas; L.lobes = a->lobes; L.emission = a->emission; L.umask = a->umask;
    return L;
}

ESR_API int esr_lts_combine_fwd(const esr_lts_args_t *a, float *off_hat, float *emo_hat, void *stream)
{
    const int c = lts_check(a);
    if (c <= 0) return c;
    if (!off_hat || !emo_hat) return ESR_EINVAL;
    LtsParams L = lts_params(a);
    L.off_hat = off_hat; L.emo_hat = emo_hat;
    lts_combine_kernel<false><<<a->n_pts, 256, 0, esr_stream(stream)>>>(L);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_lts_combine_bwd(const esr_lts_args_t *a, const float *g_off_hat, const float *g_emo_hat,
                                const esr_lts_grads_t *g, void *stream)
{
    const int c = lts_check(a);
    if (c <= 0) return c;
    if (!g_off_hat || !g_emo_hat || !g || !g->d_off_m || !g->d_emo_m || !g->d_last2 || !g->d_base ||
        !g->d_rough || !g->d_metal || !g->d_emission || !g->d_mus || !g->d_lambdas || !g->d_lobes)
        return ESR_EINVAL;
    LtsParams L = lts_params(a);
    L.g_off_hat = g_off_hat; L.g_emo_hat = g_emo_hat;
    L.d_off_m = g->d_off_m; L.d_emo_m = g->d_emo_m; L.d_last2 = g->d_last2; L.d_base = g->d_base;
    L.d_rough = g->d_rough; L.d_metal = g->d_metal; L.d_emission = g->d_emission; L.d_mus = g->d_mus;
    L.d_lambdas = g->d_lambdas; L.d_lobes = g->d_lobes;
    lts_combine_kernel<true><<<a->n_pts, 256, 0, esr_stream(stream)>>>(L);
    ESR_CHECK_LAUNCH();
    return 0;
}
