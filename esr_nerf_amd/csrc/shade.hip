// Element-wise stages between the MLP passes, compositing, and the trainer loss.
//
// Reference algorithm (paths under the reference tree):
//   app/fine/model/voxurff.py:243-256  lin_rgb = softplus(emo) + softplus(off).detach() | softplus(off)
//   app/fine/model/voxurff.py:783-788  tonemapper input [lin, sin(lin*2^i), cos(lin*2^i)]
//   app/fine/model/voxurff.py:258-272  weights * rgb -> segment_coo(sum) per ray
//   app/fine/fine.py:355-382 + utils2/image.py:14-26   loss of one training step
//
// All buffers are tile-major [tile][row][32 samples]: consecutive lanes touch
// consecutive addresses on every access.  Compositing reduces inside the wave with a
// segmented shuffle scan (samples are sorted by ray) and issues one float atomic per
// (ray segment, channel) -- the MI355X replacement of torch_scatter.segment_coo.
#include "esr_common.h"

namespace {

constexpr int XT_ROWS = 48;

__device__ __forceinline__ float softplus_grad(float z) { return z > 20.f ? 1.f : esr_sigmoid(z); }

__global__ void __launch_bounds__(256) tone_in_fwd_kernel(const float *__restrict__ z_off,
                                                          const float *__restrict__ z_emo, int tiles_on,
                                                          int tiles_all, float *__restrict__ lin,
                                                          float *__restrict__ Xt)
{
    const int total = tiles_all * 32;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < total; j += gridDim.x * blockDim.x) {
        const int t = j >> 5, s = j & 31;
        const size_t z4 = (size_t)t * 4 * 32 + s;
        float *X = Xt + (size_t)t * XT_ROWS * 32 + s;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = esr_softplus(z_off[z4 + c * 32]);
            if (t < tiles_on) v = esr_softplus(z_emo[z4 + c * 32]) + v;
            lin[z4 + c * 32] = v;
            X[c * 32] = v;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const float a = v * (float)(1 << i);
                float sn, cs;
                esr_sincos(a, sn, cs);
                X[(3 + c * 5 + i) * 32] = sn;
                X[(18 + c * 5 + i) * 32] = cs;
            }
        }
        lin[z4 + 96] = 0.f;
        for (int r = 33; r < XT_ROWS; ++r) X[r * 32] = 0.f;
    }
}

// segmented inclusive scan over the 64 lanes (segments = runs of equal key)
__device__ __forceinline__ float seg_scan(float v, int key, int lane)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float u = __shfl_up(v, off);
        const int ku = __shfl_up(key, off);
        if (lane >= off && ku == key) v += u;
    }
    return v;
}

__global__ void __launch_bounds__(256) composite_fwd_kernel(const float *__restrict__ zt,
                                                            const float *__restrict__ lin,
                                                            const int32_t *__restrict__ rec_ray,
                                                            const float *__restrict__ rec_w, int tiles_all,
                                                            float *__restrict__ rgb,
                                                            float *__restrict__ srgb_marched,
                                                            float *__restrict__ lin_marched)
{
    const int lane = esr_lane();
    const int total = tiles_all * 32;                     // multiple of 32; waves cover 64
    const int stride = gridDim.x * blockDim.x;
    for (int j0 = (blockIdx.x * blockDim.x + threadIdx.x) - lane; j0 < total; j0 += stride) {
        const int j = j0 + lane;
        const bool in = j < total;
        const int t = j >> 5, s = j & 31;
        const size_t z4 = (size_t)t * 4 * 32 + s;
        const int ray = in ? rec_ray[j] : -1;
        const float w = (in && ray >= 0) ? rec_w[j] : 0.f;
        const int ray_next = __shfl_down(ray, 1);
        const bool tail = ray >= 0 && (lane == 63 || ray_next != ray);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float col = 0.f, l = 0.f;
            if (in) {
                col = esr_sigmoid(zt[z4 + c * 32]);
                rgb[z4 + c * 32] = col;
                l = lin[z4 + c * 32];
            }
            const float a = seg_scan(w * col, ray, lane);
            const float b = seg_scan(w * l, ray, lane);
            if (tail) {
                atomicAdd(&srgb_marched[3 * ray + c], a);
                atomicAdd(&lin_marched[3 * ray + c], b);
            }
        }
        if (in) rgb[z4 + 96] = 0.f;
    }
}

__global__ void __launch_bounds__(256) composite_bwd_kernel(const float *__restrict__ g_srgb,
                                                            const float *__restrict__ g_lin,
                                                            const float *__restrict__ rgb,
                                                            const float *__restrict__ lin,
                                                            const int32_t *__restrict__ rec_ray,
                                                            const float *__restrict__ rec_w, int tiles_all,
                                                            float *__restrict__ dweight,
                                                            float *__restrict__ dzt)
{
    const int total = tiles_all * 32;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < total; j += gridDim.x * blockDim.x) {
        const int t = j >> 5, s = j & 31;
        const size_t z4 = (size_t)t * 4 * 32 + s;
        const int ray = rec_ray[j];
        float dw = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float d = 0.f;
            if (ray >= 0) {
                const float col = rgb[z4 + c * 32];
                const float gs = g_srgb[3 * ray + c];
                dw += gs * col + g_lin[3 * ray + c] * lin[z4 + c * 32];
                d = rec_w[j] * gs * col * (1.f - col);
            }
            dzt[z4 + c * 32] = d;
        }
        dzt[z4 + 96] = 0.f;
        dweight[j] = dw;
    }
}

// (the derivative's cos(a) / sin(a) factors are the forward's own rows of Xt: the 30 accurate sin / cos evaluations per
// sample of a recomputation were most of this kernel)
__global__ void __launch_bounds__(256) tone_in_bwd_kernel(
    const float *__restrict__ dXt, const float *__restrict__ Xt, const float *__restrict__ g_lin, const float *__restrict__ lin,
    const float *__restrict__ z_off, const float *__restrict__ z_emo, const int32_t *__restrict__ rec_ray,
    const float *__restrict__ rec_w, int tiles_on, int tiles_all, float *__restrict__ dz)
{
    const int total = tiles_all * 32;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < total; j += gridDim.x * blockDim.x) {
        const int t = j >> 5, s = j & 31;
        const size_t z4 = (size_t)t * 4 * 32 + s;
        const float *dX = dXt + (size_t)t * 64 * 32 + s;
        const float *X = Xt + (size_t)t * XT_ROWS * 32 + s;
        const int ray = rec_ray[j];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float d = 0.f;
            if (ray >= 0) {
                d = rec_w[j] * g_lin[3 * ray + c] + dX[c * 32];
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    const float f = (float)(1 << i);
                    d += f * (dX[(3 + c * 5 + i) * 32] * X[(18 + c * 5 + i) * 32] - dX[(18 + c * 5 + i) * 32] * X[(3 + c * 5 + i) * 32]);
                }
                const float z = (t < tiles_on) ? z_emo[z4 + c * 32] : z_off[z4 + c * 32];
                d *= softplus_grad(z);
            }
            dz[z4 + c * 32] = d;
        }
        dz[z4 + 96] = 0.f;
    }
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__global__ void __launch_bounds__(256) loss_kernel(const float *__restrict__ srgb_m,
                                                   const float *__restrict__ lin_m,
                                                   const float *__restrict__ last,
                                                   const float *__restrict__ rgbs, int n_rays, float white_bg,
                                                   float w_lin, float w_ent, float scale, float *__restrict__ loss,
                                                   float *__restrict__ g_srgb, float *__restrict__ g_lin,
                                                   float *__restrict__ g_last)
{
    // scale: a rank's share n_local / n_global of a data-parallel batch (1: the whole batch) -- applied to every term
    const float inv = scale == 1.f ? 1.f / (3.f * (float)n_rays) : scale / (3.f * (float)n_rays);
    w_ent *= scale;
    float acc = 0.f;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n_rays; r += gridDim.x * blockDim.x) {
        const float al = last[r];
        const float bg = al * white_bg;
        float gl = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float gt = rgbs[3 * r + c];
            // srgb term
            const float ps = srgb_m[3 * r + c] + bg;
            const float xs = fminf(fmaxf(ps, 0.f), 1.f);
            acc += (xs - gt) * (xs - gt) * inv;
            const float gs = (ps >= 0.f && ps <= 1.f) ? 2.f * (xs - gt) * inv : 0.f;
            g_srgb[3 * r + c] = gs;
            // linear term through the sRGB transfer curve
            const float pl = lin_m[3 * r + c] + bg;
            const float l0 = fmaxf(pl, 0.f);
            const bool sat = gt >= 1.f;
            const float x = sat ? fminf(l0, 1.f) : l0;
            const bool low = x <= 0.0031308f;
            const float y = low ? 12.92f * x : 1.055f * powf(x, 1.f / 2.4f) - 0.055f;
            acc += w_lin * (y - gt) * (y - gt) * inv;
            float g = w_lin * 2.f * (y - gt) * inv;
            g *= low ? 12.92f : 1.055f * (1.f / 2.4f) * powf(x, 1.f / 2.4f - 1.f);
            if (sat && !(l0 <= 1.f)) g = 0.f;       // clamp(max=1) passes the gradient for l0 <= 1
            if (!(pl >= 0.f)) g = 0.f;              // clamp(min=0)
            g_lin[3 * r + c] = g;
            gl += (gs + g) * white_bg;
        }
        if (r == n_rays - 1) {                      // reference quirk: entropy of the LAST ray only
            const float p = fminf(fmaxf(al, 1e-6f), 1.f - 1e-6f);
            acc += w_ent * -(p * logf(p) + (1.f - p) * logf(1.f - p));
            if (al >= 1e-6f && al <= 1.f - 1e-6f) gl += w_ent * -(logf(p) - logf(1.f - p));
        }
        g_last[r] = gl;
    }
    acc = wave_sum(acc);
    if (esr_lane() == 0 && acc != 0.f) atomicAdd(loss, acc);
}

// One mean-reduced two-operand term of the LTS / PDRA trainer losses (lts.py:362-379, pdra.py:408-457)
// with its gradients; rows selected by row_mask == mask_value, b == nullptr stands for zeros.
__global__ void __launch_bounds__(256) pair_loss_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                        int64_t rows, int cols, const uint8_t *__restrict__ row_mask,
                                                        int mask_value, const int32_t *__restrict__ count_dev, int kind,
                                                        float w_value, float w_a, float w_b, float *__restrict__ loss,
                                                        float *__restrict__ ga, float *__restrict__ gb)
{
    const int64_t total = rows * cols;
    const int64_t n_sel = count_dev ? (int64_t)count_dev[0] * cols : total;
    const float inv = n_sel > 0 ? 1.f / (float)n_sel : 0.f;
    float acc = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const bool sel = !row_mask || (int)row_mask[i / cols] == mask_value;
        float g = 0.f;
        if (sel) {
            const float d = a[i] - (b ? b[i] : 0.f);
            if (kind == 0) { acc += d * d * inv; g = 2.f * d * inv; }
            else { acc += fabsf(d) * inv; g = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * inv; }
        }
        if (ga) ga[i] = w_a * g;
        if (gb) gb[i] = -w_b * g;
    }
    // one atomic per WORKGROUP (and at most 256 workgroups: see the launch): same-address float atomics serialise in the
    // L2 -- one per wave of a 1500-workgroup grid made this kernel 35 us at C4
    __shared__ float part[4];
    acc = wave_sum(acc);
    if (esr_lane() == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float tot = ((part[0] + part[1]) + (part[2] + part[3])) * w_value;
        if (tot != 0.f) atomicAdd(loss, tot);
    }
}

// ---- image rendering (forward_evaluate) helpers --------------------------------------------------
// aux [tiles,8,32]: rows 0-2 camera-space normal colour ((n @ pos_rt) * (1,-1,-1) + 1) / 2 with n the unit
// normal the feature tile already holds (fine: the radius-1 finite difference, X rows 31 + axis*4 + 1 in the
// reference's (z,y,x) order, voxurff.py:431-435; coarse: rows 24-26, voxurfc.py:395-397), row 4: step_id *
// stepdist (depth integrand, voxurff.py:437-441), rest 0.
struct EvalAux {
    const float *X;
    const int32_t *rec_ray, *rec_step;
    int tiles, xrows;
    int nrow[3];               // X rows holding the unit normal's world x, y, z components
    float rt[9], stepdist;
    float *aux;
};
__global__ void __launch_bounds__(256) eval_aux_kernel(EvalAux A)
{
    const int total = A.tiles * 32;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < total; j += gridDim.x * blockDim.x) {
        const int t = j >> 5, s = j & 31;
        float *o = A.aux + (size_t)t * 8 * 32 + s;
        const bool ok = A.rec_ray[j] >= 0;
        const float *X = A.X + (size_t)t * A.xrows * 32 + s;
        float n[3] = {0.f, 0.f, 0.f};
        if (ok) { n[0] = X[A.nrow[0] * 32]; n[1] = X[A.nrow[1] * 32]; n[2] = X[A.nrow[2] * 32]; }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = n[0] * A.rt[0 * 3 + c] + n[1] * A.rt[1 * 3 + c] + n[2] * A.rt[2 * 3 + c];
            v = (v * (c == 0 ? 1.f : -1.f) + 1.f) / 2.f;
            o[c * 32] = ok ? v : 0.f;
        }
        o[3 * 32] = 0.f;
        o[4 * 32] = ok ? (float)A.rec_step[j] * A.stepdist : 0.f;
        o[5 * 32] = 0.f; o[6 * 32] = 0.f; o[7 * 32] = 0.f;
    }
}

__global__ void __launch_bounds__(256) eval_disp_kernel(const float *__restrict__ depth3, const float *__restrict__ last,
                                                        float far_, int n, float *__restrict__ depth,
                                                        float *__restrict__ disp)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float d = depth3[3 * i];
        depth[i] = d;
        disp[i] = 1.f / (d + last[i] * far_);
    }
}

// ---- LTS-stage helpers ------------------------------------------------------------------------
// out = act(z) on the first n_ch rows of [tiles, rows, 32] tiles (act 0: softplus, 1: sigmoid);
// backward: dz = g * act'(z)
template <bool BWD>
__global__ void __launch_bounds__(256) act_kernel(const float *__restrict__ z, const float *__restrict__ g,
                                                  int tiles, int rows, int n_ch, int act, float *__restrict__ out)
{
    const int64_t total = (int64_t)tiles * rows * 32;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int row = (int)((i >> 5) % rows);
        float v = 0.f;
        if (row < n_ch) {
            const float x = z[i];
            if (!BWD) v = act ? esr_sigmoid(x) : esr_softplus(x);
            else {
                const float sg = esr_sigmoid(x);
                v = g[i] * (act ? sg * (1.f - sg) : (x > 20.f ? 1.f : sg));
            }
        }
        out[i] = v;
    }
}

// Up to ESR_ACT_MAX_JOBS activation jobs in one launch (blockIdx.y = job); a backward job also gathers the row-major
// gradient sources of its head (include/esr_hip.h: esr_act_job_t) -- the index_put / index_add_ / cat / permute-copy glue
// that stood in front of every act_bwd launch.
struct ActBatch {
    int n;
    esr_act_job_t job[ESR_ACT_MAX_JOBS];
};
__global__ void __launch_bounds__(256) act_batch_kernel(ActBatch B)
{
    const esr_act_job_t &J = B.job[blockIdx.y];
    const int64_t total = (int64_t)J.tiles * J.rows * 32;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int row = (int)((i >> 5) % J.rows);
        float v = 0.f;
        if (row < J.n_ch) {
            const float x = J.z[i];
            if (!J.bwd) v = J.act ? esr_sigmoid(x) : esr_softplus(x);
            else {
                const int64_t slot = (i / (32 * (int64_t)J.rows)) * 32 + (i & 31);
                float g = J.g_tile ? J.g_tile[i] : 0.f;
                if (J.src && row < J.src_c) {
                    const int64_t k = J.inv ? (int64_t)J.inv[slot] : slot;
                    if (k >= 0 && k < J.n_src) g += J.src[k * J.src_c + row];
                }
                if (J.pt1) {
                    const int p = J.pt1[slot] - 1;
                    if (p >= 0) {
#pragma unroll
                        for (int e = 0; e < 3; ++e)
                            if (J.ex[e] && row >= J.ex_col0[e] && row < J.ex_col0[e] + J.ex_c[e])
                                g += J.ex[e][(int64_t)p * J.ex_c[e] + (row - J.ex_col0[e])];
                    }
                }
                const float sg = esr_sigmoid(x);
                v = g * (J.act ? sg * (1.f - sg) : (x > 20.f ? 1.f : sg));
            }
        }
        J.out[i] = v;
    }
}

struct PairBatch {
    int n;
    esr_pair_job_t job[ESR_PAIR_MAX_JOBS];
    float *loss;
};
__global__ void __launch_bounds__(256) pair_loss_batch_kernel(PairBatch B)
{
    const esr_pair_job_t &J = B.job[blockIdx.y];
    const int64_t total = J.rows * J.cols;
    const int64_t n_sel = J.count_dev ? (int64_t)J.count_dev[0] * J.cols : total;
    const float inv = n_sel > 0 ? 1.f / (float)n_sel : 0.f;
    float acc = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const bool sel = !J.row_mask || (int)J.row_mask[i / J.cols] == J.mask_value;
        float g = 0.f;
        if (sel) {
            const float d = J.a[i] - (J.b ? J.b[i] : 0.f);
            if (J.kind == 0) { acc += d * d * inv; g = 2.f * d * inv; }
            else { acc += fabsf(d) * inv; g = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * inv; }
        }
        if (J.ga) J.ga[i] = J.w_a * g;
        if (J.gb) J.gb[i] = -J.w_b * g;
    }
    __shared__ float part[4];
    acc = wave_sum(acc);
    if (esr_lane() == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float tot = ((part[0] + part[1]) + (part[2] + part[3])) * J.w_value;
        if (tot != 0.f) atomicAdd(B.loss, tot);
    }
}

// out[ray, c] += w * v[c] for a 3-channel tile-major quantity (segmented wave reduction)
__global__ void __launch_bounds__(256) composite3_fwd_kernel(const float *__restrict__ v, int rows,
                                                             const int32_t *__restrict__ rec_ray,
                                                             const float *__restrict__ rec_w, int tiles,
                                                             float *__restrict__ out)
{
    const int lane = esr_lane();
    const int total = tiles * 32;
    const int stride = gridDim.x * blockDim.x;
    for (int j0 = (blockIdx.x * blockDim.x + threadIdx.x) - lane; j0 < total; j0 += stride) {
        const int j = j0 + lane;
        const bool in = j < total;
        const int t = j >> 5, s = j & 31;
        const int ray = in ? rec_ray[j] : -1;
        const float w = (in && ray >= 0) ? rec_w[j] : 0.f;
        const int ray_next = __shfl_down(ray, 1);
        const bool tail = ray >= 0 && (lane == 63 || ray_next != ray);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float x = in ? v[((size_t)t * rows + c) * 32 + s] : 0.f;
            const float a = seg_scan(w * x, ray, lane);
            if (tail) atomicAdd(&out[3 * ray + c], a);
        }
    }
}

// dv[c] (+)= w * g[ray, c];  dweight (+)= sum_c g[ray, c] * v[c]
__global__ void __launch_bounds__(256) composite3_bwd_kernel(const float *__restrict__ g,
                                                             const float *__restrict__ v, int rows,
                                                             const int32_t *__restrict__ rec_ray,
                                                             const float *__restrict__ rec_w, int tiles,
                                                             int accumulate, float *__restrict__ dv,
                                                             float *__restrict__ dweight)
{
    const int total = tiles * 32;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < total; j += gridDim.x * blockDim.x) {
        const int t = j >> 5, s = j & 31;
        const int ray = rec_ray[j];
        float dw = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const size_t i = ((size_t)t * rows + c) * 32 + s;
            float d = 0.f;
            if (ray >= 0) {
                d = rec_w[j] * g[3 * ray + c];
                dw += g[3 * ray + c] * v[i];
            }
            dv[i] = (accumulate & 1) ? dv[i] + d : d;
        }
        dweight[j] = (accumulate & 2) ? dweight[j] + dw : dw;
    }
}

// LTS renderer: lin = softplus(z_off) + [on] softplus(z_emo) with NO detach (esrnerf.py:751-757):
// dz_off on every tile, dz_emo on the on-tiles.  dlin_extra [tiles,4,32] (optional) is added.
__global__ void __launch_bounds__(256) lts_tone_in_bwd_kernel(
    const float *__restrict__ dXt, const float *__restrict__ Xt, const float *__restrict__ g_lin, const float *__restrict__ lin,
    const float *__restrict__ z_off, const float *__restrict__ z_emo, const int32_t *__restrict__ rec_ray,
    const float *__restrict__ rec_w, int tiles_on, int tiles_all, float *__restrict__ dz_off,
    float *__restrict__ dz_emo)
{
    const int total = tiles_all * 32;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < total; j += gridDim.x * blockDim.x) {
        const int t = j >> 5, s = j & 31;
        const size_t z4 = (size_t)t * 4 * 32 + s;
        const float *dX = dXt + (size_t)t * 64 * 32 + s;
        const float *X = Xt + (size_t)t * XT_ROWS * 32 + s;
        const int ray = rec_ray[j];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float d = 0.f;
            if (ray >= 0) {
                d = rec_w[j] * g_lin[3 * ray + c] + dX[c * 32];
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    const float f = (float)(1 << i);
                    d += f * (dX[(3 + c * 5 + i) * 32] * X[(18 + c * 5 + i) * 32] - dX[(18 + c * 5 + i) * 32] * X[(3 + c * 5 + i) * 32]);
                }
            }
            dz_off[z4 + c * 32] = d * softplus_grad(z_off[z4 + c * 32]);
            if (t < tiles_on) dz_emo[z4 + c * 32] = d * softplus_grad(z_emo[z4 + c * 32]);
        }
        dz_off[z4 + 96] = 0.f;
        if (t < tiles_on) dz_emo[z4 + 96] = 0.f;
    }
}

__global__ void __launch_bounds__(256) sample_points_kernel(esr_scene_t sc, const float *__restrict__ rays_o,
                                                            const float *__restrict__ rays_d,
                                                            const int32_t *__restrict__ rec_ray,
                                                            const int32_t *__restrict__ rec_step, int n,
                                                            float *__restrict__ pts)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int ray = rec_ray[i];
        float p[3] = {0.f, 0.f, 0.f};
        if (ray >= 0) {
            const RayGeom g = esr_ray_geom(rays_o, rays_d, ray, sc.xyz_min, sc.xyz_max, sc.near_, 1e9f, sc.stepdist);
            esr_ray_point(g.start, g.dir, sc.stepdist, rec_step[i], p);
        }
        pts[3 * i] = p[0]; pts[3 * i + 1] = p[1]; pts[3 * i + 2] = p[2];
    }
}

}  // namespace

ESR_API int esr_fine_tone_in_fwd(const float *z_off, const float *z_emo, int32_t tiles_on,
                                 int32_t tiles_all, float *lin, float *Xt, void *stream)
{
    if (tiles_all < 0 || tiles_on < 0 || tiles_on > tiles_all) return ESR_EINVAL;
    if (tiles_all == 0) return 0;
    if (!z_off || (tiles_on && !z_emo) || !lin || !Xt) return ESR_EINVAL;
    tone_in_fwd_kernel<<<esr_grid_for((int64_t)tiles_all * 32, 256), 256, 0, esr_stream(stream)>>>(
        z_off, z_emo, tiles_on, tiles_all, lin, Xt);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_fine_composite_fwd(const float *zt, const float *lin, const int32_t *rec_ray,
                                   const float *rec_w, int32_t tiles_all, float *rgb, float *srgb_marched,
                                   float *lin_marched, void *stream)
{
    if (tiles_all < 0) return ESR_EINVAL;
    if (tiles_all == 0) return 0;
    if (!zt || !lin || !rec_ray || !rec_w || !rgb || !srgb_marched || !lin_marched) return ESR_EINVAL;
    composite_fwd_kernel<<<esr_grid_for(((int64_t)tiles_all * 32 + 63) / 64 * 64, 256), 256, 0,
                           esr_stream(stream)>>>(zt, lin, rec_ray, rec_w, tiles_all, rgb, srgb_marched,
                                                 lin_marched);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_fine_composite_bwd(const float *g_srgb, const float *g_lin, const float *rgb,
                                   const float *lin, const int32_t *rec_ray, const float *rec_w,
                                   int32_t tiles_all, float *dweight, float *dzt, void *stream)
{
    if (tiles_all < 0) return ESR_EINVAL;
    if (tiles_all == 0) return 0;
    if (!g_srgb || !g_lin || !rgb || !lin || !rec_ray || !rec_w || !dweight || !dzt) return ESR_EINVAL;
    composite_bwd_kernel<<<esr_grid_for((int64_t)tiles_all * 32, 256), 256, 0, esr_stream(stream)>>>(
        g_srgb, g_lin, rgb, lin, rec_ray, rec_w, tiles_all, dweight, dzt);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_fine_tone_in_bwd(const float *dXt, const float *Xt, const float *g_lin, const float *lin,
                                 const float *z_off, const float *z_emo, const int32_t *rec_ray,
                                 const float *rec_w, int32_t tiles_on, int32_t tiles_all, float *dz,
                                 void *stream)
{
    if (tiles_all < 0 || tiles_on < 0 || tiles_on > tiles_all) return ESR_EINVAL;
    if (tiles_all == 0) return 0;
    if (!dXt || !Xt || !g_lin || !lin || !z_off || (tiles_on && !z_emo) || !rec_ray || !rec_w || !dz) return ESR_EINVAL;
    tone_in_bwd_kernel<<<esr_grid_for((int64_t)tiles_all * 32, 256), 256, 0, esr_stream(stream)>>>(
        dXt, Xt, g_lin, lin, z_off, z_emo, rec_ray, rec_w, tiles_on, tiles_all, dz);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_fine_loss_fwd_bwd_dp(const float *srgb_marched, const float *lin_marched,
                                     const float *alphainv_last, const float *rgbs, int32_t n_rays,
                                     float white_bg, float weight_linear, float weight_entropy_last, float scale,
                                     float *loss, float *g_srgb, float *g_lin, float *g_last, void *stream)
{
    if (n_rays < 0) return ESR_EINVAL;
    if (n_rays == 0) return 0;
    if (!srgb_marched || !lin_marched || !alphainv_last || !rgbs || !loss || !g_srgb || !g_lin || !g_last)
        return ESR_EINVAL;
    loss_kernel<<<esr_grid_for(n_rays, 256), 256, 0, esr_stream(stream)>>>(
        srgb_marched, lin_marched, alphainv_last, rgbs, n_rays, white_bg, weight_linear,
        weight_entropy_last, scale, loss, g_srgb, g_lin, g_last);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_fine_loss_fwd_bwd(const float *srgb_marched, const float *lin_marched,
                                  const float *alphainv_last, const float *rgbs, int32_t n_rays,
                                  float white_bg, float weight_linear, float weight_entropy_last,
                                  float *loss, float *g_srgb, float *g_lin, float *g_last, void *stream)
{
    return esr_fine_loss_fwd_bwd_dp(srgb_marched, lin_marched, alphainv_last, rgbs, n_rays, white_bg, weight_linear,
                                    weight_entropy_last, 1.f, loss, g_srgb, g_lin, g_last, stream);
}

ESR_API int esr_eval_aux(const float *X, int32_t xrows, int32_t row_nx, int32_t row_ny, int32_t row_nz,
                         const int32_t *rec_ray, const int32_t *rec_step, int32_t tiles,
                         const float *pos_rt_host, float stepdist, float *aux, void *stream)
{
    if (tiles < 0 || xrows < 1 || row_nx < 0 || row_ny < 0 || row_nz < 0 || row_nx >= xrows || row_ny >= xrows ||
        row_nz >= xrows)
        return ESR_EINVAL;
    if (tiles == 0) return 0;
    if (!X || !rec_ray || !rec_step || !pos_rt_host || !aux) return ESR_EINVAL;
    EvalAux A = {};
    A.X = X; A.rec_ray = rec_ray; A.rec_step = rec_step; A.tiles = tiles; A.xrows = xrows; A.stepdist = stepdist;
    A.aux = aux; A.nrow[0] = row_nx; A.nrow[1] = row_ny; A.nrow[2] = row_nz;
    for (int i = 0; i < 9; ++i) A.rt[i] = pos_rt_host[i];
    eval_aux_kernel<<<esr_grid_for((int64_t)tiles * 32, 256), 256, 0, esr_stream(stream)>>>(A);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_eval_disp(const float *depth3, const float *alphainv_last, float far_, int32_t n_rays,
                          float *depth, float *disp, void *stream)
{
    if (n_rays < 0) return ESR_EINVAL;
    if (n_rays == 0) return 0;
    if (!depth3 || !alphainv_last || !depth || !disp) return ESR_EINVAL;
    eval_disp_kernel<<<esr_grid_for(n_rays, 256), 256, 0, esr_stream(stream)>>>(depth3, alphainv_last, far_, n_rays,
                                                                                depth, disp);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_pair_loss_fwd_bwd(const float *a, const float *b, int64_t rows, int32_t cols,
                                  const uint8_t *row_mask, int mask_value, const int32_t *count_dev, int kind,
                                  float w_value, float w_a, float w_b, float *loss, float *ga, float *gb,
                                  void *stream)
{
    if (rows < 0 || cols < 1 || (kind != 0 && kind != 1)) return ESR_EINVAL;
    if (rows == 0) return 0;
    if (!a || !loss) return ESR_EINVAL;
    pair_loss_kernel<<<esr_grid_for(rows * cols, 256, 256), 256, 0, esr_stream(stream)>>>(
        a, b, rows, cols, row_mask, mask_value, count_dev, kind, w_value, w_a, w_b, loss, ga, gb);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_act_batch(const esr_act_job_t *jobs, int32_t n_jobs, void *stream)
{
    if (n_jobs < 0 || n_jobs > ESR_ACT_MAX_JOBS || (n_jobs > 0 && !jobs)) return ESR_EINVAL;
    ActBatch B = {};
    int64_t most = 0;
    for (int i = 0; i < n_jobs; ++i) {
        const esr_act_job_t &J = jobs[i];
        if (J.tiles < 0 || J.rows < 1 || J.n_ch < 0 || J.n_ch > J.rows || (J.act != 0 && J.act != 1)) return ESR_EINVAL;
        if (J.tiles == 0) continue;
        if (!J.z || !J.out) return ESR_EINVAL;
        if (J.bwd) {
            if (!J.g_tile && !J.src && !J.pt1) return ESR_EINVAL;
            if (J.src && (J.src_c < 1 || J.n_src < 0)) return ESR_EINVAL;
            for (int e = 0; e < 3; ++e)
                if (J.pt1 && J.ex[e] && (J.ex_c[e] < 1 || J.ex_col0[e] < 0)) return ESR_EINVAL;
        }
        B.job[B.n++] = J;
        const int64_t tot = (int64_t)J.tiles * J.rows * 32;
        most = tot > most ? tot : most;
    }
    if (B.n == 0) return 0;
    act_batch_kernel<<<dim3(esr_grid_for(most, 256, 1024), B.n), 256, 0, esr_stream(stream)>>>(B);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_pair_loss_batch(const esr_pair_job_t *jobs, int32_t n_jobs, float *loss, void *stream)
{
    if (n_jobs < 0 || n_jobs > ESR_PAIR_MAX_JOBS || (n_jobs > 0 && (!jobs || !loss))) return ESR_EINVAL;
    PairBatch B = {};
    B.loss = loss;
    int64_t most = 0;
    for (int i = 0; i < n_jobs; ++i) {
        const esr_pair_job_t &J = jobs[i];
        if (J.rows < 0 || J.cols < 1 || (J.kind != 0 && J.kind != 1)) return ESR_EINVAL;
        if (J.rows == 0) continue;
        if (!J.a) return ESR_EINVAL;
        B.job[B.n++] = J;
        most = J.rows * J.cols > most ? J.rows * J.cols : most;
    }
    if (B.n == 0) return 0;
    pair_loss_batch_kernel<<<dim3(esr_grid_for(most, 256, 128), B.n), 256, 0, esr_stream(stream)>>>(B);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_act_fwd(const float *z, int32_t tiles, int32_t rows, int32_t n_ch, int act, float *out,
                        void *stream)
{
    if (tiles < 0 || rows < 1 || n_ch < 0 || n_ch > rows || (act != 0 && act != 1)) return ESR_EINVAL;
    if (tiles == 0) return 0;
    if (!z || !out) return ESR_EINVAL;
    act_kernel<false><<<esr_grid_for((int64_t)tiles * rows * 32, 256), 256, 0, esr_stream(stream)>>>(
        z, nullptr, tiles, rows, n_ch, act, out);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_act_bwd(const float *z, const float *g, int32_t tiles, int32_t rows, int32_t n_ch, int act,
                        float *dz, void *stream)
{
    if (tiles < 0 || rows < 1 || n_ch < 0 || n_ch > rows || (act != 0 && act != 1)) return ESR_EINVAL;
    if (tiles == 0) return 0;
    if (!z || !g || !dz) return ESR_EINVAL;
    act_kernel<true><<<esr_grid_for((int64_t)tiles * rows * 32, 256), 256, 0, esr_stream(stream)>>>(
        z, g, tiles, rows, n_ch, act, dz);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_composite3_fwd(const float *v, int32_t rows, const int32_t *rec_ray, const float *rec_w,
                               int32_t tiles, float *out, void *stream)
{
    if (tiles < 0 || rows < 3) return ESR_EINVAL;
    if (tiles == 0) return 0;
    if (!v || !rec_ray || !rec_w || !out) return ESR_EINVAL;
    composite3_fwd_kernel<<<esr_grid_for(((int64_t)tiles * 32 + 63) / 64 * 64, 256), 256, 0, esr_stream(stream)>>>(
        v, rows, rec_ray, rec_w, tiles, out);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_composite3_bwd(const float *g, const float *v, int32_t rows, const int32_t *rec_ray,
                               const float *rec_w, int32_t tiles, int accumulate, float *dv, float *dweight,
                               void *stream)
{
    if (tiles < 0 || rows < 3) return ESR_EINVAL;
    if (tiles == 0) return 0;
    if (!g || !v || !rec_ray || !rec_w || !dv || !dweight) return ESR_EINVAL;
    composite3_bwd_kernel<<<esr_grid_for((int64_t)tiles * 32, 256), 256, 0, esr_stream(stream)>>>(
        g, v, rows, rec_ray, rec_w, tiles, accumulate, dv, dweight);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_lts_tone_in_bwd(const float *dXt, const float *Xt, const float *g_lin, const float *lin, const float *z_off,
                                const float *z_emo, const int32_t *rec_ray, const float *rec_w,
                                int32_t tiles_on, int32_t tiles_all, float *dz_off, float *dz_emo, void *stream)
{
    if (tiles_all < 0 || tiles_on < 0 || tiles_on > tiles_all) return ESR_EINVAL;
    if (tiles_all == 0) return 0;
    if (!dXt || !Xt || !g_lin || !lin || !z_off || (tiles_on && (!z_emo || !dz_emo)) || !rec_ray || !rec_w || !dz_off)
        return ESR_EINVAL;
    lts_tone_in_bwd_kernel<<<esr_grid_for((int64_t)tiles_all * 32, 256), 256, 0, esr_stream(stream)>>>(
        dXt, Xt, g_lin, lin, z_off, z_emo, rec_ray, rec_w, tiles_on, tiles_all, dz_off, dz_emo);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_sample_points(const esr_scene_t *scene, const float *rays_o, const float *rays_d,
                              const int32_t *rec_ray, const int32_t *rec_step, int32_t n, float *pts,
                              void *stream)
{
    if (!scene || n < 0) return ESR_EINVAL;
    if (n == 0) return 0;
    if (!rays_o || !rays_d || !rec_ray || !rec_step || !pts) return ESR_EINVAL;
    sample_points_kernel<<<esr_grid_for(n, 256), 256, 0, esr_stream(stream)>>>(*scene, rays_o, rays_d, rec_ray,
                                                                               rec_step, n, pts);
    ESR_CHECK_LAUNCH();
    return 0;
}
