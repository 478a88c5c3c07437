// Fused ray march of the fine stage: sampler -> in-box test -> mask cache ->
// SDF tap -> NeuS "interp" alpha -> alpha>thres -> front-to-back transmittance
// with the early stop -> weight>thres, and its backward.
//
// Reference algorithm (paths under the reference tree):
//   app/fine/model/voxurff.py:186-213 (forward_training, up to "app mask 1")
//   app/fine/model/voxurff.py:623-654 + render_utils_kernel.cu:12-79,167-194 (sample_ray)
//   app/utils/base/module.py:104-114 (MaskCache.forward)
//   app/utils/base/functions.py:72-105 (neus_alpha_from_sdf_scatter_interp)
//   render_utils_kernel.cu:577-605,654-677 (alpha2weight fwd/bwd)
//
// MI355X design: ONE 64-lane wavefront per ray.  The reference materialises M0,
// M1, M2 sized tensors in HBM between ~25 launches, compacts with boolean masks
// (host syncs) and composites with one thread per ray.  Here a ray's samples live
// in lanes/LDS: compaction is ballot + mbcnt, the NeuS neighbour exchange is an
// LDS read, the transmittance recurrence is a v_readlane loop in the reference's
// serial order (so the early stop is bit-identical), and nothing but the final
// survivors (16 B each) touches HBM.  Three instantiations: COUNT (sizes), FILL
// (records) and BWD (recompute + reverse scan + trilinear atomic scatter).
#include "esr_common.h"

namespace {

enum { MARCH_COUNT = 0, MARCH_FILL = 1, MARCH_BWD = 2 };

struct MarchParams {
    esr_scene_t sc;
    const float *rays_o, *rays_d, *mask_density, *sdf;
    const float *viewdirs;   // GA only: the batch's view directions (the reference indexes viewdirs[ray_id])
    int n_rays;
    int cap;                 // per-wave LDS capacity in samples (multiple of 64)
    // COUNT
    int32_t *cnt3;
    float *alphainv_last;
    float *cumw;             // coarse: per-ray sum of the final weights (white_bg = 1 - cumw)
    int32_t *stats;          // [n_rays,3] in-box / mask-cache / alpha survivors per ray (summed by plan_kernel)
    esr_plan_t *plan;
    // FILL / BWD
    const int32_t *off3;
    int32_t *rec_ray, *rec_step;
    float *rec_w, *rec_sdf;
    // BWD
    const float *dweight, *dlast;
    float *grad_sdf;
    float *dsdf_rec;         // optional [n_tiles*32]: the value-tap gradient of a RECORDED sample is added here (the feature
                             // backward scatters it with its own SDF taps) instead of 8 atomics into grad_sdf
    int dsdf_acc;            // 0: every recorded slot is overwritten; 1: added to what the caller put there
    // march cache [n_rays][5][cap] (sdf | step | alpha | T | info of every mask-cache survivor, compacted per ray): written
    // by the COUNT pass, it lets FILL be a plain copy and BWD start at its reverse scan -- the walk (two trilinear
    // fetches per step) and the serial transmittance loop run once per step instead of three times
    float *cache;
    const float *last_in;    // BWD from the cache: alphainv_last of the COUNT pass
    // COARSE + GA: the dense central-difference gradient grid [X,Y,Z,3] the coarse renderer samples (voxurfc.py:204-206)
    // and, in the backward, its gradient buffer (summed into the SDF grid by esr_central_grad_bwd)
    const float *gg;
    float *grad_gg;
};
constexpr int CACHE_ARR = 5;

__device__ __forceinline__ float neus_alpha(float pc, float nc)
{
    float r = (fmaxf(pc - nc, 0.f) + 1e-5f) / (pc + 1e-5f);
    return fminf(fmaxf(r, 0.f), 1.f);
}

// 0.5 * dist * (viewdir . grad) with grad = (f(+1 voxel) - f(-1 voxel)) / clamped index distance / voxel_size per grid
// axis (= world axis: grid axis 0 is world x); operation order of functions.py:53-55 and voxurff.py:706-711
__device__ __forceinline__ float grad_alpha_ic(const float *__restrict__ sdf, const int dims[3], const float ind[3],
                                               const float vd[3], float voxel_size, float dist)
{
#pragma clang fp contract(off)
    float gv[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float ixp[3], ixm[3];
        const float ap = tap_index(ind, dims, a, 1.0f, ixp), am = tap_index(ind, dims, a, -1.0f, ixm);
        const float fd = esr_tri_fetch1(sdf, dims, ixp) - esr_tri_fetch1(sdf, dims, ixm);
        gv[a] = __fdiv_rn(__fdiv_rn(fd, ap - am), voxel_size);
    }
    const float dotp = (vd[0] * gv[0] + vd[1] * gv[1]) + vd[2] * gv[2];
    return (dotp * dist) * 0.5f;
}

__device__ __forceinline__ void grad_alpha_ic_bwd(float *__restrict__ grad_sdf, const int dims[3], const float ind[3],
                                                  const float vd[3], float voxel_size, float dist, float dic)
{
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float ixp[3], ixm[3];
        const float ap = tap_index(ind, dims, a, 1.0f, ixp), am = tap_index(ind, dims, a, -1.0f, ixm);
        const float coef = dic * 0.5f * dist * vd[a] / (ap - am) / voxel_size;
        if (coef != 0.f) {
            esr_tri_scatter1(grad_sdf, dims, ixp, coef);
            esr_tri_scatter1(grad_sdf, dims, ixm, -coef);
        }
    }
}

// Coarse renderer (voxurfc.py:204-210): the gradient is a trilinear sample of the dense central-difference grid
// (3 channels, stored channel-last), iter_cos = (viewdir . gradient) * dist * 0.5 (functions.py:53-55).
__device__ __forceinline__ float coarse_alpha_ic(const float *__restrict__ gg, const int dims[3], const float ind[3],
                                                 const float vd[3], float dist)
{
#pragma clang fp contract(off)
    const Tri t = esr_tri_setup(ind);
    float gv[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int cx = 0; cx < 2; ++cx)
#pragma unroll
        for (int cy = 0; cy < 2; ++cy)
#pragma unroll
            for (int cz = 0; cz < 2; ++cz) {
                const int x = t.i0[0] + cx, y = t.i0[1] + cy, z = t.i0[2] + cz;
                const bool inb = (x >= 0) & (x < dims[0]) & (y >= 0) & (y < dims[1]) & (z >= 0) & (z < dims[2]);
                const float w = esr_corner_w(t, ind, cx, cy, cz);
                const float *q = gg + (inb ? (((int64_t)x * dims[1] + y) * dims[2] + z) * 3 : 0);   // esr_ld_or0's reasoning
                const float q0 = inb ? q[0] : 0.f, q1 = inb ? q[1] : 0.f, q2 = inb ? q[2] : 0.f;
                gv[0] += q0 * w; gv[1] += q1 * w; gv[2] += q2 * w;
            }
    const float dotp = (vd[0] * gv[0] + vd[1] * gv[1]) + vd[2] * gv[2];
    return (dotp * dist) * 0.5f;
}

__device__ __forceinline__ void coarse_alpha_ic_bwd(float *__restrict__ grad_gg, const int dims[3], const float ind[3],
                                                    const float vd[3], float dist, float dic)
{
    const Tri t = esr_tri_setup(ind);
    const float k = dic * 0.5f * dist;
#pragma unroll
    for (int cx = 0; cx < 2; ++cx)
#pragma unroll
        for (int cy = 0; cy < 2; ++cy)
#pragma unroll
            for (int cz = 0; cz < 2; ++cz) {
                const int x = t.i0[0] + cx, y = t.i0[1] + cy, z = t.i0[2] + cz;
                const bool inb = (x >= 0) & (x < dims[0]) & (y >= 0) & (y < dims[1]) & (z >= 0) & (z < dims[2]);
                const float w = esr_corner_w(t, ind, cx, cy, cz);
                if (inb && w != 0.f) {
                    float *q = grad_gg + (((int64_t)x * dims[1] + y) * dims[2] + z) * 3;
#pragma unroll
                    for (int a = 0; a < 3; ++a)
                        if (vd[a] != 0.f) atomicAdd(q + a, k * vd[a] * w);
                }
            }
}

// The reference's alpha2weight walks a ray's samples in order: T_i = T_cum;  T_cum = float(double(T_cum) * (1.0 - alpha_i));
// stop once T_cum < 1e-3 (render_utils_kernel.cu; the early stop and every rounding step decide which samples survive,
// so the order and the mixed precision are kept).  One wave per ray: the samples of `todo` (one bit per lane) are visited
// in lane order; every wave instruction of the loop serves ONE sample, so the loop is kept to the dependent chain:
// 1 - alpha is formed for all lanes beforehand (in double) and read with two v_readlane, and the comparison is done in
// float -- (double)t < 1e-3 <=> t < 1e-3f for a float t, because the float nearest to 1e-3 lies above it.
// Returns whether this lane's sample was visited; `mine` = its T_i.
__device__ __forceinline__ bool serial_transmittance(float alpha, unsigned long long todo, int lane, float &t_cum,
                                                     bool &stopped, float &mine)
{
    static_assert((double)1e-3f >= 1e-3, "float threshold must not lie below the double one");
    const double om = 1.0 - (double)alpha;
    const unsigned om_lo = (unsigned)__double_as_longlong(om), om_hi = (unsigned)(__double_as_longlong(om) >> 32);
    unsigned long long rem = todo;
    while (rem) {
        const int i = __ffsll((long long)rem) - 1;
        rem &= rem - 1;
        const double omi = __longlong_as_double(((long long)__builtin_amdgcn_readlane(om_hi, i) << 32) |
                                                (unsigned)__builtin_amdgcn_readlane(om_lo, i));
        mine = (lane == i) ? t_cum : mine;
        t_cum = (float)((double)t_cum * omi);
        if (t_cum < 1e-3f) { stopped = true; break; }
    }
    return ((todo & ~rem) >> lane) & 1ull;
}

// COARSE (VoxurfC.forward_training, voxurfc.py:207-219): no alpha threshold, and alpha2weight runs a SECOND
// time over the weight > thres survivors of the first pass; weights and alphainv_last come from that pass.
// GA = cfg neus_alpha: "grad" (app/utils/base/functions.py:45-69): the two section SDFs of a sample are extrapolated
// from ITS OWN value and finite-difference gradient, sdf -+ 0.5 * dist * (viewdir . grad), instead of interpolated
// towards its surviving neighbours; grad = the radius-1 clamped central differences of sample_sdf_grad
// (voxurff.py:670-721), so a sample reads 7 trilinear taps and its backward scatters through all 7.
template <int MODE, bool COARSE, bool GA = false>
__global__ void __launch_bounds__(256) march_kernel(MarchParams P)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = esr_lane();
    const int wave_in_blk = threadIdx.x >> 6;
    const int waves_per_blk = blockDim.x >> 6;
    const int r = blockIdx.x * waves_per_blk + wave_in_blk;
    if (r >= P.n_rays) return;                        // whole wave leaves together

    constexpr int NARR = ((MODE == MARCH_BWD) ? 5 : 2) + (GA ? 1 : 0);
    float *base = reinterpret_cast<float *>(smem_raw) + (size_t)wave_in_blk * NARR * P.cap;
    float *ic1 = GA ? base + (NARR - 1) * P.cap : nullptr;   // 0.5 * dist * (viewdir . grad) of mask-cache survivors
    float *sdf1 = base;                                // SDF of mask-cache survivors
    int *step1 = reinterpret_cast<int *>(base + P.cap);
    float *alpha1 = (MODE == MARCH_BWD) ? base + 2 * P.cap : nullptr;   // later: d/d prev-midpoint
    float *T1 = (MODE == MARCH_BWD) ? base + 3 * P.cap : nullptr;       // later: d/d next-midpoint
    int *info1 = (MODE == MARCH_BWD) ? reinterpret_cast<int *>(base + 4 * P.cap) : nullptr;

    const esr_scene_t &sc = P.sc;
    const int gdims[3] = {sc.gx, sc.gy, sc.gz};
    const int mdims[3] = {sc.mx, sc.my, sc.mz};
    const RayGeom g = esr_ray_geom(P.rays_o, P.rays_d, r, sc.xyz_min, sc.xyz_max, sc.near_, 1e9f,
                                   sc.stepdist);
    if (g.n_steps > P.cap) {
        if (MODE == MARCH_COUNT && lane == 0) {
            atomicOr(&P.plan->overflow, 1);
            P.cnt3[r] = 0;
            P.alphainv_last[r] = 1.f;
            P.stats[3 * r] = 0; P.stats[3 * r + 1] = 0; P.stats[3 * r + 2] = 0;
        }
        return;
    }

    float vd[3] = {0.f, 0.f, 0.f};
    if (GA) { vd[0] = P.viewdirs[3 * r]; vd[1] = P.viewdirs[3 * r + 1]; vd[2] = P.viewdirs[3 * r + 2]; }
    float *const csdf = P.cache ? P.cache + (size_t)r * CACHE_ARR * P.cap : nullptr;
    int *const cstep = reinterpret_cast<int *>(csdf + P.cap);
    float *const calpha = csdf + 2 * P.cap, *const cT = csdf + 3 * P.cap;
    int *const cinfo = reinterpret_cast<int *>(csdf + 4 * P.cap);
    if (MODE == MARCH_FILL && P.cache) {              // records straight from the COUNT pass's cache
        const int n1c = P.stats[3 * r + 1], ob = P.off3[r];
        for (int j = lane; j < n1c; j += 64) {
            const int rec = cinfo[j] >> 8;
            if (rec) {
                const int o = ob + rec - 1;
                P.rec_ray[o] = r;
                P.rec_step[o] = cstep[j];
                P.rec_w[o] = cT[j] * calpha[j];
                P.rec_sdf[o] = csdf[j];
            }
        }
        return;
    }
    // ---- phase 1: walk the ray, keep in-box & mask-cache survivors in LDS ----
    int n0 = 0, n1 = 0;
    float tc = 1.f, tc2 = 1.f, wsum = 0.f;
    int n2 = 0, n3 = 0;
    const int out_base = (MODE != MARCH_COUNT) ? P.off3[r] : 0;
    if (MODE == MARCH_BWD && P.cache) {               // LDS arrays of the backward from the cache
        n1 = P.stats[3 * r + 1];
        for (int j = lane; j < n1; j += 64) {
            sdf1[j] = csdf[j]; step1[j] = cstep[j]; alpha1[j] = calpha[j]; T1[j] = cT[j]; info1[j] = cinfo[j];
        }
        tc = tc2 = P.last_in[r];
    } else {
    for (int c0 = 0; c0 < g.n_steps; c0 += 64) {
        const int step = c0 + lane;
        bool ok = step < g.n_steps;
        float p[3], idx[3], s = 0.f;
        esr_ray_point(g.start, g.dir, sc.stepdist, step, p);
        ok = ok && !esr_out_of_box(p, sc.xyz_min, sc.xyz_max);
        n0 += __popcll(__ballot(ok));
        if (ok) {
            esr_world_to_index(p, sc.mask_min, sc.mask_max, mdims, idx);
            const float dens = esr_tri_fetch1(P.mask_density, mdims, idx);
            const float a = 1.f - expf(-esr_softplus(dens + sc.act_shift));
            ok = a >= sc.mask_thres;
        }
        float ic = 0.f;
        if (ok) {
            esr_world_to_index(p, sc.xyz_min, sc.xyz_max, gdims, idx);
            s = esr_tri_fetch1(P.sdf, gdims, idx);
            if (GA) ic = COARSE ? coarse_alpha_ic(P.gg, gdims, idx, vd, sc.stepdist)
                                : grad_alpha_ic(P.sdf, gdims, idx, vd, sc.voxel_size, sc.stepdist);
        }
        const unsigned long long b = __ballot(ok);
        if (ok) {
            const int pos = n1 + __popcll(b & ((1ull << lane) - 1ull));
            sdf1[pos] = s;
            step1[pos] = step;
            if (GA) ic1[pos] = ic;
            if (MODE == MARCH_COUNT && P.cache) { csdf[pos] = s; cstep[pos] = step; }
        }
        n1 += __popcll(b);
    }
    __threadfence_block();

    // ---- phase 2: alpha, thresholds, transmittance in serial order ----------
    bool stopped = false, stopped2 = false;
    for (int c0 = 0; c0 < n1; c0 += 64) {
        const int j = c0 + lane;
        const bool ok = j < n1;
        float s = 0.f, alpha = 0.f;
        if (ok) {
            s = sdf1[j];
            const float prv = GA ? s - ic1[j] : (j > 0) ? (sdf1[j - 1] + s) * 0.5f : s;
            const float nxt = GA ? s + ic1[j] : (j < n1 - 1) ? (s + sdf1[j + 1]) * 0.5f : s;
            alpha = neus_alpha(esr_sigmoid(prv * sc.s_val), esr_sigmoid(nxt * sc.s_val));
        }
        const bool v2 = ok && (COARSE || alpha > sc.fast_thres);
        const unsigned long long b2 = __ballot(v2);
        n2 += __popcll(b2);
        float myT = 1.f;
        bool proc = false;
        if (!stopped) proc = serial_transmittance(alpha, b2, lane, tc, stopped, myT);
        float w = proc ? myT * alpha : 0.f;
        const bool v3 = proc && w > sc.fast_thres;
        const unsigned long long b3 = __ballot(v3);
        const int rank = n3 + __popcll(b3 & ((1ull << lane) - 1ull));
        if (COARSE) {                     // second transmittance pass over the survivors, same serial order
            myT = 1.f;
            proc = false;
            if (!stopped2) proc = serial_transmittance(alpha, b3, lane, tc2, stopped2, myT);
            w = proc ? myT * alpha : 0.f;
            wsum += v3 ? w : 0.f;
        }
        if (MODE == MARCH_FILL && v3) {
            const int o = out_base + rank;
            P.rec_ray[o] = r;
            P.rec_step[o] = step1[j];
            P.rec_w[o] = w;
            P.rec_sdf[o] = s;
        }
        if (MODE == MARCH_BWD && ok) {
            alpha1[j] = alpha;
            T1[j] = myT;
            info1[j] = (v2 ? 1 : 0) | (proc ? 2 : 0) | (v3 ? ((rank + 1) << 8) : 0);
        }
        if (MODE == MARCH_COUNT && P.cache && ok) {
            calpha[j] = alpha;
            cT[j] = myT;
            cinfo[j] = (v2 ? 1 : 0) | (proc ? 2 : 0) | (v3 ? ((rank + 1) << 8) : 0);
        }
        n3 += __popcll(b3);
    }
    }   // (walk + serial pass; skipped by the backward when the COUNT pass's cache is given)

    if (MODE == MARCH_COUNT) {
        if (COARSE && P.cumw) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) wsum += __shfl_xor(wsum, off);
        }
        if (lane == 0) {
            P.cnt3[r] = n3;
            P.alphainv_last[r] = COARSE ? tc2 : tc;
            if (COARSE && P.cumw) P.cumw[r] = wsum;
            // per-ray statistics, summed by plan_kernel: 3 x n_rays atomics on three addresses of the plan header
            // serialised in the L2 and were 70 % of this kernel's time
            P.stats[3 * r] = n0; P.stats[3 * r + 1] = n1; P.stats[3 * r + 2] = n2;
        }
        return;
    }
    if (MODE != MARCH_BWD) return;

    // ---- backward: reverse scan over the processed samples -------------------
    __threadfence_block();
    float back = P.dlast[r] * (COARSE ? tc2 : tc);
    const int nchunk = (n1 + 63) >> 6;
    for (int c = nchunk - 1; c >= 0; --c) {
        const int j = c * 64 + lane;
        const bool ok = j < n1;
        const int info = ok ? info1[j] : 0;
        const bool proc = info & 2;
        const float alpha = ok ? alpha1[j] : 0.f;
        const float T = ok ? T1[j] : 1.f;
        const float gw = (info >> 8) ? P.dweight[out_base + (info >> 8) - 1] : 0.f;
        const float x = proc ? gw * (T * alpha) : 0.f;
        float v = x;                                  // inclusive suffix sum over lanes >= lane
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const float u = __shfl_down(v, off);
            if (lane + off < 64) v += u;
        }
        const float myback = back + (v - x);
        back += esr_readlane(v, 0);
        float dprev = 0.f, dnext = 0.f;
        if (proc) {
            const float dalpha =
                (float)((double)(gw * T) - (double)myback / ((double)(1.0f - alpha) + 1e-10));
            const float s = sdf1[j];
            const float prv = GA ? s - ic1[j] : (j > 0) ? (sdf1[j - 1] + s) * 0.5f : s;
            const float nxt = GA ? s + ic1[j] : (j < n1 - 1) ? (s + sdf1[j + 1]) * 0.5f : s;
            const float pc = esr_sigmoid(prv * sc.s_val), nc = esr_sigmoid(nxt * sc.s_val);
            const float num = fmaxf(pc - nc, 0.f) + 1e-5f, den = pc + 1e-5f;
            const float ratio = num / den;
            if (ratio >= 0.f && ratio <= 1.f) {       // clip passes the gradient on [0,1]
                const float pos = (pc - nc > 0.f) ? 1.f : 0.f;   // relu'
                const float dr_dpc = (pos * den - num) / (den * den);
                const float dr_dnc = -pos / den;
                dprev = dalpha * dr_dpc * sc.s_val * pc * (1.f - pc);
                dnext = dalpha * dr_dnc * sc.s_val * nc * (1.f - nc);
            }
        }
        if (ok) { alpha1[j] = dprev; T1[j] = dnext; }
    }
    __threadfence_block();
    for (int c0 = 0; c0 < n1; c0 += 64) {
        const int j = c0 + lane;
        if (j >= n1) continue;
        if (GA) {        // prev = sdf - ic, next = sdf + ic: d sdf = dprev + dnext, d ic = dnext - dprev
            const float ds_ = alpha1[j] + T1[j], dic = T1[j] - alpha1[j];
            if (ds_ != 0.f || dic != 0.f) {
                float p[3], idx[3];
                esr_ray_point(g.start, g.dir, sc.stepdist, step1[j], p);
                esr_world_to_index(p, sc.xyz_min, sc.xyz_max, gdims, idx);
                if (ds_ != 0.f) esr_tri_scatter1(P.grad_sdf, gdims, idx, ds_);
                if (dic != 0.f) {
                    if (COARSE) coarse_alpha_ic_bwd(P.grad_gg, gdims, idx, vd, sc.stepdist, dic);
                    else grad_alpha_ic_bwd(P.grad_sdf, gdims, idx, vd, sc.voxel_size, sc.stepdist, dic);
                }
            }
            continue;
        }
        float ds = alpha1[j] * ((j > 0) ? 0.5f : 1.f) + T1[j] * ((j < n1 - 1) ? 0.5f : 1.f);
        if (j + 1 < n1) ds += 0.5f * alpha1[j + 1];
        if (j > 0) ds += 0.5f * T1[j - 1];
        const int rec = info1[j] >> 8;
        if (P.dsdf_rec && rec) {                      // one record per slot: plain store / read-modify-write
            float *q = P.dsdf_rec + out_base + rec - 1;
            *q = P.dsdf_acc ? *q + ds : ds;
            continue;
        }
        if (ds != 0.f) {
            float p[3], idx[3];
            esr_ray_point(g.start, g.dir, sc.stepdist, step1[j], p);
            esr_world_to_index(p, sc.xyz_min, sc.xyz_max, gdims, idx);
            esr_tri_scatter1(P.grad_sdf, gdims, idx, ds);
        }
    }
}

// One workgroup: exclusive offsets, emissive-on rays first, off rays from the
// next multiple of 32.  Rays are walked in chunks of 1024 with COALESCED loads (the first version gave every thread
// a contiguous run of rays: lane-strided dependent loads, 0.43 ms per LTS step for its 8 k + 25.6 k rays); both
// classes are scanned at once (two counters per thread), each ray first gets the offset inside its class, and the
// off rays are shifted by the padded on-total in a second coalesced sweep.
__device__ __forceinline__ int2 block_scan2(int2 v, int2 *wave_tot, int tid)   // exclusive scan over 1024 threads
{
    int2 inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int ax = __shfl_up(inc.x, off), ay = __shfl_up(inc.y, off);
        if ((tid & 63) >= off) { inc.x += ax; inc.y += ay; }
    }
    if ((tid & 63) == 63) wave_tot[tid >> 6] = inc;
    __syncthreads();
    if (tid < 64) {
        int2 t = tid < 16 ? wave_tot[tid] : make_int2(0, 0);
        int2 ti = t;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
            const int ax = __shfl_up(ti.x, off), ay = __shfl_up(ti.y, off);
            if (tid >= off) { ti.x += ax; ti.y += ay; }
        }
        if (tid < 16) wave_tot[tid] = make_int2(ti.x - t.x, ti.y - t.y);       // exclusive wave bases
        if (tid == 15) wave_tot[16] = ti;                                       // chunk totals
    }
    __syncthreads();
    const int2 wb = wave_tot[tid >> 6];
    return make_int2(wb.x + inc.x - v.x, wb.y + inc.y - v.y);
}

// TOTALS = false: the offsets only (the statistics and the range flag came from plan_totals_kernel; n_on / n_off / tiles are
// written again, with the same values)
template <bool TOTALS>
__global__ void __launch_bounds__(1024) plan_kernel(const int32_t *__restrict__ cnt3,
                                                    const int64_t *__restrict__ em_modes,
                                                    const int32_t *__restrict__ stats, int n_rays,
                                                    int32_t *__restrict__ off3, esr_plan_t *plan,
                                                    const unsigned *__restrict__ range_flag)
{
    __shared__ int2 wave_tot[17];
    const int tid = threadIdx.x;
    // the split-fp16 forward kernels' sticky range flag (mlp_split.hip: esr_mlp_split_range_flag) rides to the host in bit 1
    // of the header's overflow word: no dispatch and no copy of its own
    if (TOTALS && tid == 0 && range_flag && *range_flag) atomicOr(&plan->overflow, 2);
    if constexpr (TOTALS) {   // survivor statistics m0, m1, m2 = sums of the per-ray counts (element j of stats is of class j % 3)
        int s[3] = {0, 0, 0};
        // 16-byte loads, unrolled: one workgroup has nobody to hide a load behind -- 75 dependent dword round trips for the
        // 25.6 k secondary rays were most of this kernel's 58 us, which sit right in front of the host's read of the plan
        const int n3 = 3 * n_rays;
        const int n4 = (reinterpret_cast<uintptr_t>(stats) & 15) ? 0 : n3 >> 2;   // (an unaligned array: the dword loop below)
        const int4 *st4 = reinterpret_cast<const int4 *>(stats);
#pragma unroll 4
        for (int q = tid; q < n4; q += 1024) {
            const int4 v = st4[q];
            const int c0 = (4 * q) % 3;                                        // class of v.x; the others follow cyclically
            const int vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = (c0 + e) % 3;
                s[0] += c == 0 ? vv[e] : 0; s[1] += c == 1 ? vv[e] : 0; s[2] += c == 2 ? vv[e] : 0;
            }
        }
        for (int j = 4 * n4 + tid; j < n3; j += 1024) {                       // tail (< 4 elements)
            const int v = stats[j], c = j % 3;
            s[0] += c == 0 ? v : 0; s[1] += c == 1 ? v : 0; s[2] += c == 2 ? v : 0;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            s[0] += __shfl_xor(s[0], off); s[1] += __shfl_xor(s[1], off); s[2] += __shfl_xor(s[2], off);
        }
        if ((tid & 63) == 0 && (s[0] | s[1] | s[2])) {
            atomicAdd(&plan->m0, s[0]); atomicAdd(&plan->m1, s[1]); atomicAdd(&plan->m2, s[2]);
        }
    }
    int2 run = make_int2(0, 0);                        // totals of the chunks before this one (on, off)
    // chunks of 4096 rays, FOUR CONSECUTIVE rays per thread (two 16-byte loads of the counts, four of the modes): a
    // quarter of the block scans and barriers of one-ray-per-thread chunks (48 -> ~20 us for the 25.6 k secondary rays)
    constexpr int RPT = 4;
    for (int c0 = 0; c0 < n_rays; c0 += 1024 * RPT) {
        const int i0 = c0 + tid * RPT;
        int cnt[RPT];
        bool on[RPT];
        int2 v = make_int2(0, 0);
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const bool live = i0 + k < n_rays;
            const int ic = live ? i0 + k : n_rays - 1;   // unconditional loads of a clamped index (esr_ld_or0's reasoning:
            const int cv = cnt3[ic];                      // `live ? a[i] : 0` is one exec-masked load + wait per element)
            const int64_t mv = em_modes[ic];
            cnt[k] = live ? cv : 0;
            on[k] = live && mv == 1;
            v.x += on[k] ? cnt[k] : 0;
            v.y += on[k] ? 0 : cnt[k];
        }
        int2 ex = block_scan2(v, wave_tot, tid);       // exclusive over the threads' four-ray sums
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            if (i0 + k < n_rays) off3[i0 + k] = on[k] ? run.x + ex.x : run.y + ex.y;
            ex.x += on[k] ? cnt[k] : 0;
            ex.y += on[k] ? 0 : cnt[k];
        }
        const int2 tot = wave_tot[16];
        run.x += tot.x; run.y += tot.y;
        __syncthreads();                               // wave_tot is rewritten by the next chunk
    }
    const int base = ((run.x + 31) / 32) * 32;
#pragma unroll 4
    for (int i = tid; i < n_rays; i += 1024) {
        const bool off = em_modes[i] != 1;
        const int o = off3[i];                       // (unconditional load: see esr_ld_or0)
        if (off) off3[i] = o + base;
    }
    if (tid == 0) {
        plan->n_on = run.x;
        plan->tiles_on = (run.x + 31) / 32;
        plan->n_off = run.y;
        plan->tiles_all = (run.x + 31) / 32 + (run.y + 31) / 32;
    }
}

// The numbers the HOST waits for (n_on, n_off, m0..m2, the overflow word) are plain sums: many workgroups, five atomics each,
// ~5 us where the one-workgroup scan above takes 17 (8 k rays) to 48 us (25.6 k secondary rays).  The caller reads the header
// back right behind this kernel and enqueues the scan (plan_kernel<false>: the offsets, which only the NEXT launches need)
// behind the copy: it runs while the host wakes up and enqueues.  tiles_on / tiles_all are the host's to derive.
__global__ void __launch_bounds__(1024) plan_totals_kernel(const int32_t *__restrict__ cnt3, const int64_t *__restrict__ em_modes,
                                                           const int32_t *__restrict__ stats, int n_rays, esr_plan_t *plan,
                                                           const unsigned *__restrict__ range_flag)
{
    __shared__ int part[16][5];
    const int tid = threadIdx.x;
    if (blockIdx.x == 0 && tid == 0 && range_flag && *range_flag) atomicOr(&plan->overflow, 2);
    int v[5] = {0, 0, 0, 0, 0};                         // on, off, m0, m1, m2
#pragma unroll 2
    for (int i = blockIdx.x * 1024 + tid; i < n_rays; i += gridDim.x * 1024) {
        const int c = cnt3[i];
        const bool on = em_modes[i] == 1;
        v[0] += on ? c : 0; v[1] += on ? 0 : c;
        v[2] += stats[3 * i]; v[3] += stats[3 * i + 1]; v[4] += stats[3 * i + 2];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int k = 0; k < 5; ++k) v[k] += __shfl_xor(v[k], off);
    if ((tid & 63) == 0)
#pragma unroll
        for (int k = 0; k < 5; ++k) part[tid >> 6][k] = v[k];
    __syncthreads();
    if (tid < 5) {
        int t = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += part[w][tid];
        int32_t *dst = tid == 0 ? &plan->n_on : tid == 1 ? &plan->n_off : tid == 2 ? &plan->m0 : tid == 3 ? &plan->m1 : &plan->m2;
        if (t) atomicAdd(dst, t);
    }
}

int march_cap(const esr_scene_t *sc) { return ((sc->max_steps + 63) / 64) * 64; }

template <int MODE, bool COARSE = false, bool GA = false>
int launch_march(MarchParams &P, hipStream_t s)
{
    if (P.n_rays == 0) return 0;
    constexpr int NARR = ((MODE == MARCH_BWD) ? 5 : 2) + (GA ? 1 : 0);
    const size_t per_wave = (size_t)NARR * P.cap * sizeof(float);
    int wpb = 4;
    while (wpb > 1 && per_wave * wpb > 64 * 1024) wpb >>= 1;
    if (per_wave * wpb > 160 * 1024) return ESR_ECAP;
    static std::atomic<uint64_t> optin{0};             // one wave per block beyond 64 KB (cap > ~3.2k steps in BWD mode)
    if (int rc = esr_lds_optin(reinterpret_cast<const void *>(&march_kernel<MODE, COARSE, GA>), per_wave * wpb, optin)) return rc;
    const int grid = (P.n_rays + wpb - 1) / wpb;
    march_kernel<MODE, COARSE, GA><<<grid, wpb * 64, per_wave * wpb, s>>>(P);
    ESR_CHECK_LAUNCH();
    return 0;
}

}  // namespace

ESR_API int esr_fine_plan_begin(esr_plan_t *plan, void *stream)
{
    if (!plan) return ESR_EINVAL;
    return (int)hipMemsetAsync(plan, 0, sizeof(esr_plan_t), esr_stream(stream));
}

ESR_API int esr_fine_march_count(const esr_scene_t *scene, const float *rays_o, const float *rays_d,
                                 const float *mask_density, const float *sdf, int32_t n_rays,
                                 int32_t *cnt3, float *alphainv_last, int32_t *ray_stats, esr_plan_t *plan,
                                 void *stream)
{
    if (!scene || n_rays < 0 || !plan) return ESR_EINVAL;
    if (n_rays && (!rays_o || !rays_d || !mask_density || !sdf || !cnt3 || !alphainv_last || !ray_stats))
        return ESR_EINVAL;
    MarchParams P = {};
    P.sc = *scene; P.rays_o = rays_o; P.rays_d = rays_d; P.mask_density = mask_density; P.sdf = sdf;
    P.n_rays = n_rays; P.cap = march_cap(scene); P.cnt3 = cnt3; P.alphainv_last = alphainv_last;
    P.stats = ray_stats; P.plan = plan;
    return launch_march<MARCH_COUNT>(P, esr_stream(stream));
}

ESR_API int esr_fine_plan(const int32_t *cnt3, const int64_t *em_modes, const int32_t *ray_stats,
                          int32_t n_rays, int32_t *off3, esr_plan_t *plan, void *stream)
{
    if (n_rays < 0 || !plan) return ESR_EINVAL;
    if (n_rays && (!cnt3 || !em_modes || !ray_stats || !off3)) return ESR_EINVAL;
    plan_kernel<true><<<1, 1024, 0, esr_stream(stream)>>>(cnt3, em_modes, ray_stats, n_rays, off3, plan, esr_split_range_flag_ptr());
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_fine_plan_totals(const int32_t *cnt3, const int64_t *em_modes, const int32_t *ray_stats, int32_t n_rays,
                                 esr_plan_t *plan, void *stream)
{
    if (n_rays < 0 || !plan) return ESR_EINVAL;
    if (n_rays && (!cnt3 || !em_modes || !ray_stats)) return ESR_EINVAL;
    const int grid = n_rays < 2048 ? 1 : (n_rays + 2047) / 2048 < 64 ? (n_rays + 2047) / 2048 : 64;
    plan_totals_kernel<<<grid, 1024, 0, esr_stream(stream)>>>(cnt3, em_modes, ray_stats, n_rays, plan, esr_split_range_flag_ptr());
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_fine_plan_offsets(const int32_t *cnt3, const int64_t *em_modes, int32_t n_rays, int32_t *off3, esr_plan_t *plan,
                                  void *stream)
{
    if (n_rays < 0 || !plan) return ESR_EINVAL;
    if (n_rays && (!cnt3 || !em_modes || !off3)) return ESR_EINVAL;
    plan_kernel<false><<<1, 1024, 0, esr_stream(stream)>>>(cnt3, em_modes, nullptr, n_rays, off3, plan, nullptr);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_fine_march_fill(const esr_scene_t *scene, const float *rays_o, const float *rays_d,
                                const float *mask_density, const float *sdf, int32_t n_rays,
                                const int32_t *off3, int32_t *rec_ray, int32_t *rec_step, float *rec_w,
                                float *rec_sdf, void *stream)
{
    if (!scene || n_rays < 0) return ESR_EINVAL;
    if (n_rays && (!rays_o || !rays_d || !mask_density || !sdf || !off3 || !rec_ray || !rec_step ||
                   !rec_w || !rec_sdf))
        return ESR_EINVAL;
    MarchParams P = {};
    P.sc = *scene; P.rays_o = rays_o; P.rays_d = rays_d; P.mask_density = mask_density; P.sdf = sdf;
    P.n_rays = n_rays; P.cap = march_cap(scene); P.off3 = off3; P.rec_ray = rec_ray;
    P.rec_step = rec_step; P.rec_w = rec_w; P.rec_sdf = rec_sdf;
    return launch_march<MARCH_FILL>(P, esr_stream(stream));
}

ESR_API int esr_fine_march_bwd(const esr_scene_t *scene, const float *rays_o, const float *rays_d,
                               const float *mask_density, const float *sdf, int32_t n_rays,
                               const int32_t *off3, const float *dweight, const float *dlast,
                               float *grad_sdf, void *stream)
{
    if (!scene || n_rays < 0) return ESR_EINVAL;
    if (n_rays && (!rays_o || !rays_d || !mask_density || !sdf || !off3 || !dweight || !dlast || !grad_sdf))
        return ESR_EINVAL;
    MarchParams P = {};
    P.sc = *scene; P.rays_o = rays_o; P.rays_d = rays_d; P.mask_density = mask_density; P.sdf = sdf;
    P.n_rays = n_rays; P.cap = march_cap(scene); P.off3 = off3; P.dweight = dweight; P.dlast = dlast;
    P.grad_sdf = grad_sdf;
    return launch_march<MARCH_BWD>(P, esr_stream(stream));
}

ESR_API int esr_fine_march_bwd_rec(const esr_scene_t *scene, const float *rays_o, const float *rays_d,
                                   const float *mask_density, const float *sdf, int32_t n_rays,
                                   const int32_t *off3, const float *dweight, const float *dlast,
                                   float *grad_sdf, float *dsdf_rec, int32_t accumulate, void *stream)
{
    if (!scene || n_rays < 0) return ESR_EINVAL;
    if (n_rays && (!rays_o || !rays_d || !mask_density || !sdf || !off3 || !dweight || !dlast || !grad_sdf || !dsdf_rec))
        return ESR_EINVAL;
    MarchParams P = {};
    P.sc = *scene; P.rays_o = rays_o; P.rays_d = rays_d; P.mask_density = mask_density; P.sdf = sdf;
    P.n_rays = n_rays; P.cap = march_cap(scene); P.off3 = off3; P.dweight = dweight; P.dlast = dlast;
    P.grad_sdf = grad_sdf; P.dsdf_rec = dsdf_rec; P.dsdf_acc = accumulate;
    return launch_march<MARCH_BWD>(P, esr_stream(stream));
}

// ---- the three passes sharing a cache (see MarchParams::cache) ----------------------------------------------------
ESR_API int64_t esr_fine_march_cache_floats(const esr_scene_t *scene, int32_t n_rays)
{
    if (!scene || n_rays < 0) return ESR_EINVAL;
    return (int64_t)n_rays * CACHE_ARR * march_cap(scene);
}

ESR_API int esr_fine_march_count_cached(const esr_scene_t *scene, const float *rays_o, const float *rays_d,
                                        const float *mask_density, const float *sdf, int32_t n_rays, int32_t *cnt3,
                                        float *alphainv_last, int32_t *ray_stats, esr_plan_t *plan, float *cache,
                                        void *stream)
{
    if (!scene || n_rays < 0 || !plan) return ESR_EINVAL;
    if (n_rays && (!rays_o || !rays_d || !mask_density || !sdf || !cnt3 || !alphainv_last || !ray_stats || !cache))
        return ESR_EINVAL;
    MarchParams P = {};
    P.sc = *scene; P.rays_o = rays_o; P.rays_d = rays_d; P.mask_density = mask_density; P.sdf = sdf;
    P.n_rays = n_rays; P.cap = march_cap(scene); P.cnt3 = cnt3; P.alphainv_last = alphainv_last;
    P.stats = ray_stats; P.plan = plan; P.cache = cache;
    return launch_march<MARCH_COUNT>(P, esr_stream(stream));
}

ESR_API int esr_fine_march_fill_cached(const esr_scene_t *scene, const float *rays_o, const float *rays_d, int32_t n_rays,
                                       const int32_t *off3, const int32_t *ray_stats, const float *cache,
                                       int32_t *rec_ray, int32_t *rec_step, float *rec_w, float *rec_sdf, void *stream)
{
    if (!scene || n_rays < 0) return ESR_EINVAL;
    if (n_rays && (!rays_o || !rays_d || !off3 || !ray_stats || !cache || !rec_ray || !rec_step || !rec_w || !rec_sdf))
        return ESR_EINVAL;
    MarchParams P = {};
    P.sc = *scene; P.rays_o = rays_o; P.rays_d = rays_d;
    P.n_rays = n_rays; P.cap = march_cap(scene); P.off3 = off3; P.stats = const_cast<int32_t *>(ray_stats);
    P.cache = const_cast<float *>(cache); P.rec_ray = rec_ray; P.rec_step = rec_step; P.rec_w = rec_w; P.rec_sdf = rec_sdf;
    return launch_march<MARCH_FILL>(P, esr_stream(stream));
}

ESR_API int esr_fine_march_bwd_cached(const esr_scene_t *scene, const float *rays_o, const float *rays_d, int32_t n_rays,
                                      const int32_t *off3, const int32_t *ray_stats, const float *alphainv_last,
                                      const float *cache, const float *dweight, const float *dlast, float *grad_sdf,
                                      float *dsdf_rec, int32_t accumulate, void *stream)
{
    if (!scene || n_rays < 0) return ESR_EINVAL;
    if (n_rays && (!rays_o || !rays_d || !off3 || !ray_stats || !alphainv_last || !cache || !dweight || !dlast || !grad_sdf))
        return ESR_EINVAL;
    MarchParams P = {};
    P.sc = *scene; P.rays_o = rays_o; P.rays_d = rays_d;
    P.n_rays = n_rays; P.cap = march_cap(scene); P.off3 = off3; P.stats = const_cast<int32_t *>(ray_stats);
    P.last_in = alphainv_last; P.cache = const_cast<float *>(cache); P.dweight = dweight; P.dlast = dlast;
    P.grad_sdf = grad_sdf; P.dsdf_rec = dsdf_rec; P.dsdf_acc = accumulate;
    return launch_march<MARCH_BWD>(P, esr_stream(stream));
}

// ---- cfg neus_alpha: "grad" (functions.py:45-69): the same three entry points with the batch's view directions ----
ESR_API int esr_fine_march_count_ga(const esr_scene_t *scene, const float *rays_o, const float *rays_d, const float *viewdirs,
                                    const float *mask_density, const float *sdf, int32_t n_rays, int32_t *cnt3,
                                    float *alphainv_last, int32_t *ray_stats, esr_plan_t *plan, void *stream)
{
    if (!scene || n_rays < 0 || !plan) return ESR_EINVAL;
    if (n_rays && (!rays_o || !rays_d || !viewdirs || !mask_density || !sdf || !cnt3 || !alphainv_last || !ray_stats))
        return ESR_EINVAL;
    MarchParams P = {};
    P.sc = *scene; P.rays_o = rays_o; P.rays_d = rays_d; P.viewdirs = viewdirs; P.mask_density = mask_density; P.sdf = sdf;
    P.n_rays = n_rays; P.cap = march_cap(scene); P.cnt3 = cnt3; P.alphainv_last = alphainv_last;
    P.stats = ray_stats; P.plan = plan;
    return launch_march<MARCH_COUNT, false, true>(P, esr_stream(stream));
}

ESR_API int esr_fine_march_fill_ga(const esr_scene_t *scene, const float *rays_o, const float *rays_d, const float *viewdirs,
                                   const float *mask_density, const float *sdf, int32_t n_rays, const int32_t *off3,
                                   int32_t *rec_ray, int32_t *rec_step, float *rec_w, float *rec_sdf, void *stream)
{
    if (!scene || n_rays < 0) return ESR_EINVAL;
    if (n_rays && (!rays_o || !rays_d || !viewdirs || !mask_density || !sdf || !off3 || !rec_ray || !rec_step ||
                   !rec_w || !rec_sdf))
        return ESR_EINVAL;
    MarchParams P = {};
    P.sc = *scene; P.rays_o = rays_o; P.rays_d = rays_d; P.viewdirs = viewdirs; P.mask_density = mask_density; P.sdf = sdf;
    P.n_rays = n_rays; P.cap = march_cap(scene); P.off3 = off3; P.rec_ray = rec_ray;
    P.rec_step = rec_step; P.rec_w = rec_w; P.rec_sdf = rec_sdf;
    return launch_march<MARCH_FILL, false, true>(P, esr_stream(stream));
}

ESR_API int esr_fine_march_bwd_ga(const esr_scene_t *scene, const float *rays_o, const float *rays_d, const float *viewdirs,
                                  const float *mask_density, const float *sdf, int32_t n_rays, const int32_t *off3,
                                  const float *dweight, const float *dlast, float *grad_sdf, void *stream)
{
    if (!scene || n_rays < 0) return ESR_EINVAL;
    if (n_rays && (!rays_o || !rays_d || !viewdirs || !mask_density || !sdf || !off3 || !dweight || !dlast || !grad_sdf))
        return ESR_EINVAL;
    MarchParams P = {};
    P.sc = *scene; P.rays_o = rays_o; P.rays_d = rays_d; P.viewdirs = viewdirs; P.mask_density = mask_density; P.sdf = sdf;
    P.n_rays = n_rays; P.cap = march_cap(scene); P.off3 = off3; P.dweight = dweight; P.dlast = dlast;
    P.grad_sdf = grad_sdf;
    return launch_march<MARCH_BWD, false, true>(P, esr_stream(stream));
}

// ---- coarse stage (VoxurfC): same march on the SMOOTHED sdf grid, two transmittance passes -------------
ESR_API int esr_coarse_march_count(const esr_scene_t *scene, const float *rays_o, const float *rays_d,
                                   const float *mask_density, const float *sdf_smooth, int32_t n_rays,
                                   int32_t *cnt3, float *alphainv_last, float *cum_weights, int32_t *ray_stats,
                                   esr_plan_t *plan, void *stream)
{
    if (!scene || n_rays < 0 || !plan) return ESR_EINVAL;
    if (n_rays && (!rays_o || !rays_d || !mask_density || !sdf_smooth || !cnt3 || !alphainv_last || !cum_weights ||
                   !ray_stats))
        return ESR_EINVAL;
    MarchParams P = {};
    P.sc = *scene; P.rays_o = rays_o; P.rays_d = rays_d; P.mask_density = mask_density; P.sdf = sdf_smooth;
    P.n_rays = n_rays; P.cap = march_cap(scene); P.cnt3 = cnt3; P.alphainv_last = alphainv_last;
    P.cumw = cum_weights; P.stats = ray_stats; P.plan = plan;
    return launch_march<MARCH_COUNT, true>(P, esr_stream(stream));
}

ESR_API int esr_coarse_march_fill(const esr_scene_t *scene, const float *rays_o, const float *rays_d,
                                  const float *mask_density, const float *sdf_smooth, int32_t n_rays,
                                  const int32_t *off3, int32_t *rec_ray, int32_t *rec_step, float *rec_w,
                                  float *rec_sdf, void *stream)
{
    if (!scene || n_rays < 0) return ESR_EINVAL;
    if (n_rays && (!rays_o || !rays_d || !mask_density || !sdf_smooth || !off3 || !rec_ray || !rec_step ||
                   !rec_w || !rec_sdf))
        return ESR_EINVAL;
    MarchParams P = {};
    P.sc = *scene; P.rays_o = rays_o; P.rays_d = rays_d; P.mask_density = mask_density; P.sdf = sdf_smooth;
    P.n_rays = n_rays; P.cap = march_cap(scene); P.off3 = off3; P.rec_ray = rec_ray;
    P.rec_step = rec_step; P.rec_w = rec_w; P.rec_sdf = rec_sdf;
    return launch_march<MARCH_FILL, true>(P, esr_stream(stream));
}

ESR_API int esr_coarse_march_bwd(const esr_scene_t *scene, const float *rays_o, const float *rays_d,
                                 const float *mask_density, const float *sdf_smooth, int32_t n_rays,
                                 const int32_t *off3, const float *dweight, const float *dlast,
                                 float *grad_sdf_smooth, void *stream)
{
    if (!scene || n_rays < 0) return ESR_EINVAL;
    if (n_rays && (!rays_o || !rays_d || !mask_density || !sdf_smooth || !off3 || !dweight || !dlast ||
                   !grad_sdf_smooth))
        return ESR_EINVAL;
    MarchParams P = {};
    P.sc = *scene; P.rays_o = rays_o; P.rays_d = rays_d; P.mask_density = mask_density; P.sdf = sdf_smooth;
    P.n_rays = n_rays; P.cap = march_cap(scene); P.off3 = off3; P.dweight = dweight; P.dlast = dlast;
    P.grad_sdf = grad_sdf_smooth;
    return launch_march<MARCH_BWD, true>(P, esr_stream(stream));
}

// ---- coarse stage with cfg neus_alpha: "grad" (voxurfc.py:171-174, 204-210): the section SDFs are extrapolated with the
// trilinear sample of the dense gradient grid `gg` [X,Y,Z,3] (esr_central_grad_fwd of the UNSMOOTHED grid); the backward
// adds d/d gg into grad_gg, which esr_central_grad_bwd folds into the SDF gradient ----
ESR_API int esr_coarse_march_count_ga(const esr_scene_t *scene, const float *rays_o, const float *rays_d, const float *viewdirs,
                                      const float *mask_density, const float *sdf_smooth, const float *gg, int32_t n_rays,
                                      int32_t *cnt3, float *alphainv_last, float *cum_weights, int32_t *ray_stats,
                                      esr_plan_t *plan, void *stream)
{
    if (!scene || n_rays < 0 || !plan) return ESR_EINVAL;
    if (n_rays && (!rays_o || !rays_d || !viewdirs || !mask_density || !sdf_smooth || !gg || !cnt3 || !alphainv_last ||
                   !cum_weights || !ray_stats))
        return ESR_EINVAL;
    MarchParams P = {};
    P.sc = *scene; P.rays_o = rays_o; P.rays_d = rays_d; P.viewdirs = viewdirs; P.mask_density = mask_density;
    P.sdf = sdf_smooth; P.gg = gg;
    P.n_rays = n_rays; P.cap = march_cap(scene); P.cnt3 = cnt3; P.alphainv_last = alphainv_last;
    P.cumw = cum_weights; P.stats = ray_stats; P.plan = plan;
    return launch_march<MARCH_COUNT, true, true>(P, esr_stream(stream));
}

ESR_API int esr_coarse_march_fill_ga(const esr_scene_t *scene, const float *rays_o, const float *rays_d, const float *viewdirs,
                                     const float *mask_density, const float *sdf_smooth, const float *gg, int32_t n_rays,
                                     const int32_t *off3, int32_t *rec_ray, int32_t *rec_step, float *rec_w,
                                     float *rec_sdf, void *stream)
{
    if (!scene || n_rays < 0) return ESR_EINVAL;
    if (n_rays && (!rays_o || !rays_d || !viewdirs || !mask_density || !sdf_smooth || !gg || !off3 || !rec_ray ||
                   !rec_step || !rec_w || !rec_sdf))
        return ESR_EINVAL;
    MarchParams P = {};
    P.sc = *scene; P.rays_o = rays_o; P.rays_d = rays_d; P.viewdirs = viewdirs; P.mask_density = mask_density;
    P.sdf = sdf_smooth; P.gg = gg;
    P.n_rays = n_rays; P.cap = march_cap(scene); P.off3 = off3; P.rec_ray = rec_ray;
    P.rec_step = rec_step; P.rec_w = rec_w; P.rec_sdf = rec_sdf;
    return launch_march<MARCH_FILL, true, true>(P, esr_stream(stream));
}

ESR_API int esr_coarse_march_bwd_ga(const esr_scene_t *scene, const float *rays_o, const float *rays_d, const float *viewdirs,
                                    const float *mask_density, const float *sdf_smooth, const float *gg, int32_t n_rays,
                                    const int32_t *off3, const float *dweight, const float *dlast,
                                    float *grad_sdf_smooth, float *grad_gg, void *stream)
{
    if (!scene || n_rays < 0) return ESR_EINVAL;
    if (n_rays && (!rays_o || !rays_d || !viewdirs || !mask_density || !sdf_smooth || !gg || !off3 || !dweight || !dlast ||
                   !grad_sdf_smooth || !grad_gg))
        return ESR_EINVAL;
    MarchParams P = {};
    P.sc = *scene; P.rays_o = rays_o; P.rays_d = rays_d; P.viewdirs = viewdirs; P.mask_density = mask_density;
    P.sdf = sdf_smooth; P.gg = gg;
    P.n_rays = n_rays; P.cap = march_cap(scene); P.off3 = off3; P.dweight = dweight; P.dlast = dlast;
    P.grad_sdf = grad_sdf_smooth; P.grad_gg = grad_gg;
    return launch_march<MARCH_BWD, true, true>(P, esr_stream(stream));
}
