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

constexpr int XROWS = 104;  // rows of an X tile (96 + third colour group at 96-101)
constexpr int DXROWS = 64;  // rows of a dX tile (0..42 used)
constexpr int ROW_COL = 0, ROW_SDF = 6, ROW_FEAT = 7, ROW_NRM = 31, ROW_XYZ = 43, ROW_SIN = 46,
              ROW_COS = 61, ROW_VD = 76, ROW_VSIN = 79, ROW_VCOS = 82;

constexpr int MAX_SRC = 4;
constexpr int COLOR_ROW[3] = {0, 88, 96};          // first row of the three 6-row colour groups

struct FeatParams {
    esr_scene_t sc;
    // sample positions: march records (pts == nullptr) or explicit points
    const float *rays_o, *rays_d, *viewdirs;
    const int32_t *rec_ray, *rec_step;
    const float *rec_sdf;
    const float *pts, *pt_viewdirs, *pt_sdf;
    int n_pts;
    const float *sdf;
    const float *color_on[3], *color_off[3];         // grid per colour group on on-tiles / off-tiles
    int tiles_on, tiles_all;
    float *X, *gnorm;
    void *X16;                                        // bf16 row-quad tile of the bf16 engine (feat_fwd_kernel<., true>)
    // backward
    int n_src;
    const float *dX[MAX_SRC];
    float *gcol_on[MAX_SRC], *gcol_off[MAX_SRC];
    int src_t0[MAX_SRC], src_t1[MAX_SRC];
    const float *dsdf_extra;                          // [tiles*32] added to the SDF-value row, or null
    float *dsdf_out;                                  // explicit mode: gradient w.r.t. pt_sdf
    // gradient of the EXACT trilinear interpolant at the sample (value, d/dx, d/dy, d/dz in world units; esr_expgrad_fwd's
    // outputs), scattered here with esr_expgrad_bwd's weights: g4 [tiles*32][4] and / or (g4_self) the sample's own
    // SDF-value gradient as the value component; g4_zero_pad: out-of-grid corners dropped instead of border-replicated
    const float *g4;
    int g4_self, g4_zero_pad;
    float *grad_sdf;
    int t_begin, t_end;                               // backward: tiles that have a gradient source at all
};

// position, view direction and SDF value of sample j; false for padding lanes
__device__ __forceinline__ bool sample_inputs(const FeatParams &P, int j, float p[3], float vd[3], float &sdfv)
{
    if (P.pts) {
        if (j >= P.n_pts) return false;
#pragma unroll
        for (int a = 0; a < 3; ++a) { p[a] = P.pts[3 * j + a]; vd[a] = P.pt_viewdirs[3 * j + a]; }
        sdfv = P.pt_sdf[j];
        return true;
    }
    const int ray = P.rec_ray[j];
    if (ray < 0) return false;
    const RayGeom g = esr_ray_geom(P.rays_o, P.rays_d, ray, P.sc.xyz_min, P.sc.xyz_max, P.sc.near_, 1e9f,
                                   P.sc.stepdist);
    esr_ray_point(g.start, g.dir, P.sc.stepdist, P.rec_step[j], p);
#pragma unroll
    for (int a = 0; a < 3; ++a) vd[a] = P.viewdirs[3 * ray + a];
    sdfv = P.rec_sdf[j];
    return true;
}

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
                // unconditional loads (esr_ld_or0's reasoning): an out-of-grid corner reads voxel 0 with weight 0
                const float2 *v = reinterpret_cast<const float2 *>(
                    g + (inb ? (((int64_t)x * dims[1] + y) * dims[2] + z) * 6 : 0));
                const float2 a = v[0], b = v[1], c = v[2];
                w = inb ? w : 0.f;
                out[0] += a.x * w; out[1] += a.y * w; out[2] += b.x * w;
                out[3] += b.y * w; out[4] += c.x * w; out[5] += c.y * w;
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


// The 8 stencil taps of one axis (+-0.5 .. +-2 voxels, clamped) differ only in their coordinate ALONG the axis: their
// 8 x 8 corner fetches fall on a few cells along the axis x the same 2 x 2 perpendicular corners.  Those cells are read
// once into a lane-private LDS strip ("bar", [along 7][perp 4]) and each tap takes its 2 x 4 corners from there with two
// ds_read_b128 -- 84 gathers (and cell addresses) per sample instead of 192; the taps were 83 of the kernel's 155 us.
// 7 cells from the base cell of the lowest tap always suffice when no displacement exceeds 2 voxels: clamping is
// 1-Lipschitz, so the taps span <= 4 voxels (+ a few ulps of the reference's normalise / de-normalise round trip, which
// is why it is 7 and not 6) and the base cells differ by <= 5.  Other configurations take the direct kernel (BAR = false).
// The tap's arithmetic is esr_tri_fetch1's, operand for operand (an out-of-grid corner is a stored 0 with weight 0
// instead of a skipped term), so the features are bit-identical to the direct form.
constexpr int BAR_ALONG = 7, BAR_STRIDE = 4 * BAR_ALONG;    // 28 floats per lane: 16 lanes' b128 reads on distinct banks
constexpr float BAR_MAX_DISP = 2.0f;

__device__ __forceinline__ void bar_fill(const float *__restrict__ sdf, const int dims[3], const int i0[3], int axis,
                                         int b0, float *bar)
{
    const int pb = axis == 0 ? 1 : 0, pc = axis == 2 ? 1 : 2;        // the two perpendicular axes, in x,y,z order
#pragma unroll
    for (int o = 0; o < BAR_ALONG; ++o)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                int c[3];
                c[axis] = b0 + o; c[pb] = i0[pb] + cb; c[pc] = i0[pc] + cc;
                const bool inb = (c[0] >= 0) & (c[0] < dims[0]) & (c[1] >= 0) & (c[1] < dims[1]) & (c[2] >= 0) &
                                 (c[2] < dims[2]);
                bar[o * 4 + cb * 2 + cc] = esr_ld_or0(sdf, ((int64_t)c[0] * dims[1] + c[1]) * dims[2] + c[2], inb);
            }
}

__device__ __forceinline__ float bar_fetch1(int axis, int b0, const float *bar, const float idx[3])
{
    const Tri t = esr_tri_setup(idx);
    const int o = t.i0[axis] - b0;                                    // 0 .. 5 (see above)
    const float4 lo = *reinterpret_cast<const float4 *>(bar + o * 4), hi = *reinterpret_cast<const float4 *>(bar + o * 4 + 4);
    const float v[2][4] = {{lo.x, lo.y, lo.z, lo.w}, {hi.x, hi.y, hi.z, hi.w}};
    float acc = 0.f;
#pragma unroll
    for (int cx = 0; cx < 2; ++cx)
#pragma unroll
        for (int cy = 0; cy < 2; ++cy)
#pragma unroll
            for (int cz = 0; cz < 2; ++cz) {
                const int ca = axis == 0 ? cx : axis == 1 ? cy : cz;
                const int cb = axis == 0 ? cy : cx, cc = axis == 2 ? cy : cz;
                // explicit fma: the direct form compiles to one per corner; left to the compiler this loop became
                // packed multiplies + separate adds (1-ulp differences in half the taps)
                acc = __builtin_fmaf(v[ca][cb * 2 + cc], esr_corner_w(t, idx, cx, cy, cz), acc);
            }
    return acc;
}

template <bool BAR>
__global__ void __launch_bounds__(256) feat_fwd_kernel(FeatParams P)
{
    const esr_scene_t &sc = P.sc;
    const int gdims[3] = {sc.gx, sc.gy, sc.gz};
    const int total = P.tiles_all * 32;
    __shared__ __attribute__((aligned(16))) float bar_lds[256 * BAR_STRIDE];
    float *bar = bar_lds + threadIdx.x * BAR_STRIDE;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < total; j += gridDim.x * blockDim.x) {
        const int t = j >> 5, s = j & 31;
        // tile rows through buffer descriptors (see esr_common.h); X(row, v) writes X[t][row][s].
        // A wave covers the two tiles t2, t2 + 1 (one per half): one wave-uniform descriptor over both.
        const int t2 = __builtin_amdgcn_readfirstlane(t);
        const rsrc_t RX = make_rsrc(P.X + (size_t)t2 * XROWS * 32, 2 * XROWS * 32 * 4);
        const rsrc_t RG = make_rsrc(P.gnorm + (size_t)t2 * 4 * 32, 2 * 4 * 32 * 4);
        const int xoff = ((t - t2) * XROWS * 32 + s) * 4, goff = ((t - t2) * 4 * 32 + s) * 4;
        auto X = [&](int row, float v) { bstore1(RX, v, xoff, row * 128); };        // (not nt: the MLP kernels read X next)
        auto G = [&](int row, float v) { bstore1(RG, v, goff, row * 128); };
        float p[3], ind[3], unit[3], vdir[3], sdfv;
        if (!sample_inputs(P, j, p, vdir, sdfv)) {       // padding lane: inert zeros
            for (int r = 0; r < XROWS; ++r) X(r, 0.f);
            for (int k = 0; k < 4; ++k) G(k, 0.f);
            continue;
        }
        const bool on_tile = t < P.tiles_on;
        esr_world_to_index(p, sc.xyz_min, sc.xyz_max, gdims, ind);
        {
#pragma clang fp contract(off)
#pragma unroll
            for (int a = 0; a < 3; ++a) unit[a] = __fdiv_rn(p[a] - sc.xyz_min[a], sc.xyz_max[a] - sc.xyz_min[a]);
        }
        // colour groups (rows 0-5, 88-93, 96-101): each fed by the grid configured for this tile type
#pragma unroll
        for (int gi = 0; gi < 3; ++gi) {
            const float *grid = on_tile ? P.color_on[gi] : P.color_off[gi];
            float col[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (grid) tri_fetch6(grid, gdims, ind, col);
#pragma unroll
            for (int c = 0; c < 6; ++c) X(COLOR_ROW[gi] + c, col[c]);
        }
        X(94, 0.f); X(95, 0.f); X(102, 0.f); X(103, 0.f);
        X(85, 0.f); X(86, 0.f); X(87, 0.f);
        X(ROW_SDF, sdfv);
        // 24-tap SDF stencil: reference axis order is (z, y, x) = grid axes (2, 1, 0)
        float grad[3][4];
#pragma unroll
        for (int ar = 0; ar < 3; ++ar) {
            const int axis = 2 - ar;
            float ixm[4][3], ixp[4][3], cm[4], cp[4];
            int b0 = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                cm[k] = tap_index(ind, gdims, axis, -sc.grad_feat[k], ixm[k]);
                cp[k] = tap_index(ind, gdims, axis, sc.grad_feat[k], ixp[k]);
            }
            if constexpr (BAR) {
                float low = ixm[0][axis];
#pragma unroll
                for (int k = 1; k < 4; ++k) low = fminf(low, ixm[k][axis]);
                b0 = (int)floorf(low);
                const Tri t0 = esr_tri_setup(ixm[0]);            // perpendicular base cells: the same for every tap
                bar_fill(P.sdf, gdims, t0.i0, axis, b0, bar);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float fm = BAR ? bar_fetch1(axis, b0, bar, ixm[k]) : esr_tri_fetch1(P.sdf, gdims, ixm[k]);
                const float fp = BAR ? bar_fetch1(axis, b0, bar, ixp[k]) : esr_tri_fetch1(P.sdf, gdims, ixp[k]);
                X(ROW_FEAT + (2 * ar) * 4 + k, fm);
                X(ROW_FEAT + (2 * ar + 1) * 4 + k, fp);
                // + 1e-12: the LTS renderer's guard (esrnerf.py:1560) for taps that clamp onto each other
                // (points pushed outside the box); a no-op in fp32 for in-box samples, where cp - cm >= 0.5
                grad[ar][k] = (fp - fm) / ((cp[k] - cm[k]) + 1e-12f) / sc.voxel_size;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float nrm = sqrtf(grad[0][k] * grad[0][k] + grad[1][k] * grad[1][k] + grad[2][k] * grad[2][k]);
            const float den = fmaxf(nrm, 1e-12f);
            G(k, nrm);
#pragma unroll
            for (int ar = 0; ar < 3; ++ar) X(ROW_NRM + ar * 4 + k, grad[ar][k] / den);
        }
        // positional encodings (coordinate-major, frequencies 1,2,4,8,16)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            X(ROW_XYZ + c, unit[c]);
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const float a = unit[c] * (float)(1 << i);
                float sn, cs;
                esr_sincos(a, sn, cs);
                X(ROW_SIN + c * 5 + i, sn);
                X(ROW_COS + c * 5 + i, cs);
            }
            const float v = vdir[c];
            float sn, cs;
            esr_sincos(v, sn, cs);
            X(ROW_VD + c, v);
            X(ROW_VSIN + c, sn);
            X(ROW_VCOS + c, cs);
        }
    }
}

// feat_fwd16_kernel: the bf16 engine's input tile.  Instead of the fp32 tile the kernel writes the rows as bf16 in the ROW-QUAD layout of
// the saved hidden tiles (mlp_common.h: store_tiles_bf16 -- [row / 4][32 sample slots][4 rows], 8 bytes per (quad, sample)):
// the forward's first-layer operand is then two 8-byte loads per 16 input rows instead of 16 dword loads + conversions, and
// the first-layer weight gradient stages the tile exactly like a hidden layer's.  24 quads = rows 0..95 (85..87, 94, 95
// zero) + quads 24, 25 = the first eight rows with the SECOND colour group in rows 0..5 (the non-emissive net's detached
// pass over emissive-on tiles reads its colours from rows 88..93).  The values are the fp32 rows rounded to bf16 -- what
// the bf16 kernels made of them on load -- so nothing downstream changes.  Of the fp32 tile only the normal rows 31..42
// are still written (the feature backward reads them).  Rows are produced in ascending order so that a quad can leave as
// soon as its four rows exist (the fp32 kernel keeps its own order: this one holds ~10 more registers, which costs that
// kernel its fourth wave per SIMD: 0.095 -> 0.122 ms at C2).
constexpr int X16_QUADS = 26;
__global__ void __launch_bounds__(256) feat_fwd16_kernel(FeatParams P)
{
    constexpr bool BAR = true, X16M = true;          // (the stencil bars' reach is checked by the host entry)
    const esr_scene_t &sc = P.sc;
    const int gdims[3] = {sc.gx, sc.gy, sc.gz};
    const int total = P.tiles_all * 32;
    __shared__ __attribute__((aligned(16))) float bar_lds[256 * BAR_STRIDE];
    float *bar = bar_lds + threadIdx.x * BAR_STRIDE;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < total; j += gridDim.x * blockDim.x) {
        const int t = j >> 5, s = j & 31;
        // tile rows through buffer descriptors (see esr_common.h); X(row, v) writes X[t][row][s].
        // A wave covers the two tiles t2, t2 + 1 (one per half): one wave-uniform descriptor over both.
        const int t2 = __builtin_amdgcn_readfirstlane(t);
        const rsrc_t RX = make_rsrc(P.X + (size_t)t2 * XROWS * 32, 2 * XROWS * 32 * 4);
        const rsrc_t RG = make_rsrc(P.gnorm + (size_t)t2 * 4 * 32, 2 * 4 * 32 * 4);
        const int xoff = ((t - t2) * XROWS * 32 + s) * 4, goff = ((t - t2) * 4 * 32 + s) * 4;
        auto X = [&](int row, float v) { bstore1(RX, v, xoff, row * 128); };        // (not nt: the MLP kernels read X next)
        auto G = [&](int row, float v) { bstore1(RG, v, goff, row * 128); };
        // bf16 tile: this sample's 8 bytes of quad q
        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
        uint2 *const x16 = X16M ? reinterpret_cast<uint2 *>(P.X16) + ((size_t)t * X16_QUADS * 32 +
                                                                     (8 * ((s >> 1) & 3) + 2 * (s >> 3) + (s & 1)))
                                : nullptr;
        auto quad = [&](int q, float a, float b, float c, float d) {
            bf16x2_t lo, hi;
            lo[0] = (__bf16)a; lo[1] = (__bf16)b; hi[0] = (__bf16)c; hi[1] = (__bf16)d;
            x16[q * 32] = make_uint2(__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi));
        };
        float pend[4] = {0.f, 0.f, 0.f, 0.f};
        auto put = [&](int row, float v) {               // rows arrive in ascending order (row: compile-time constant)
            if constexpr (X16M) {
                pend[row & 3] = v;
                if ((row & 3) == 3) quad(row >> 2, pend[0], pend[1], pend[2], pend[3]);
                if (row >= ROW_NRM && row < ROW_NRM + 12) X(row, v);
            } else {
                X(row, v);
            }
        };
        float p[3], ind[3], unit[3], vdir[3], sdfv;
        if (!sample_inputs(P, j, p, vdir, sdfv)) {       // padding lane: inert zeros
            if constexpr (X16M) {
                for (int q = 0; q < X16_QUADS; ++q) quad(q, 0.f, 0.f, 0.f, 0.f);
                for (int r = ROW_NRM; r < ROW_NRM + 12; ++r) X(r, 0.f);
            } else {
                for (int r = 0; r < XROWS; ++r) X(r, 0.f);
            }
            for (int k = 0; k < 4; ++k) G(k, 0.f);
            continue;
        }
        const bool on_tile = t < P.tiles_on;
        esr_world_to_index(p, sc.xyz_min, sc.xyz_max, gdims, ind);
        {
#pragma clang fp contract(off)
#pragma unroll
            for (int a = 0; a < 3; ++a) unit[a] = __fdiv_rn(p[a] - sc.xyz_min[a], sc.xyz_max[a] - sc.xyz_min[a]);
        }
        // colour groups (rows 0-5, 88-93, 96-101): each fed by the grid configured for this tile type
        float col[3][6];
#pragma unroll
        for (int gi = 0; gi < 3; ++gi) {
            const float *grid = on_tile ? P.color_on[gi] : P.color_off[gi];
#pragma unroll
            for (int c = 0; c < 6; ++c) col[gi][c] = 0.f;
            if (grid && !(X16M && gi == 2)) tri_fetch6(grid, gdims, ind, col[gi]);
        }
#pragma unroll
        for (int c = 0; c < 6; ++c) put(COLOR_ROW[0] + c, col[0][c]);
        put(ROW_SDF, sdfv);
        // 24-tap SDF stencil: reference axis order is (z, y, x) = grid axes (2, 1, 0)
        float grad[3][4], feat7 = 0.f;
#pragma unroll
        for (int ar = 0; ar < 3; ++ar) {
            const int axis = 2 - ar;
            float ixm[4][3], ixp[4][3], cm[4], cp[4], fm[4], fp[4];
            int b0 = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                cm[k] = tap_index(ind, gdims, axis, -sc.grad_feat[k], ixm[k]);
                cp[k] = tap_index(ind, gdims, axis, sc.grad_feat[k], ixp[k]);
            }
            if constexpr (BAR) {
                float low = ixm[0][axis];
#pragma unroll
                for (int k = 1; k < 4; ++k) low = fminf(low, ixm[k][axis]);
                b0 = (int)floorf(low);
                const Tri t0 = esr_tri_setup(ixm[0]);            // perpendicular base cells: the same for every tap
                bar_fill(P.sdf, gdims, t0.i0, axis, b0, bar);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                fm[k] = BAR ? bar_fetch1(axis, b0, bar, ixm[k]) : esr_tri_fetch1(P.sdf, gdims, ixm[k]);
                fp[k] = BAR ? bar_fetch1(axis, b0, bar, ixp[k]) : esr_tri_fetch1(P.sdf, gdims, ixp[k]);
                // + 1e-12: the LTS renderer's guard (esrnerf.py:1560) for taps that clamp onto each other
                // (points pushed outside the box); a no-op in fp32 for in-box samples, where cp - cm >= 0.5
                grad[ar][k] = (fp[k] - fm[k]) / ((cp[k] - cm[k]) + 1e-12f) / sc.voxel_size;
            }
            if (ar == 0) feat7 = fm[0];
#pragma unroll
            for (int k = 0; k < 4; ++k) put(ROW_FEAT + (2 * ar) * 4 + k, fm[k]);
#pragma unroll
            for (int k = 0; k < 4; ++k) put(ROW_FEAT + (2 * ar + 1) * 4 + k, fp[k]);
        }
        float den[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float nrm = sqrtf(grad[0][k] * grad[0][k] + grad[1][k] * grad[1][k] + grad[2][k] * grad[2][k]);
            den[k] = fmaxf(nrm, 1e-12f);
            G(k, nrm);
        }
#pragma unroll
        for (int ar = 0; ar < 3; ++ar)
#pragma unroll
            for (int k = 0; k < 4; ++k) put(ROW_NRM + ar * 4 + k, grad[ar][k] / den[k]);
        // positional encodings (coordinate-major, frequencies 1,2,4,8,16)
        float cs[15], vs[3], vc[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) put(ROW_XYZ + c, unit[c]);
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int i = 0; i < 5; ++i) {                    // the sine rows leave as they are formed, the cosines wait their turn
                float sn;
                esr_sincos(unit[c] * (float)(1 << i), sn, cs[c * 5 + i]);
                put(ROW_SIN + c * 5 + i, sn);
            }
#pragma unroll
        for (int q = 0; q < 15; ++q) put(ROW_COS + q, cs[q]);
#pragma unroll
        for (int c = 0; c < 3; ++c) esr_sincos(vdir[c], vs[c], vc[c]);
#pragma unroll
        for (int c = 0; c < 3; ++c) put(ROW_VD + c, vdir[c]);
#pragma unroll
        for (int c = 0; c < 3; ++c) put(ROW_VSIN + c, vs[c]);
#pragma unroll
        for (int c = 0; c < 3; ++c) put(ROW_VCOS + c, vc[c]);
        put(85, 0.f); put(86, 0.f); put(87, 0.f);
#pragma unroll
        for (int c = 0; c < 6; ++c) put(COLOR_ROW[1] + c, col[1][c]);
        put(94, 0.f); put(95, 0.f);
        if constexpr (X16M) {
            quad(24, col[1][0], col[1][1], col[1][2], col[1][3]);
            quad(25, col[1][4], col[1][5], sdfv, feat7);
        } else {
#pragma unroll
            for (int c = 0; c < 6; ++c) X(COLOR_ROW[2] + c, col[2][c]);
            X(102, 0.f); X(103, 0.f);
        }
    }
}

// ---- backward: LDS accumulation window --------------------------------------------
// A direct scatter issues 248 float atomics per sample (25 SDF taps x 8 corners + 8 x 6
// colour channels): 130 M atomics per C2 step, 3.5 ms, atomic-rate bound.  Consecutive
// samples of a ray are 0.5 voxel apart, so the 32 samples of a tile touch only a few
// hundred distinct cells.  One wave owns one tile (two lanes per sample share the taps),
// accumulates into a dense per-wave window of the grid held in LDS (ds_add_f32), then
// flushes the non-zero cells with z-contiguous global atomics -- ~10x fewer global
// atomics, better shaped.  Cells outside the window (ray changes inside a tile, long
// diagonal tiles) fall back to direct global atomics, so the result never depends on
// the window fitting.
// The window cells are DOUBLES: ds_add_f32 retires ~1 lane per 3 clocks on gfx950 (80 ns per wave
// instruction, tools/ubench/lds_atomic.hip) while ds_add_f64 runs at the integer-atomic rate, 8x faster; the
// f32 form kept the LDS pipe 60 % busy for the whole kernel (SQ_LDS_IDX_ACTIVE).  As a side effect the
// in-window sum is exact to ~2^-52 and rounded to fp32 once at the flush.
constexpr int WIN_CELLS = 2560;           // 20 KB of doubles per wave (2 workgroups x 4 waves x 20 KB per CU)

// explicit LDS address space: with a generic pointer hipcc merges the LDS and the global
// branch of window_add into one flat_atomic_add_f32 on a selected address
typedef __attribute__((address_space(3))) double lds_cell;

// One accumulation window per SEGMENT of the tile: a run of consecutive samples of one ray (or, for explicit points, of
// one spatially coherent run).  At the start of training a tile is 32 consecutive samples of ONE ray; later (large s_val) a
// ray keeps ~20 samples, and the secondary rays of the LTS stages 7 (median): a tile then holds pieces of 2-10 rays that
// can sit anywhere in the grid.  Everything about the segments lives in per-lane registers (both lane halves hold the
// same values) and is built with SEGMENTED scans over the 32 samples -- 5 shuffle steps whatever the number of
// segments -- and every window gets exactly the LDS cells its box needs (exclusive prefix sum of the box sizes).
// Round 2's form kept <= 4 windows in wave-uniform registers (one 6-step wave reduction per window and bound: 48-174
// shuffles per tile, 35 spilled scalar registers), gave each an EQUAL share of the LDS and shaved boxes to fit: the fifth ray
// of a tile and everything outside a shaved box went to global atomics tap by tap (12 % resp. ~40 % of the C4 secondary
// pass's tiles, which ran 5x slower per tile than the primary pass).  A window that does not fit now gets no cells at all
// (its lanes use global atomics): no bounds tests anywhere.
struct Segs {
    unsigned heads;                    // wave-uniform: bit s set = sample s starts a segment
    int start, last;                   // per lane: first / last sample (0..31) of this lane's segment
    int mn[3], mx[3];                  // per lane: bounds of the (clamped) base cells of the segment's samples
};
struct LaneWin {                       // the window of THIS lane's segment in one phase
    lds_cell *lds;                     // its first cell
    int lo[3], wd[3];                  // origin and extent in grid cells (x, y, z); z fastest
    int base, cells;                   // first LDS word, number of words (cells x channels)
    int total;                         // wave-uniform: words asked for by all segments of the tile
    bool has;                          // the window fits (else: this lane scatters straight to global memory)
};

// Ordering of one wave's own LDS accesses: the hardware executes a wave's DS instructions in
// order, so waiting for lgkmcnt (and stopping compiler motion) is enough.  __threadfence_block()
// would also wait vmcnt(0), i.e. drain the global atomics of the previous flush (~us each).
__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// segments of the tile: a new one starts where the key changes (ray id / run id), after a padding lane, and -- cut8 --
// at every 8th sample of a segment (see the kernel: ray pieces whose box does not fit)
__device__ __forceinline__ Segs segs_build(int key, bool valid, bool cut8, const int i0c[3], int s, int h)
{
    Segs S;
    const int pk = __shfl_up(key, 1);
    const bool pv = __shfl_up(valid ? 1 : 0, 1) != 0;
    bool head = valid && (s == 0 || !pv || key != pk);
    S.heads = (unsigned)__ballot(head);                              // low word: one bit per sample
    const unsigned upto = (2u << s) - 1u;                            // bits 0..s  (s = 31: 2u << 31 == 0, - 1 -> all ones)
    unsigned below = S.heads & upto;
    S.start = below ? 31 - __clz(below) : s;
    if (cut8) {
        head = head || (valid && ((s - S.start) & 7) == 0);
        S.heads = (unsigned)__ballot(head);
        below = S.heads & upto;
        S.start = below ? 31 - __clz(below) : s;
    }
    const unsigned above = S.heads & ~upto;
    const int endx = above ? __ffs(above) - 1 : 32;                  // first sample of the next segment
    const unsigned vmask = (unsigned)__ballot(valid);
    const unsigned inseg = vmask & (endx == 32 ? 0xffffffffu : ((1u << endx) - 1u)) & ~((1u << S.start) - 1u);
    S.last = inseg ? 31 - __clz(inseg) : s;
    // segmented min-scan; half 0 scans the cells, half 1 their negatives (= max): 3 values x 5 steps for both bounds
    int v[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) v[a] = valid ? (h ? -i0c[a] : i0c[a]) : 0x3fffffff;
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int o = __shfl_up(v[a], off);
            if (s - off >= S.start) v[a] = min(v[a], o);             // (same segment => same lane half)
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int r = __shfl(v[a], 32 * h + S.last);                 // the segment's last sample holds the whole scan
        const int other = __shfl_xor(r, 32);
        S.mn[a] = h == 0 ? r : other;
        S.mx[a] = h == 0 ? -other : -r;
    }
    return S;
}

// this lane's window for one phase: the segment's box [mn - below, mx + above] clipped to the grid, `ch` words per cell
__device__ __forceinline__ LaneWin seg_window(const Segs &S, bool valid, const int dims[3], int below, int above, int ch,
                                              lds_cell *lds, int s, int h)
{
    LaneWin w;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        w.lo[a] = max(S.mn[a] - below, 0);
        w.wd[a] = max(min(S.mx[a] + above, dims[a] - 1) - w.lo[a] + 1, 0);
    }
    // (a box too large to count in 32 bits is simply "does not fit")
    const long long want = valid ? (long long)w.wd[0] * w.wd[1] * w.wd[2] * ch : 0;
    w.cells = (int)(want < (long long)WIN_CELLS + 1 ? want : (long long)WIN_CELLS + 1);
    int inc = (valid && s == S.start) ? w.cells : 0;                 // one contribution per segment, at its head
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) {
        const int o = __shfl_up(inc, off);
        if (s >= off) inc += o;
    }
    w.base = __shfl(inc, 32 * h + S.start) - w.cells;
    w.total = __builtin_amdgcn_readlane(inc, 31);
    w.has = valid && w.base + w.cells <= WIN_CELLS;
    w.lds = lds + w.base;
    return w;
}

__device__ __forceinline__ void lds_add(lds_cell *p, float v)
{
    __hip_atomic_fetch_add(p, (double)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// every in-grid tap of a lane lies inside its window by construction (the box spans the segment's base cells plus the
// phase's margins): no bounds tests; a lane without a window goes to global memory
__device__ __forceinline__ void cell_add(const LaneWin &w, int off, float *__restrict__ g, int64_t gi, float v)
{
    if (w.has) lds_add(w.lds + off, v);
    else atomicAdd(g + gi, v);
}

__device__ __forceinline__ void windows_zero(const LaneWin &w, lds_cell *lds, int lane)
{
    const int n = min(w.total, WIN_CELLS);
    for (int i = lane; i < n; i += 64) lds[i] = 0.0;
}

// non-zero cells of every window that fitted -> z-contiguous global atomics
__device__ __forceinline__ void windows_flush(const Segs &S, const LaneWin &w, int ch, lds_cell *lds,
                                              float *__restrict__ g, const int dims[3], int lane)
{
    unsigned hm = S.heads;
    while (hm) {                                                     // wave-uniform loop over the segments
        const int hs = __builtin_amdgcn_readfirstlane(__ffs(hm) - 1);
        hm &= hm - 1;
        const int base = __builtin_amdgcn_readlane(w.base, hs), n = __builtin_amdgcn_readlane(w.cells, hs);
        if (base + n > WIN_CELLS) continue;
        const int lo0 = __builtin_amdgcn_readlane(w.lo[0], hs), lo1 = __builtin_amdgcn_readlane(w.lo[1], hs),
                  lo2 = __builtin_amdgcn_readlane(w.lo[2], hs);
        const int wd1 = __builtin_amdgcn_readlane(w.wd[1], hs), wd2 = __builtin_amdgcn_readlane(w.wd[2], hs);
        const int row = wd2 * ch;                         // floats per (x,y) column, contiguous in memory too
        // i / row and xy / wd[1] through the (wave-uniform) reciprocals: gfx950 has no integer divide, the two signed
        // divisions were ~50 of this loop's ~70 vector instructions.  floor((i + 0.5) * (1 / d)) == i / d for
        // 0 <= i < 2^21: the product is within 1.2e-7 relative of (i + 0.5) / d, which lies >= 0.5 / d from an integer.
        static_assert(WIN_CELLS * 6 < (1 << 21), "reciprocal division below is exact for i < 2^21");
        const float r_row = 1.0f / (float)max(row, 1), r_wy = 1.0f / (float)max(wd1, 1);
        for (int i = lane; i < n; i += 64) {
            const float v = (float)lds[base + i];
            if (v != 0.f) {
                const int xy = (int)(((float)i + 0.5f) * r_row), r = i - xy * row;
                const int wx = (int)(((float)xy + 0.5f) * r_wy), wy = xy - wx * wd1;
                atomicAdd(&g[(((int64_t)(lo0 + wx) * dims[1] + (lo1 + wy)) * dims[2] + lo2) * ch + r], v);
            }
        }
    }
}

__global__ void __launch_bounds__(256, 2) feat_bwd_kernel(FeatParams P)
{
    extern __shared__ __attribute__((aligned(16))) double win_all[];
    const esr_scene_t &sc = P.sc;
    const int gdims[3] = {sc.gx, sc.gy, sc.gz};
    const int lane = esr_lane();
    const int s = lane & 31, h = lane >> 5;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    lds_cell *const lds = (lds_cell *)(win_all + (threadIdx.x >> 6) * WIN_CELLS);
    // The march record and the ray of the NEXT tile travel one tile ahead (8 registers): record -> ray -> position is a chain
    // of two dependent loads that every tile used to start with, with nothing to hide it behind.
    int nx_ray = -1, nx_step = 0;
    float nx_o[3] = {0.f, 0.f, 0.f}, nx_d[3] = {0.f, 0.f, 1.f};
    auto fetch_rec = [&](int tt) {
        if (P.pts) return;                                                 // (explicit points: independent loads, below)
        const int jj = (tt < P.t_end ? tt : P.t_end - 1) * 32 + s;         // past the end: the last tile again, unused
        nx_ray = P.rec_ray[jj];
        nx_step = P.rec_step[jj];
    };
    auto fetch_ray = [&]() {
        if (P.pts) return;
        const int rc = max(nx_ray, 0);                                     // padding lanes read ray 0 (unconditional loads)
#pragma unroll
        for (int a = 0; a < 3; ++a) { nx_o[a] = P.rays_o[3 * rc + a]; nx_d[a] = P.rays_d[3 * rc + a]; }
    };
    if (P.t_begin + wave < P.t_end) { fetch_rec(P.t_begin + wave); fetch_ray(); }
    for (int t = P.t_begin + wave; t < P.t_end; t += nwaves) {
        const int j = t * 32 + s;
        const int ray_ = nx_ray, step_ = nx_step;
        float ro_[3], rd_[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) { ro_[a] = nx_o[a]; rd_[a] = nx_d[a]; }
        fetch_rec(t + nwaves);
        const float *Xt = P.X + (size_t)t * XROWS * 32 + s;
        const float *Gn = P.gnorm + (size_t)t * 4 * 32 + s;
        const bool on_tile = t < P.tiles_on;
        float p[3] = {0.f, 0.f, 0.f}, ind[3] = {0.f, 0.f, 0.f};
        int i0[3] = {0, 0, 0};
        bool valid;
        if (P.pts) {
            valid = j < P.n_pts;
            if (valid) {
#pragma unroll
                for (int a = 0; a < 3; ++a) p[a] = P.pts[3 * j + a];
            }
        } else {                                         // sample_inputs' arithmetic on the prefetched record and ray
            valid = ray_ >= 0;
            float t_min, t_max, start[3], dir[3];
            esr_ray_trange(ro_, rd_, sc.xyz_min, sc.xyz_max, sc.near_, 1e9f, t_min, t_max);
            esr_ray_start_dir(ro_, rd_, t_min, esr_ray_norm(rd_), start, dir);
            esr_ray_point(start, dir, sc.stepdist, step_, p);
        }
        // Gradient rows of this sample, summed over the nets that consumed the tile (rows 6-42 are shared), and the saved
        // normals: ALL of a tile's loads are issued here, before anything waits.  (They used to be fetched where they are
        // used, behind wave-uniform branches: ~60 load -> wait round trips per tile were 0.08 of this kernel's 0.29 ms at
        // C2 -- tools/variant.sh with the loads replaced by constants.)  Lane half h needs the rows of its two bars only.
        int nact = 0, k_one = 0;
#pragma unroll
        for (int k = 0; k < MAX_SRC; ++k)
            if (k < P.n_src && t >= P.src_t0[k] && t < P.src_t1[k]) { if (!nact) k_one = k; ++nact; }
        float r_n[3][4], r_dn[3][4], r_nrm[4], r_df[2][2][4], r_dsdf = 0.f, r_d3[3] = {0.f, 0.f, 0.f};
        auto dx_rows = [&](const float *dXt, bool first) {
            auto put = [&](float &dst, int row) { const float v = dXt[row * 32]; dst = first ? v : dst + v; };
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#pragma unroll
                for (int ar = 0; ar < 3; ++ar) put(r_dn[ar][k], ROW_NRM + ar * 4 + k);
#pragma unroll
                for (int bar = 0; bar < 2; ++bar) {
                    const int ar = bar == 0 ? h : 2;
                    put(r_df[bar][0][k], ROW_FEAT + (2 * ar) * 4 + k);
                    put(r_df[bar][1][k], ROW_FEAT + (2 * ar + 1) * 4 + k);
                }
            }
            put(r_dsdf, ROW_SDF);
        };
        if (P.grad_sdf) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                r_nrm[k] = Gn[k * 32];
#pragma unroll
                for (int ar = 0; ar < 3; ++ar) r_n[ar][k] = Xt[(ROW_NRM + ar * 4 + k) * 32];
            }
            if (nact == 1) {
                dx_rows(P.dX[k_one] + (size_t)t * DXROWS * 32 + s, true);
            } else {
                bool first = true;
#pragma unroll
                for (int k = 0; k < MAX_SRC; ++k)
                    if (k < P.n_src && t >= P.src_t0[k] && t < P.src_t1[k]) {
                        dx_rows(P.dX[k] + (size_t)t * DXROWS * 32 + s, first);
                        first = false;
                    }
            }
            if (P.dsdf_extra) r_dsdf += P.dsdf_extra[j];
        }
        float4 r_g4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (P.grad_sdf && P.g4) r_g4 = *reinterpret_cast<const float4 *>(P.g4 + 4 * (size_t)j);
        if (nact == 1) {             // the colour rows of the tile's only net (several nets: fetched in their passes)
#pragma unroll
            for (int c = 0; c < 3; ++c) r_d3[c] = P.dX[k_one][((size_t)t * DXROWS + ROW_COL + 3 * h + c) * 32 + s];
        }
        if (valid) {
            esr_world_to_index(p, sc.xyz_min, sc.xyz_max, gdims, ind);
#pragma unroll
            for (int a = 0; a < 3; ++a) i0[a] = (int)floorf(ind[a]);
        }
        // The reference clamps every coordinate of a tap to the grid (not only the displaced one), which
        // matters for explicit points that a perturbation pushed outside the box: the bars are anchored
        // at the clamped position (identical to `ind` for in-box samples).
        float indc[3] = {0.f, 0.f, 0.f};
        int i0c[3] = {0, 0, 0};
        if (valid) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                indc[a] = fminf(fmaxf(ind[a], 0.f), (float)(gdims[a] - 1));
                i0c[a] = (int)floorf(indc[a]);
            }
        }
        // One accumulation window per ray piece of the tile.  Explicit points carry no ray id, but the large explicit passes
        // of the LTS stages (the perturbed re-evaluation of every surviving sample, esrnerf.py:807-830) list their points
        // in ray order: the tile is cut into runs wherever consecutive points jump by more than 3 cells -- each run gets
        // a tight window like a ray piece does.
        int key = 0;
        if (P.pts) {
            const int px = __shfl_up(i0c[0], 1), py = __shfl_up(i0c[1], 1), pz = __shfl_up(i0c[2], 1);
            const bool pv = __shfl_up(valid ? 1 : 0, 1) != 0;
            const bool jump = valid && s > 0 && pv &&
                              (abs(i0c[0] - px) > 3 || abs(i0c[1] - py) > 3 || abs(i0c[2] - pz) > 3);
            const unsigned runs = (unsigned)__ballot(jump);                  // low half: one bit per sample
            key = __popc(runs & ((2u << s) - 1u));
        } else {
            key = valid ? ray_ : 0;
        }
        Segs SG = segs_build(key, valid, false, i0c, s, h);
        // A ray piece that runs diagonally through the grid has a bounding box far larger than the cells it touches
        // (32 samples = 16 voxels of path along (1,1,1): 15^3 SDF cells, 11^3 x 6 colour floats).  When the tile's boxes
        // do not fit, every piece is cut into runs of 8 samples with a window each: 8 samples span <= 4 voxels.
        {
            const LaneWin probe = P.grad_sdf ? seg_window(SG, valid, gdims, 2, 3, 1, lds, s, h)
                                             : seg_window(SG, valid, gdims, 0, 1, 6, lds, s, h);
            if (probe.total > WIN_CELLS) SG = segs_build(key, valid, true, i0c, s, h);
        }
        // ---- phase 1: SDF grid.  The value tap and the 24 stencil taps of a sample touch only three
        // 6x2x2 "bars" of cells (one per axis, sharing the central 2x2x2): the taps of an axis are first
        // reduced in registers onto the 6 cells along that axis, then spread over the 2x2 perpendicular
        // corners -- 72 LDS atomics per sample instead of 200 (LDS float atomics retire ~1 lane per 1.5
        // clocks and were 59 % of this kernel's wave time).  Lane half 0 owns the z bar and the lower half
        // of the x bar, lane half 1 the y bar and the upper half of the x bar.
        if (P.grad_sdf) {       // null: the SDF grid is frozen (re-lighting fine-tune), colour phase only
        const LaneWin w = seg_window(SG, valid, gdims, 2, 3, 1, lds, s, h);
        windows_zero(w, lds, lane);
        lds_fence();
        if (valid) {
            // gradient w.r.t. the finite-difference vectors (through F.normalize), all 3 axes x 4 radii
            // (only the two reference axes of this lane's bars: ar = h for bar 0, ar = 2 for bar 1; one reciprocal per radius
            // instead of three divisions -- v_rcp_f32 is within 1 ulp, these are gradients)
            float through[2][4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float nrm = r_nrm[k];
                float dot = 0.f;
#pragma unroll
                for (int ar = 0; ar < 3; ++ar) dot += r_n[ar][k] * r_dn[ar][k];
                const bool big = nrm > 1e-12f;           // projection for |g| > eps, plain 1/eps scaling below it
                const float inv = big ? __builtin_amdgcn_rcpf(nrm) : 1e12f;
                const float n0 = h ? r_n[1][k] : r_n[0][k], d0 = h ? r_dn[1][k] : r_dn[0][k];
                through[0][k] = (big ? d0 - n0 * dot : d0) * inv;
                through[1][k] = (big ? r_dn[2][k] - r_n[2][k] * dot : r_dn[2][k]) * inv;
            }
            const float inv_vox = __builtin_amdgcn_rcpf(sc.voxel_size);
            // perpendicular (centre) weights per grid axis: [axis][low/high corner]
            float wc[3][2];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                wc[a][0] = (float)(i0c[a] + 1) - indc[a];
                wc[a][1] = indc[a] - (float)i0c[a];
            }
            float d_sdf = r_dsdf;
            if (P.dsdf_out || P.g4_self) {    // explicit points: the SDF value is an input, not a tap of the stencil
                if (P.dsdf_out && h == 0) P.dsdf_out[j] = d_sdf;         // (g4_self: it goes to the grid through the exact
                d_sdf = 0.f;                                             //  interpolant below)
            }
#pragma unroll
            for (int bar = 0; bar < 2; ++bar) {
                // bar 0: z (h=0) or y (h=1); bar 1: x, split between the halves.  ar: reference axis order
                const int ar = bar == 0 ? h : 2;
                const int axis = 2 - ar;
                const int iA = axis == 0 ? i0c[0] : (axis == 1 ? i0c[1] : i0c[2]);
                const int dimA = axis == 0 ? gdims[0] : (axis == 1 ? gdims[1] : gdims[2]);
                const float indA = axis == 0 ? ind[0] : (axis == 1 ? ind[1] : ind[2]);
                float acc6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                // a tap at index ixA gives cell c the hat weight max(0, 1 - |ixA - c|): (fl + 1 - ixA) on its base cell, (ixA - fl)
                // on the next, 0 elsewhere -- three instructions per cell instead of two compares, two selects and two adds
                const float baseA = (float)(iA - 2);
                auto deposit = [&](float ixA, float d) {
                    const float t = ixA - baseA;                     // exact: small integers and an index < 2^12
#pragma unroll
                    for (int o = 0; o < 6; ++o) acc6[o] = fmaf(fmaxf(1.f - fabsf(t - (float)o), 0.f), d, acc6[o]);
                };
                if (bar == 0 && h == 0) deposit(indA, d_sdf);          // the SDF value tap rides on the z bar
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float cm, cp, ixm, ixp;
                    {
#pragma clang fp contract(off)
                        const float top = (float)(dimA - 1);
                        cm = fminf(fmaxf(indA - sc.grad_feat[k], 0.f), top);
                        cp = fminf(fmaxf(indA + sc.grad_feat[k], 0.f), top);
#ifdef FEAT_EXP_BWD_ROUNDTRIP
                        ixm = __fdiv_rn((__fdiv_rn(cm, top) * 2.0f - 1.0f) + 1.0f, 2.0f) * top;
                        ixp = __fdiv_rn((__fdiv_rn(cp, top) * 2.0f - 1.0f) + 1.0f, 2.0f) * top;
#else
                        // The forward reproduces the reference's normalise / de-normalise round trip of the tap index
                        // bit for bit (tap_index: it decides which float the tap lands on).  Here the index only
                        // splits a gradient between two neighbouring cells: the round trip moves it by <= 2 ulp, i.e.
                        // the two shares by ~1e-7 of the value, continuously (a tap within an ulp of a cell boundary
                        // has weight ~0 on the far side) -- the two correctly rounded divisions per tap (16 per lane,
                        // ~160 vector instructions) are not worth that.
                        ixm = cm;
                        ixp = cp;
#endif
                    }
                    const float thr = through[bar][k] * __builtin_amdgcn_rcpf((cp - cm) + 1e-12f) * inv_vox;
                    const float dfm = r_df[bar][0][k] - thr;
                    const float dfp = r_df[bar][1][k] + thr;
                    deposit(ixm, dfm);
                    deposit(ixp, dfp);
                }
                // spread over the 2x2 perpendicular corners (axes pb < pc are the two axes != axis)
                const int pb = axis == 0 ? 1 : 0, pc = axis == 2 ? 1 : 2;
                const int o_lo = (bar == 1 && h == 1) ? 3 : 0, o_hi = (bar == 1 && h == 0) ? 3 : 6;
                // window strides of the three grid axes (1 channel) and this lane's base cell in its window
                const int st[3] = {w.wd[1] * w.wd[2], w.wd[2], 1};
                const int sA = axis == 0 ? st[0] : (axis == 1 ? st[1] : st[2]);
                const int sB = pb == 0 ? st[0] : st[1], sC = pc == 1 ? st[1] : st[2];
                const int gst[3] = {gdims[1] * gdims[2], gdims[2], 1};                 // the same for the grid itself
                const int gA = axis == 0 ? gst[0] : (axis == 1 ? gst[1] : gst[2]);
                const int gB = pb == 0 ? gst[0] : gst[1], gC = pc == 1 ? gst[1] : gst[2];
                const int cell0 = ((i0c[0] - w.lo[0]) * w.wd[1] + (i0c[1] - w.lo[1])) * w.wd[2] + (i0c[2] - w.lo[2]);
#pragma unroll
                for (int o = 0; o < 6; ++o) {
                    if (o < o_lo || o >= o_hi || acc6[o] == 0.f) continue;
                    const int cA = iA - 2 + o;
                    if (cA < 0 || cA >= dimA) continue;
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            const float wgt = (pb == 0 ? wc[0][b] : wc[1][b]) * (pc == 1 ? wc[1][c] : wc[2][c]);
                            const int qb = (pb == 0 ? i0c[0] : i0c[1]) + b, qc = (pc == 1 ? i0c[1] : i0c[2]) + c;
                            const bool inb = (qb < (pb == 0 ? gdims[0] : gdims[1])) & (qc < (pc == 1 ? gdims[1] : gdims[2]));
                            if (inb && wgt != 0.f)
                                cell_add(w, cell0 + ((o - 2) * sA + b * sB + c * sC), P.grad_sdf,
                                         (int64_t)cA * gA + (int64_t)qb * gB + (int64_t)qc * gC, acc6[o] * wgt);
                        }
                }
            }
        }
        // ---- the exact interpolant's gradient (esr_expgrad_bwd's scatter, lts.hip, folded in: the LTS stages ran it as
        // launches of their own over the same samples, 8 scattered float atomics per sample): value weights
        // w_x w_y w_z and derivative weights +-w_y w_z ... on the 8 corners of the UNCLAMPED base cell, corners
        // border-replicated (or dropped: g4_zero_pad).  A clamped corner is the sample's clamped base cell or that + 1:
        // inside the window.  Lane half h takes the corners with cx = h.
        if (valid && (P.g4 || P.g4_self)) {
            float gd[3] = {r_g4.y, r_g4.z, r_g4.w}, wq[3][2];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float fl = floorf(ind[a]);
                wq[a][0] = (fl + 1.f) - ind[a];
                wq[a][1] = ind[a] - fl;
                gd[a] *= (float)(gdims[a] - 1) / (sc.xyz_max[a] - sc.xyz_min[a]);
            }
            const float gv = P.g4_self ? r_g4.x + (r_dsdf) : r_g4.x;
            const int cell0 = ((i0c[0] - w.lo[0]) * w.wd[1] + (i0c[1] - w.lo[1])) * w.wd[2] + (i0c[2] - w.lo[2]);
            const int cx = h;
#pragma unroll
            for (int cy = 0; cy < 2; ++cy)
#pragma unroll
                for (int cz = 0; cz < 2; ++cz) {
                    const int ux = i0[0] + cx, uy = i0[1] + cy, uz = i0[2] + cz;
                    const int x = min(max(ux, 0), gdims[0] - 1), y = min(max(uy, 0), gdims[1] - 1), z = min(max(uz, 0), gdims[2] - 1);
                    const bool inb = (ux == x) & (uy == y) & (uz == z);
                    const float sx = cx ? 1.f : -1.f, sy = cy ? 1.f : -1.f, sz = cz ? 1.f : -1.f;
                    const float wx = cx ? wq[0][1] : wq[0][0];
                    const float c0 = wx * wq[1][cy] * wq[2][cz];
                    const float c1 = sx * wq[1][cy] * wq[2][cz], c2 = wx * sy * wq[2][cz], c3 = wx * wq[1][cy] * sz;
                    const float t = gv * c0 + gd[0] * c1 + gd[1] * c2 + gd[2] * c3;
                    if (t != 0.f && !(P.g4_zero_pad && !inb))
                        cell_add(w, cell0 + ((x - i0c[0]) * w.wd[1] + (y - i0c[1])) * w.wd[2] + (z - i0c[2]), P.grad_sdf,
                                 ((int64_t)x * gdims[1] + y) * gdims[2] + z, t);
                }
        }
        lds_fence();
        windows_flush(SG, w, 1, lds, P.grad_sdf, gdims, lane);
        lds_fence();
        }
        // ---- phase 2: colour grids, one pass per net that read a colour group (3 channels per lane half)
        for (int k = 0; k < P.n_src; ++k) {
            float *gcol = on_tile ? P.gcol_on[k] : P.gcol_off[k];
            if (!gcol || t < P.src_t0[k] || t >= P.src_t1[k]) continue;       // wave-uniform
            const float *dXt = P.dX[k] + (size_t)t * DXROWS * 32 + s;
            const LaneWin w = seg_window(SG, valid, gdims, 0, 1, 6, lds, s, h);
            windows_zero(w, lds, lane);
            lds_fence();
            if (valid) {
                float d3[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) d3[c] = nact == 1 ? r_d3[c] : dXt[(ROW_COL + 3 * h + c) * 32];
                Tri tr = esr_tri_setup(ind);
                // (an in-grid corner of a sample is its clamped base cell or that cell + 1, also for points outside the box)
                const int cell0 = (((tr.i0[0] - w.lo[0]) * w.wd[1] + (tr.i0[1] - w.lo[1])) * w.wd[2] + (tr.i0[2] - w.lo[2])) * 6 + 3 * h;
                const int sx = w.wd[1] * w.wd[2] * 6, sy = w.wd[2] * 6;
#pragma unroll
                for (int cx = 0; cx < 2; ++cx)
#pragma unroll
                    for (int cy = 0; cy < 2; ++cy)
#pragma unroll
                        for (int cz = 0; cz < 2; ++cz) {
                            int x = tr.i0[0] + cx, y = tr.i0[1] + cy, z = tr.i0[2] + cz;
                            bool inb = (x >= 0) & (x < gdims[0]) & (y >= 0) & (y < gdims[1]) & (z >= 0) & (z < gdims[2]);
                            float wgt = esr_corner_w(tr, ind, cx, cy, cz);
                            if (inb && wgt != 0.f) {
#pragma unroll
                                for (int c = 0; c < 3; ++c)
                                    cell_add(w, cell0 + (cx * sx + cy * sy + cz * 6 + c), gcol,
                                             (((int64_t)x * gdims[1] + y) * gdims[2] + z) * 6 + 3 * h + c, d3[c] * wgt);
                            }
                        }
            }
            lds_fence();
            windows_flush(SG, w, 6, lds, gcol, gdims, lane);
            lds_fence();
        }
        fetch_ray();                                     // the next tile's ray (its record arrived long ago)
    }
}

}  // namespace

static int feat_common(const esr_scene_t *scene, const esr_feat_args_t *a, FeatParams &P)
{
    if (!scene || !a || a->tiles_all < 0 || a->tiles_on < 0 || a->tiles_on > a->tiles_all) return ESR_EINVAL;
    if (a->tiles_all == 0) return 0;
    if (!a->sdf) return ESR_EINVAL;
    if (a->pts) {
        if (!a->pt_viewdirs || !a->pt_sdf || a->n_pts < 0 || a->n_pts > a->tiles_all * 32) return ESR_EINVAL;
    } else if (!a->rays_o || !a->rays_d || !a->viewdirs || !a->rec_ray || !a->rec_step || !a->rec_sdf) {
        return ESR_EINVAL;
    }
    P.sc = *scene; P.rays_o = a->rays_o; P.rays_d = a->rays_d; P.viewdirs = a->viewdirs;
    P.rec_ray = a->rec_ray; P.rec_step = a->rec_step; P.rec_sdf = a->rec_sdf;
    P.pts = a->pts; P.pt_viewdirs = a->pt_viewdirs; P.pt_sdf = a->pt_sdf; P.n_pts = a->n_pts;
    P.sdf = a->sdf; P.tiles_on = a->tiles_on; P.tiles_all = a->tiles_all;
    for (int g = 0; g < 3; ++g) { P.color_on[g] = a->color_on[g]; P.color_off[g] = a->color_off[g]; }
    return 1;
}

ESR_API int esr_fine_feat_fwd(const esr_scene_t *scene, const esr_feat_args_t *args, float *X, float *gnorm,
                              void *stream)
{
    FeatParams P = {};
    const int c = feat_common(scene, args, P);
    if (c <= 0) return c;
    if (!X || !gnorm) return ESR_EINVAL;
    P.X = X; P.gnorm = gnorm;
    bool bar = true;                                    // every stencil radius within the bar's reach?
    for (int k = 0; k < 4; ++k) bar = bar && scene->grad_feat[k] >= 0.f && scene->grad_feat[k] <= BAR_MAX_DISP;
#ifdef ESR_FEAT_DIRECT
    bar = false;                                        // developer build: the direct form, for bit comparisons
#endif
    const int grid = esr_grid_for((int64_t)P.tiles_all * 32, 256, 256 * 16);
    if (bar) feat_fwd_kernel<true><<<grid, 256, 0, esr_stream(stream)>>>(P);
    else feat_fwd_kernel<false><<<grid, 256, 0, esr_stream(stream)>>>(P);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int64_t esr_fine_feat_x16_bytes(int32_t n_tiles) { return n_tiles < 0 ? (int64_t)ESR_EINVAL : (int64_t)n_tiles * X16_QUADS * 256; }

// The bf16 engine's form: the input tile as bf16 in the row-quad layout (comment above feat_fwd_kernel); X receives the
// normal rows 31..42 only.  Stencil radii beyond the bars' reach: ESR_ECAP (use esr_fine_feat_fwd).
ESR_API int esr_fine_feat_fwd_x16(const esr_scene_t *scene, const esr_feat_args_t *args, float *X, float *gnorm, void *X16,
                                  void *stream)
{
    FeatParams P = {};
    const int c = feat_common(scene, args, P);
    if (c <= 0) return c;
    if (!X || !gnorm || !X16) return ESR_EINVAL;
    for (int k = 0; k < 4; ++k)
        if (!(scene->grad_feat[k] >= 0.f && scene->grad_feat[k] <= BAR_MAX_DISP)) return ESR_ECAP;
    P.X = X; P.gnorm = gnorm; P.X16 = X16;
    const int grid = esr_grid_for((int64_t)P.tiles_all * 32, 256, 256 * 16);
    feat_fwd16_kernel<<<grid, 256, 0, esr_stream(stream)>>>(P);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_fine_feat_bwd(const esr_scene_t *scene, const esr_feat_args_t *args, const float *X,
                              const float *gnorm, const esr_feat_bwd_src_t *src, int32_t n_src,
                              const float *dsdf_extra, float *grad_sdf, float *dsdf_out, const float *grad4,
                              int32_t grad4_mode, void *stream)
{
    FeatParams P = {};
    const int c = feat_common(scene, args, P);
    if (c <= 0) return c;
    if (!X || !gnorm || !src || n_src < 1 || n_src > MAX_SRC) return ESR_EINVAL;
    P.X = const_cast<float *>(X); P.gnorm = const_cast<float *>(gnorm);
    P.n_src = n_src;
    for (int k = 0; k < n_src; ++k) {
        if (!src[k].dX) return ESR_EINVAL;
        P.dX[k] = src[k].dX; P.gcol_on[k] = src[k].grad_color_on; P.gcol_off[k] = src[k].grad_color_off;
        P.src_t0[k] = src[k].t0; P.src_t1[k] = src[k].t1;
    }
    P.dsdf_extra = dsdf_extra; P.dsdf_out = dsdf_out; P.grad_sdf = grad_sdf;
    if ((grad4 || (grad4_mode & 2)) && !grad_sdf) return ESR_EINVAL;
    P.g4 = grad4; P.g4_zero_pad = grad4_mode & 1; P.g4_self = (grad4_mode & 2) ? 1 : 0;
    // the scatter reduces a sample's stencil onto three 6-cell bars (and sizes its LDS windows for them): radii beyond
    // 2 voxels would fall off the bars -- refused rather than dropped (the forward has a direct form for any radius)
    if (grad_sdf)
        for (int k = 0; k < 4; ++k)
            if (!(scene->grad_feat[k] >= 0.f && scene->grad_feat[k] <= BAR_MAX_DISP)) return ESR_ECAP;
    // tiles outside every source's range receive no gradient: the launch covers the sources' window only (the fine
    // engine scatters the on-tiles while the off net's input gradients are still being computed)
    P.t_begin = 0; P.t_end = P.tiles_all;
    if (!dsdf_extra && !dsdf_out && !grad4) {
        int lo = P.tiles_all, hi = 0;
        for (int k = 0; k < n_src; ++k) {
            lo = src[k].t0 < lo ? src[k].t0 : lo;
            hi = src[k].t1 > hi ? src[k].t1 : hi;
        }
        P.t_begin = lo < 0 ? 0 : lo;
        P.t_end = hi > P.tiles_all ? P.tiles_all : hi;
        if (P.t_end <= P.t_begin) return 0;
    }
    // one wave per tile, 4 waves (4 x 20 KB LDS windows) per workgroup
    feat_bwd_kernel<<<esr_grid_for((int64_t)(P.t_end - P.t_begin) * 64, 256, 256 * 2), 256,
                      4 * WIN_CELLS * sizeof(double), esr_stream(stream)>>>(P);
    ESR_CHECK_LAUNCH();
    return 0;
}
