// Shared pieces of the tiny-MLP engine (mlp.hip: f32 matrix cores, mlp_bf16.hip: bf16 matrix cores):
// network descriptions, the X-row <-> reference-column maps, the packed-buffer layout, buffer-descriptor
// loads/stores and the accumulator-tile helpers.  See mlp.hip for the design notes.
#pragma once
#include "esr_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int MAX_HID_TILES = 6;      // widest hidden layer: 192 features

// feature index a (k-pair p, lane half h) register slot stands for in a hidden layer
__host__ __device__ constexpr int hid_feature(int p, int h)
{
    return 32 * (p >> 4) + ((p & 15) & 3) + 8 * ((p & 15) >> 2) + 4 * h;
}
// row of a 32x32 accumulator tile held by register r of lane half h
__host__ __device__ constexpr int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ---- network descriptions ---------------------------------------------------
struct NetDesc {
    int n_layers;           // linear layers (hidden + output)
    int in_dim, in_kp;      // reference input width, k-pairs of the first layer (multiple of 4)
    int xrows;              // rows of the input tile
    int out_dim;
    int hid_tiles;          // hidden width / 32
    int zrows;              // rows of the output / output-gradient tile (4 or 8)
    int cw;                 // rows of the colour group at the head of the input tile (re-targetable, see crow)
};
constexpr int XC_ROWS = 72; // coarse feature tile: 12 colour | 12 alt colour | normal3 | xyz3 sin15 cos15 | view 9 | pad
constexpr int X_ROWS = 104; // feature tile: 96 rows + a third colour group (rows 96-101)
__host__ __device__ constexpr NetDesc net_desc(int kind)
{
    return kind == ESR_MLP_RADIANCE ? NetDesc{4, 85, 44, X_ROWS, 3, 6, 4, 6}   // pbr/module.py:6-21
         : kind == ESR_MLP_TONEMAP  ? NetDesc{2, 33, 20, 48, 3, 6, 4, 6}       // pbr/module.py:24-39
         : kind == ESR_MLP_BRDF     ? NetDesc{4, 76, 40, X_ROWS, 5, 4, 8, 6}   // pbr/module.py:42-65
         : kind == ESR_MLP_EMIT     ? NetDesc{4, 76, 40, X_ROWS, 3, 4, 4, 6}   // EmissionNet, pbr/module.py:68-83
         :                            NetDesc{3, 57, 36, XC_ROWS, 3, 4, 4, 12}; // coarse rgbnet, voxurfc.py:134-149
}
// rows of the input tile that carry a gradient back to a grid (colour | sdf | feat24 | normal12 for the sample nets, the
// 33 inputs of the tone mapper, colour12 | colour12 | normal3 for the coarse net): the input-gradient pass writes dX rows
// 0 .. dx_rows-1 rounded up to 4 -- rows 0-31 as one 32x32 tile, the rest as 4-row v_mfma_f32_4x4x1 passes
__host__ __device__ constexpr int dx_rows(int kind)
{
    return kind == ESR_MLP_TONEMAP ? 33 : kind == ESR_MLP_COARSE ? 27 : 43;
}
__host__ __device__ constexpr int dx_pass4(int kind) { return dx_rows(kind) > 32 ? (dx_rows(kind) - 32 + 3) / 4 : 0; }
__host__ __device__ constexpr bool kind_ok(int kind) { return kind >= 0 && kind <= ESR_MLP_COARSE; }

// X-tile row -> column of the reference's first-layer weight (-1: no column)
__host__ __device__ inline int in_colmap(int kind, int row)
{
    if (kind == ESR_MLP_TONEMAP) return row < 33 ? row : -1;
    if (kind == ESR_MLP_COARSE) {
        // reference order (voxurfc.py:228-250): colour12 | xyz3 sin15 cos15 | vd3 sin3 cos3 | normal3
        if (row < 12) return row;
        if (row < 24) return -1;                 // the other net's colour group
        if (row < 27) return 54 + (row - 24);    // normal
        if (row < 30) return 12 + (row - 27);    // xyz
        if (row < 45) return 15 + (row - 30);    // sin
        if (row < 60) return 30 + (row - 45);    // cos
        if (row < 69) return 45 + (row - 60);    // viewdir, sin, cos
        return -1;
    }
    if (kind == ESR_MLP_BRDF || kind == ESR_MLP_EMIT) {
        // reference order (esrnerf.py:761-765): colour6 | xyz3 sin15 cos15 | sdf | feat24 | normal12
        if (row < 6) return row;
        if (row == 6) return 39;
        if (row < 31) return 40 + (row - 7);
        if (row < 43) return 64 + (row - 31);
        if (row < 46) return 6 + (row - 43);
        if (row < 61) return 9 + (row - 46);
        if (row < 76) return 24 + (row - 61);
        return -1;                           // no view-direction input
    }
    if (row < 6) return row;                 // colour
    if (row == 6) return 48;                 // sdf
    if (row < 31) return 49 + (row - 7);     // feat24
    if (row < 43) return 73 + (row - 31);    // normal12
    if (row < 46) return 6 + (row - 43);     // xyz
    if (row < 61) return 9 + (row - 46);     // sin
    if (row < 76) return 24 + (row - 61);    // cos
    if (row < 85) return 39 + (row - 76);    // viewdir, sin, cos
    return -1;
}

// Packed buffer layout (floats).  Forward part per layer l:
//   Wf_l [tiles_out][kp/4][64 lanes][4]   bias Bf_l [tiles_out][2 halves][16]
// Backward part per layer l: Wb_l [tiles_in][kpo/4][64][4]   (tiles_in: rows of the layer INPUT)
struct PackLayout {
    int n_layers;
    int kp[4], tiles_out[4], in_dim[4], out_dim[4];
    int kpo[4], tiles_in[4];
    int64_t off_wf[4], off_bf[4], off_wb[4];
    // output layer for the f32 forward as v_mfma_f32_4x4x1 operands (see lds4_layer): [pass][quad][2 halves][4 rows][4]
    // and its bias [8]; the 32x32 forms above stay for the bf16 engine's bias loads
    int n_pass4;
    int64_t off_w4, off_b4;
    // first layer, transposed, rows 32 .. 32 + 4 * n_passx - 1 of the input tile as 4x4x1 operands (see dx4_layer):
    // [pass][quad][2 halves][4 rows][4]
    int n_passx;
    int64_t off_wx4;
    int64_t total;
};
__host__ __device__ constexpr PackLayout pack_layout(int kind)
{
    const NetDesc d = net_desc(kind);
    PackLayout L = {};
    L.n_layers = d.n_layers;
    int64_t o = 0;
    for (int l = 0; l < d.n_layers; ++l) {
        const bool first = l == 0, last = l == d.n_layers - 1;
        const int hkp = 16 * d.hid_tiles, hid = 32 * d.hid_tiles;
        L.kp[l] = first ? d.in_kp : hkp;
        L.in_dim[l] = first ? d.in_dim : hid;
        L.out_dim[l] = last ? d.out_dim : hid;
        L.tiles_out[l] = last ? 1 : d.hid_tiles;
        L.kpo[l] = last ? 4 : hkp;                       // contraction pairs of the transposed product
        L.tiles_in[l] = first ? 2 : d.hid_tiles;         // dX: rows 0..63 only
        L.off_wf[l] = o; o += (int64_t)L.tiles_out[l] * L.kp[l] * 64;
        L.off_bf[l] = o; o += (int64_t)L.tiles_out[l] * 32;
        L.off_wb[l] = o; o += (int64_t)L.tiles_in[l] * L.kpo[l] * 64;
    }
    L.n_pass4 = d.zrows / 4;
    L.off_w4 = o; o += (int64_t)L.n_pass4 * d.hid_tiles * 4 * 32;
    L.off_b4 = o; o += 8;
    L.n_passx = dx_pass4(kind);
    L.off_wx4 = o; o += (int64_t)L.n_passx * d.hid_tiles * 4 * 32;
    L.total = o;
    return L;
}

struct PackArgs {
    int kind;
    const float *w[4], *b[4];
    float *out;
    __bf16 *out16;          // bf16 twin (pack16_body), or NULL
    _Float16 *outs;         // split-fp16 planes of the forward weights (packs_body; mlp_split.hip), or NULL
    unsigned *range;        // the split kernels' range flag (mlp_split.hip), or NULL: ORed with 1 when 64 w leaves fp16's range
};

// One element of the packed fp32 buffer.  KIND is a template argument so that the layout table is a set of immediates:
// with a run-time kind the table was built per thread and indexed dynamically -> 256 B of scratch per lane, 48 MB of
// scratch writes to pack 0.75 MB of weights (round 3's profiles: Scratch_Size 256 for pack_kernel).
template <int KIND>
__device__ __forceinline__ void pack_body(const PackArgs &A, int64_t e)
{
    constexpr PackLayout L = pack_layout(KIND);
    if (e >= L.off_wx4) {                                // first layer transposed, input rows >= 32, 4x4x1 operand order
        int64_t i = e - L.off_wx4;
        const int sub = i & 3, r = (i >> 2) & 3, h = (i >> 4) & 1;
        i >>= 5;
        constexpr int nq = L.in_dim[1] / 8;                      // quads of k-registers (hidden width / 2 registers)
        const int pass = (int)(i / nq), q = (int)(i % nq);
        const int col = in_colmap(KIND, 32 + 4 * pass + r), u = hid_feature(4 * q + sub, h);
        A.out[e] = (col >= 0 && col < L.in_dim[0]) ? A.w[0][(int64_t)u * L.in_dim[0] + col] : 0.f;
        return;
    }
    if (e >= L.off_w4) {                                 // output layer, 4x4x1 operand order
        constexpr int ll = L.n_layers - 1, hid = L.in_dim[ll], outd = L.out_dim[ll];
        float v = 0.f;
        if (e < L.off_b4) {
            int64_t i = e - L.off_w4;
            const int sub = i & 3, r = (i >> 2) & 3, h = (i >> 4) & 1;
            i >>= 5;
            constexpr int nq = hid / 8;                          // quads of k-registers per pass (hid/2 registers)
            const int pass = (int)(i / nq), q = (int)(i % nq);
            const int c = 4 * pass + r, u = hid_feature(4 * q + sub, h);
            if (c < outd) v = A.w[ll][(int64_t)c * hid + u];
        } else {
            const int c = (int)(e - L.off_b4);
            if (c < outd) v = A.b[ll][c];
        }
        A.out[e] = v;
        return;
    }
    float v = 0.f;
    bool done = false;
#pragma unroll
    for (int l = 0; l < L.n_layers; ++l) {               // (unrolled: every layer's constants are immediates)
        if (done || (l + 1 < L.n_layers && e >= L.off_wf[l + 1])) continue;
        done = true;
        const bool first = l == 0, last = l == L.n_layers - 1;
        const float *W = A.w[l];
        const int ind = L.in_dim[l], outd = L.out_dim[l];
        if (e < L.off_bf[l]) {                               // forward weights
            int64_t i = e - L.off_wf[l];
            const int sub = i & 3; i >>= 2;
            const int lane = i & 63; i >>= 6;
            const int q = (int)(i % (L.kp[l] / 4)), it = (int)(i / (L.kp[l] / 4));
            const int p = 4 * q + sub, h = lane >> 5;
            const int row = 32 * it + (lane & 31);
            const int col = first ? in_colmap(KIND, 2 * p + h) : hid_feature(p, h);
            if (row < outd && col >= 0 && col < ind) v = W[(int64_t)row * ind + col];
        } else if (e < L.off_wb[l]) {                        // bias in accumulator order
            int64_t i = e - L.off_bf[l];
            const int r = i & 15, h = (i >> 4) & 1, it = (int)(i >> 5);
            const int row = 32 * it + acc_row(r, h);
            if (row < outd) v = A.b[l][row];
        } else {                                             // transposed weights for dgrad
            int64_t i = e - L.off_wb[l];
            const int sub = i & 3; i >>= 2;
            const int lane = i & 63; i >>= 6;
            const int q = (int)(i % (L.kpo[l] / 4)), it = (int)(i / (L.kpo[l] / 4));
            const int p = 4 * q + sub, h = lane >> 5;
            const int orow = last ? (2 * p + h) : hid_feature(p, h);          // output feature of layer l
            const int irow = 32 * it + (lane & 31);                           // input feature / X row
            const int col = first ? in_colmap(KIND, irow) : irow;
            if (orow < outd && col >= 0 && col < ind) v = W[(int64_t)orow * ind + col];
        }
    }
    A.out[e] = v;
}

// ---- packed bf16 buffer (the bf16 engine, mlp_bf16.hip) ------------------------------------------------------------
// A 32x32x16 bf16 MFMA takes 8 consecutive k values per lane (k block = lane >> 5); k-step j = 2 * tile + jj, slot i of
// half h <-> feature kfeat16(j, h, i) (see mlp_bf16.hip).
__host__ __device__ constexpr int kfeat16(int j, int h, int i)
{
    return 32 * (j >> 1) + 16 * (j & 1) + 4 * h + (i & 3) + 8 * (i >> 2);
}
// Packed bf16 buffer (elements).  Forward part of layer l: [ks][tiles_out][64 lanes][8]; transposed part
// (dgrad): [kso][tiles_in][64][8].  Biases are read from the fp32 packed buffer of esr_mlp_pack.
struct Pack16Layout {
    int n_layers;
    int ks[4], tiles_out[4], kso[4], tiles_in[4], in_dim[4], out_dim[4];
    int64_t off_wf[4], off_wb[4];
    int64_t total;
};
__host__ __device__ constexpr Pack16Layout pack16_layout(int kind)
{
    const NetDesc d = net_desc(kind);
    Pack16Layout L = {};
    L.n_layers = d.n_layers;
    int64_t o = 0;
    for (int l = 0; l < d.n_layers; ++l) {
        const bool first = l == 0, last = l == d.n_layers - 1;
        const int hid = 32 * d.hid_tiles;
        L.ks[l] = first ? (2 * d.in_kp + 15) / 16 : hid / 16;
        L.in_dim[l] = first ? d.in_dim : hid;
        L.out_dim[l] = last ? d.out_dim : hid;
        L.tiles_out[l] = last ? 1 : d.hid_tiles;
        L.kso[l] = last ? 1 : hid / 16;
        L.tiles_in[l] = first ? 2 : d.hid_tiles;
        L.off_wf[l] = o; o += (int64_t)L.ks[l] * L.tiles_out[l] * 512;
        L.off_wb[l] = o; o += (int64_t)L.kso[l] * L.tiles_in[l] * 512;
    }
    L.total = o;
    return L;
}
template <int KIND>
__device__ __forceinline__ void pack16_body(const PackArgs &A, int64_t e)
{
    constexpr Pack16Layout L = pack16_layout(KIND);
    float v = 0.f;
    bool done = false;
#pragma unroll
    for (int l = 0; l < L.n_layers; ++l) {
        if (done || (l + 1 < L.n_layers && e >= L.off_wf[l + 1])) continue;
        done = true;
        const bool first = l == 0, last = l == L.n_layers - 1;
        const float *W = A.w[l];
        const int ind = L.in_dim[l], outd = L.out_dim[l];
        if (e < L.off_wb[l]) {                               // forward weights, order [k-step][out tile]
            int64_t i = e - L.off_wf[l];
            const int slot = i & 7; i >>= 3;
            const int lane = i & 63; i >>= 6;
            const int it = (int)(i % L.tiles_out[l]), j = (int)(i / L.tiles_out[l]);
            const int h = lane >> 5, row = 32 * it + (lane & 31);
            const int col = first ? in_colmap(KIND, 16 * j + 8 * h + slot) : kfeat16(j, h, slot);
            if (row < outd && col >= 0 && col < ind) v = W[(int64_t)row * ind + col];
        } else {                                             // transposed weights, order [k-step][in tile]
            int64_t i = e - L.off_wb[l];
            const int slot = i & 7; i >>= 3;
            const int lane = i & 63; i >>= 6;
            const int it = (int)(i % L.tiles_in[l]), j = (int)(i / L.tiles_in[l]);
            const int h = lane >> 5;
            const int orow = last ? (8 * h + slot) : kfeat16(j, h, slot);      // output feature of layer l
            const int irow = 32 * it + (lane & 31);                            // input feature / X row
            const int col = first ? in_colmap(KIND, irow) : irow;
            if (orow < outd && col >= 0 && col < ind) v = W[(int64_t)orow * ind + col];
        }
    }
    A.out16[e] = (__bf16)v;
}

// ---- split-fp16 planes (the f32 engine's forward on the 16-bit matrix cores, mlp_split.hip) ---------------------------------
// 64 w = w1 + w2 with w1 = fp16(64 w), w2 = fp16(64 w - w1): two fp16 planes carry 22 bits of the fp32 mantissa.  The
// power of two (SPLIT_W_SCALE) lifts the residual into fp16's NORMAL range for every |w| >= 2^-9 (unscaled it would be a
// subnormal below |w| = 0.125, i.e. for every weight of a 192-wide layer); both planes carry the SAME scale, so all three
// products of a k-step go into one accumulator and the kernels fold 1/64 into their bias / scale multiply.  Chunk order: the kernel stages ONE step = (layer, pair of output tiles) at a time, so a step's chunks are
// contiguous: [layer][pair][tile of the pair][plane][k-step], each chunk [64 lanes][8] as in the bf16 buffer (lane: output
// row 32 it + lane % 32, k block lane / 32; slot i: input feature kfeat16(j, h, i), first layer: in_colmap).
constexpr float SPLIT_W_SCALE = 64.f, SPLIT_W_INV = 1.f / 64.f;
struct SplitLayout {
    int n_layers;
    int ks[4], tiles_out[4], pairs[4], in_dim[4], out_dim[4];
    int off_chunk[4];          // first 1-KB chunk of the layer
    int total_chunks;
};
__host__ __device__ constexpr SplitLayout split_layout(int kind)
{
    const Pack16Layout P = pack16_layout(kind);
    SplitLayout L = {};
    L.n_layers = P.n_layers;
    int o = 0;
    for (int l = 0; l < P.n_layers; ++l) {
        L.ks[l] = P.ks[l]; L.tiles_out[l] = P.tiles_out[l]; L.pairs[l] = (P.tiles_out[l] + 1) / 2;
        L.in_dim[l] = P.in_dim[l]; L.out_dim[l] = P.out_dim[l];
        L.off_chunk[l] = o;
        o += P.tiles_out[l] * 2 * P.ks[l];
    }
    L.total_chunks = o;
    return L;
}
// The TRANSPOSED weights as split planes, for the input-gradient chain (mlp_split.hip: mlp_dgrad_split_kernel): "layer" q of
// this layout is the transpose of network layer NL-1-q (the chain runs output -> input); its k dimension is that layer's
// OUTPUT features (kfeat16 order; the output layer's 3 rows in slots 8 h + i of a single k-step), its rows the layer's
// INPUT features (X rows through in_colmap for the first layer: 2 tiles).  Appended to the forward planes.
__host__ __device__ constexpr SplitLayout split_layout_t(int kind)
{
    const Pack16Layout P = pack16_layout(kind);
    SplitLayout L = {};
    L.n_layers = P.n_layers;
    int o = 0;
    for (int q = 0; q < P.n_layers; ++q) {
        const int l = P.n_layers - 1 - q;
        L.ks[q] = P.kso[l]; L.tiles_out[q] = P.tiles_in[l]; L.pairs[q] = (P.tiles_in[l] + 1) / 2;
        L.in_dim[q] = P.in_dim[l]; L.out_dim[q] = P.out_dim[l];
        L.off_chunk[q] = o;
        o += P.tiles_in[l] * 2 * P.kso[l];
    }
    L.total_chunks = o;
    return L;
}
// A weight whose first plane fp16(64 w) is not finite (|w| >= 1023.5, inf, NaN) cannot be carried by the planes: the sticky
// range flag tells the host, which runs the step on the f32 MFMA kernels instead (fine_engine.py: same-step fallback).
__device__ __forceinline__ void split_weight_range(const PackArgs &A, float v64)
{
    if (A.range && !(fabsf(v64) < 65504.f)) atomicOr(A.range, 1u);
}
// The planes buffer ends with one fp32 slot: the net's GRADIENT GAIN BOUND (split_gain_kernel, mlp.hip), read by the split
// input-gradient kernel.  Offset in fp16 elements:
__host__ __device__ constexpr int64_t split_gain_offset(int kind)
{
    return ((int64_t)split_layout(kind).total_chunks + split_layout_t(kind).total_chunks) * 512;
}
constexpr int SPLIT_GAIN_PAD = 8;                       // fp16 elements reserved for the slot (16 bytes)
constexpr float SPLIT_GAIN_MAX = 262144.f;              // 2^18: beyond it the scaled chain would lose bits -> range flag
template <int KIND>
__device__ __forceinline__ void packst_body(const PackArgs &A, int64_t e)
{
    constexpr SplitLayout L = split_layout_t(KIND);
    constexpr int64_t BASE = (int64_t)split_layout(KIND).total_chunks * 512;
    float v = 0.f;
    int plane = 0;
    bool done = false;
#pragma unroll
    for (int q = 0; q < L.n_layers; ++q) {
        if (done || (q + 1 < L.n_layers && e >= (int64_t)L.off_chunk[q + 1] * 512)) continue;
        done = true;
        const int l = L.n_layers - 1 - q;
        const bool first = l == 0, last = l == L.n_layers - 1;
        int64_t i = e - (int64_t)L.off_chunk[q] * 512;
        const int slot = i & 7; i >>= 3;
        const int lane = i & 63; i >>= 6;
        const int j = (int)(i % L.ks[q]); i /= L.ks[q];
        plane = (int)(i & 1);
        const int it = (int)(i >> 1);
        const int h = lane >> 5;
        const int orow = last ? (8 * h + slot) : kfeat16(j, h, slot);          // output feature of network layer l
        const int irow = 32 * it + (lane & 31);                                // its input feature / X row
        const int col = first ? in_colmap(KIND, irow) : irow;
        if (orow < L.out_dim[q] && col >= 0 && col < L.in_dim[q]) v = A.w[l][(int64_t)orow * L.in_dim[q] + col];
    }
    v *= SPLIT_W_SCALE;
    split_weight_range(A, v);
    const _Float16 w1 = (_Float16)v;
    A.outs[BASE + e] = plane == 0 ? w1 : (_Float16)(v - (float)w1);
}

template <int KIND>
__device__ __forceinline__ void packs_body(const PackArgs &A, int64_t e)
{
    constexpr SplitLayout L = split_layout(KIND);
    float v = 0.f;
    int plane = 0;
    bool done = false;
#pragma unroll
    for (int l = 0; l < L.n_layers; ++l) {
        if (done || (l + 1 < L.n_layers && e >= (int64_t)L.off_chunk[l + 1] * 512)) continue;
        done = true;
        const bool first = l == 0;
        int64_t i = e - (int64_t)L.off_chunk[l] * 512;
        const int slot = i & 7; i >>= 3;
        const int lane = i & 63; i >>= 6;
        const int j = (int)(i % L.ks[l]); i /= L.ks[l];
        plane = (int)(i & 1);
        const int it = (int)(i >> 1);                                           // (pair, tile of the pair) = tile index
        const int h = lane >> 5, row = 32 * it + (lane & 31);
        const int col = first ? in_colmap(KIND, 16 * j + 8 * h + slot) : kfeat16(j, h, slot);
        if (row < L.out_dim[l] && col >= 0 && col < L.in_dim[l]) v = A.w[l][(int64_t)row * L.in_dim[l] + col];
    }
    v *= SPLIT_W_SCALE;
    split_weight_range(A, v);
    const _Float16 w1 = (_Float16)v;
    A.outs[e] = plane == 0 ? w1 : (_Float16)(v - (float)w1);
}

// Every net of a step in ONE launch (esr_mlp_pack_batch): blockIdx.y = job; fp32 elements first, then the bf16 twin's.
constexpr int MAX_PACK_JOBS = 8;
struct PackBatch {
    int n;
    PackArgs job[MAX_PACK_JOBS];
};
template <int KIND>
__device__ __forceinline__ void pack_job(const PackArgs &A)
{
    constexpr int64_t N32 = pack_layout(KIND).total, N16 = pack16_layout(KIND).total;
    constexpr int64_t NS = (int64_t)split_layout(KIND).total_chunks * 512, NST = (int64_t)split_layout_t(KIND).total_chunks * 512;
    const int64_t n = N32 + N16 + (A.outs ? NS + NST : 0);
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        if (e < N32) { if (A.out) pack_body<KIND>(A, e); }
        else if (e < N32 + N16) { if (A.out16) pack16_body<KIND>(A, e - N32); }
        else if (e < N32 + N16 + NS) packs_body<KIND>(A, e - N32 - N16);
        else packst_body<KIND>(A, e - N32 - N16 - NS);
    }
}
__global__ void __launch_bounds__(256) pack_kernel(PackBatch B)
{
    const PackArgs &A = B.job[blockIdx.y];
    switch (A.kind) {                                    // (block-uniform)
    case ESR_MLP_RADIANCE: pack_job<ESR_MLP_RADIANCE>(A); break;
    case ESR_MLP_TONEMAP:  pack_job<ESR_MLP_TONEMAP>(A); break;
    case ESR_MLP_BRDF:     pack_job<ESR_MLP_BRDF>(A); break;
    case ESR_MLP_EMIT:     pack_job<ESR_MLP_EMIT>(A); break;
    default:               pack_job<ESR_MLP_COARSE>(A); break;
    }
}

// ---- MFMA building blocks ------------------------------------------------------
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
// 16 independent 4x4 outer products per instruction: D_b[i][j] += A_b[i] * B_b[j] for block b = lane / 4, with A row i
// supplied by lane 4b+i, B column j by lane 4b+j, and D[i][j] in register i of lane 4b+j; 2 passes = 8 cycles, the same
// 64 FLOP/clk/SIMD as the 32x32x2 form (MI355X_MICROARCH.md, Matrix cores)
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
}

// acc[n / KP4] += (packed weight quad n) . B-quad (n % KP4), n = 0 .. NT*KP4-1.
// The weight stream (byte offset `woff` in the packed buffer) is explicitly
// double-buffered in groups of G 16-B loads (16 MFMAs = ~1k cycles of matrix work per
// group) with a scheduling barrier per group so the loads stay one group ahead.
template <int KP4, int NT, typename BF>
__device__ __forceinline__ void stream_layer(rsrc_t W, int woff, BF bget, f32x16 (&acc)[NT], int lane)
{
    constexpr int NTOT = NT * KP4, G = 4, NG = (NTOT + G - 1) / G;
    const int voff = lane * 16;
    float4 buf[2][G];
#pragma unroll
    for (int i = 0; i < G; ++i)
        if (i < NTOT) buf[0][i] = bload4(W, voff, woff + i * 1024);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int n = (g + 1) * G + i;
            if (n < NTOT) buf[(g + 1) & 1][i] = bload4(W, voff, woff + n * 1024);
        }
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int n = g * G + i;
            if (n < NTOT) {
                const int it = n / KP4, q = n % KP4;
                const float4 a = buf[g & 1][i];
                acc[it] = mfma32(a.x, bget(4 * q + 0), acc[it]);
                acc[it] = mfma32(a.y, bget(4 * q + 1), acc[it]);
                acc[it] = mfma32(a.z, bget(4 * q + 2), acc[it]);
                acc[it] = mfma32(a.w, bget(4 * q + 3), acc[it]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// acc[it] += Wp[it] . B      B given as KP per-lane registers
template <int KP, int NT>
__device__ __forceinline__ void layer_from_regs(rsrc_t W, int woff, const float (&B)[KP],
                                                f32x16 (&acc)[NT], int lane)
{
    static_assert(KP % 4 == 0, "k-pairs come in quads");
    stream_layer<KP / 4, NT>(W, woff, [&](int k) { return B[k]; }, acc, lane);
}

// acc[it] += Wp[it] . prev    prev = NP accumulator tiles of the previous layer (32*NP features)
template <int NP, int NT>
__device__ __forceinline__ void layer_from_acc(rsrc_t W, int woff, const f32x16 (&prev)[NP],
                                               f32x16 (&acc)[NT], int lane)
{
    stream_layer<NP * 4, NT>(W, woff, [&](int k) { return prev[k >> 4][k & 15]; }, acc, lane);
}

// The same stream with its FIRST group of weight quads already in registers (`pre`, issued by stream_prefetch).  The
// forward kernel issues that group -- and the next layer's bias -- right after the last MFMA of the previous layer,
// BEFORE that layer's epilogue (ReLU, mask words, ~100 stores): the L2 round trips run under the epilogue instead of
// after it, and because the loads are then OLDER than the epilogue's stores the counted vmcnt wait in front of the
// first MFMA does not drain those stores (vmcnt retires in order and counts stores on gfx950).
constexpr int STREAM_G = 4;
struct StreamPre { float4 q[STREAM_G]; };
template <int NTOT>
__device__ __forceinline__ StreamPre stream_prefetch(rsrc_t W, int woff, int lane)
{
    StreamPre P;
#pragma unroll
    for (int i = 0; i < STREAM_G; ++i) P.q[i] = (i < NTOT) ? bload4(W, lane * 16, woff + i * 1024) : make_float4(0.f, 0.f, 0.f, 0.f);
    return P;
}
template <int KP4, int NT, typename BF>
__device__ __forceinline__ void stream_layer_pre(rsrc_t W, int woff, const StreamPre &pre, BF bget, f32x16 (&acc)[NT], int lane)
{
    constexpr int NTOT = NT * KP4, G = STREAM_G, NG = (NTOT + G - 1) / G;
    const int voff = lane * 16;
    float4 buf[2][G];
#pragma unroll
    for (int i = 0; i < G; ++i) buf[0][i] = pre.q[i];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int n = (g + 1) * G + i;
            if (n < NTOT) buf[(g + 1) & 1][i] = bload4(W, voff, woff + n * 1024);
        }
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int n = g * G + i;
            if (n < NTOT) {
                const int it = n / KP4, q = n % KP4;
                const float4 a = buf[g & 1][i];
                acc[it] = mfma32(a.x, bget(4 * q + 0), acc[it]);
                acc[it] = mfma32(a.y, bget(4 * q + 1), acc[it]);
                acc[it] = mfma32(a.z, bget(4 * q + 2), acc[it]);
                acc[it] = mfma32(a.w, bget(4 * q + 3), acc[it]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Small-M products on v_mfma_f32_4x4x1 (16 independent 4x4 outer products per instruction, 8.5 cycles measured with 4
// accumulators in flight: tools/ubench/mfma4_loop.hip): rows [4 * pass, 4 * pass + 4) of   out[row][s] = sum_u Wt[row][u]
// * cur[u][s]   for the 32 samples of the tile.  Used for the output layer of the forward (3-5 rows) and for the input
// gradient's rows above 31 (1-11 rows): a 32x32 tile would spend 6144 cycles on them, a pass costs 96 x 8.5.
//   A operand: the weight of row 4 * pass + lane % 4 for THIS half's unit of register k -- 8 distinct 16-B pieces per quad
//   of registers, selected by (half, lane % 4); B operand: register k of `cur`.  Each half of the wave sums its own
//   16 * HT units; the caller adds the halves (one __shfl_xor 32 per row).
// The operands come from LDS (copied there once per workgroup by lds4_preload): a quad is only ~35 cycles of matrix
// work, so streaming them from L2 one group ahead -- as the 32x32 layers do with their 1 k cycles per group -- left the
// loop waiting on every group.
// x(lane) + x(lane ^ 32) in every lane: one v_permlane32_swap + one add (ds_bpermute, which __shfl_xor compiles to, cost
// ~500 cycles per row behind its lgkmcnt wait: 22 us of the radiance input-gradient kernel's 400 for 12 rows)
__device__ __forceinline__ float half_sum(float x)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// rows [0, 4 * NP) of a [rows][32] tile from the per-half partial sums of lds4_layer (+ optional per-row bias): both
// halves get the full sums, half 0 stores the even rows and half 1 the odd ones -- 2 * NP full-wave stores through the
// tile's buffer descriptor (row0: first row)
template <int NP, bool BIAS, bool NT>
__device__ __forceinline__ void store_rows4(rsrc_t T, int row0, const f32x4 (&z)[NP], const float *bias, int lane)
{
    const int h = lane >> 5;
    const int voff = ((row0 + h) * 32 + (lane & 31)) * 4;
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int c = 0; c < 4; c += 2) {
            float v0 = half_sum(z[p][c]), v1 = half_sum(z[p][c + 1]);
            if (BIAS) { v0 += bias[4 * p + c]; v1 += bias[4 * p + c + 1]; }
            if (NT) bstore1_nt(T, h ? v1 : v0, voff, (4 * p + c) * 128);
            else bstore1(T, h ? v1 : v0, voff, (4 * p + c) * 128);
        }
}

template <int NF4>
__device__ __forceinline__ void lds4_preload(rsrc_t W, int woff, float4 *lds)
{
    for (int i = threadIdx.x; i < NF4; i += blockDim.x) lds[i] = bload4(W, i * 16, woff);
}
template <int HT, int NP>
__device__ __forceinline__ void lds4_layer(const float4 *lds, const f32x16 (&cur)[HT], f32x4 (&z)[NP], int lane)
{
    // explicitly double-buffered in groups of G LDS reads (left to itself the compiler issues one read, waits for it and
    // runs its 4 products: ~120 cycles of LDS latency per 34 cycles of matrix work)
    constexpr int NQ = HT * 4, NTOT = NQ * NP, G = 6, NG = (NTOT + G - 1) / G;
    const float4 *mine = lds + ((lane >> 5) * 4 + (lane & 3));
    f32x4 acc[NP][4];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[p][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 buf[2][G];
    // flat index n = q * NP + p: consecutive products go to different passes' accumulators
#pragma unroll
    for (int i = 0; i < G; ++i)
        if (i < NTOT) buf[0][i] = mine[((i % NP) * NQ + i / NP) * 8];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int n = (g + 1) * G + i;
            if (n < NTOT) buf[(g + 1) & 1][i] = mine[((n % NP) * NQ + n / NP) * 8];
        }
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int n = g * G + i;
            if (n < NTOT) {
                const int p = n % NP, q = n / NP;
                const float4 a = buf[g & 1][i];
                acc[p][0] = mfma4(a.x, cur[(4 * q + 0) >> 4][(4 * q + 0) & 15], acc[p][0]);
                acc[p][1] = mfma4(a.y, cur[(4 * q + 1) >> 4][(4 * q + 1) & 15], acc[p][1]);
                acc[p][2] = mfma4(a.z, cur[(4 * q + 2) >> 4][(4 * q + 2) & 15], acc[p][2]);
                acc[p][3] = mfma4(a.w, cur[(4 * q + 3) >> 4][(4 * q + 3) & 15], acc[p][3]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) z[p] = (acc[p][0] + acc[p][1]) + (acc[p][2] + acc[p][3]);
}

// bias (packed in accumulator order at byte offset boff)
template <int NT>
__device__ __forceinline__ void load_bias(rsrc_t W, int boff, f32x16 (&acc)[NT], int lane)
{
    const int voff = (lane >> 5) * 64;
#pragma unroll
    for (int it = 0; it < NT; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = bload4(W, voff, boff + it * 128 + q * 16);
            acc[it][4 * q + 0] = v.x; acc[it][4 * q + 1] = v.y;
            acc[it][4 * q + 2] = v.z; acc[it][4 * q + 3] = v.w;
        }
    }
}

// ReLU of accumulator tiles as ONE integer instruction per element: max((int)bits, 0) -- non-negative floats are
// non-negative integers in the same order, every negative float (and -0) has the sign bit and becomes +0.  fmaxf on an
// MFMA result compiles to a canonicalising max + the max (two instructions); rounds 1-3 used an inline-asm v_max_f32
// instead, which hipcc's hazard recogniser cannot see into: nothing inserted the wait states a vector instruction needs
// behind the MFMA that wrote its operand, and round 3's bf16 tone-mapper kernel read an accumulator before its (single,
// 8-pass) MFMA had written it.  The integer form is compiler-visible: the dependency on the MFMA is tracked, the hazard
// nops are inserted where a schedule needs them (tests/test_isa.py checks the generated code: no hand-written vector
// instruction reads an MFMA result, and the ReLU is one v_max_i32 per element), and nothing is canonicalised.
template <int NT>
__device__ __forceinline__ void relu_tiles(f32x16 (&acc)[NT])
{
#pragma unroll
    for (int it = 0; it < NT; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int b = __float_as_int(acc[it][r]);
            acc[it][r] = __int_as_float(b > 0 ? b : 0);
        }
}

// per-lane byte offset inside a tile-major tile: row 4h, sample s
__device__ __forceinline__ int tile_voff(int lane) { return ((lane >> 5) * 4 * 32 + (lane & 31)) * 4; }
// scalar byte offset of accumulator register r of tile `it` (row = 32it + (r&3) + 8(r>>2))
__host__ __device__ constexpr int tile_soff(int it, int r) { return (32 * it + (r & 3) + 8 * (r >> 2)) * 128; }

// tile-major store of NT accumulator tiles through descriptor T (based at the tile).  STREAM: non-temporal (tiles a
// much later kernel reads); otherwise default policy (tiles the NEXT kernel reads: they stay in L2 / the memory-side cache)
template <int NT, bool STREAM = true>
__device__ __forceinline__ void store_tiles(rsrc_t T, const f32x16 (&acc)[NT], int lane)
{
    const int voff = tile_voff(lane);
#pragma unroll
    for (int it = 0; it < NT; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (STREAM) bstore1_nt(T, acc[it][r], voff, tile_soff(it, r));
            else bstore1(T, acc[it][r], voff, tile_soff(it, r));
        }
}

// acc = (saved activation > 0) ? acc : 0
template <int NT>
__device__ __forceinline__ void mask_by_saved(rsrc_t T, f32x16 (&acc)[NT], int lane)
{
    const int voff = tile_voff(lane);
#pragma unroll
    for (int it = 0; it < NT; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float a = bload1(T, voff, tile_soff(it, r));
            acc[it][r] = a > 0.f ? acc[it][r] : 0.f;
        }
}

// ReLU sign bits of NT accumulator tiles, 16 bits per tile, two tiles per word.  The backward
// needs only (h > 0): 3 words per lane instead of re-reading 96 floats (24 KB per tile).
template <int NT>
__device__ __forceinline__ void store_relu_mask(rsrc_t T, const f32x16 (&acc)[NT], int lane)
{
    static_assert(NT % 2 == 0, "two tiles per mask word");
#pragma unroll
    for (int wd = 0; wd < NT / 2; ++wd) {
        // (x > 0) as an integer clamp of the float's bits to [0, 1] (v_med3_i32: +0, -0 and negatives give 0) shifted
        // into place with v_lshl_or_b32: two VALU instructions per element and no VCC round trip (the compare /
        // select / or form compiled to ~3.5 instructions plus hazard nops)
        unsigned m = 0;
#pragma unroll
        for (int b = 0; b < 32; ++b) {
            int one;                     // (asm: hipcc folds the C form back into compare + select)
            asm("v_med3_i32 %0, %1, 0, 1" : "=v"(one) : "v"(__float_as_int(acc[2 * wd + (b >> 4)][b & 15])));
            m |= (unsigned)one << b;
        }
        __builtin_amdgcn_raw_buffer_store_b32(m, T, lane * 4, wd * 256, 0);
    }
}
template <int NT>
__device__ __forceinline__ void load_relu_mask(rsrc_t T, unsigned (&m)[NT / 2], int lane)
{
#pragma unroll
    for (int wd = 0; wd < NT / 2; ++wd) m[wd] = __builtin_amdgcn_raw_buffer_load_b32(T, lane * 4, wd * 256, 0);
}
template <int NT>
__device__ __forceinline__ void apply_relu_mask(const unsigned (&m)[NT / 2], f32x16 (&acc)[NT])
{
#pragma unroll
    for (int it = 0; it < NT; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // sign-extended one-bit field (v_bfe_i32: 0 or -1) ANDed onto the value: two instructions, no VCC
            const int keep = ((int)(m[it >> 1] << (31 - ((it & 1) * 16 + r)))) >> 31;
            acc[it][r] = __int_as_float(__float_as_int(acc[it][r]) & keep);
        }
}

template <int NT>
__device__ __forceinline__ void zero_tiles(f32x16 (&acc)[NT])
{
#pragma unroll
    for (int it = 0; it < NT; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[it][r] = 0.f;
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// bf16 saved tiles (H, dZ of the bf16 engine), ROW-QUAD layout: [row / 4][32 sample slots][4 rows] -- 8 bytes per (quad, sample).
// Registers 4q .. 4q+3 of an accumulator tile are four CONSECUTIVE rows (acc_row), so a lane stores a tile with 4 eight-byte
// stores (512 contiguous bytes per wave instruction) instead of 16 two-byte ones: the two-byte form made the bf16 forward /
// input-gradient kernels store-issue bound (tools/ubench/fwd16_stamps.hip: 6.3 k of a layer step's ~9 k cycles in the
// epilogue).  The weight-gradient kernel's stager transposes 64-byte pieces back to rows on its way to LDS (mlp.hip).
template <int NT>
__device__ __forceinline__ void store_tiles_bf16(rsrc_t T, const f32x16 (&acc)[NT], int lane)
{
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    // lane = 32 h + s: quad 2q + h of tile it.  Inside a quad's 256 bytes the 32 sample slots are ordered so that the
    // weight-gradient stager's four 16-B loads (samples 8m+2k, 8m+2k+1 for k = 0..3 of its 8-sample block m) are each a
    // full 64-byte run over four neighbouring threads: slot(s) = 8 ((s >> 1) & 3) + 2 (s >> 3) + (s & 1)
    const int s_ = lane & 31;
    const int voff = ((lane >> 5) * 32 + 8 * ((s_ >> 1) & 3) + 2 * (s_ >> 3) + (s_ & 1)) * 8;
#pragma unroll
    for (int it = 0; it < NT; ++it)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
            bf16x2 lo, hi;
            lo[0] = (__bf16)acc[it][4 * q + 0]; lo[1] = (__bf16)acc[it][4 * q + 1];
            hi[0] = (__bf16)acc[it][4 * q + 2]; hi[1] = (__bf16)acc[it][4 * q + 3];
            u32x2 v;
            v[0] = __builtin_bit_cast(unsigned, lo); v[1] = __builtin_bit_cast(unsigned, hi);
            __builtin_amdgcn_raw_buffer_store_b64(v, T, voff, (8 * it + 2 * q) * 256, ESR_NT_AUX);
        }
}

// the same tile store from ALREADY packed values: hb[2 * it + jj] = registers 8 jj .. 8 jj + 7 of tile `it` rounded to bf16
// (the next layer's B operands); quad q of the tile = elements 4 (q & 1) .. + 3 of hb[2 * it + (q >> 1)]
template <int NT>
__device__ __forceinline__ void store_tiles_bf16_packed(rsrc_t T, const bf16x8 (&hb)[2 * NT], int lane)
{
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    const int s_ = lane & 31;
    const int voff = ((lane >> 5) * 32 + 8 * ((s_ >> 1) & 3) + 2 * (s_ >> 3) + (s_ & 1)) * 8;
#pragma unroll
    for (int it = 0; it < NT; ++it)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const u32x4_ w = __builtin_bit_cast(u32x4_, hb[2 * it + (q >> 1)]);
            u32x2 v;
            v[0] = w[2 * (q & 1)]; v[1] = w[2 * (q & 1) + 1];
            __builtin_amdgcn_raw_buffer_store_b64(v, T, voff, (8 * it + 2 * q) * 256, ESR_NT_AUX);
        }
}

__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8 pack8(float4 lo, float4 hi)
{
    bf16x8 v;
    v[0] = (__bf16)lo.x; v[1] = (__bf16)lo.y; v[2] = (__bf16)lo.z; v[3] = (__bf16)lo.w;
    v[4] = (__bf16)hi.x; v[5] = (__bf16)hi.y; v[6] = (__bf16)hi.z; v[7] = (__bf16)hi.w;
    return v;
}

}  // namespace
