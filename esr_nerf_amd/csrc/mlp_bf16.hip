// Tiny-MLP engine on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16) -- the build-side precision choice of
// BASELINE.json's bf16 configurations (C3, C5; the reference itself is fp32 everywhere).
//
// Same networks, same "transposed" chain as mlp.hip (one wave owns 32 samples end to end, the accumulator of
// layer l is the B operand of layer l+1; here eight waves of a workgroup share each layer's weights through LDS, see
// Shared16 below), with bf16 OPERANDS and fp32 ACCUMULATION: weights are rounded to bf16
// by esr_mlp_pack_bf16, activations are rounded when they become an MFMA operand (v_cvt_pk_bf16_f32); biases and
// accumulators are fp32; the tiles saved for the backward (hidden activations H, hidden gradients dZ) are stored as
// bf16 in a row-quad order (mlp_common.h: store_tiles_bf16) -- they are only ever read back as bf16 operands of the
// weight-gradient kernel, and they are most of this engine's HBM traffic; the network inputs X, outputs z / dz and the input gradient dX stay fp32, so the feature and
// shading kernels of the fp32 path are shared unchanged.
//
// A 32x32x16 bf16 MFMA takes 8 consecutive k values per lane (k block = lane >> 5).  Accumulator register r
// of lane-half h holds row (r & 3) + 8 (r >> 2) + 4 h, so the 16 rows 16 jj .. 16 jj + 15 of a 32-row tile are
// exactly registers 8 jj .. 8 jj + 7 of both halves: k-step j = 2 * tile + jj, slot i of half h <-> feature
// 32 (j >> 1) + 16 (j & 1) + 4 h + (i & 3) + 8 (i >> 2).  That permutation is folded into the weight packing.
//
// Matrix time drops 16x against the f32 cores, so these kernels are bound by everything else: the epilogues (ReLU, masks,
// stores of the saved tiles), the X loads and the barriers between layer steps (tools/ubench/fwd16_stamps.hip), not by
// the MFMA pipe.
#include "mlp_common.h"

#include <type_traits>

// in-kernel time stamps for tools/ubench/fwd16_stamps.hip (nothing in the product build)
#ifndef ESR_STAMP16
#define ESR_STAMP16(i)
#endif

namespace {

// ReLU masks of the bf16 engine: one u32 per lane and tile PAIR, written by the forward from packed bf16 pairs -- register r
// of tile `it` is bit 8 (it & 1) + (r >> 1) + 16 (r & 1) of word it / 2 (the f32 engine's order is 16 (it & 1) + r)
__host__ __device__ constexpr int mask_bit16(int it, int r) { return 8 * (it & 1) + (r >> 1) + 16 * (r & 1); }
template <int NT>
__device__ __forceinline__ void apply_relu_mask16(const unsigned (&m)[NT / 2], f32x16 (&acc)[NT])
{
#pragma unroll
    for (int it = 0; it < NT; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // sign-extended one-bit field (v_bfe_i32: 0 or -1) ANDed onto the value: two instructions, no VCC
            const int keep = ((int)(m[it >> 1] << (31 - mask_bit16(it, r)))) >> 31;
            acc[it][r] = __int_as_float(__float_as_int(acc[it][r]) & keep);
        }
}

// registers 8 jj .. 8 jj + 7 of an accumulator tile, rounded to bf16: the B operand of k-step 2 * tile + jj
__device__ __forceinline__ bf16x8 acc_to_b(const f32x16 &t, int jj)
{
    bf16x8 b;
#pragma unroll
    for (int i = 0; i < 8; ++i) b[i] = (__bf16)t[8 * jj + i];
    return b;
}

struct Fwd16Args {
    const float *packed32;     // esr_mlp_pack buffer (biases)
    const __bf16 *packed16;
    const float *X;
    int t0, t1;
    float *H[3];
    unsigned *M[3];
    int save, crow;
    float *zout;
    // radiance nets: the input tile as bf16 in the row-quad layout (feat.hip: feat_fwd16_kernel; 26 quads x 256 B per tile,
    // quads 24 / 25 = the first eight rows with the colour group of rows 88..93) -- then X is not read
    const void *X16;
};
// Several passes of the same net KIND in ONE launch (esr_mlp_fwd_fine_bf16: the fine stage's three radiance forward passes,
// esr_mlp_dgrad_fine_bf16: its two input-gradient passes): a workgroup works on exactly one segment (its weights stay in
// LDS), segments get a share of the launch's workgroups proportional to their tiles.  X, H, M, dZ, dX are shared arrays.
constexpr int MAX_SEG16 = 3;
struct Seg16 {
    const float *packed32;
    const __bf16 *packed16;
    int t0, t1, save, crow;
    float *zout;
    int b0, nb;                // workgroups [b0, b0 + nb) of the launch
};
struct Fwd16Batch {
    Fwd16Args base;            // X, H, M (and, for nseg == 0, the single pass)
    int skew;                  // see layer_skew()
    int nseg;
    Seg16 seg[MAX_SEG16];
};
__device__ __forceinline__ Fwd16Args pick_seg16(const Fwd16Batch &B, int &b0, int &nb)
{
    Fwd16Args A = B.base;
    b0 = 0; nb = (int)gridDim.x;
#pragma unroll
    for (int k = 0; k < MAX_SEG16; ++k)
        if (k < B.nseg && (int)blockIdx.x >= B.seg[k].b0) {
            const Seg16 &S = B.seg[k];
            A.packed32 = S.packed32; A.packed16 = S.packed16; A.t0 = S.t0; A.t1 = S.t1; A.save = S.save; A.crow = S.crow;
            A.zout = S.zout; b0 = S.b0; nb = S.nb;
        }
    return A;
}

// ---- forward / input gradients with the weights SHARED through LDS ----------------------------------------------------
// The first version of these kernels (one wave = one tile end to end, as in mlp.hip) let every wave stream the whole
// weight set from L2 for every 32-sample tile, 16 loads ahead at most: 8 MFMAs (256 cycles) of work per L2 round trip (~1.5 k cycles) -- matrix pipe 17-19 % busy, 14 TB/s of L2-to-CU
// traffic (profiles/r02_p_c2bf16_*).  Here one workgroup of EIGHT waves per CU (two per SIMD) owns eight tiles at a time and
// stages every layer's weights in LDS once for all eight: while a layer runs from one LDS buffer, the eight waves copy the
// NEXT layer's weights into the other (plain 16-B loads in three stages, 12 registers, see StagePlan), one raw s_barrier
// per layer.  Per layer and tile group: 74 KB from L2 instead of 8 x 74; operands by ds_read_b128 (conflict-free: the LDS
// image is the packed buffer's [chunk][64 lanes][16 B]); biases from LDS.  Same arithmetic, same packed buffers, same
// saved tiles as the streaming kernels.
constexpr int SHW = 8;                                      // waves per workgroup = tiles per group

template <int KIND, bool BWD> struct Shared16 {
    static constexpr NetDesc D = net_desc(KIND);
    static constexpr Pack16Layout L = pack16_layout(KIND);
    static constexpr int NL = D.n_layers;
    // layer order of the pass: forward 0 .. NL-1; backward NL-1 .. 0 (transposed parts)
    static constexpr int layer(int step) { return BWD ? NL - 1 - step : step; }
    static constexpr int chunks(int l) { return BWD ? L.kso[l] * L.tiles_in[l] : L.ks[l] * L.tiles_out[l]; }   // 1-KB chunks
    static constexpr int64_t off(int l) { return (BWD ? L.off_wb[l] : L.off_wf[l]) * 2; }                        // bytes
    static constexpr int max_chunks()
    {
        int m = 0;
        for (int l = 0; l < NL; ++l) m = chunks(l) > m ? chunks(l) : m;
        return m;
    }
    static constexpr int BUF = max_chunks() * 1024;                                      // bytes per LDS weight buffer
    static constexpr int BIAS_FLOATS = 32 * MAX_HID_TILES;                               // per layer, accumulator order
    static constexpr int LDS_BYTES = 2 * BUF + NL * BIAS_FLOATS * 4;
};

// Cooperative copy of the NEXT layer's CH 1-KB chunks into the other LDS buffer, in three stages spread over the current
// layer's products: thread tid moves the 16-B pieces tid + 512 p.  Each stage's loads are issued a third of a layer
// (~770 matrix cycles) before their ds_write, 3 x 4 registers.  (An LDS-DMA loader wave instead of this fills 25 GB/s per
// CU -- 2.9 us per 72-KB layer, more than the layer's matrix time; plain loads spread over all waves do not have that limit.)
template <int CH>
struct StagePlan {
    static constexpr int PIECES = CH * 64;                     // 16-B pieces
    static constexpr int PASSES = (PIECES + 64 * SHW - 1) / (64 * SHW);
    static constexpr int PER = (PASSES + 2) / 3;               // passes per stage
    static_assert(PER <= 3, "stage registers");
};
template <int CH, int S>
__device__ __forceinline__ void stage_load(rsrc_t W, int woff_bytes, int tid, u32x4 (&pre)[3])
{
    using P = StagePlan<CH>;
#pragma unroll
    for (int k = 0; k < P::PER; ++k) {
        const int p = S * P::PER + k;
        if (p < P::PASSES) pre[k] = __builtin_amdgcn_raw_buffer_load_b128(W, (tid + 64 * SHW * p) * 16, woff_bytes, 0);
    }
}
template <int CH, int S>
__device__ __forceinline__ void stage_store(unsigned char *dst, int tid, const u32x4 (&pre)[3])
{
    using P = StagePlan<CH>;
#pragma unroll
    for (int k = 0; k < P::PER; ++k) {
        const int p = S * P::PER + k;
        if (p < P::PASSES && (tid + 64 * SHW * p) < P::PIECES)
            *reinterpret_cast<u32x4 *>(dst + (size_t)(tid + 64 * SHW * p) * 16) = pre[k];
    }
}

// acc[it] += W[j][it] . B(j), weights read from the LDS image of the packed layer (explicitly double-buffered reads);
// interleaved at thirds of the layer: the staging of the NEXT layer's CHN chunks (packed byte offset noff) into `ndst`
template <int KS, int NT, int CHN, typename BF>
__device__ __forceinline__ void lds_layer16(const unsigned char *wsrc, BF bget, f32x16 (&acc)[NT], int lane, int tid,
                                            rsrc_t W, int noff, unsigned char *ndst)
{
    constexpr int NTOT = KS * NT, G = NTOT >= 12 ? 4 : NTOT >= 6 ? 2 : 1, NG = (NTOT + G - 1) / G;
    constexpr int G1 = NG / 3, G2 = (2 * NG) / 3;
    static_assert(CHN == 0 || (G1 >= 1 && G2 > G1 && G2 < NG), "three distinct staging points inside the layer");
    const u32x4 *mine = reinterpret_cast<const u32x4 *>(wsrc) + lane;
    u32x4 buf[2][G], pre[3];
#pragma unroll
    for (int i = 0; i < G; ++i)
        if (i < NTOT) buf[0][i] = mine[i * 64];
    if (CHN > 0) stage_load<CHN, 0>(W, noff, tid, pre);
    bf16x8 b = {};
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int n = (g + 1) * G + i;
            if (n < NTOT) buf[(g + 1) & 1][i] = mine[n * 64];
        }
        if (CHN > 0 && g == G1) { stage_store<CHN, 0>(ndst, tid, pre); stage_load<CHN, 1>(W, noff, tid, pre); }
        if (CHN > 0 && g == G2) { stage_store<CHN, 1>(ndst, tid, pre); stage_load<CHN, 2>(W, noff, tid, pre); }
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int n = g * G + i;
            if (n < NTOT) {
                const int j = n / NT, it = n % NT;
                if (it == 0) b = bget(j);
                acc[it] = mfma16(__builtin_bit_cast(bf16x8, buf[g & 1][i]), b, acc[it]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (CHN > 0) stage_store<CHN, 2>(ndst, tid, pre);
}
// The same layer TILE BY TILE (round 4): output tile `it` receives all its KS k-steps before the next tile starts, so it is
// finished early and its epilogue -- done(it, acc): bias, ReLU, mask bits, rounding, stores, ~80 vector instructions -- is
// issued while the NEXT tile's MFMAs execute and while the SIMD's other wave streams its own.  In the k-step-major order
// of lds_layer16 all NT tiles finish together at the end of the layer: the eight waves of the workgroup then sat in
// their epilogues at the same time with the matrix pipe idle, and in their products with the vector lanes idle
// (tools/ubench/fwd16_stamps.hip, round 4: products 3.0-3.8 k, epilogue 5.1-5.8 k, barrier wait 2.6 k cycles per layer).
// Two accumulators alternate (tile it -> a[it & 1]); the weights' LDS chunk of (j, it) is j * NT + it as before.
// (compile-time loop: the tile index reaches done() as an integral_constant, so every register-array index in the
// epilogue is a constant whatever the optimiser decides about unrolling)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
template <int KS, int NT, int CHN, typename BF, typename DONE>
__device__ __forceinline__ void lds_layer16_tiled(const unsigned char *wsrc, BF bget, DONE done, int lane, int tid,
                                                  rsrc_t W, int noff, unsigned char *ndst)
{
    constexpr int NTOT = KS * NT, G = NTOT >= 12 ? 4 : NTOT >= 6 ? 2 : 1, NG = (NTOT + G - 1) / G;
    constexpr int G1 = NG / 3, G2 = (2 * NG) / 3;
    static_assert(CHN == 0 || (G1 >= 1 && G2 > G1 && G2 < NG), "three distinct staging points inside the layer");
    const u32x4 *mine = reinterpret_cast<const u32x4 *>(wsrc) + lane;
    u32x4 buf[2][G], pre[3];
    static_for<0, G>([&](auto I) {
        constexpr int i = decltype(I)::value;
        if constexpr (i < NTOT) buf[0][i] = mine[((i % KS) * NT + i / KS) * 64];      // flat n = it * KS + j -> chunk j * NT + it
    });
    if (CHN > 0) stage_load<CHN, 0>(W, noff, tid, pre);
    f32x16 a[2];
    static_for<0, NG>([&](auto GI) {
        constexpr int g = decltype(GI)::value;
        static_for<0, G>([&](auto I) {
            constexpr int n = (g + 1) * G + decltype(I)::value;
            if constexpr (n < NTOT) buf[(g + 1) & 1][decltype(I)::value] = mine[((n % KS) * NT + n / KS) * 64];
        });
        if constexpr (CHN > 0 && g == G1) { stage_store<CHN, 0>(ndst, tid, pre); stage_load<CHN, 1>(W, noff, tid, pre); }
        if constexpr (CHN > 0 && g == G2) { stage_store<CHN, 1>(ndst, tid, pre); stage_load<CHN, 2>(W, noff, tid, pre); }
        static_for<0, G>([&](auto I) {
            constexpr int i = decltype(I)::value, n = g * G + i;
            if constexpr (n < NTOT) {
                constexpr int it = n / KS, j = n % KS;
                if constexpr (j == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) a[it & 1][r] = 0.f;
                }
                a[it & 1] = mfma16(__builtin_bit_cast(bf16x8, buf[g & 1][i]), bget(j), a[it & 1]);
                if constexpr (j == KS - 1 && it > 0)               // (behind this tile's MFMAs in issue order)
                    done(std::integral_constant<int, it - 1>{}, a[(it - 1) & 1]);
            }
        });
        __builtin_amdgcn_sched_barrier(0);
    });
    done(std::integral_constant<int, NT - 1>{}, a[(NT - 1) & 1]);
    if (CHN > 0) stage_store<CHN, 2>(ndst, tid, pre);
}
// acc += bias (accumulator order, from LDS), AFTER the layer's products.  (Initialising the accumulators with ds_read_b128
// straight into the MFMA's srcC registers gave wrong values in two registers of the last tile -- rows 11 / 15 / 16 / 20 of
// units 160-191, deterministically, with bias and weights in LDS verified correct in-kernel; zero-initialised
// accumulators + this add are exact.  Cause not established; the add costs ~100 VALU instructions per layer, beside a
// matrix pipe that is not the f32 lanes.)
template <int NT>
__device__ __forceinline__ void lds_bias_add(const float *bl, f32x16 (&acc)[NT], int lane)
{
    const float4 *b4 = reinterpret_cast<const float4 *>(bl + (lane >> 5) * 16);
#pragma unroll
    for (int it = 0; it < NT; ++it)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = b4[it * 8 + q];
            acc[it][4 * q + 0] += v.x; acc[it][4 * q + 1] += v.y; acc[it][4 * q + 2] += v.z; acc[it][4 * q + 3] += v.w;
        }
}
// workgroup barrier between layer steps: LDS contents change hands here, nothing may be moved across it
__device__ __forceinline__ void layer_barrier()
{
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xc07f);                    // lgkmcnt(0): this wave's LDS reads of the layer are done
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// The two waves of a SIMD run the same instruction stream from the same barrier: left alone they stay in phase -- both in a
// tile's MFMAs (taking turns on the matrix pipe), then both in a tile's epilogue (taking turns on the vector lanes) -- and
// the per-SIMD time is the SUM of matrix and vector time.  Holding back the second wave of every SIMD (waves 4-7) by
// about one tile's products after each barrier puts its MFMAs under the first wave's epilogues and vice versa.
// skew: units of 64 clocks (s_sleep 1); skew16() below (6 measured best at C3 / C5; 0 = off).
__device__ __forceinline__ void layer_skew(int wv, int skew)
{
    if (wv >= SHW / 2)
        for (int i = 0; i < skew; ++i) __builtin_amdgcn_s_sleep(1);
}
constexpr int skew16() { return 6; }

template <int KIND>
__global__ void __launch_bounds__(64 * SHW, 1) mlp_fwd16s_kernel(Fwd16Batch AB)
{
    int blk0, nblk;
    const Fwd16Args A = pick_seg16(AB, blk0, nblk);
    using S = Shared16<KIND, false>;
    constexpr NetDesc D = S::D;
    constexpr int NL = S::NL, NHID = NL - 1, HT = D.hid_tiles;
    constexpr unsigned HBYTES = HT * 32 * 32 * 2, MBYTES = (HT / 2) * 256;
    constexpr PackLayout L32 = pack_layout(KIND);
    constexpr Pack16Layout L = S::L;
    constexpr int KS1 = L.ks[0];
    extern __shared__ __attribute__((aligned(16))) unsigned char wl[];          // buffer 0 | buffer 1 | biases
    float *bias_l = reinterpret_cast<float *>(wl + 2 * S::BUF);
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, s = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntiles = A.t1 - A.t0, ngroups = (ntiles + SHW - 1) / SHW;
    // biases of every layer into LDS (accumulator order, as packed for the fp32 engine)
    for (int i = tid; i < NL * S::BIAS_FLOATS; i += 64 * SHW) {
        const int l = i / S::BIAS_FLOATS, k = i % S::BIAS_FLOATS;
        bias_l[i] = k < L32.tiles_out[l] * 32 ? A.packed32[L32.off_bf[l] + k] : 0.f;
    }
    const rsrc_t W16 = make_rsrc(A.packed16, (unsigned)(L.total * 2));
    {   // layer 0 into buffer 0 (and, for a two-layer net, layer 1 into buffer 1: then nothing moves in the loop)
        u32x4 pre[3];
        stage_load<S::chunks(0), 0>(W16, (int)S::off(0), tid, pre); stage_store<S::chunks(0), 0>(wl, tid, pre);
        stage_load<S::chunks(0), 1>(W16, (int)S::off(0), tid, pre); stage_store<S::chunks(0), 1>(wl, tid, pre);
        stage_load<S::chunks(0), 2>(W16, (int)S::off(0), tid, pre); stage_store<S::chunks(0), 2>(wl, tid, pre);
        if (NL == 2) {
            stage_load<S::chunks(1), 0>(W16, (int)S::off(1), tid, pre); stage_store<S::chunks(1), 0>(wl + S::BUF, tid, pre);
            stage_load<S::chunks(1), 1>(W16, (int)S::off(1), tid, pre); stage_store<S::chunks(1), 1>(wl + S::BUF, tid, pre);
            stage_load<S::chunks(1), 2>(W16, (int)S::off(1), tid, pre); stage_store<S::chunks(1), 2>(wl + S::BUF, tid, pre);
        }
    }
    layer_barrier();
    layer_skew(wv, AB.skew);
    int cur_buf = 0;                                       // LDS buffer holding the layer about to run
    // The smaller nets (everything but the 256-register radiance instance) request the NEXT tile group's input rows while
    // this one runs: the workgroup's eight waves march in step behind the layer barriers, so at the top of a group all of
    // them waited for these loads at once (mlp_dgrad16s_kernel does the same with its gradients and masks).
    constexpr bool PREFETCH = KIND != ESR_MLP_RADIANCE;
    float xn[PREFETCH ? KS1 * 8 : 1];
    const int xvoff = (h * 8 * 32 + s) * 4;
    auto fetch = [&](int tg) {
        const int tt = A.t0 + tg * SHW + wv;
        const int t = tt < A.t1 ? tt : A.t1 - 1;
        const rsrc_t RX = make_rsrc(A.X + (size_t)t * D.xrows * 32, D.xrows * 32 * 4);
        const int coff = A.crow * 128;
#pragma unroll
        for (int j = 0; j < KS1; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = 16 * j + 8 * h + i;
                xn[PREFETCH ? j * 8 + i : 0] = bload1(RX, xvoff + (row < D.cw ? coff : 0), (16 * j + i) * 128);
            }
    };
    if (PREFETCH && (int)blockIdx.x - blk0 < ngroups) fetch((int)blockIdx.x - blk0);
    // radiance instance with the bf16 input tile: the first layer's operand IS the tile's bytes -- two 8-byte loads per 16
    // input rows (quads 4 j + 2 h and + 1 of this lane's sample slot) instead of 16 dword loads and 16 conversions, and
    // the next tile group's go out as soon as the first layer has consumed the registers (nothing extra is held)
    const bool x16 = KIND == ESR_MLP_RADIANCE && A.X16 != nullptr;
    bf16x8 B1[KS1];
    auto fetch16 = [&](int tg) {
        const int tt = A.t0 + tg * SHW + wv;
        const int t = tt < A.t1 ? tt : A.t1 - 1;
        const uint2 *q0 = reinterpret_cast<const uint2 *>(A.X16) + ((size_t)t * 26 * 32 + (8 * ((s >> 1) & 3) + 2 * (s >> 3) + (s & 1)));
#pragma unroll
        for (int j = 0; j < KS1; ++j) {
            int q = 4 * j + 2 * h;
            if (j == 0 && h == 0 && A.crow == 88) q = 24;          // colour rows 0..5 <- the group of rows 88..93
            const uint2 a = q0[q * 32], b = q0[(q + 1) * 32];
            u32x4 v = {a.x, a.y, b.x, b.y};
            B1[j] = __builtin_bit_cast(bf16x8, v);
        }
    };
    if (x16 && (int)blockIdx.x - blk0 < ngroups) fetch16((int)blockIdx.x - blk0);
    for (int tg = (int)blockIdx.x - blk0; tg < ngroups; tg += nblk) {
        const int tt = A.t0 + tg * SHW + wv;
        const bool live = tt < A.t1;                       // a wave past the range runs on the last tile, stores nothing
        const int t = live ? tt : A.t1 - 1;
        ESR_STAMP16(0);
        if (x16) {
            // (B1 was requested behind the previous group's first layer, or before the loop)
        } else if constexpr (PREFETCH) {
#pragma unroll
            for (int j = 0; j < KS1; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i) B1[j][i] = (__bf16)xn[j * 8 + i];
            fetch(tg + nblk < ngroups ? tg + nblk : tg);   // (past the end: this group again, never used)
        } else {
            const rsrc_t RX = make_rsrc(A.X + (size_t)t * D.xrows * 32, D.xrows * 32 * 4);
            const int coff = A.crow * 128;
#pragma unroll
            for (int j = 0; j < KS1; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int row = 16 * j + 8 * h + i;
                    B1[j][i] = (__bf16)bload1(RX, xvoff + (row < D.cw ? coff : 0), (16 * j + i) * 128);
                }
        }
        const bool save = A.save && live;
        // A layer's input and output live as PACKED bf16 (8 registers per 32 rows, 48 for a 192-wide layer): that is what
        // the next layer's MFMAs consume and what the saved H tile holds, so each accumulator is rounded ONCE, in its
        // tile's epilogue, and the fp32 tile dies there (rounds 2-3 kept the whole fp32 layer through the next layer:
        // 96 + 96 accumulator registers + 32 + 12 + 24 = 260 -> 12 spilled).  Two sets alternate: a layer reads one
        // while its tiles' epilogues fill the other.
        bf16x8 hbA[2 * HT], hbB[2 * HT];
        const int s_ = lane & 31;
        const int hvoff = ((lane >> 5) * 32 + 8 * ((s_ >> 1) & 3) + 2 * (s_ >> 3) + (s_ & 1)) * 8;      // store_tiles_bf16's slot order
        unsigned mword = 0;
        // epilogue of ONE finished tile of layer l: bias (LDS, accumulator order), ReLU, the tile's 16 mask bits,
        // rounding into `hout`, the tile's four 8-byte row-quad stores; masks leave as one word per tile pair
        auto tile_epilogue = [&](auto LC, auto IT, f32x16 &acc, bf16x8 (&hout)[2 * HT]) __attribute__((always_inline)) {
            constexpr int l = decltype(LC)::value, it = decltype(IT)::value;
            const float4 *b4 = reinterpret_cast<const float4 *>(bias_l + l * S::BIAS_FLOATS + (lane >> 5) * 16) + it * 8;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = b4[q];
                acc[4 * q + 0] += v.x; acc[4 * q + 1] += v.y; acc[4 * q + 2] += v.z; acc[4 * q + 3] += v.w;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int bts = __float_as_int(acc[r]);
                acc[r] = __int_as_float(bts > 0 ? bts : 0);
            }
            hout[2 * it] = acc_to_b(acc, 0);
            hout[2 * it + 1] = acc_to_b(acc, 1);
            if (save) {
                // ReLU mask from the PACKED values, two elements per instruction pair: min(u16, 1) per half (a rounded
                // ReLU output is +0 or a positive bf16; a normal fp32 never rounds to zero, bf16 has fp32's exponent
                // range) shifted into place -- dword k of the tile (registers 2k, 2k + 1) lands on bits mask_bit16(it, 2k)
                // and + 16.  One vector instruction per element instead of two (v_med3 + v_lshl_or on the fp32 value).
                const u32x4 w0 = __builtin_bit_cast(u32x4, hout[2 * it]), w1 = __builtin_bit_cast(u32x4, hout[2 * it + 1]);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    unsigned one;     // (asm: hipcc turns the C form into compare + select per half; the operand is a conversion's result -- a VALU value, no MFMA hazard)
                    asm("v_pk_min_u16 %0, %1, %2" : "=v"(one) : "v"(k < 4 ? w0[k & 3] : w1[k & 3]), "s"(0x00010001u));
                    mword |= one << mask_bit16(it, 2 * k);
                }
                if ((it & 1) || it == HT - 1) {
                    __builtin_amdgcn_raw_buffer_store_b32(mword, make_rsrc(A.M[l] + (size_t)t * (MBYTES / 4), MBYTES), lane * 4,
                                                          (it >> 1) * 256, 0);
                    mword = 0;
                }
            }
            if (save && A.save == 1) {                             // save == 2: ReLU masks only (the weight gradients recompute the layer)
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                const rsrc_t RH = make_rsrc(A.H[l] + (size_t)t * (HBYTES / 4), HBYTES);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const u32x4 w = __builtin_bit_cast(u32x4, hout[2 * it + (q >> 1)]);
                    u32x2 v;
                    v[0] = w[2 * (q & 1)]; v[1] = w[2 * (q & 1) + 1];
                    __builtin_amdgcn_raw_buffer_store_b64(v, RH, hvoff, (8 * it + 2 * q) * 256, ESR_NT_AUX);
                }
            }
        };
        lds_layer16_tiled<KS1, HT, (NL == 2 ? 0 : S::chunks(1))>(
            wl + (NL == 2 ? 0 : cur_buf * S::BUF), [&](int j) { return B1[j]; },
            [&](auto IT, f32x16 &acc) __attribute__((always_inline)) { tile_epilogue(std::integral_constant<int, 0>{}, IT, acc, hbA); },
            lane, tid, W16, (int)S::off(1), wl + (cur_buf ^ 1) * S::BUF);
        if (x16) fetch16(tg + nblk < ngroups ? tg + nblk : tg);    // (past the end: this group again, never used)
        ESR_STAMP16(1);
        ESR_STAMP16(2);
        if (NL != 2) { layer_barrier(); layer_skew(wv, AB.skew); cur_buf ^= 1; }
        ESR_STAMP16(3);
        auto hidden = [&](auto LC, bf16x8 (&hin)[2 * HT], bf16x8 (&hout)[2 * HT]) __attribute__((always_inline)) {
            constexpr int l = decltype(LC)::value;
            lds_layer16_tiled<2 * HT, HT, S::chunks(l + 1)>(
                wl + cur_buf * S::BUF, [&](int j) { return hin[j]; },
                [&](auto IT, f32x16 &acc) __attribute__((always_inline)) { tile_epilogue(LC, IT, acc, hout); },
                lane, tid, W16, (int)S::off(l + 1), wl + (cur_buf ^ 1) * S::BUF);
            ESR_STAMP16(4 + 3 * (l - 1));
            ESR_STAMP16(5 + 3 * (l - 1));
            layer_barrier();
            layer_skew(wv, AB.skew);
            cur_buf ^= 1;
            ESR_STAMP16(6 + 3 * (l - 1));
        };
        if constexpr (NHID > 1) hidden(std::integral_constant<int, 1>{}, hbA, hbB);
        if constexpr (NHID > 2) hidden(std::integral_constant<int, 2>{}, hbB, hbA);
        static_assert(NHID <= 3, "hidden steps are spelled out");
        bf16x8 (&hlast)[2 * HT] = (NHID == 2) ? hbB : hbA;         // output of the last hidden layer
        f32x16 out[1];
        zero_tiles<1>(out);
        // (meanwhile layer 0 of the NEXT tile group is staged: the same weights, only the buffer differs)
        lds_layer16<2 * HT, 1, (NL == 2 ? 0 : S::chunks(0))>(wl + (NL == 2 ? S::BUF : cur_buf * S::BUF),
                                                           [&](int j) { return hlast[j]; }, out, lane, tid,
                                                           W16, (int)S::off(0), wl + (cur_buf ^ 1) * S::BUF);
        ESR_STAMP16(10);
        lds_bias_add<1>(bias_l + NHID * S::BIAS_FLOATS, out, lane);
        const rsrc_t RZ = make_rsrc(A.zout + (size_t)t * D.zrows * 32, live ? D.zrows * 32 * 4 : 0);
        const int zvoff = (D.zrows == 8) ? (4 * h * 32 + s) * 4 : ((h ? D.zrows : 0) * 32 + s) * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) bstore1(RZ, (D.zrows == 8 || r < 3) ? out[0][r] : 0.f, zvoff, r * 128);
        ESR_STAMP16(11);
        if (NL != 2) { layer_barrier(); layer_skew(wv, AB.skew); cur_buf ^= 1; }
        ESR_STAMP16(12);
    }
}

struct Dgrad16Args {
    const __bf16 *packed16;
    const float *dz;
    int t0, t1;
    const unsigned *M[3];
    float *dZ[3];
    float *dX;
};
struct Dgrad16Batch {
    Dgrad16Args base;
    int nseg;
    Seg16 seg[MAX_SEG16];      // packed16, t0, t1, b0, nb used
};
__device__ __forceinline__ Dgrad16Args pick_dseg16(const Dgrad16Batch &B, int &b0, int &nb)
{
    Dgrad16Args A = B.base;
    b0 = 0; nb = (int)gridDim.x;
#pragma unroll
    for (int k = 0; k < MAX_SEG16; ++k)
        if (k < B.nseg && (int)blockIdx.x >= B.seg[k].b0) {
            A.packed16 = B.seg[k].packed16; A.t0 = B.seg[k].t0; A.t1 = B.seg[k].t1; b0 = B.seg[k].b0; nb = B.seg[k].nb;
        }
    return A;
}

// ---- input gradients with the weights shared through LDS (the forward's scheme, layers in reverse) -------------------
template <int KIND>
__global__ void __launch_bounds__(64 * SHW, 1) mlp_dgrad16s_kernel(Dgrad16Batch AB)
{
    int blk0, nblk;
    const Dgrad16Args A = pick_dseg16(AB, blk0, nblk);
    using S = Shared16<KIND, true>;
    constexpr NetDesc D = S::D;
    constexpr int NL = S::NL, NHID = NL - 1, HT = D.hid_tiles;
    constexpr unsigned HBYTES = HT * 32 * 32 * 2, MBYTES = (HT / 2) * 256;
    constexpr Pack16Layout L = S::L;
    extern __shared__ __attribute__((aligned(16))) unsigned char wl[];          // buffer 0 | buffer 1
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, s = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntiles = A.t1 - A.t0, ngroups = (ntiles + SHW - 1) / SHW;
    const rsrc_t W16 = make_rsrc(A.packed16, (unsigned)(L.total * 2));
    {   // step 0 (the output layer's transpose) into buffer 0; a two-layer net: its other layer into buffer 1, for good
        u32x4 pre[3];
        stage_load<S::chunks(NL - 1), 0>(W16, (int)S::off(NL - 1), tid, pre); stage_store<S::chunks(NL - 1), 0>(wl, tid, pre);
        stage_load<S::chunks(NL - 1), 1>(W16, (int)S::off(NL - 1), tid, pre); stage_store<S::chunks(NL - 1), 1>(wl, tid, pre);
        stage_load<S::chunks(NL - 1), 2>(W16, (int)S::off(NL - 1), tid, pre); stage_store<S::chunks(NL - 1), 2>(wl, tid, pre);
        if (NL == 2) {
            stage_load<S::chunks(0), 0>(W16, (int)S::off(0), tid, pre); stage_store<S::chunks(0), 0>(wl + S::BUF, tid, pre);
            stage_load<S::chunks(0), 1>(W16, (int)S::off(0), tid, pre); stage_store<S::chunks(0), 1>(wl + S::BUF, tid, pre);
            stage_load<S::chunks(0), 2>(W16, (int)S::off(0), tid, pre); stage_store<S::chunks(0), 2>(wl + S::BUF, tid, pre);
        }
    }
    layer_barrier();
    int cur_buf = 0;
    // The output gradients and ReLU masks of the NEXT tile group are requested while this one runs (13 registers): the
    // eight waves of the workgroup march in step behind the layer barriers, so at the top of a group all of them waited
    // for these loads at once -- nobody to hide them behind.
    constexpr int ZN = D.zrows < 8 ? D.zrows : 8;                              // dz rows 0 .. ZN-1 live in half 0's slots
    float zn[ZN];
    unsigned mn[NHID][HT / 2];
    auto fetch = [&](int tg) {
        const int tt = A.t0 + tg * SHW + wv;
        const int t = tt < A.t1 ? tt : A.t1 - 1;
        const rsrc_t RZ = make_rsrc(A.dz + (size_t)t * D.zrows * 32, D.zrows * 32 * 4);
        const int zoff = h == 0 ? s * 4 : 0x7ffffff0;                          // (half 1: k = 8 .. 15, zeros -- out of range)
#pragma unroll
        for (int i = 0; i < ZN; ++i) zn[i] = bload1(RZ, zoff, i * 128);
#pragma unroll
        for (int l = 0; l < NHID; ++l)
            load_relu_mask<HT>(make_rsrc(A.M[l] + (size_t)t * (MBYTES / 4), MBYTES), mn[l], lane);
    };
    if ((int)blockIdx.x - blk0 < ngroups) fetch((int)blockIdx.x - blk0);
    for (int tg = (int)blockIdx.x - blk0; tg < ngroups; tg += nblk) {
        const int tt = A.t0 + tg * SHW + wv;
        const bool live = tt < A.t1;                       // a wave past the range runs on the last tile, stores nothing
        const int t = live ? tt : A.t1 - 1;
        bf16x8 B0;                                                           // slot i of half h <-> dz row 8 h + i
#pragma unroll
        for (int i = 0; i < 8; ++i) B0[i] = (__bf16)(i < ZN ? zn[i] : 0.f);
        unsigned msk[NHID][HT / 2];
#pragma unroll
        for (int l = 0; l < NHID; ++l)
#pragma unroll
            for (int q = 0; q < HT / 2; ++q) msk[l][q] = mn[l][q];
        fetch(tg + nblk < ngroups ? tg + nblk : tg);       // (past the end: this group again, never used)
        const unsigned hb = live ? HBYTES : 0u;            // zero-record descriptors drop the stores of a wave past the range
        f32x16 cur[HT];
        zero_tiles<HT>(cur);
        lds_layer16<1, HT, (NL == 2 ? 0 : S::chunks(NL - 2))>(wl + (NL == 2 ? 0 : cur_buf * S::BUF), [&](int) { return B0; }, cur, lane,
                                                            tid, W16, (int)S::off(NL - 2), wl + (cur_buf ^ 1) * S::BUF);
        apply_relu_mask16<HT>(msk[NHID - 1], cur);
        if (A.dZ[NHID - 1])                                // (a NULL dZ[l] is not stored: its weight gradient recomputes it)
            store_tiles_bf16<HT>(make_rsrc(A.dZ[NHID - 1] + (size_t)t * (HBYTES / 4), hb), cur, lane);
        if (NL != 2) { layer_barrier(); cur_buf ^= 1; }
        auto hidden = [&](auto LC) {
            constexpr int l = decltype(LC)::value;                               // layer l's transpose: dZ[l] -> dZ[l - 1]
            f32x16 nxt[HT];
            zero_tiles<HT>(nxt);
            lds_layer16<2 * HT, HT, S::chunks(l - 1)>(wl + cur_buf * S::BUF, [&](int j) { return acc_to_b(cur[j >> 1], j & 1); },
                                                     nxt, lane, tid, W16, (int)S::off(l - 1), wl + (cur_buf ^ 1) * S::BUF);
            apply_relu_mask16<HT>(msk[l - 1], nxt);
            if (A.dZ[l - 1]) store_tiles_bf16<HT>(make_rsrc(A.dZ[l - 1] + (size_t)t * (HBYTES / 4), hb), nxt, lane);
#pragma unroll
            for (int it = 0; it < HT; ++it) cur[it] = nxt[it];
            layer_barrier();
            cur_buf ^= 1;
        };
        if constexpr (NHID > 2) hidden(std::integral_constant<int, NHID - 1>{});
        if constexpr (NHID > 1) hidden(std::integral_constant<int, 1>{});
        static_assert(NHID <= 3, "hidden steps are spelled out");
        f32x16 dx[2];
        zero_tiles<2>(dx);
        // first layer's transpose; meanwhile step 0 of the NEXT tile group is staged
        lds_layer16<2 * HT, 2, (NL == 2 ? 0 : S::chunks(NL - 1))>(wl + (NL == 2 ? S::BUF : cur_buf * S::BUF),
                                                                [&](int j) { return acc_to_b(cur[j >> 1], j & 1); }, dx, lane, tid,
                                                                W16, (int)S::off(NL - 1), wl + (cur_buf ^ 1) * S::BUF);
        store_tiles<2, false>(make_rsrc(A.dX + (size_t)t * 64 * 32, live ? 64 * 32 * 4 : 0), dx, lane);     // (the scatter reads dX next)
        if (NL != 2) { layer_barrier(); cur_buf ^= 1; }
    }
}

template <int KIND>
int launch_dgrad16s(const Dgrad16Args &A, hipStream_t s)
{
    using S = Shared16<KIND, true>;
    static std::atomic<uint64_t> optin{0};
    if (int rc = esr_lds_optin(reinterpret_cast<const void *>(&mlp_dgrad16s_kernel<KIND>), 2 * S::BUF, optin)) return rc;
    const int groups = (A.t1 - A.t0 + SHW - 1) / SHW;
    Dgrad16Batch B = {};
    B.base = A;
    mlp_dgrad16s_kernel<KIND><<<groups < 256 ? groups : 256, 64 * SHW, 2 * S::BUF, s>>>(B);
    ESR_CHECK_LAUNCH();
    return 0;
}

// workgroups per segment proportional to its tile groups (every non-empty segment >= 1); returns the grid
int share_blocks16(Seg16 *seg, int nseg)
{
    int groups[MAX_SEG16], total = 0;
    for (int k = 0; k < nseg; ++k) { groups[k] = (seg[k].t1 - seg[k].t0 + SHW - 1) / SHW; total += groups[k]; }
    const int grid = total < 256 ? total : 256;
    int given = 0;
    for (int k = 0; k < nseg; ++k) {
        int n = (int)((int64_t)grid * groups[k] / (total > 0 ? total : 1));
        if (n < 1) n = 1;
        if (n > groups[k]) n = groups[k];
        seg[k].nb = n;
        given += n;
    }
    for (int guard = 0; given != grid && guard < 1024; ++guard) {
        int pick = -1;
        double best = 0.0;
        for (int k = 0; k < nseg; ++k) {
            if (given < grid) {
                if (seg[k].nb >= groups[k]) continue;
                const double load = (double)groups[k] / seg[k].nb;
                if (pick < 0 || load > best) { pick = k; best = load; }
            } else {
                if (seg[k].nb <= 1) continue;
                const double load = (double)groups[k] / (seg[k].nb - 1);
                if (pick < 0 || load < best) { pick = k; best = load; }
            }
        }
        if (pick < 0) break;
        seg[pick].nb += given < grid ? 1 : -1;
        given += given < grid ? 1 : -1;
    }
    int b0 = 0;
    for (int k = 0; k < nseg; ++k) { seg[k].b0 = b0; b0 += seg[k].nb; }
    return b0;
}

bool crow_ok(int kind, int crow)
{
    return kind == ESR_MLP_COARSE ? (crow == 0 || crow == 12) : (crow == 0 || crow == 88 || crow == 96);
}

// one workgroup (7 compute waves + loader) per CU, persistent over tile groups; > 64 KB of LDS needs the opt-in
template <int KIND>
int launch_fwd16s(const Fwd16Args &A, hipStream_t s)
{
    using S = Shared16<KIND, false>;
    static std::atomic<uint64_t> optin{0};
    if (int rc = esr_lds_optin(reinterpret_cast<const void *>(&mlp_fwd16s_kernel<KIND>), S::LDS_BYTES, optin)) return rc;
    const int groups = (A.t1 - A.t0 + SHW - 1) / SHW;
    Fwd16Batch B = {};
    B.base = A;
    B.skew = skew16();
    mlp_fwd16s_kernel<KIND><<<groups < 256 ? groups : 256, 64 * SHW, S::LDS_BYTES, s>>>(B);
    ESR_CHECK_LAUNCH();
    return 0;
}


}  // namespace

ESR_API int64_t esr_mlp_packed_bf16_elems(int kind)
{
    if (!kind_ok(kind)) return ESR_EINVAL;
    return pack16_layout(kind).total;
}

ESR_API int esr_mlp_pack_bf16(int kind, const esr_mlp_weights_t *w, void *packed16, void *stream)
{
    if (!kind_ok(kind) || !w || !packed16) return ESR_EINVAL;
    PackBatch B = {};
    B.n = 1;
    PackArgs &A = B.job[0];
    A.kind = kind;
    A.out16 = static_cast<__bf16 *>(packed16);
    for (int l = 0; l < net_desc(kind).n_layers; ++l) {
        if (!w->w[l]) return ESR_EINVAL;
        A.w[l] = w->w[l];
    }
    // (fp32 part skipped: out == NULL; the element range still starts with it)
    pack_kernel<<<dim3(esr_grid_for(pack_layout(kind).total + pack16_layout(kind).total, 256, 1024), 1), 256, 0, esr_stream(stream)>>>(B);   // (fp32 part skipped: out == NULL)
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_mlp_fwd_bf16(int kind, const float *packed32, const void *packed16, const float *X, int32_t t0,
                             int32_t t1, float *const *H, uint32_t *const *M, int save, int color_row0,
                             float *zout, void *stream)
{
    if (!kind_ok(kind) || t0 < 0 || t1 < t0 || !crow_ok(kind, color_row0)) return ESR_EINVAL;
    if (t1 == t0) return 0;
    if (!packed32 || !packed16 || !X || !zout) return ESR_EINVAL;
    const int nhid = net_desc(kind).n_layers - 1;
    Fwd16Args A = {};
    A.packed32 = packed32; A.packed16 = static_cast<const __bf16 *>(packed16); A.X = X; A.t0 = t0; A.t1 = t1;
    A.save = save == 2 ? 2 : save ? 1 : 0; A.crow = color_row0; A.zout = zout;
    if (save) {
        if (!M || (save != 2 && !H)) return ESR_EINVAL;
        for (int l = 0; l < nhid; ++l) {
            if (!M[l] || (save != 2 && !H[l])) return ESR_EINVAL;
            A.H[l] = save != 2 ? H[l] : nullptr; A.M[l] = M[l];
        }
    }
    hipStream_t s = esr_stream(stream);
    switch (kind) {
    case ESR_MLP_RADIANCE: return launch_fwd16s<ESR_MLP_RADIANCE>(A, s);
    case ESR_MLP_TONEMAP:  return launch_fwd16s<ESR_MLP_TONEMAP>(A, s);
    case ESR_MLP_BRDF:     return launch_fwd16s<ESR_MLP_BRDF>(A, s);
    case ESR_MLP_EMIT:     return launch_fwd16s<ESR_MLP_EMIT>(A, s);
    default:               return launch_fwd16s<ESR_MLP_COARSE>(A, s);
    }
}

// The fine stage's three radiance forward passes of a step (bf16 engine) as ONE launch: see esr_mlp_fwd_fine.
ESR_API int esr_mlp_fwd_fine_bf16(const float *packed32_off, const void *packed16_off, const float *packed32_emo,
                                  const void *packed16_emo, const float *X, const void *X16, int32_t t_on, int32_t t_all,
                                  float *const *H, uint32_t *const *M, int color_row_detached, float *z_off, float *z_emo,
                                  void *stream)
{
    if (t_on < 0 || t_all < t_on || !crow_ok(ESR_MLP_RADIANCE, color_row_detached)) return ESR_EINVAL;
    if (t_all == 0) return 0;
    if (!packed32_off || !packed16_off || !packed32_emo || !packed16_emo || (!X && !X16) || !H || !M || !z_off || !z_emo)
        return ESR_EINVAL;
    if (X16 && color_row_detached != 0 && color_row_detached != 88) return ESR_EINVAL;     // (the bf16 tile carries rows 88..93 only)
    using S = Shared16<ESR_MLP_RADIANCE, false>;
    Fwd16Batch B = {};
    B.skew = skew16();
    B.base.X = X; B.base.X16 = X16;
    for (int l = 0; l < 3; ++l) {
        if (!H[l] || !M[l]) return ESR_EINVAL;
        B.base.H[l] = H[l]; B.base.M[l] = M[l];
    }
    const __bf16 *p16o = static_cast<const __bf16 *>(packed16_off), *p16e = static_cast<const __bf16 *>(packed16_emo);
    int n = 0;
    if (t_on > 0) B.seg[n++] = Seg16{packed32_off, p16o, 0, t_on, 0, color_row_detached, z_off, 0, 0};
    if (t_all > t_on) B.seg[n++] = Seg16{packed32_off, p16o, t_on, t_all, 1, 0, z_off, 0, 0};
    if (t_on > 0) B.seg[n++] = Seg16{packed32_emo, p16e, 0, t_on, 1, 0, z_emo, 0, 0};
    B.nseg = n;
    const int grid = share_blocks16(B.seg, n);
    static std::atomic<uint64_t> optin{0};
    if (int rc = esr_lds_optin(reinterpret_cast<const void *>(&mlp_fwd16s_kernel<ESR_MLP_RADIANCE>), S::LDS_BYTES, optin)) return rc;
    mlp_fwd16s_kernel<ESR_MLP_RADIANCE><<<grid, 64 * SHW, S::LDS_BYTES, esr_stream(stream)>>>(B);
    ESR_CHECK_LAUNCH();
    return 0;
}

// Both radiance nets' input gradients (bf16 engine) as ONE launch: emissive net on [0, t_on), non-emissive on [t_on, t_all).
ESR_API int esr_mlp_dgrad_fine_bf16(const void *packed16_emo, const void *packed16_off, const float *dz, int32_t t_on,
                                    int32_t t_all, const uint32_t *const *M, float *const *dZ, float *dX, void *stream)
{
    if (t_on < 0 || t_all < t_on) return ESR_EINVAL;
    if (t_all == 0) return 0;
    if (!packed16_emo || !packed16_off || !dz || !M || !dZ || !dX) return ESR_EINVAL;
    using S = Shared16<ESR_MLP_RADIANCE, true>;
    Dgrad16Batch B = {};
    B.base.dz = dz; B.base.dX = dX;
    for (int l = 0; l < 3; ++l) {
        if (!M[l]) return ESR_EINVAL;
        B.base.M[l] = M[l]; B.base.dZ[l] = dZ[l];
    }
    int n = 0;
    if (t_on > 0) B.seg[n++] = Seg16{nullptr, static_cast<const __bf16 *>(packed16_emo), 0, t_on, 0, 0, nullptr, 0, 0};
    if (t_all > t_on) B.seg[n++] = Seg16{nullptr, static_cast<const __bf16 *>(packed16_off), t_on, t_all, 0, 0, nullptr, 0, 0};
    B.nseg = n;
    const int grid = share_blocks16(B.seg, n);
    static std::atomic<uint64_t> optin{0};
    if (int rc = esr_lds_optin(reinterpret_cast<const void *>(&mlp_dgrad16s_kernel<ESR_MLP_RADIANCE>), 2 * S::BUF, optin)) return rc;
    mlp_dgrad16s_kernel<ESR_MLP_RADIANCE><<<grid, 64 * SHW, 2 * S::BUF, esr_stream(stream)>>>(B);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_mlp_dgrad_bf16(int kind, const void *packed16, const float *dz, int32_t t0, int32_t t1,
                               const uint32_t *const *M, float *const *dZ, float *dX, void *stream)
{
    if (!kind_ok(kind) || t0 < 0 || t1 < t0) return ESR_EINVAL;
    if (t1 == t0) return 0;
    if (!packed16 || !dz || !M || !dZ || !dX) return ESR_EINVAL;
    const int nhid = net_desc(kind).n_layers - 1;
    Dgrad16Args A = {};
    A.packed16 = static_cast<const __bf16 *>(packed16); A.dz = dz; A.t0 = t0; A.t1 = t1; A.dX = dX;
    for (int l = 0; l < nhid; ++l) {
        if (!M[l]) return ESR_EINVAL;
        A.M[l] = M[l]; A.dZ[l] = dZ[l];                 // a NULL dZ[l] is computed but not stored
    }
    hipStream_t s = esr_stream(stream);
    switch (kind) {
    case ESR_MLP_RADIANCE: return launch_dgrad16s<ESR_MLP_RADIANCE>(A, s);
    case ESR_MLP_TONEMAP:  return launch_dgrad16s<ESR_MLP_TONEMAP>(A, s);
    case ESR_MLP_BRDF:     return launch_dgrad16s<ESR_MLP_BRDF>(A, s);
    case ESR_MLP_EMIT:     return launch_dgrad16s<ESR_MLP_EMIT>(A, s);
    default:               return launch_dgrad16s<ESR_MLP_COARSE>(A, s);
    }
}
