// Tiny-MLP engine on the f32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Reference: app/utils/pbr/module.py:6-39 (RadianceNet 85-192-192-192-3 + softplus,
// TonemapNet 33-192-3 + sigmoid) evaluated per surviving sample in
// app/fine/model/voxurff.py:243-256, and torch autograd for the backward.
//
// MI355X design (not a GEMM-library call, not a warp-shaped tiling):
//  * "Transposed" formulation  H_l^T = W_l . H_{l-1}^T : the weight matrix is the MFMA
//    A operand, a tile of 32 SAMPLES is the N dimension.  The 32x32 accumulator of
//    layer l then has the sample on the lane and the feature in the register --
//    exactly the B-operand layout layer l+1 needs (B[k][n]: lane = n + 32*(k&1)).
//    The whole 4-layer chain therefore runs out of registers: no LDS round trip, no
//    transposes, no barriers; one 64-lane wave owns 32 samples end to end.  The k
//    index a register stands for is a fixed permutation of the feature index; it is
//    folded into the weight packing (esr_mlp_pack), which also bakes in the column
//    permutation between the reference's 85-d input order and the X tile rows.
//  * Weights are pre-packed so that the A operand of 4 consecutive k-steps is one
//    coalesced 16-B-per-lane load (1 KiB per wave instruction) served by L1/L2: the
//    MFMA rate needs only 16 B/clk/CU of weights.
//  * fp32 MFMA is bit-for-bit a k-ordered fmaf chain, so results stay within 1e-6 of
//    the CPU oracle; there is no reduced-precision path on gfx950 for f32 inputs.
//  * Hidden activations are saved tile-major [tile][feature][32] (each store = two
//    full 128-B lines) and re-read by the backward instead of being recomputed: the
//    f32 matrix rate (157 TF) is the binding roof, HBM (8 TB/s) has headroom.
//  * Weight gradients contract over SAMPLES: a workgroup keeps the whole dW block of a layer in
//    the accumulator registers of its four compute waves across its tile range; a fifth LOADER
//    wave streams the two operand tiles of every sample tile HBM -> LDS by LDS-DMA into a
//    3-buffer ring (mlp_wgrad_dma_kernel); partial blocks go to per-workgroup slabs that a
//    second kernel sums.  (bf16 operand modes: the register-staged mlp_wgrad_kernel.)
#include "esr_common.h"

#include "mlp_common.h"

// In-kernel time stamps: nothing in the product build; tools/ubench/fwd_stamps.hip defines ESR_STAMP before including
// this file to record s_memtime at the layer seams of one traced wave per SIMD.
#ifndef ESR_DSTAMP
#define ESR_DSTAMP(i)
#endif
#ifndef ESR_STAMP
#define ESR_STAMP(i)
#endif

namespace {

struct FwdArgs {
    const float *packed, *X;
    int t0, t1;
    float *H[3];
    unsigned *M[3];
    int save, crow;            // crow: X row of the 6-row colour group this net reads (0, 88 or 96)
    float *zout;
    int t_det;                 // tiles t < t_det are evaluated detached: nothing saved, colour group crow_det
    int crow_det;
    // A SECOND net of the same kind in the same launch (esr_mlp_fwd_fine: the fine stage's emissive net beside the
    // non-emissive one): after the first net's tiles [t0, t1) the waves go on with tiles [t0_2, t0_2 + n2) under
    // `packed2` (colour group crow2, save mode save2, output zout2).  One ramp-up and one tail instead of two, and a
    // wave's 12 tiles mix detached (light epilogue) and saved ones.
    const float *packed2;
    float *zout2;
    int t0_2, n2, crow2, save2;
};

// Waves per SIMD of the input-gradient kernel: the tone mapper's chain is short (one hidden layer), a tile's loads are
// not amortised, and its 142 VGPRs allow three; so do the 128-wide material nets' 160.  (The tone mapper's forward
// kernel spills at three: 188 instead of 110 us.)
constexpr int mlp_occ(int kind) { return (kind == ESR_MLP_TONEMAP || kind == ESR_MLP_BRDF || kind == ESR_MLP_EMIT) ? 3 : 2; }

template <int KIND>
__global__ void __launch_bounds__(256, 2) mlp_fwd_kernel(FwdArgs A)
{
    constexpr NetDesc D = net_desc(KIND);
    constexpr int NHID = D.n_layers - 1;
    constexpr int KP1 = D.in_kp;
    constexpr int HT = D.hid_tiles;
    constexpr unsigned HBYTES = HT * 32 * 32 * 4, MBYTES = (HT / 2) * 256;
    constexpr PackLayout L = pack_layout(KIND);
    const int lane = esr_lane();
    const int h = lane >> 5, s = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const rsrc_t W1 = make_rsrc(A.packed, (unsigned)(L.total * 4));
    const rsrc_t W2 = make_rsrc(A.packed2 ? A.packed2 : A.packed, (unsigned)(L.total * 4));
    constexpr int NP4 = D.zrows / 4;
    constexpr int NW4 = NP4 * HT * 4 * 8;
    __shared__ float4 w4s[2 * NW4];                      // output layers as 4x4x1 operands (lds4_layer), both nets
    lds4_preload<NW4>(W1, (int)L.off_w4 * 4, w4s);
    if (A.n2 > 0) lds4_preload<NW4>(W2, (int)L.off_w4 * 4, w4s + NW4);
    __syncthreads();
    const int n1 = A.t1 - A.t0, nv = n1 + A.n2;
    for (int v = wave; v < nv; v += nwaves) {
        const bool second = v >= n1;                         // wave-uniform
        const int t = second ? A.t0_2 + (v - n1) : A.t0 + v;
        const rsrc_t W = second ? W2 : W1;
        const float4 *w4 = w4s + (second ? NW4 : 0);
        const rsrc_t RX = make_rsrc(A.X + (size_t)t * D.xrows * 32, D.xrows * 32 * 4);
        const int xvoff = (h * 32 + s) * 4;
        const bool det = !second && t < A.t_det;             // wave-uniform
        const int coff = (second ? A.crow2 : det ? A.crow_det : A.crow) * 128;
        const int sv = second ? A.save2 : det ? 0 : A.save;  // 1: hidden tiles + ReLU masks, 2: masks only
        const bool save = sv != 0;
        float *const zdst = second ? A.zout2 : A.zout;
        ESR_STAMP(0);
        float B1[KP1];
#pragma unroll
        for (int p = 0; p < KP1; ++p)
            B1[p] = bload1(RX, xvoff, 2 * p * 128 + ((2 * p < D.cw) ? coff : 0));
        // Two accumulator sets used alternately: set (l & 1) receives layer l.  The bias and the first weight
        // group of layer l+1 are requested right after the last MFMA of layer l, before its epilogue (see
        // stream_prefetch): the other set is dead at that point (it was layer l's input).
        f32x16 acc2[2][HT];
        load_bias<HT>(W, (int)L.off_bf[0] * 4, acc2[0], lane);
        StreamPre pre = stream_prefetch<HT * (KP1 / 4)>(W, (int)L.off_wf[0] * 4, lane);
        stream_layer_pre<KP1 / 4, HT>(W, (int)L.off_wf[0] * 4, pre, [&](int k) { return B1[k]; }, acc2[0], lane);
        ESR_STAMP(1);
        f32x4 z4[NP4];
        float bias4[D.zrows];
#pragma unroll
        for (int l = 0; l < NHID; ++l) {
            f32x16 (&cur)[HT] = acc2[l & 1];
            f32x16 (&nxt)[HT] = acc2[(l + 1) & 1];
            if (l + 1 < NHID) {
                load_bias<HT>(W, (int)L.off_bf[l + 1] * 4, nxt, lane);
                pre = stream_prefetch<HT * HT * 4>(W, (int)L.off_wf[l + 1] * 4, lane);
            } else {
#pragma unroll
                for (int c = 0; c < D.zrows; ++c) bias4[c] = bload1(W, 0, ((int)L.off_b4 + c) * 4);
            }
            __builtin_amdgcn_sched_barrier(0);            // keep the requests in front of the epilogue's stores
            __builtin_amdgcn_s_setprio(3);                // (see the note on wave priorities above the kernel)
            relu_tiles<HT>(cur);
            if (save) {
                if (sv == 1) store_tiles<HT>(make_rsrc(A.H[l] + (size_t)t * (HBYTES / 4), HBYTES), cur, lane);
                store_relu_mask<HT>(make_rsrc(A.M[l] + (size_t)t * (MBYTES / 4), MBYTES), cur, lane);
            }
            __builtin_amdgcn_s_setprio(0);
            ESR_STAMP(2 + 2 * l);
            if (l + 1 < NHID)
                stream_layer_pre<HT * 4, HT>(W, (int)L.off_wf[l + 1] * 4, pre,
                                             [&](int k) { return cur[k >> 4][k & 15]; }, nxt, lane);
            else
                lds4_layer<HT, NP4>(w4, cur, z4, lane);
            ESR_STAMP(3 + 2 * l);
        }
        // each half of the wave holds the sum over ITS 16*HT units: add the halves, then the bias (rows >= out_dim
        // have zero weights and bias: the padding row of the output tile is written as 0)
        store_rows4<NP4, true, false>(make_rsrc(zdst + (size_t)t * D.zrows * 32, D.zrows * 32 * 4), 0, z4, bias4, lane);
    }
}

struct DgradArgs {
    const float *packed, *dz;
    int t0, t1;
    const unsigned *M[3];
    float *dZ[3];
    float *dX;
    // tiles t >= t_split run under `packed2` (esr_mlp_dgrad_fine: emissive net on the on-tiles, non-emissive net on the
    // off-tiles, one launch); packed2 == NULL: one net
    const float *packed2;
    int t_split;
};

template <int KIND>
__global__ void __launch_bounds__(256, mlp_occ(KIND)) mlp_dgrad_kernel(DgradArgs A)
{
    constexpr NetDesc D = net_desc(KIND);
    constexpr int NHID = D.n_layers - 1;
    constexpr int HT = D.hid_tiles;
    constexpr unsigned HBYTES = HT * 32 * 32 * 4, MBYTES = (HT / 2) * 256;
    constexpr PackLayout L = pack_layout(KIND);
    const int lane = esr_lane();
    const int h = lane >> 5, s = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const rsrc_t W1 = make_rsrc(A.packed, (unsigned)(L.total * 4));
    const rsrc_t W2 = make_rsrc(A.packed2 ? A.packed2 : A.packed, (unsigned)(L.total * 4));
    constexpr int NPX = L.n_passx;
    constexpr int NWX = NPX > 0 ? NPX * HT * 4 * 8 : 1;
    __shared__ float4 wx4s[2 * NWX];                            // first layer transposed, input rows >= 32 (lds4_layer), both nets
    if constexpr (NPX > 0) {
        lds4_preload<NWX>(W1, (int)L.off_wx4 * 4, wx4s);
        if (A.packed2) lds4_preload<NWX>(W2, (int)L.off_wx4 * 4, wx4s + NWX);
        __syncthreads();
    }
    for (int t = A.t0 + wave; t < A.t1; t += nwaves) {
        const bool second = A.packed2 && t >= A.t_split;         // wave-uniform
        const rsrc_t W = second ? W2 : W1;
        const float4 *wx4 = wx4s + (second ? NWX : 0);
        ESR_DSTAMP(0);
        const rsrc_t RZ = make_rsrc(A.dz + (size_t)t * D.zrows * 32, D.zrows * 32 * 4);
        float B0[4];                                                         // pair p <-> rows 2p, 2p+1
#pragma unroll
        for (int p = 0; p < 4; ++p) B0[p] = (2 * p < D.zrows) ? bload1(RZ, (h * 32 + s) * 4, 2 * p * 128) : 0.f;
        unsigned msk[NHID][HT / 2];                                          // all layers' ReLU masks up front
#pragma unroll
        for (int l = 0; l < NHID; ++l)
            load_relu_mask<HT>(make_rsrc(A.M[l] + (size_t)t * (MBYTES / 4), MBYTES), msk[l], lane);
        f32x16 cur[HT];
        zero_tiles<HT>(cur);
        layer_from_regs<4, HT>(W, (int)L.off_wb[NHID] * 4, B0, cur, lane);
        ESR_DSTAMP(1);
        __builtin_amdgcn_s_setprio(3);
        apply_relu_mask<HT>(msk[NHID - 1], cur);
        if (A.dZ[NHID - 1]) store_tiles<HT>(make_rsrc(A.dZ[NHID - 1] + (size_t)t * (HBYTES / 4), HBYTES), cur, lane);
        __builtin_amdgcn_s_setprio(0);
        ESR_DSTAMP(2);
#pragma unroll
        for (int l = NHID - 1; l >= 1; --l) {
            f32x16 nxt[HT];
            zero_tiles<HT>(nxt);
            layer_from_acc<HT, HT>(W, (int)L.off_wb[l] * 4, cur, nxt, lane);
            ESR_DSTAMP(3 + 2 * (NHID - 1 - l));
            __builtin_amdgcn_s_setprio(3);
            apply_relu_mask<HT>(msk[l - 1], nxt);
            if (A.dZ[l - 1]) store_tiles<HT>(make_rsrc(A.dZ[l - 1] + (size_t)t * (HBYTES / 4), HBYTES), nxt, lane);
            __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int it = 0; it < HT; ++it) cur[it] = nxt[it];
            ESR_DSTAMP(4 + 2 * (NHID - 1 - l));
        }
        // rows 0-31 as a 32x32 tile; the rows above that still lead to a grid (dx_rows) as 4-row 4x4x1 passes: a
        // second 32x32 tile would spend 96 MFMAs on 1 (tone mapper) to 11 (sample nets) useful rows
        f32x16 dx[1];
        zero_tiles<1>(dx);
        layer_from_acc<HT, 1>(W, (int)L.off_wb[0] * 4, cur, dx, lane);
        ESR_DSTAMP(7);
        store_tiles<1, false>(make_rsrc(A.dX + (size_t)t * 64 * 32, 64 * 32 * 4), dx, lane);    // (the scatter reads dX next)
        ESR_DSTAMP(8);
        if constexpr (NPX > 0) {
            f32x4 x4[NPX];
            lds4_layer<HT, NPX>(wx4, cur, x4, lane);
            ESR_DSTAMP(9);
            store_rows4<NPX, false, false>(make_rsrc(A.dX + (size_t)t * 64 * 32, 64 * 32 * 4), 32, x4, nullptr, lane);
            ESR_DSTAMP(10);
        }
    }
}

// dW[rowA][col(rowB)] += sum_samples A[rowA][s] * B[rowB][s];  db[rowA] += sum_s A[rowA][s]
struct WgradArgs {
    const float *A; int RA;      // [tiles][RA][32]
    const float *B; int RB;      // [tiles][RB][32]
    int t0, t1;
    float *gw; int ld; int out_rows;
    int kind; int first;         // column map: first layer uses in_colmap(kind, row)
    float *gb;
    float *slab;                 // [gridDim.x * WK][out_rows][ld] partial sums
    int b_tile_rows;             // rows per B tile in memory (>= RB; the X tile has extra colour rows)
    int crow;                    // first layer: B rows 0..cw-1 are read from X rows crow..crow+cw-1
    int cw8;                     // colour-group rows x 8 (float4 units per row)
    int wg0, nwg;                // workgroups [wg0, wg0 + nwg) of the launch work on this job
    int cfg;                     // kernel shape of this job (layer_cfg) -- read by the unified launch (mlp_wgrad_uni192_kernel)
    const float *amax;           // split-fp16 kernel: max |dz| of the step (device), the source of the gradient operand's scale
};

// One launch can carry up to MAX_JOBS jobs of the same kernel shape (the same layer of the emissive and the non-emissive
// net, both hidden layers of a net, the 3-row output layers of all three nets ...): each job gets a share of the 256
// workgroups proportional to its tiles.  Why: a weight-gradient launch has a FIXED cost of ~25-35 us whatever its
// tile count (tools/wgrad_scaling.py: 135 us per 4-layer net) -- the partial dW of every workgroup goes to a slab
// (256 x 147 KB written, then read by the reduction) and the ring has to fill and drain.  With J jobs per launch each
// job runs on 256/J workgroups: J times fewer, J times longer launches, and J times less slab traffic per layer.
// (16 since the end of round 4: the sixteen radiance jobs of an LTS flush -- two nets x four layers x two passes -- are one
//  launch; C5 2.97 -> 2.91 ms, C4 unchanged, A/B on one box with tools/variant.sh -DESR_MAX_JOBS=8.  The batch is a kernel
//  argument: 16 x 152 B, inside the 4 KB limit.)
#ifndef ESR_MAX_JOBS
#define ESR_MAX_JOBS 16
#endif
constexpr int MAX_JOBS = ESR_MAX_JOBS;
struct WgradBatch {
    int n;
    WgradArgs job[MAX_JOBS];
};
__device__ __forceinline__ WgradArgs pick_job(const WgradBatch &B)
{
    WgradArgs W = B.job[0];            // static indices only: scalar loads + selects, no private-memory copy of the batch
#pragma unroll
    for (int k = 1; k < MAX_JOBS; ++k)
        if (k < B.n && (int)blockIdx.x >= B.job[k].wg0) W = B.job[k];
    return W;
}

// Cooperative weight-gradient kernel.  One workgroup covers the WHOLE dW of a layer:
// wave (wm, wn, wk) owns the MI x NJ block of 32x32 output tiles at (wm, wn) and the
// wk-th share of each tile's 32 samples.  Per sample tile the workgroup stages A (dZ)
// and B (H / X) ONCE from HBM with fully coalesced 16-B loads into LDS (row stride 36
// floats: the 16 lanes of a ds_read_b128 group then hit 16 distinct 4-bank groups), and
// every wave reads its MFMA operands from LDS.  Global loads of tile t+1 are in flight
// while tile t is multiplied (register-staged, write-after-compute, one barrier per
// tile).  The predecessor of this kernel let every wave fetch its operand rows straight
// from global memory with a 128-B lane stride: 2x redundant traffic and 64 cache lines
// per load instruction -- 49 TF.  Accumulators (up to 144 registers) live across the
// whole tile range and are written once to the workgroup's slab (see the flush below).
constexpr int LDS_STRIDE = 36;        // floats per staged row (32 samples + 4 pad)

// (f32 operands: mlp_wgrad_dma_kernel below.)  MODE 1-3: bf16 operands (v_mfma_f32_32x32x16_bf16, 16 samples per k-step), fp32
// accumulation -- the weight-gradient kernel of the bf16 configurations, where the saved hidden tiles (H, dZ) are
// bf16 in the row-quad layout of store_tiles_bf16 (the stager turns 64-B pieces back into rows; LDS then holds
// [row][32] bf16 and a 16-B read IS an 8-sample operand) while the
// network input X and the output gradient dz are fp32 (rounded on the way from LDS to the operand registers):
//   1 = output layer (A = dz fp32, B = H bf16), 2 = hidden layer (both bf16), 3 = first layer (A = dZ bf16, B = X fp32).
template <int MI, int NJ, int WM, int WN, int WK, int MODE>
__device__ __forceinline__ void wgrad_reg_body(const WgradArgs &W)
{
    static_assert(MODE >= 1 && MODE <= 3, "f32 operands: mlp_wgrad_dma_kernel");
    constexpr bool A16 = MODE == 2 || MODE == 3, B16 = MODE == 1 || MODE == 2;
    constexpr int NW = WM * WN * WK, NT = 64 * NW;
    constexpr int RAP = WM * MI * 32, RBP = WN * NJ * 32;          // staged rows (padded to tiles)
    constexpr int BUF = (RAP + RBP) * LDS_STRIDE;                  // floats per LDS buffer
    constexpr int STRIDE16 = 20;                                   // floats per staged bf16 row (64 B + 16 B pad)
    constexpr int LA = A16 ? 4 : RAP * 8 / NT, LB = B16 ? 4 : RBP * 8 / NT;       // 16-B loads per thread (bf16: one 64-B unit)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, rl = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = w % WK, wn = (w / WK) % WN, wm = w / (WK * WN);
    const int nsplit = W.nwg, split = blockIdx.x - W.wg0;

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i) zero_tiles<NJ>(acc[i]);
    float bsum[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) bsum[i] = 0.f;

    // Three-stage pipeline: tile t is multiplied out of LDS while tile t+1 sits in registers waiting
    // to be committed and tile t+2's HBM loads are in flight (two register sets, static names).  With
    // one tile of prefetch the 192x96 and 4x192 layers were latency-bound: their MFMA work per tile
    // (<= 4.6k cycles) is shorter than an HBM round trip under load.
    float4 ga[2][LA], gb[2][LB];
    // Staging loads are range-checked BUFFER loads (out-of-range rows of a short operand read as
    // zero): no per-load branches, so the whole step is one basic block and hipcc keeps a COUNTED
    // vmcnt at the commit (with predicated plain loads it fell back to vmcnt(0) and drained the
    // prefetch every iteration).
    static_assert((A16 ? RAP <= NT : LA * NT == RAP * 8) && (B16 ? RBP <= NT : LB * NT == RBP * 8),
                  "staging covers the padded tiles exactly");
    // Past the end of the tile range the descriptor gets ZERO records: the loads still issue (so the
    // vmcnt arithmetic is the same on every trip) but touch no memory and return zeros.
    auto issue = [&](int t, float4 (&ra)[LA], float4 (&rb_)[LB]) {
        const bool live = t < W.t1;
        const int tc = live ? t : W.t0;
        constexpr unsigned RBA = A16 ? 64u : 128u, RBB = B16 ? 64u : 128u;          // bytes per stored row
        const rsrc_t SA = make_rsrc(reinterpret_cast<const char *>(W.A) + (size_t)tc * W.RA * RBA,
                                    live ? (unsigned)W.RA * RBA : 0u);
        const rsrc_t SB = make_rsrc(reinterpret_cast<const char *>(W.B) + (size_t)tc * W.b_tile_rows * RBB,
                                    live ? (unsigned)W.b_tile_rows * RBB : 0u);
        // bf16 operand (row-quad layout, mlp_common.h: store_tiles_bf16): thread u < rows takes quad u / 4, samples
        // 8 (u % 4) .. + 7 -- four 16-B pieces of 2 samples x 4 rows each, piece k at + 64 k (so that four neighbouring
        // threads read a full 64-byte run per load)
#pragma unroll
        for (int k = 0; k < LA; ++k)
            ra[k] = A16 ? bload4(SA, (tid >> 2) * 256 + k * 64 + (tid & 3) * 16, 0) : bload4(SA, (tid + k * NT) * 16, 0);
#pragma unroll
        for (int k = 0; k < LB; ++k) {
            if (B16) {
                rb_[k] = bload4(SB, (tid >> 2) * 256 + k * 64 + (tid & 3) * 16, 0);
            } else {
                const int q = tid + k * NT;                              // float4 index: row q/8
                // rows 0-5 of a first layer come from the net's colour group of the X tile
                const int src = (q < W.cw8) ? q + W.crow * 8 : q;
                rb_[k] = bload4(SB, src * 16, 0);   // (tiles shorter than the staged block read as zero)
            }
        }
    };
    // 2 samples x 4 rows per piece -> row r of the thread's quad: dword k = {row r of sample 2k, row r of sample 2k + 1}
    auto quad_rows = [&](float *Lrows, int rows, const float4 *p) {
        if (tid >= rows) return;                                         // one 64-byte unit per thread, `rows` units per tile
        const int quad = tid >> 2, blk = tid & 3;                        // LDS rows 4 quad .. + 3, 16-B piece blk of each
        unsigned w[4][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned d0 = __float_as_uint(p[k].x), d1 = __float_as_uint(p[k].y);      // sample 2k:     rows 01 | 23
            const unsigned d2 = __float_as_uint(p[k].z), d3 = __float_as_uint(p[k].w);      // sample 2k + 1: rows 01 | 23
            w[0][k] = __builtin_amdgcn_perm(d2, d0, 0x05040100u);        // low halves:  row 0
            w[1][k] = __builtin_amdgcn_perm(d2, d0, 0x07060302u);        // high halves: row 1
            w[2][k] = __builtin_amdgcn_perm(d3, d1, 0x05040100u);        // row 2
            w[3][k] = __builtin_amdgcn_perm(d3, d1, 0x07060302u);        // row 3
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            *reinterpret_cast<float4 *>(Lrows + (4 * quad + r) * STRIDE16 + blk * 4) =
                make_float4(__uint_as_float(w[r][0]), __uint_as_float(w[r][1]), __uint_as_float(w[r][2]), __uint_as_float(w[r][3]));
    };
    auto commit = [&](int buf, const float4 (&ra)[LA], const float4 (&rb_)[LB]) {
        float *La = lds + buf * BUF, *Lb = La + RAP * (A16 ? STRIDE16 : LDS_STRIDE);
        if (A16) {
            quad_rows(La, RAP, ra);
        } else {
#pragma unroll
            for (int k = 0; k < LA; ++k) {
                const int q = tid + k * NT;
                *reinterpret_cast<float4 *>(La + (q >> 3) * LDS_STRIDE + (q & 7) * 4) = ra[k];
            }
        }
        if (B16) {
            quad_rows(Lb, RBP, rb_);
        } else {
#pragma unroll
            for (int k = 0; k < LB; ++k) {
                const int q = tid + k * NT;
                *reinterpret_cast<float4 *>(Lb + (q >> 3) * LDS_STRIDE + (q & 7) * 4) = rb_[k];
            }
        }
    };
    auto compute = [&](int cur) {
        {
            static_assert(WK <= 2, "a tile has two 16-sample k-steps");
            constexpr int NU16 = 2 / WK;
            // fp32-staged operand: row stride LDS_STRIDE floats, 8 samples = two float4; bf16-staged: STRIDE16
            // floats per row, 8 samples = one 16-B read that IS the MFMA operand
            constexpr int SA_ = A16 ? STRIDE16 : LDS_STRIDE, SB_ = B16 ? STRIDE16 : LDS_STRIDE;
            const float *La = lds + cur * BUF + (wm * MI * 32 + rl) * SA_ + (A16 ? 4 : 8) * h;
            const float *Lb = lds + cur * BUF + RAP * SA_ + (wn * NJ * 32 + rl) * SB_ + (B16 ? 4 : 8) * h;
#pragma unroll
            for (int uu = 0; uu < NU16; ++uu) {
                const int u = wk * NU16 + uu;
                bf16x8 a[MI], b[NJ];
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    if (A16) {
                        a[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4 *>(La + i * 32 * SA_ + 8 * u));
                        float t8 = 0.f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) t8 += (float)a[i][e];
                        bsum[i] += t8;
                    } else {
                        const float4 lo = *reinterpret_cast<const float4 *>(La + i * 32 * SA_ + 16 * u);
                        const float4 hi = *reinterpret_cast<const float4 *>(La + i * 32 * SA_ + 16 * u + 4);
                        bsum[i] += ((lo.x + lo.y) + (lo.z + lo.w)) + ((hi.x + hi.y) + (hi.z + hi.w));
                        a[i] = pack8(lo, hi);
                    }
                }
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if (B16)
                        b[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const float4 *>(Lb + j * 32 * SB_ + 8 * u));
                    else
                        b[j] = pack8(*reinterpret_cast<const float4 *>(Lb + j * 32 * SB_ + 16 * u),
                                     *reinterpret_cast<const float4 *>(Lb + j * 32 * SB_ + 16 * u + 4));
                }
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = mfma16(a[i], b[j], acc[i][j]);
            }
        }
    };

    int t = W.t0 + split;
    issue(t, ga[0], gb[0]);
    commit(0, ga[0], gb[0]);
    issue(t + nsplit, ga[1], gb[1]);                                 // tile t+1 -> register set 1
    __syncthreads();
    // invariant at the top of a step on tile t: LDS buffer `cur` holds t, register set `nx` holds t+1
    for (int cur = 0; t < W.t1;) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {                       // unrolled by 2: static register sets
            if (t < W.t1) {
                const int nx = half ^ 1, fr = half;                  // set holding t+1 / set free for t+2
                issue(t + 2 * nsplit, ga[fr], gb[fr]);               // unconditional: counted vmcnt below
                compute(cur);
                commit(cur ^ 1, ga[nx], gb[nx]);                     // (zeros past the end: never read)
                __syncthreads();
                cur ^= 1;
                t += nsplit;
            }
        }
    }
    // flush: accumulator column = B row (lane), accumulator row = A row (register).  Every wave
    // stores its block into its workgroup's private slab with plain 128-B-contiguous stores; a
    // second tiny kernel sums the slabs.  (1024 waves atomically adding into the same 147 KB at
    // kernel end ran at a fraction of the atomic rate: same-address contention.)
    float *S = W.slab + (size_t)(split * WK + wk) * W.out_rows * W.ld;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int rb = 32 * (NJ * wn + j) + rl;
        const int col = (rb < W.RB) ? (W.first ? in_colmap(W.kind, rb) : rb) : -1;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ra = 32 * (MI * wm + i) + acc_row(r, h);
                if (col >= 0 && col < W.ld && ra < W.out_rows) S[(size_t)ra * W.ld + col] = acc[i][j][r];
            }
    }
    if (W.gb && wn == 0) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int ra = 32 * (MI * wm + i) + rl;
            const float tot = bsum[i] + __shfl_xor(bsum[i], 32);
            if (h == 0 && ra < W.out_rows) atomicAdd(&W.gb[ra], tot);
        }
    }
}

template <int MI, int NJ, int WM, int WN, int WK, int MODE>
__global__ void __launch_bounds__(64 * WM * WN * WK, 1) mlp_wgrad_kernel(WgradBatch WB)
{
    wgrad_reg_body<MI, NJ, WM, WN, WK, MODE>(pick_job(WB));
}

// ---- f32 weight gradients, operands staged by LDS-DMA -------------------------------------------------
// Same decomposition as mlp_wgrad_kernel (one workgroup = the whole dW of a layer over a tile range, wave
// (wm, wn, wk) owns an MI x NJ block of 32x32 accumulators), but the two operand tiles of a sample tile go
// from HBM straight into LDS (`buffer_load_dwordx4 ... lds`, 1 KiB = 8 rows per wave instruction) instead of
// through staging registers.  Why: in the register-staged kernel hipcc re-used the staging registers of the
// tile in flight for LDS-read operands and had to wait for those loads first (`s_waitcnt vmcnt(11..8)` at
// the top of every step, i.e. a prefetch distance of half a tile instead of two): 68 % matrix-core busy at
// 352 VGPRs.  Here nothing in the loop has a VGPR destination in memory, the waits are counted by hand
// (cdna_hip_programming.md: LDS-DMA + counted vmcnt + raw s_barrier), and 96 staging registers are gone.
//
// LDS image: 3 buffers x (RAP + RBP) rows x 128 B, unpadded (an LDS-DMA piece is lane-linear).  Bank
// conflicts of the row-per-lane ds_read_b128 are removed by an XOR swizzle applied on the SOURCE side:
// 16-B chunk c of row r is stored at chunk position c ^ ((r >> 1) & 7), so 16 consecutive rows read at the
// same logical chunk hit 16 distinct 4-bank groups.
// Three buffers, one barrier per tile in the MIDDLE of the tile (schedule: comment above the main loop).
//
// SPLIT: the same staging, fp32 operands in HBM and LDS, but the products run on the 16-bit matrix cores
// (v_mfma_f32_32x32x16_f16, 16x the f32 MFMA rate): every operand value is cut into two fp16 planes on its way from LDS to
// the operand registers, x = x1 + x2 with x1 = fp16(x) and x2 = fp16(x - x1) (the subtraction is exact), and
//   A.B = A1.B1 + A1.B2 + A2.B1   (+ A2.B2, < 2^-22 relative: dropped)
// goes into ONE fp32 accumulator.  The gradient operand A (dZ / dz, ~1e-3 .. 1e-7) is first multiplied by a power of two
// s chosen from the step's max |dz| (W.amax) so that max |dz| s is in [2^7, 2^8): fp16's range then holds 128x that
// maximum (hidden-layer gradients exceed the output gradient by the weights' row sums), and a residual x2 below fp16's
// normal range (|x| s < 2^-3) is rounded to 2^-25 ABSOLUTE, 2^-32 of the maximum -- far below the fp32 accumulation's own
// error.  The activation operand B (H / X, O(1)) is not scaled: a residual below the normal range means |x| < 0.125 and
// an absolute error of 3e-8.  The accumulator is divided by s at the flush; the bias gradient sums the raw fp32 values.
// Accuracy: tests/test_gpu_split.py (vs float64: as the f32 MFMA kernel).  Cost: 8 samples per lane and operand row are
// converted by ~4 VALU instructions each -- the kernel is HBM-bound with them.
typedef _Float16 wg_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void wg_split8(const float4 &lo, const float4 &hi, float s, wg_f16x8 &p1, wg_f16x8 &p2)
{
    // (C form: an asm form of four instructions per value pair made the loop 13 % shorter and the launch no faster -- it waits for HBM)
    const float v[8] = {lo.x * s, lo.y * s, lo.z * s, lo.w * s, hi.x * s, hi.y * s, hi.z * s, hi.w * s};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const _Float16 h1 = (_Float16)v[e];
        p1[e] = h1;
        p2[e] = (_Float16)(v[e] - (float)h1);
    }
}
__device__ __forceinline__ f32x16 wg_mfma_f16(wg_f16x8 a, wg_f16x8 b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
// power of two s with amax * s in [2^7, 2^8) (amax == 0 or not finite: 1)
__device__ __forceinline__ float wg_scale(float amax)
{
    if (!(amax > 0.f) || !(amax < 3.0e38f)) return 1.f;
    int e;
    frexpf(amax, &e);                     // amax = m 2^e, m in [0.5, 1)
    e = 8 - e;
    e = e > 120 ? 120 : e < -120 ? -120 : e;
    return ldexpf(1.f, e);
}

template <int MI, int NJ, int WM, int WN, int WK, bool SPLIT = false>
__device__ __forceinline__ void wgrad_dma_body(const WgradArgs &W)
{
    constexpr int NW = WM * WN * WK;                 // compute waves (one per SIMD); wave NW is the loader
    constexpr int RAP = WM * MI * 32, RBP = WN * NJ * 32, ROWS = RAP + RBP;
    constexpr int PIECES = ROWS / 8;                 // 1-KiB pieces (8 rows x 128 B) per tile
    static_assert(PIECES <= 60, "the loader counts a whole tile on vmcnt");
    constexpr int BUF = ROWS * 32;                   // floats per LDS buffer
    constexpr int NU = 4 / WK;                       // 8-sample groups per compute wave per tile
    extern __shared__ __attribute__((aligned(16))) float lds[];
    typedef __attribute__((address_space(3))) void lds_void;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, rl = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = w % WK, wn = (w / WK) % WN, wm = w / (WK * WN);
    const int nsplit = W.nwg, split = blockIdx.x - W.wg0;

    // rows that no DMA ever fills (operands shorter than the staged block) must not hold NaN patterns
    for (int i = tid; i < 3 * BUF; i += 64 * (NW + 1)) lds[i] = 0.f;
    __syncthreads();

    // Barrier schedule, identical in every wave: one before the first tile, then one in the MIDDLE of every tile.
    //   barrier before tile 0:   tile 0 has landed
    //   barrier inside tile i:   tile i+1 has landed (the loader waited for it) and every compute wave is done with
    //                            tile i-1, whose buffer the loader refills with tile i+2 right after the barrier
    if (w == NW) {
        // ---- loader wave: streams whole tiles HBM -> LDS, nothing else.  A compute wave that issued its own pieces
        // paid 60-180 cycles of matrix-pipe idle per LDS-DMA instruction (in-order issue: the next MFMA waits behind
        // it); from a wave of its own the same instructions interleave with the MFMAs of the SIMD's compute wave.
        const unsigned lds0 = (unsigned)(uintptr_t)(lds_void *)lds;
        const int cw = W.cw8 / 8;
        auto load_tile = [&](int t_next, int fb) {
            const bool live = t_next < W.t1;                     // past the range: zero records, no memory touched
            const int tc = t_next < W.t1 ? t_next : W.t0;
            const u32x4 SA = raw_rsrc(reinterpret_cast<const char *>(W.A) + (size_t)tc * W.RA * 128u,
                                      live ? (unsigned)W.RA * 128u : 0u);
            const u32x4 SB = raw_rsrc(reinterpret_cast<const char *>(W.B) + (size_t)tc * W.b_tile_rows * 128u,
                                      live ? (unsigned)W.b_tile_rows * 128u : 0u);
#pragma unroll
            for (int p = 0; p < PIECES; ++p) {
                const int R = 8 * p + (lane >> 3);                       // row of the staged image
                const int c = (lane & 7) ^ ((R >> 1) & 7);               // source chunk of this LDS slot
                int row = R;                                             // A rows >= RA fail the range check
                if (8 * p >= RAP) row = (R - RAP < cw) ? R - RAP + W.crow : R - RAP;   // first layer: colour rows
                lds_dma16(8 * p < RAP ? SA : SB, lds0 + (unsigned)(fb * BUF + p * 256) * 4u, row * 128 + c * 16);
            }
        };
        int t = W.t0 + split;
        load_tile(t, 0);
        load_tile(t + nsplit, 1);
        __builtin_amdgcn_s_waitcnt(wait_vm(PIECES));
        __builtin_amdgcn_s_barrier();
        for (int fb = 2; t < W.t1; t += nsplit) {
            __builtin_amdgcn_s_waitcnt(wait_vm(0));
            __builtin_amdgcn_s_barrier();
            load_tile(t + 2 * nsplit, fb);
            fb = fb == 2 ? 0 : fb + 1;
        }
        __builtin_amdgcn_s_waitcnt(wait_vm(0));          // nothing may land in LDS after the workgroup is gone
        return;
    }

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i) zero_tiles<NJ>(acc[i]);
    float bsum[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) bsum[i] = 0.f;
    // operand reads of one 8-sample group: row (wm*MI*32 + i*32 + rl), logical chunk 2u + h, swizzled by the row
    const int swz = (rl >> 1) & 7;
    const int a_row = (wm * MI * 32 + rl) * 32, b_row = (RAP + wn * NJ * 32 + rl) * 32;
    auto lds_read = [&](int buf, int uu, float4 (&a)[MI], float4 (&b)[NJ]) {
        const int ch = ((2 * (wk * NU + uu) + h) ^ swz) * 4;
        const float *La = lds + buf * BUF + a_row + ch, *Lb = lds + buf * BUF + b_row + ch;
#pragma unroll
        for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const float4 *>(La + i * 32 * 32);
#pragma unroll
        for (int j = 0; j < NJ; ++j) b[j] = *reinterpret_cast<const float4 *>(Lb + j * 32 * 32);
    };
    // Operands are register double-buffered by group; the last group of a tile reads the first group of the next
    // tile (readable since the mid-tile barrier), so no LDS latency is exposed at the tile seam.
    static_assert(NU == 2 || NU == 4, "two halves of whole groups");
    float osc = 1.f;                                      // SPLIT: 1 / (scale of the gradient operand)
    if constexpr (SPLIT) {
        // one k-step of the 16-bit MFMA = two 8-sample groups (lane half h: chunks 2u + h of both, the same samples for A
        // and B).  Per k-step: the A rows of the wave become planes once (MI x 2 x 4 registers), the B rows per output
        // column block j; raw fp32 operands are prefetched one block ahead (A: one k-step ahead).
        constexpr int KS = NU / 2;
        const float sc = wg_scale(W.amax ? *W.amax : 0.f);
        osc = 1.f / sc;
        auto read2 = [&](int buf, int ks, int row0, float4 &lo, float4 &hi) {
            const int c0 = ((2 * (wk * NU + 2 * ks) + h) ^ swz) * 4, c1 = ((2 * (wk * NU + 2 * ks + 1) + h) ^ swz) * 4;
            lo = *reinterpret_cast<const float4 *>(lds + buf * BUF + row0 + c0);
            hi = *reinterpret_cast<const float4 *>(lds + buf * BUF + row0 + c1);
        };
        float4 ral[MI], rah[MI], rbl[2], rbh[2];
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < MI; ++i) read2(0, 0, a_row + i * 32 * 32, ral[i], rah[i]);
        read2(0, 0, b_row, rbl[0], rbh[0]);
        int t = W.t0 + split;
        for (int buf = 0; t < W.t1; t += nsplit) {
            const int nb = buf == 2 ? 0 : buf + 1;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks == KS - 1) __builtin_amdgcn_s_barrier();          // the next tile has landed (see the schedule above)
                const int nbuf = ks + 1 < KS ? buf : nb, nks = ks + 1 < KS ? ks + 1 : 0;
                wg_f16x8 a1[MI], a2[MI];
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    bsum[i] += ((ral[i].x + ral[i].y) + (ral[i].z + ral[i].w)) + ((rah[i].x + rah[i].y) + (rah[i].z + rah[i].w));
                    wg_split8(ral[i], rah[i], sc, a1[i], a2[i]);
                }
#pragma unroll
                for (int i = 0; i < MI; ++i) read2(nbuf, nks, a_row + i * 32 * 32, ral[i], rah[i]); // (stale but unused after the last tile)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    wg_f16x8 b1, b2;
                    wg_split8(rbl[j & 1], rbh[j & 1], 1.f, b1, b2);
                    // raw rows of the next block (next k-step: slot 0 again) -- in flight under this block's MFMAs
                    if (j + 1 < NJ) read2(buf, ks, b_row + (j + 1) * 32 * 32, rbl[(j + 1) & 1], rbh[(j + 1) & 1]);
                    else read2(nbuf, nks, b_row, rbl[0], rbh[0]);
#pragma unroll
                    for (int i = 0; i < MI; ++i) {
                        acc[i][j] = wg_mfma_f16(a1[i], b1, acc[i][j]);
                        acc[i][j] = wg_mfma_f16(a1[i], b2, acc[i][j]);
                        acc[i][j] = wg_mfma_f16(a2[i], b1, acc[i][j]);
                    }
                }
            }
            buf = nb;
        }
    } else {
    float4 a0[MI], b0[NJ], a1[MI], b1[NJ];
    __builtin_amdgcn_s_barrier();
    lds_read(0, 0, a0, b0);
    int t = W.t0 + split;
    for (int buf = 0; t < W.t1; t += nsplit) {
        const int nb = buf == 2 ? 0 : buf + 1;
#pragma unroll
        for (int uu = 0; uu < NU; ++uu) {
            float4 (&a)[MI] = (uu & 1) ? a1 : a0;
            float4 (&b)[NJ] = (uu & 1) ? b1 : b0;
            float4 (&an)[MI] = (uu & 1) ? a0 : a1;
            float4 (&bn)[NJ] = (uu & 1) ? b0 : b1;
            if (uu == NU / 2) __builtin_amdgcn_s_barrier();
            if (uu + 1 < NU) lds_read(buf, uu + 1, an, bn);
            else lds_read(nb, 0, an, bn);                 // (stale but unused after the last tile)
            __builtin_amdgcn_sched_barrier(0);            // or hipcc sinks the reads next to their first use
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                bsum[i] += (a[i].x + a[i].y) + (a[i].z + a[i].w);
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    acc[i][j] = mfma32(a[i].x, b[j].x, acc[i][j]);
                    acc[i][j] = mfma32(a[i].y, b[j].y, acc[i][j]);
                    acc[i][j] = mfma32(a[i].z, b[j].z, acc[i][j]);
                    acc[i][j] = mfma32(a[i].w, b[j].w, acc[i][j]);
                }
            }
        }
        buf = nb;
    }
    }
    __builtin_amdgcn_s_waitcnt(wait_vm_lgkm0(0));

    float *S = W.slab + (size_t)(split * WK + wk) * W.out_rows * W.ld;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int rb = 32 * (NJ * wn + j) + rl;
        const int col = (rb < W.RB) ? (W.first ? in_colmap(W.kind, rb) : rb) : -1;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ra = 32 * (MI * wm + i) + acc_row(r, h);
                if (col >= 0 && col < W.ld && ra < W.out_rows) S[(size_t)ra * W.ld + col] = SPLIT ? acc[i][j][r] * osc : acc[i][j][r];
            }
    }
    if (W.gb && wn == 0) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int ra = 32 * (MI * wm + i) + rl;
            const float tot = bsum[i] + __shfl_xor(bsum[i], 32);
            if (h == 0 && ra < W.out_rows) atomicAdd(&W.gb[ra], tot);
        }
    }
}

template <int MI, int NJ, int WM, int WN, int WK, bool SPLIT = false>
__global__ void __launch_bounds__(64 * (WM * WN * WK + 1), 1) mlp_wgrad_dma_kernel(WgradBatch WB)
{
    wgrad_dma_body<MI, NJ, WM, WN, WK, SPLIT>(pick_job(WB));
}

// Every f32 weight-gradient job of the 192-wide nets of a step in ONE launch: hidden layers (192 x 192), first layers
// (192 x <= 96) and the 3-row output layers each keep their own shape (all three are 4 compute waves + the loader
// wave; a workgroup works on exactly one job, so the switch is workgroup-uniform) and get a share of the 256 workgroups
// proportional to tiles x cost per tile.  Why: every launch pays a fixed 25-35 us (slab round trip, ring fill and
// drain, tail) whatever its tile count -- three launches + three reductions per step became one + one.
// (round 4, end: the 128-wide nets' shapes -- the BRDF / emission jobs of an LTS step -- ride in the same launches: three more
//  launches + their fill / drain per flush gone)
enum { UNI_HID192 = 0, UNI_FIRST192 = 1, UNI_OUT192 = 2, UNI_FIRST192_X16 = 3, UNI_HID128 = 4, UNI_FIRST128 = 5, UNI_OUT128 = 6,
       N_UNI_CFG = 7 };
__global__ void __launch_bounds__(320, 1) mlp_wgrad_uni192_kernel(WgradBatch WB)
{
    const WgradArgs W = pick_job(WB);
    if (W.cfg == UNI_HID192) wgrad_dma_body<3, 3, 2, 2, 1>(W);
    else if (W.cfg == UNI_FIRST192) wgrad_dma_body<3, 3, 2, 1, 2>(W);
    else if (W.cfg == UNI_HID128) wgrad_dma_body<2, 2, 2, 2, 1>(W);
    else if (W.cfg == UNI_FIRST128) wgrad_dma_body<2, 3, 2, 1, 2>(W);
    else if (W.cfg == UNI_OUT128) wgrad_dma_body<1, 2, 1, 2, 2>(W);
    else wgrad_dma_body<1, 3, 1, 2, 2>(W);
}
// the bf16 engine's 192-wide jobs the same way (register-staged bodies, four waves each; UNI_FIRST192_X16: the first layer
// reading the bf16 input tile).  Three launches + their fill / drain became one.
__global__ void __launch_bounds__(256, 1) mlp_wgrad_uni192b_kernel(WgradBatch WB)
{
    const WgradArgs W = pick_job(WB);
    if (W.cfg == UNI_HID192) wgrad_reg_body<3, 3, 2, 2, 1, 2>(W);
    else if (W.cfg == UNI_FIRST192) wgrad_reg_body<3, 3, 2, 1, 2, 3>(W);
    else if (W.cfg == UNI_FIRST192_X16) wgrad_reg_body<3, 3, 2, 1, 2, 2>(W);
    else if (W.cfg == UNI_HID128) wgrad_reg_body<2, 2, 2, 2, 1, 2>(W);
    else if (W.cfg == UNI_FIRST128) wgrad_reg_body<2, 3, 2, 1, 2, 3>(W);
    else if (W.cfg == UNI_OUT128) wgrad_reg_body<1, 2, 1, 2, 2, 1>(W);
    else wgrad_reg_body<1, 3, 1, 2, 2, 1>(W);
}
// the same launch with the products on the 16-bit matrix cores (wgrad_dma_body<..., SPLIT>)
__global__ void __launch_bounds__(320, 1) mlp_wgrad_uni192s_kernel(WgradBatch WB)
{
    const WgradArgs W = pick_job(WB);
    if (W.cfg == UNI_HID192) wgrad_dma_body<3, 3, 2, 2, 1, true>(W);
    else if (W.cfg == UNI_FIRST192) wgrad_dma_body<3, 3, 2, 1, 2, true>(W);
    else if (W.cfg == UNI_HID128) wgrad_dma_body<2, 2, 2, 2, 1, true>(W);
    else if (W.cfg == UNI_FIRST128) wgrad_dma_body<2, 3, 2, 1, 2, true>(W);
    else if (W.cfg == UNI_OUT128) wgrad_dma_body<1, 2, 1, 2, 2, true>(W);
    else wgrad_dma_body<1, 3, 1, 2, 2, true>(W);
}

// gw[e] += sum over the partial slabs of every job of a launch; 32 slabs per thread, groups combined with one atomic
struct ReduceArgs {
    int nseg;
    const float *slab[MAX_JOBS];
    int n_partials[MAX_JOBS], n_elems[MAX_JOBS];
    float *gw[MAX_JOBS];
    int64_t first[MAX_JOBS + 1];   // work-item prefix: segment l owns items [first[l], first[l+1])
};
constexpr int REDUCE_PG = 32;

__global__ void __launch_bounds__(256) wgrad_reduce_kernel(ReduceArgs R)
{
    const int64_t total = R.first[R.nseg];
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        int l = 0;
#pragma unroll
        for (int k = 1; k < MAX_JOBS; ++k)
            if (k < R.nseg && i >= R.first[k]) l = k;
        const int64_t j = i - R.first[l];
        const int n_elems = R.n_elems[l], n_partials = R.n_partials[l];
        const int e = (int)(j % n_elems), g = (int)(j / n_elems);
        const int p1 = min((g + 1) * REDUCE_PG, n_partials);
        const float *slab = R.slab[l];
        float acc = 0.f;
        for (int p = g * REDUCE_PG; p < p1; ++p) acc += slab[(size_t)p * n_elems + e];
        atomicAdd(&R.gw[l][e], acc);
    }
}

int mlp_grid(int n_tiles, int occ = 2)
{
    // `occ` workgroups (4 waves each) per CU resident; one tile per wave per trip
    int wg = (n_tiles + 3) / 4;
    const int cap = 256 * occ;
    if (wg > cap) wg = cap;
    if (wg < 1) wg = 1;
    return wg;
}

// Share of the launch's workgroups (one per CU) for every job, proportional to its tiles; slab regions back to back.
template <int WK>
int plan_batch(WgradBatch &B, float *scratch, int64_t slab_floats, ReduceArgs &R, int64_t &used_out)
{
    int64_t tiles = 0;
    for (int j = 0; j < B.n; ++j) tiles += B.job[j].t1 - B.job[j].t0;
    const int grid = tiles < 256 ? (int)tiles : 256;
    int given = 0;
    for (int j = 0; j < B.n; ++j) {
        const int tj = B.job[j].t1 - B.job[j].t0;
        int n = (int)((int64_t)grid * tj / tiles);
        if (n < 1) n = 1;
        B.job[j].nwg = n;
        given += n;
    }
    for (int j = 0; given != grid; j = (j + 1) % B.n) {            // hand out / take back the rounding remainder
        WgradArgs &W = B.job[j];
        if (given < grid && W.nwg < W.t1 - W.t0) { ++W.nwg; ++given; }
        else if (given > grid && W.nwg > 1) { --W.nwg; --given; }
    }
    int64_t used = 0;
    int wg0 = 0;
    R = ReduceArgs{};
    for (int j = 0; j < B.n; ++j) {
        WgradArgs &W = B.job[j];
        W.wg0 = wg0;
        wg0 += W.nwg;
        const int n_elems = W.out_rows * W.ld;
        W.slab = scratch + used;
        used += (int64_t)W.nwg * WK * n_elems;
        R.slab[j] = W.slab; R.n_partials[j] = W.nwg * WK; R.n_elems[j] = n_elems; R.gw[j] = W.gw;
        R.first[j + 1] = R.first[j] + (int64_t)n_elems * ((W.nwg * WK + REDUCE_PG - 1) / REDUCE_PG);
    }
    R.nseg = B.n;
    used_out = used;
    if (used > slab_floats) return ESR_ECAP;
    return wg0;                                                     // = grid
}

int launch_reduce(const ReduceArgs &R, hipStream_t s)
{
    wgrad_reduce_kernel<<<esr_grid_for(R.first[R.nseg], 256, 2048), 256, 0, s>>>(R);
    ESR_CHECK_LAUNCH();
    return 0;
}

// The slab workspace of one esr_mlp_wgrad_batch call.  Every weight-gradient launch used to be followed by its own
// reduction (12-15 us each, and the next launch could not start before it had finished): the launches of a call now take
// consecutive regions of the workspace and ONE reduction (segments of all launches, <= MAX_JOBS per reduce launch) runs
// at the end -- or earlier, when the next launch's slabs no longer fit behind the pending ones.
struct SlabPool {
    float *base;
    int64_t total, used;
    hipStream_t s;
    ReduceArgs pending[4 * MAX_JOBS];
    int n_pending;
    SlabPool(float *b, int64_t t, hipStream_t st) : base(b), total(t), used(0), s(st), n_pending(0) {}
    int flush()
    {
        ReduceArgs M = {};
        for (int q = 0; q < n_pending; ++q) {
            const ReduceArgs &R = pending[q];
            for (int j = 0; j < R.nseg; ++j) {
                if (M.nseg == MAX_JOBS) {
                    if (int rc = launch_reduce(M, s)) return rc;
                    M = ReduceArgs{};
                }
                const int k = M.nseg++;
                M.slab[k] = R.slab[j]; M.n_partials[k] = R.n_partials[j]; M.n_elems[k] = R.n_elems[j]; M.gw[k] = R.gw[j];
                M.first[k + 1] = M.first[k] + (R.first[j + 1] - R.first[j]);
            }
        }
        n_pending = 0;
        used = 0;
        return M.nseg ? launch_reduce(M, s) : 0;
    }
    // plan(scratch, floats, R, used) -> grid or error: behind the pending slabs if it fits, else after a flush
    template <class Plan>
    int place(Plan plan, ReduceArgs &R, int64_t &n)
    {
        int grid = plan(base + used, total - used, R, n);
        if (grid == ESR_ECAP && used > 0) {
            if (int rc = flush()) return rc;
            grid = plan(base, total, R, n);
        }
        return grid;
    }
    int launched(const ReduceArgs &R, int64_t n)
    {
        pending[n_pending++] = R;
        used += n;
        return n_pending == 4 * MAX_JOBS ? flush() : 0;
    }
};

template <int MI, int NJ, int WM, int WN, int WK, int MODE = 0>
int launch_wgrad(WgradBatch &B, SlabPool &P)
{
    hipStream_t s = P.s;
    for (int j = 0; j < B.n; ++j)
        if (B.job[j].RA > WM * MI * 32 || B.job[j].RB > WN * NJ * 32) return ESR_ECAP;
    constexpr int NT = 64 * WM * WN * WK;
    constexpr size_t lds_bytes = 2 * (size_t)(WM * MI * 32 + WN * NJ * 32) * LDS_STRIDE * sizeof(float);
    static std::atomic<uint64_t> optin{0};
    if (int rc = esr_lds_optin(reinterpret_cast<const void *>(&mlp_wgrad_kernel<MI, NJ, WM, WN, WK, MODE>), lds_bytes, optin))
        return rc;
    ReduceArgs R;
    int64_t n = 0;
    const int grid = P.place([&](float *sc, int64_t fl, ReduceArgs &R_, int64_t &n_) { return plan_batch<WK>(B, sc, fl, R_, n_); }, R, n);     // one workgroup per CU (LDS-bound residency)
    if (grid < 0) return grid;
    mlp_wgrad_kernel<MI, NJ, WM, WN, WK, MODE><<<grid, NT, lds_bytes, s>>>(B);
    ESR_CHECK_LAUNCH();
    return P.launched(R, n);
}

template <int MI, int NJ, int WM, int WN, int WK, bool SPLIT = false>
int launch_wgrad_dma(WgradBatch &B, SlabPool &P)
{
    hipStream_t s = P.s;
    for (int j = 0; j < B.n; ++j)
        if (B.job[j].RA > WM * MI * 32 || B.job[j].RB > WN * NJ * 32) return ESR_ECAP;
    constexpr int NT = 64 * (WM * WN * WK + 1);           // compute waves + the loader wave
    constexpr size_t lds_bytes = 3 * (size_t)(WM * MI * 32 + WN * NJ * 32) * 32 * sizeof(float);
    static_assert(lds_bytes <= 160 * 1024, "three staged tiles must fit the CU's LDS");
    static std::atomic<uint64_t> optin{0};
    if (int rc = esr_lds_optin(reinterpret_cast<const void *>(&mlp_wgrad_dma_kernel<MI, NJ, WM, WN, WK, SPLIT>), lds_bytes, optin))
        return rc;
    ReduceArgs R;
    int64_t n = 0;
    const int grid = P.place([&](float *sc, int64_t fl, ReduceArgs &R_, int64_t &n_) { return plan_batch<WK>(B, sc, fl, R_, n_); }, R, n);     // one workgroup per CU (LDS-bound residency)
    if (grid < 0) return grid;
    mlp_wgrad_dma_kernel<MI, NJ, WM, WN, WK, SPLIT><<<grid, NT, lds_bytes, s>>>(B);
    ESR_CHECK_LAUNCH();
    return P.launched(R, n);
}

// f32: LDS-DMA staging (MODE 4: products on the 16-bit matrix cores from split planes); bf16 operand modes: the register-staged kernel
template <int MI, int NJ, int WM, int WN, int WK, int MODE>
int launch_wgrad_any(WgradBatch &B, SlabPool &P)
{
    if constexpr (MODE == 0) return launch_wgrad_dma<MI, NJ, WM, WN, WK>(B, P);
    else if constexpr (MODE == 4) return launch_wgrad_dma<MI, NJ, WM, WN, WK, true>(B, P);
    else return launch_wgrad<MI, NJ, WM, WN, WK, MODE>(B, P);
}

// ---- unified launch of the 192-wide f32 jobs --------------------------------------------------------------
// relative cost of one sample tile per job shape (hidden : first : output), from the per-shape launches' times at C2
// (4.4 : 2.2 : 0.8 us per tile per workgroup; the output shape is bound by its loader wave, not by matrix work);
// The split-fp16 launch is bound by the bytes it streams, not by matrix work: the shares follow the tile sizes
// (48 : 36 : 25 KB per sample tile; measured at C2: "1,0.85,0.6" 0.466 ms, "1,1,0.6" 0.480-0.489, the f32 table 0.850; C4: 0.954 / 0.986 / -).
// variant: 0 = f32 MFMA, 1 = split fp16 planes, 2 = bf16 operands (bytes per tile: hidden 24.6 KB, first layer 18-24 KB,
// output layer 12.8 KB); the fourth entry is the bf16 first layer on the bf16 input tile
const double *uni_cost(int variant)
{
    // (fifth to seventh: the 128-wide nets' hidden / first / output shapes -- bytes per tile 0.67 / 0.58 / 0.35 of the 192-wide
    //  hidden shape's, MACs 0.44 / 0.33 / 0.03)
    static const double c[3][N_UNI_CFG] = {{1.0, 0.5, 0.2, 0.5, 0.45, 0.35, 0.15}, {1.0, 0.8, 0.6, 0.8, 0.65, 0.55, 0.4},
                                     {1.0, 1.0, 0.7, 0.9, 0.65, 0.6, 0.45}};
    return c[variant];
}
constexpr int uni_wk(int cfg) { return (cfg == UNI_HID192 || cfg == UNI_HID128) ? 1 : 2; }      // (both first-layer forms and the output layer: k split in two)

// workgroups per job proportional to tiles x cost (every job >= 1, none more than its tiles); slab regions back to back
int plan_uni(WgradBatch &B, float *scratch, int64_t slab_floats, ReduceArgs &R, int64_t &used_out, int variant)
{
    const double *cost = uni_cost(variant);
    double w[MAX_JOBS], wt = 0.0;
    int64_t tiles = 0;
    for (int j = 0; j < B.n; ++j) {
        const int tj = B.job[j].t1 - B.job[j].t0;
        w[j] = tj * cost[B.job[j].cfg];
        wt += w[j];
        tiles += tj;
    }
    const int grid = tiles < 256 ? (int)tiles : 256;
    if (grid < B.n) return ESR_ECAP;
    int given = 0;
    for (int j = 0; j < B.n; ++j) {
        const int tj = B.job[j].t1 - B.job[j].t0;
        int n = (int)(grid * w[j] / wt);
        if (n < 1) n = 1;
        if (n > tj) n = tj;
        B.job[j].nwg = n;
        given += n;
    }
    // remainder: one workgroup at a time to the most loaded job / from the least loaded one
    for (int guard = 0; given != grid && guard < 4096; ++guard) {
        int pick = -1;
        double best = 0.0;
        for (int j = 0; j < B.n; ++j) {
            const WgradArgs &W = B.job[j];
            if (given < grid) {
                if (W.nwg >= W.t1 - W.t0) continue;
                const double load = w[j] / W.nwg;
                if (pick < 0 || load > best) { pick = j; best = load; }
            } else {
                if (W.nwg <= 1) continue;
                const double load = w[j] / (W.nwg - 1);
                if (pick < 0 || load < best) { pick = j; best = load; }
            }
        }
        if (pick < 0) break;
        B.job[pick].nwg += given < grid ? 1 : -1;
        given += given < grid ? 1 : -1;
    }
    int64_t used = 0;
    int wg0 = 0;
    R = ReduceArgs{};
    for (int j = 0; j < B.n; ++j) {
        WgradArgs &W = B.job[j];
        W.wg0 = wg0;
        wg0 += W.nwg;
        const int n_elems = W.out_rows * W.ld, wk = uni_wk(W.cfg);
        W.slab = scratch + used;
        used += (int64_t)W.nwg * wk * n_elems;
        R.slab[j] = W.slab; R.n_partials[j] = W.nwg * wk; R.n_elems[j] = n_elems; R.gw[j] = W.gw;
        R.first[j + 1] = R.first[j] + (int64_t)n_elems * ((W.nwg * wk + REDUCE_PG - 1) / REDUCE_PG);
    }
    R.nseg = B.n;
    used_out = used;
    if (used > slab_floats) return ESR_ECAP;
    return wg0;
}

// V: 0 = f32 MFMA (LDS-DMA), 1 = the same with split fp16 products, 2 = bf16 operands (register-staged)
template <int V>
int launch_wgrad_uni(WgradBatch &B, SlabPool &P)
{
    hipStream_t s = P.s;
    const void *kern = V == 1 ? reinterpret_cast<const void *>(&mlp_wgrad_uni192s_kernel)
                     : V == 2 ? reinterpret_cast<const void *>(&mlp_wgrad_uni192b_kernel)
                              : reinterpret_cast<const void *>(&mlp_wgrad_uni192_kernel);
    for (int j = 0; j < B.n; ++j) {
        const WgradArgs &W = B.job[j];
        const bool w128 = W.cfg == UNI_HID128 || W.cfg == UNI_FIRST128 || W.cfg == UNI_OUT128;
        const int rap = (W.cfg == UNI_OUT192 || W.cfg == UNI_OUT128) ? 32 : w128 ? 128 : 192;
        const int rbp = (W.cfg == UNI_FIRST192 || W.cfg == UNI_FIRST192_X16 || W.cfg == UNI_FIRST128) ? 96 : w128 ? 128 : 192;
        if (W.RA > rap || W.RB > rbp) return ESR_ECAP;
    }
    constexpr size_t lds_bytes = V == 2 ? 2 * (size_t)(192 + 192) * LDS_STRIDE * sizeof(float)       // two register-staged buffers
                                        : 3 * (size_t)(192 + 192) * 32 * sizeof(float);              // the largest shape's ring
    static std::atomic<uint64_t> optin{0};
    if (int rc = esr_lds_optin(kern, lds_bytes, optin)) return rc;
    ReduceArgs R;
    int64_t n = 0;
    const int grid = P.place([&](float *sc, int64_t fl, ReduceArgs &R_, int64_t &n_) { return plan_uni(B, sc, fl, R_, n_, V); }, R, n);
    if (grid < 0) return grid;
    if (V == 1) mlp_wgrad_uni192s_kernel<<<grid, 320, lds_bytes, s>>>(B);
    else if (V == 2) mlp_wgrad_uni192b_kernel<<<grid, 256, lds_bytes, s>>>(B);
    else mlp_wgrad_uni192_kernel<<<grid, 320, lds_bytes, s>>>(B);
    ESR_CHECK_LAUNCH();
    return P.launched(R, n);
}

// kernel shapes (waves per workgroup wm x wn x wk, always 4 compute waves = 1 per SIMD):
//   192-wide nets: 192x192 -> 2x2x1, 192x<=96 (first layer) -> 2x1x2, 192x<=64 (tone mapper's first) , zrows x 192 (output) -> 1x2x2
//   128-wide nets: 128x128 -> 2x2x1, 128x96 -> 2x1x2, 8x128 -> 1x2x2
enum { CFG_HID192, CFG_FIRST192, CFG_FIRST192_64, CFG_OUT192, CFG_HID128, CFG_FIRST128, CFG_OUT128, CFG_FIRST192_X16, N_CFG };

int layer_cfg(const NetDesc &D, bool first, bool last, int RB)
{
    if (D.hid_tiles == 6) return last ? CFG_OUT192 : first ? (RB <= 64 ? CFG_FIRST192_64 : CFG_FIRST192) : CFG_HID192;
    return last ? CFG_OUT128 : first ? CFG_FIRST128 : CFG_HID128;
}

// Every shape but the tone mapper's narrow first layer (192 x <= 64) rides in the unified launches above (rounds 2-4 kept one
// launch per shape beside them, behind environment switches).  F: 0 = f32 MFMA, 1 = bf16 operands, 2 = f32 operands as split
// fp16 planes (jobs with esr_wgrad_job_t::amax)
template <int F>
int launch_first64(WgradBatch &B, SlabPool &P)
{
    return launch_wgrad_any<3, 2, 2, 1, 2, F == 1 ? 3 : F == 2 ? 4 : 0>(B, P);
}

}  // namespace

ESR_API int64_t esr_mlp_packed_floats(int kind)
{
    if (!kind_ok(kind)) return ESR_EINVAL;
    return pack_layout(kind).total;
}

ESR_API int esr_mlp_pack(int kind, const esr_mlp_weights_t *w, float *packed, void *stream)
{
    if (!kind_ok(kind) || !w || !packed) return ESR_EINVAL;
    PackBatch B = {};
    B.n = 1;
    PackArgs &A = B.job[0];
    A.kind = kind;
    const int nl = net_desc(kind).n_layers;
    for (int l = 0; l < nl; ++l) {
        if (!w->w[l] || !w->b[l]) return ESR_EINVAL;
        A.w[l] = w->w[l];
        A.b[l] = w->b[l];
    }
    A.out = packed;
    pack_kernel<<<dim3(esr_grid_for(pack_layout(kind).total, 256, 1024), 1), 256, 0, esr_stream(stream)>>>(B);
    ESR_CHECK_LAUNCH();
    return 0;
}

// Every net of a step in ONE launch: packed32[i] (and, where packed16 != NULL and packed16[i] != NULL, its bf16 twin) from
// the reference-layout tensors of w[i].  n <= 8.
ESR_API int64_t esr_mlp_packed_split_elems(int kind)
{
    if (!kind_ok(kind)) return ESR_EINVAL;
    // forward | transposed planes | the gradient gain bound (one fp32)
    return split_gain_offset(kind) + SPLIT_GAIN_PAD;
}

// element (fp16) offset of the net's gradient gain bound (one fp32) inside its planes buffer
ESR_API int64_t esr_mlp_split_gain_offset(int kind)
{
    if (!kind_ok(kind)) return ESR_EINVAL;
    return split_gain_offset(kind);
}

// The split input-gradient chain carries dz and every hidden gradient as fp16 planes of ONE power-of-two scale per tile.  How
// far a hidden gradient can exceed the output gradient is bounded by the weights alone: dZ[l-1] = mask (.) (W_l^T dZ[l]), so
// max |dZ[l-1]| <= (largest column sum of |W_l|) max |dZ[l]|.  One block per net: G = max over the hidden layers of the
// running product of those column sums (and 1), written behind the net's planes.  mlp_dgrad_split_kernel scales a tile so
// that G max |dz| stays inside fp16's range -- no overflow in the chain BY CONSTRUCTION -- and hands the weight-gradient
// kernels max |dz| max(1, G / 16) as their scale source (their headroom above it is >= 32x), so neither needs a run-time check.
// A net whose bound is beyond 2^18 (or not finite) raises the range flag: the step runs on the f32 MFMA kernels.
template <int KIND>
__device__ __forceinline__ void split_gain_job(const PackArgs &A)
{
    constexpr NetDesc D = net_desc(KIND);
    constexpr int NL = D.n_layers, hid = 32 * D.hid_tiles, RG = 4;         // 1024 threads = RG row groups x 256 columns
    static_assert(hid <= 256, "one thread per hidden unit and row group");
    __shared__ float red[RG * 256];
    const int tid = threadIdx.x, col = tid & 255, rg = tid >> 8;
    float cum = 1.f, worst = 1.f;
    for (int l = NL - 1; l >= 1; --l) {
        const int outd = l == NL - 1 ? D.out_dim : hid;
        float s = 0.f;
        if (col < hid) {
            const float *w = A.w[l] + col;
            int o = rg;
            for (; o + 7 * RG < outd; o += 8 * RG) {               // eight independent loads in flight per thread
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = w[(int64_t)(o + k * RG) * hid];
#pragma unroll
                for (int k = 0; k < 8; ++k) s += fabsf(v[k]);
            }
            for (; o < outd; o += RG) s += fabsf(w[(int64_t)o * hid]);
        }
        red[tid] = s;
        __syncthreads();
        if (tid < 256) red[tid] = (red[tid] + red[tid + 256]) + (red[tid + 512] + red[tid + 768]);     // the column's sum
        __syncthreads();
        for (int k = 128; k > 0; k >>= 1) {
            if (tid < k) { const float a = red[tid], b = red[tid + k]; red[tid] = (a != a || b != b) ? a + b : fmaxf(a, b); }   // (a NaN stays)
            __syncthreads();
        }
        cum *= red[0];
        if (!(worst >= cum)) worst = cum;                    // (also takes a NaN)
        __syncthreads();
    }
    if (tid == 0) {
        *reinterpret_cast<float *>(A.outs + split_gain_offset(KIND)) = worst;
        if (A.range && !(worst <= SPLIT_GAIN_MAX)) atomicOr(A.range, 1u);
    }
}
__global__ void __launch_bounds__(1024) split_gain_kernel(PackBatch B)
{
    const PackArgs &A = B.job[blockIdx.x];
    if (!A.outs) return;
    switch (A.kind) {                                    // (block-uniform)
    case ESR_MLP_RADIANCE: split_gain_job<ESR_MLP_RADIANCE>(A); break;
    case ESR_MLP_TONEMAP:  split_gain_job<ESR_MLP_TONEMAP>(A); break;
    case ESR_MLP_BRDF:     split_gain_job<ESR_MLP_BRDF>(A); break;
    case ESR_MLP_EMIT:     split_gain_job<ESR_MLP_EMIT>(A); break;
    default: break;
    }
}

ESR_API int esr_mlp_pack_batch(int n, const int32_t *kinds, const esr_mlp_weights_t *const *w, float *const *packed32,
                               void *const *packed16, void *const *packed_split, void *stream)
{
    if (n < 0 || n > MAX_PACK_JOBS || (n > 0 && (!kinds || !w || !packed32))) return ESR_EINVAL;
    if (n == 0) return 0;
    PackBatch B = {};
    B.n = n;
    int64_t most = 0;
    bool any_split = false;
    for (int i = 0; i < n; ++i) {
        const int kind = kinds[i];
        if (!kind_ok(kind) || !w[i] || !packed32[i]) return ESR_EINVAL;
        PackArgs &A = B.job[i];
        A.kind = kind;
        for (int l = 0; l < net_desc(kind).n_layers; ++l) {
            if (!w[i]->w[l] || !w[i]->b[l]) return ESR_EINVAL;
            A.w[l] = w[i]->w[l];
            A.b[l] = w[i]->b[l];
        }
        A.out = packed32[i];
        A.out16 = packed16 ? static_cast<__bf16 *>(packed16[i]) : nullptr;
        A.outs = packed_split ? static_cast<_Float16 *>(packed_split[i]) : nullptr;
        if (A.outs && !(kind == ESR_MLP_RADIANCE || kind == ESR_MLP_TONEMAP || kind == ESR_MLP_BRDF || kind == ESR_MLP_EMIT)) return ESR_EINVAL;
        A.range = A.outs ? esr_split_range_flag_ptr() : nullptr;
        any_split = any_split || A.outs;
        const int64_t tot = pack_layout(kind).total + pack16_layout(kind).total + (A.outs ? split_gain_offset(kind) : 0);
        most = tot > most ? tot : most;
    }
    pack_kernel<<<dim3(esr_grid_for(most, 256, 256), n), 256, 0, esr_stream(stream)>>>(B);
    ESR_CHECK_LAUNCH();
    if (any_split) {                                     // the nets' gradient gain bounds, behind their planes
        split_gain_kernel<<<n, 1024, 0, esr_stream(stream)>>>(B);
        ESR_CHECK_LAUNCH();
    }
    return 0;
}

static bool color_row_ok(int kind, int crow)
{
    return kind == ESR_MLP_COARSE ? (crow == 0 || crow == 12) : (crow == 0 || crow == 88 || crow == 96);
}

ESR_API int esr_mlp_fwd(int kind, const float *packed, const float *X, int32_t t0, int32_t t1,
                        float *const *H, uint32_t *const *M, int save, int color_row0, float *zout,
                        void *stream)
{
    if (!kind_ok(kind) || t0 < 0 || t1 < t0) return ESR_EINVAL;
    if (!color_row_ok(kind, color_row0)) return ESR_EINVAL;
    if (t1 == t0) return 0;
    if (!packed || !X || !zout) return ESR_EINVAL;
    const int nhid = net_desc(kind).n_layers - 1;
    FwdArgs A = {};
    A.packed = packed; A.X = X; A.t0 = t0; A.t1 = t1; A.save = save == 2 ? 2 : save ? 1 : 0; A.crow = color_row0;
    A.zout = zout;
    if (save) {
        if (!M || (save != 2 && !H)) return ESR_EINVAL;
        for (int l = 0; l < nhid; ++l) {
            if (!M[l] || (save != 2 && !H[l])) return ESR_EINVAL;
            A.H[l] = save != 2 ? H[l] : nullptr;
            A.M[l] = M[l];
        }
    }
    const int grid = mlp_grid(t1 - t0);
    hipStream_t s = esr_stream(stream);
    switch (kind) {
    case ESR_MLP_RADIANCE: mlp_fwd_kernel<ESR_MLP_RADIANCE><<<grid, 256, 0, s>>>(A); break;
    case ESR_MLP_TONEMAP:  mlp_fwd_kernel<ESR_MLP_TONEMAP><<<grid, 256, 0, s>>>(A); break;
    case ESR_MLP_BRDF:     mlp_fwd_kernel<ESR_MLP_BRDF><<<grid, 256, 0, s>>>(A); break;
    case ESR_MLP_EMIT:     mlp_fwd_kernel<ESR_MLP_EMIT><<<grid, 256, 0, s>>>(A); break;
    default:               mlp_fwd_kernel<ESR_MLP_COARSE><<<grid, 256, 0, s>>>(A); break;
    }
    ESR_CHECK_LAUNCH();
    return 0;
}

// One launch for a net that runs twice over disjoint tile ranges with the same weights: tiles [t0, t_mid) detached
// (nothing saved, colour rows color_row_detached), tiles [t_mid, t1) saved with colour rows 0 -- the fine stage's
// off-net (voxurff.py:244-254: `.detach()` on the on-rays, differentiable on the off-rays).
ESR_API int esr_mlp_fwd_mixed(int kind, const float *packed, const float *X, int32_t t0, int32_t t_mid, int32_t t1,
                              float *const *H, uint32_t *const *M, int color_row_detached, float *zout, void *stream)
{
    if (kind != ESR_MLP_RADIANCE || t0 < 0 || t_mid < t0 || t1 < t_mid) return ESR_EINVAL;
    if (!color_row_ok(kind, color_row_detached)) return ESR_EINVAL;
    if (t1 == t0) return 0;
    if (!packed || !X || !zout || !H || !M) return ESR_EINVAL;
    FwdArgs A = {};
    A.packed = packed; A.X = X; A.t0 = t0; A.t1 = t1; A.save = 1; A.crow = 0; A.zout = zout;
    A.t_det = t_mid; A.crow_det = color_row_detached;
    for (int l = 0; l < 3; ++l) {
        if (!H[l] || !M[l]) return ESR_EINVAL;
        A.H[l] = H[l];
        A.M[l] = M[l];
    }
    mlp_fwd_kernel<ESR_MLP_RADIANCE><<<mlp_grid(t1 - t0), 256, 0, esr_stream(stream)>>>(A);
    ESR_CHECK_LAUNCH();
    return 0;
}

// The fine stage's three radiance passes in ONE launch (voxurff.py:243-256): non-emissive net on tiles [0, t_on)
// detached (colour rows color_row_detached, nothing saved) and on [t_on, t_all) saved, emissive net on [0, t_on) saved.
// Both nets save into the same H / M arrays (disjoint tiles).
ESR_API int esr_mlp_fwd_fine(const float *packed_off, const float *packed_emo, const float *X, int32_t t_on, int32_t t_all,
                             float *const *H, uint32_t *const *M, int color_row_detached, float *z_off, float *z_emo,
                             void *stream)
{
    if (t_on < 0 || t_all < t_on) return ESR_EINVAL;
    if (!color_row_ok(ESR_MLP_RADIANCE, color_row_detached)) return ESR_EINVAL;
    if (t_all == 0) return 0;
    if (!packed_off || !packed_emo || !X || !z_off || !z_emo || !H || !M) return ESR_EINVAL;
    FwdArgs A = {};
    A.packed = packed_off; A.X = X; A.t0 = 0; A.t1 = t_all; A.save = 1; A.crow = 0; A.zout = z_off;
    A.t_det = t_on; A.crow_det = color_row_detached;
    A.packed2 = packed_emo; A.zout2 = z_emo; A.t0_2 = 0; A.n2 = t_on; A.crow2 = 0; A.save2 = 1;
    for (int l = 0; l < 3; ++l) {
        if (!H[l] || !M[l]) return ESR_EINVAL;
        A.H[l] = H[l];
        A.M[l] = M[l];
    }
    mlp_fwd_kernel<ESR_MLP_RADIANCE><<<mlp_grid(t_all + t_on), 256, 0, esr_stream(stream)>>>(A);
    ESR_CHECK_LAUNCH();
    return 0;
}

// Input gradients of the fine stage's two radiance nets in ONE launch: emissive net on tiles [0, t_on), non-emissive
// net on [t_on, t_all) (its on-tile pass is detached in the reference: no gradient).
ESR_API int esr_mlp_dgrad_fine(const float *packed_emo, const float *packed_off, const float *dz, int32_t t_on, int32_t t_all,
                               const uint32_t *const *M, float *const *dZ, float *dX, void *stream)
{
    if (t_on < 0 || t_all < t_on) return ESR_EINVAL;
    if (t_all == 0) return 0;
    if (!packed_emo || !packed_off || !dz || !M || !dZ || !dX) return ESR_EINVAL;
    DgradArgs A = {};
    A.packed = packed_emo; A.packed2 = packed_off; A.t_split = t_on; A.dz = dz; A.t0 = 0; A.t1 = t_all; A.dX = dX;
    for (int l = 0; l < 3; ++l) {
        if (!M[l]) return ESR_EINVAL;
        A.M[l] = M[l]; A.dZ[l] = dZ[l];
    }
    mlp_dgrad_kernel<ESR_MLP_RADIANCE><<<mlp_grid(t_all, mlp_occ(ESR_MLP_RADIANCE)), 256, 0, esr_stream(stream)>>>(A);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_mlp_dgrad(int kind, const float *packed, const float *dz, int32_t t0, int32_t t1,
                          const uint32_t *const *M, float *const *dZ, float *dX, void *stream)
{
    return esr_mlp_dgrad_wg(kind, packed, dz, t0, t1, M, dZ, dX, 0, stream);
}

// max_workgroups > 0 caps the grid (256 = one 4-wave workgroup per CU: half the register file and all of the LDS stay
// free for a latency-bound kernel of another stream -- the grid scatters -- to run beside the matrix work).
ESR_API int esr_mlp_dgrad_wg(int kind, const float *packed, const float *dz, int32_t t0, int32_t t1,
                             const uint32_t *const *M, float *const *dZ, float *dX, int32_t max_workgroups, void *stream)
{
    if (!kind_ok(kind) || t0 < 0 || t1 < t0 || max_workgroups < 0) return ESR_EINVAL;
    if (t1 == t0) return 0;
    if (!packed || !dz || !M || !dZ || !dX) return ESR_EINVAL;
    const int nhid = net_desc(kind).n_layers - 1;
    DgradArgs A = {};
    A.packed = packed; A.dz = dz; A.t0 = t0; A.t1 = t1; A.dX = dX;
    for (int l = 0; l < nhid; ++l) {
        if (!M[l]) return ESR_EINVAL;
        A.M[l] = M[l]; A.dZ[l] = dZ[l];      // a NULL dZ[l] is not stored (its weight gradient recomputes it)
    }
    int grid = mlp_grid(t1 - t0, mlp_occ(kind));
    if (max_workgroups > 0 && grid > max_workgroups) grid = max_workgroups;
    hipStream_t s = esr_stream(stream);
    switch (kind) {
    case ESR_MLP_RADIANCE: mlp_dgrad_kernel<ESR_MLP_RADIANCE><<<grid, 256, 0, s>>>(A); break;
    case ESR_MLP_TONEMAP:  mlp_dgrad_kernel<ESR_MLP_TONEMAP><<<grid, 256, 0, s>>>(A); break;
    case ESR_MLP_BRDF:     mlp_dgrad_kernel<ESR_MLP_BRDF><<<grid, 256, 0, s>>>(A); break;
    case ESR_MLP_EMIT:     mlp_dgrad_kernel<ESR_MLP_EMIT><<<grid, 256, 0, s>>>(A); break;
    default:               mlp_dgrad_kernel<ESR_MLP_COARSE><<<grid, 256, 0, s>>>(A); break;
    }
    ESR_CHECK_LAUNCH();
    return 0;
}

// max |x| into out[0] (atomic maximum of non-negative floats = of their bit patterns); NaN inputs are ignored by fmaxf.
// One atomic per WORKGROUP (256 of them): with one per wave, 4096 same-address atomics made an 8 MB reduction take 52 us.
__global__ void __launch_bounds__(256) absmax_kernel(const float *__restrict__ x, int64_t n, float *out)
{
    __shared__ float part[4];
    float m = 0.f;
    const int64_t n4 = n >> 2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4 *>(x)[i];
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = fmaxf(m, fabsf(x[(n4 << 2) + threadIdx.x]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
        if (m > 0.f) atomicMax(reinterpret_cast<unsigned *>(out), __float_as_uint(m));
    }
}

ESR_API int esr_absmax(const float *x, int64_t n, float *out, void *stream)
{
    if (n < 0 || !out || (n > 0 && !x) || (reinterpret_cast<uintptr_t>(x) & 15)) return ESR_EINVAL;
    if (n == 0) return 0;
    absmax_kernel<<<esr_grid_for((n + 3) / 4, 256, 256), 256, 0, esr_stream(stream)>>>(x, n, out);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int64_t esr_mlp_wgrad_scratch_floats(void) { return (int64_t)256 * 2 * 192 * 192; }

// every layer of every net job, grouped by kernel shape: one launch per shape (up to MAX_JOBS layer jobs each)
template <bool BF>
static int wgrad_jobs(const esr_wgrad_job_t *jobs, int n_jobs, float *scratch, int64_t scratch_floats, void *stream)
{
    if (!jobs || n_jobs < 0 || !scratch) return ESR_EINVAL;
    WgradBatch group[2][4];                 // the tone mapper's first layers; [0]: this call's operand type, [1]: f32 operands as split planes (jobs with amax)
    int n_group[2] = {};
    constexpr int MAX_UNI = 8;
    WgradBatch uni[MAX_UNI];
    bool uni_split[MAX_UNI];
    int n_uni = 0;
    for (int v = 0; v < 2; ++v)
        for (int g = 0; g < 4; ++g) group[v][g].n = 0;
    for (int q = 0; q < n_jobs; ++q) {
        const esr_wgrad_job_t &J = jobs[q];
        if (!kind_ok(J.kind) || J.t0 < 0 || J.t1 < J.t0) return ESR_EINVAL;
        if (!color_row_ok(J.kind, J.color_row0)) return ESR_EINVAL;
        if (J.t1 == J.t0) continue;
        if ((!J.X && !J.X16) || !J.H || !J.dZ || !J.dz || !J.gw || !J.gb) return ESR_EINVAL;
        const NetDesc D = net_desc(J.kind);
        const int hid = 32 * D.hid_tiles;
        for (int l = 0; l < D.n_layers; ++l) {
            const bool first = l == 0, last = l == D.n_layers - 1;
            WgradArgs W = {};
            W.A = last ? J.dz : J.dZ[l];        W.RA = last ? D.zrows : hid;
            W.B = first ? J.X : J.H[l - 1];     W.RB = first ? (D.xrows < 96 ? D.xrows : 96) : hid;
            W.b_tile_rows = first ? D.xrows : hid;
            W.crow = first ? J.color_row0 : 0;
            W.cw8 = D.cw * 8;
            W.t0 = J.t0; W.t1 = J.t1;
            W.gw = J.gw[l]; W.ld = first ? D.in_dim : hid; W.out_rows = last ? D.out_dim : hid;
            W.kind = J.kind; W.first = first ? 1 : 0; W.gb = J.gb[l];
            if (!W.A || !W.B || !W.gw || !W.gb) return ESR_EINVAL;
            int c = layer_cfg(D, first, last, W.RB);
            if (BF && first && J.X16 && J.kind == ESR_MLP_RADIANCE) {
                if (J.color_row0 != 0) return ESR_EINVAL;          // (the bf16 tile's alternate colour rows are the forward's only)
                W.B = static_cast<const float *>(J.X16);
                W.RB = 96; W.b_tile_rows = 104;                    // 24 row quads of operand rows, 26 quads (6656 B) per tile
                c = CFG_FIRST192_X16;
            }
            if (c != CFG_FIRST192_64) {
                W.cfg = c == CFG_HID192 ? UNI_HID192 : c == CFG_FIRST192 ? UNI_FIRST192 : c == CFG_OUT192 ? UNI_OUT192
                      : c == CFG_HID128 ? UNI_HID128 : c == CFG_FIRST128 ? UNI_FIRST128 : c == CFG_OUT128 ? UNI_OUT128 : UNI_FIRST192_X16;
                W.amax = BF ? nullptr : J.amax;                    // non-NULL: the split-fp16 kernel (esr_hip.h)
                const bool sp = !BF && J.amax != nullptr;
                if (n_uni == 0 || uni[n_uni - 1].n == MAX_JOBS || uni_split[n_uni - 1] != sp) {
                    if (n_uni == MAX_UNI) return ESR_ECAP;
                    uni_split[n_uni] = sp;
                    uni[n_uni++].n = 0;
                }
                WgradBatch &U = uni[n_uni - 1];
                U.job[U.n++] = W;
                continue;
            }
            const int v = (!BF && J.amax) ? 1 : 0;
            W.amax = J.amax;
            int g = n_group[v];
            if (g == 0 || group[v][g - 1].n == MAX_JOBS) {
                if (g == 4) return ESR_ECAP;
                g = ++n_group[v];
            }
            WgradBatch &B = group[v][g - 1];
            B.job[B.n++] = W;
        }
    }
    SlabPool P(scratch, scratch_floats, esr_stream(stream));
    for (int u = 0; u < n_uni; ++u)
        if (int rc = BF ? launch_wgrad_uni<2>(uni[u], P) : uni_split[u] ? launch_wgrad_uni<1>(uni[u], P) : launch_wgrad_uni<0>(uni[u], P)) return rc;
    for (int g = 0; g < n_group[0]; ++g)
        if (int rc = launch_first64<BF ? 1 : 0>(group[0][g], P)) return rc;
    for (int g = 0; g < n_group[1]; ++g)
        if (int rc = launch_first64<2>(group[1][g], P)) return rc;
    return P.flush();
}

template <bool BF>
static int wgrad_all(int kind, const float *X, int color_row0, const float *const *H,
                     const float *const *dZ, const float *dz, int32_t t0, int32_t t1,
                     float *const *gw, float *const *gb, float *scratch, int64_t scratch_floats,
                     void *stream)
{
    esr_wgrad_job_t J = {};
    J.kind = kind; J.color_row0 = color_row0; J.t0 = t0; J.t1 = t1;
    J.X = X; J.H = H; J.dZ = dZ; J.dz = dz; J.gw = gw; J.gb = gb;
    return wgrad_jobs<BF>(&J, 1, scratch, scratch_floats, stream);
}

ESR_API int esr_mlp_wgrad_batch(const esr_wgrad_job_t *jobs, int32_t n_jobs, int bf16_operands, float *scratch,
                                int64_t scratch_floats, void *stream)
{
    return bf16_operands ? wgrad_jobs<true>(jobs, n_jobs, scratch, scratch_floats, stream)
                         : wgrad_jobs<false>(jobs, n_jobs, scratch, scratch_floats, stream);
}

ESR_API int esr_mlp_wgrad(int kind, const float *X, int color_row0, const float *const *H,
                          const float *const *dZ, const float *dz, int32_t t0, int32_t t1,
                          float *const *gw, float *const *gb, float *scratch, int64_t scratch_floats,
                          void *stream)
{
    return wgrad_all<false>(kind, X, color_row0, H, dZ, dz, t0, t1, gw, gb, scratch, scratch_floats, stream);
}

ESR_API int esr_mlp_wgrad_bf16(int kind, const float *X, int color_row0, const float *const *H,
                               const float *const *dZ, const float *dz, int32_t t0, int32_t t1,
                               float *const *gw, float *const *gb, float *scratch, int64_t scratch_floats,
                               void *stream)
{
    return wgrad_all<true>(kind, X, color_row0, H, dZ, dz, t0, t1, gw, gb, scratch, scratch_floats, stream);
}
