// Weight gradients of the tone mapper (TonemapNet 33 -> 192 -> 3, app/utils/pbr/module.py:24-39) WITHOUT its saved
// hidden tiles: the hidden layer is recomputed, in a TRANSPOSED accumulator layout, inside the kernel that contracts
// over the samples.
//
// Why a kernel of its own.  The tone mapper runs on every surviving sample (16 384 tiles at C2) but carries 13x less
// matrix work per sample than a radiance net, so with the save-everything scheme of mlp.hip its three passes are pure
// tile traffic: the forward writes Ht (24 KB per tile), the input-gradient pass writes dZt (24 KB), the weight-gradient
// launches read both back (0.53 ms per step for 0.14 ms of matrix work at peak).  Recomputing the 33 -> 192 layer costs
// 120 MFMAs per tile -- less than moving one of those tiles.
//
// The transposition problem and how it disappears.  dW = sum over SAMPLES needs the sample index as the MFMA's k, i.e.
// in registers, while the forward chain (mlp.hip) keeps the sample on the LANE.  Here the hidden layer is evaluated as
//     Ht^T[s][u] = sum_x Xt^T[s][x] W0^T[x][u]        (A = the X tile exactly as the forward loads it, B = W0^T)
// whose 32x32 accumulator has the UNIT on the lane and the SAMPLE in the register: precisely the A-operand layout of
//     dW0[u][x] += sum_s dZt^T[s][u] Xt[x][s]
// and the natural layout for per-lane (= per-unit) FMA sums of dW1[c][u] = sum_s dzt[c][s] Ht^T[s][u], db0, and the
// 33rd input column.  Nothing is transposed through memory; the only staged data is the wave's own 6-KB X tile
// (written to LDS from the registers it was loaded into, read back row-per-lane as the B operand of dW0).
//
// A pair of waves = one tile at a time (each wave half of the hidden units), accumulators for dW0/dW1/db in registers
// across the pair's tile range (1 wave per SIMD), partial sums to a per-pair slab, summed by a small reduce kernel.  fp32 throughout
// (v_mfma_f32_32x32x2_f32): the recomputed Ht equals the forward's up to summation order.
#include "esr_common.h"

#include "mlp_common.h"

// In-kernel time stamps: nothing in the product build; tools/ubench/tone_stamps.hip defines ESR_TSTAMP before including
// this file.
#ifndef ESR_TSTAMP
#define ESR_TSTAMP(i)
#endif

namespace {

constexpr int TIN = 33, THID = 192, TOUT = 3, TKP = 17;       // inputs, hidden units, outputs, k-pairs of the input rows 0..33 (row 33: zero weight; the forward pads to 40 only because its weight stream comes in quads)
constexpr int XT_ROWS = 48;                                    // rows of the Xt tile in memory
constexpr int XS = 36;                                         // LDS row stride (floats) of the staged tiles: 16-B reads of 16
                                                               // consecutive rows then hit 16 distinct 4-bank groups
constexpr int W0T_FLOATS = 2 * TKP * THID;                     // W0^T [40][192] in LDS
constexpr int WAVE_LDS = (2 * TKP + 4) * XS;                   // per wave: Xt rows 0..39 + dzt rows 0..3
constexpr int N_DW0 = THID * TIN, N_DW1 = TOUT * THID;
constexpr int SLAB = N_DW0 + N_DW1 + THID + 4;                 // floats per wave slab: dW0 | dW1 | db0 | db1(+pad)

struct ToneWgArgs {
    const float *Xt, *dzt;                  // [tiles][48][32], [tiles][4][32]
    const float *W0, *b0, *W1;              // reference layout: [192][33], [192], [3][192]
    int t0, t1;
    float *slab;                            // [n_wave_pairs][SLAB]
};

__global__ void __launch_bounds__(256, 1) tone_wgrad_t_kernel(ToneWgArgs A)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *w0t = lds;                                                   // [40][192]: w0t[x * 192 + u] = W0[u][x]
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, ul = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *lx = lds + W0T_FLOATS + wv * WAVE_LDS;                       // this wave's staged Xt tile [40][XS]
    float *lz = lx + 2 * TKP * XS;                                      // and dzt tile [4][XS]
    for (int i = tid; i < W0T_FLOATS; i += 256) {
        const int x = i / THID, u = i % THID;
        w0t[i] = x < TIN ? A.W0[u * TIN + x] : 0.f;
    }
    __syncthreads();

    // Two waves share a sample tile: wave parity g owns hidden units [96 g, 96 g + 96) -- three 32-unit accumulator
    // tiles of everything (Ht / dZt 48 registers, dW0 48): the whole working set stays inside the 256 architected VGPRs
    // (with all 192 units in one wave the compiler spilled ~100 registers: VALU use of an accumulator tile needs it in
    // the architected half of the file).  The X tile is loaded by both (6 KB, second read from L2).
    const int g = wv & 1;
    // stationary operands: B of dHt^T (W1[c][u], pairs (0,1) and (2,-)), the hidden bias
    float w1b[3][2], b0r[3];
#pragma unroll
    for (int i3 = 0; i3 < 3; ++i3) {
        const int u = 96 * g + 32 * i3 + ul;
        w1b[i3][0] = A.W1[h * THID + u];                               // c = 0 + h
        w1b[i3][1] = h == 0 ? A.W1[2 * THID + u] : 0.f;                // c = 2 + h (row 3 does not exist)
        b0r[i3] = A.b0[u];
    }
    f32x16 dW0[3];
    zero_tiles<3>(dW0);
    float dW0c[3], dW1r[3][3], db0r[3], db1r[2] = {0.f, 0.f};
#pragma unroll
    for (int i3 = 0; i3 < 3; ++i3) {
        dW0c[i3] = 0.f; db0r[i3] = 0.f;
        dW1r[i3][0] = dW1r[i3][1] = dW1r[i3][2] = 0.f;
    }

    const int pair = blockIdx.x * 2 + (wv >> 1), npairs = gridDim.x * 2;
    // A operand of Ht^T (sample on the lane, input row in the register) = what the forward loads.  One wave per SIMD
    // has nobody to hide its global-load latency behind: the NEXT tile's 22 loads are issued before this tile's math.
    float xn[TKP], zn[2];
    auto fetch = [&](int t) {
        const bool live = t < A.t1;
        const float *X = A.Xt + (size_t)(live ? t : A.t0) * XT_ROWS * 32 + ul;
        const float *Zt = A.dzt + (size_t)(live ? t : A.t0) * 4 * 32 + ul;
#pragma unroll
        for (int j = 0; j < TKP; ++j) xn[j] = X[(2 * j + h) * 32];
        zn[0] = Zt[h * 32];
        zn[1] = h == 0 ? Zt[2 * 32] : 0.f;
    };
    if (A.t0 + pair < A.t1) fetch(A.t0 + pair);
    for (int t = A.t0 + pair; t < A.t1; t += npairs) {
        ESR_TSTAMP(0);
        float xa[TKP], za[2];
#pragma unroll
        for (int j = 0; j < TKP; ++j) xa[j] = xn[j];
        za[0] = zn[0]; za[1] = zn[1];
        fetch(t + npairs);                              // (past the range: re-reads tile t0, never used)
        db1r[0] += za[0]; db1r[1] += za[1];
#pragma unroll
        for (int j = 0; j < TKP; ++j) lx[(2 * j + h) * XS + ul] = xa[j];
        lz[h * XS + ul] = za[0];
        if (h == 0) lz[2 * XS + ul] = za[1];
        ESR_TSTAMP(1);
        // ---- Ht^T[s][u]: accumulator register r of lane (u, h) holds sample s = acc_row(r, h)
        f32x16 ht[3];
#pragma unroll
        for (int i3 = 0; i3 < 3; ++i3)
#pragma unroll
            for (int r = 0; r < 16; ++r) ht[i3][r] = b0r[i3];
#pragma unroll
        for (int j = 0; j < TKP; ++j) {                 // k-pair outer: three B values live at a time
            float wb[3];
#pragma unroll
            for (int i3 = 0; i3 < 3; ++i3) wb[i3] = w0t[(2 * j + h) * THID + 96 * g + 32 * i3 + ul];
#pragma unroll
            for (int i3 = 0; i3 < 3; ++i3) ht[i3] = mfma32(xa[j], wb[i3], ht[i3]);
            if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        ESR_TSTAMP(2);
        relu_tiles<3>(ht);
        // ---- dW1[c][u] += sum_s dzt[c][s] Ht[u][s]: per-lane sums over this lane's 16 samples; the per-sample scalars
        // of this half-wave come from LDS four samples at a time (register r = 4q + i holds sample 8q + 4h + i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 a = *reinterpret_cast<const float4 *>(lz + 0 * XS + 8 * q + 4 * h);
            const float4 b = *reinterpret_cast<const float4 *>(lz + 1 * XS + 8 * q + 4 * h);
            const float4 c = *reinterpret_cast<const float4 *>(lz + 2 * XS + 8 * q + 4 * h);
            const float za4[4] = {a.x, a.y, a.z, a.w}, zb4[4] = {b.x, b.y, b.z, b.w}, zc4[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
            for (int i3 = 0; i3 < 3; ++i3)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    dW1r[i3][0] = fmaf(za4[i], ht[i3][4 * q + i], dW1r[i3][0]);
                    dW1r[i3][1] = fmaf(zb4[i], ht[i3][4 * q + i], dW1r[i3][1]);
                    dW1r[i3][2] = fmaf(zc4[i], ht[i3][4 * q + i], dW1r[i3][2]);
                }
        }
        __builtin_amdgcn_sched_barrier(0);
        ESR_TSTAMP(3);
        float x32[16];                                  // Xt row 32 (the 33rd input) at this half-wave's 16 samples
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 d = *reinterpret_cast<const float4 *>(lx + 32 * XS + 8 * q + 4 * h);
            x32[4 * q] = d.x; x32[4 * q + 1] = d.y; x32[4 * q + 2] = d.z; x32[4 * q + 3] = d.w;
        }
        // ---- dHt^T = dzt^T W1, masked by the recomputed activation: dZt^T (same layout)
#pragma unroll
        for (int i3 = 0; i3 < 3; ++i3) {
            f32x16 d;
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = 0.f;
            d = mfma32(za[0], w1b[i3][0], d);
            d = mfma32(za[1], w1b[i3][1], d);
            float sb = 0.f, sc = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = ht[i3][r] > 0.f ? d[r] : 0.f;
                ht[i3][r] = v;                                         // Ht is dead from here: the tile now holds dZt^T
                sb += v;
                sc = fmaf(v, x32[r], sc);
            }
            db0r[i3] += sb;
            dW0c[i3] += sc;                                            // input column 32 (the 33rd row of Xt)
        }
        __builtin_amdgcn_sched_barrier(0);
        ESR_TSTAMP(4);
        // ---- dW0[u][x] += sum_s dZt^T[s][u] Xt[x][s], x = 0..31: A = the dZt^T registers, B = Xt row-per-lane from LDS
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 xb = *reinterpret_cast<const float4 *>(lx + ul * XS + 8 * q + 4 * h);
#pragma unroll
            for (int i3 = 0; i3 < 3; ++i3) {
                dW0[i3] = mfma32(ht[i3][4 * q + 0], xb.x, dW0[i3]);
                dW0[i3] = mfma32(ht[i3][4 * q + 1], xb.y, dW0[i3]);
                dW0[i3] = mfma32(ht[i3][4 * q + 2], xb.z, dW0[i3]);
                dW0[i3] = mfma32(ht[i3][4 * q + 3], xb.w, dW0[i3]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        ESR_TSTAMP(5);
    }

    // ---- flush this wave's partial sums into its pair's slab: dW0 [192][33] | dW1 [3][192] | db0 [192] | db1 [3]
    float *S = A.slab + (size_t)pair * SLAB;
#pragma unroll
    for (int i3 = 0; i3 < 3; ++i3) {
        const int ub = 96 * g + 32 * i3;
        // accumulator of dW0: column = x (lane), row = u_local = acc_row(r, h)
#pragma unroll
        for (int r = 0; r < 16; ++r) S[(ub + acc_row(r, h)) * TIN + ul] = dW0[i3][r];
        const int u = ub + ul;
        const float c32 = dW0c[i3] + __shfl_xor(dW0c[i3], 32);
        const float b = db0r[i3] + __shfl_xor(db0r[i3], 32);
        float w1[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) w1[c] = dW1r[i3][c] + __shfl_xor(dW1r[i3][c], 32);
        if (h == 0) {
            S[u * TIN + 32] = c32;
            S[N_DW0 + N_DW1 + u] = b;
#pragma unroll
            for (int c = 0; c < 3; ++c) S[N_DW0 + c * THID + u] = w1[c];
        }
    }
    float d0 = db1r[0], d1 = db1r[1];                                   // db1: rows h and 2 + h, summed over the 32 samples
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) { d0 += __shfl_xor(d0, off); d1 += __shfl_xor(d1, off); }
    if (ul == 0 && g == 0) {                                            // (both waves of a pair saw the same dzt)
        S[N_DW0 + N_DW1 + THID + h] = d0;
        if (h == 0) { S[N_DW0 + N_DW1 + THID + 2] = d1; S[N_DW0 + N_DW1 + THID + 3] = 0.f; }
    }
}


// ---- bf16 twin (the bf16 configurations C3 / C5) --------------------------------------------------------------------
// Same scheme with v_mfma_f32_32x32x16_bf16: operands rounded to bf16 where the bf16 engine rounds them (Xt, W0, dzt, W1
// when they become an operand; the recomputed Ht and dZt when THEY become an operand of dW1 / dW0 / db0 -- the values
// the saved-tile path stored as bf16), fp32 accumulation.  With it the bf16 step saves neither Ht nor dZt: 48 KB of HBM
// traffic per sample tile (1.05 GB per C3 step) and the tone mapper's share of the staged weight-gradient launches.
// Matrix work drops from 114 f32 MFMAs (7.3 k cycles) to 18 bf16 ones (0.6 k) per wave and tile, so the kernel is bound by
// its vector work (ReLU, masks, dW1 / db0 sums, bf16 packing) and runs two waves per SIMD.
//
// k-slot conventions of the 32x32x16 operands (lane = 32 h + i, 8 slots e per lane and k-step q):
//   Ht^T = Xt^T W0^T     k = input row 16 q + 8 h + e          A: lane i = sample, B: lane i = unit (16-B LDS read of w0b)
//   dHt^T = dzt^T W1     k = output channel e (half 0 only)     one k-step
//   dW0 = dZt Xt^T       k = sample acc_row(8 q + e, h): the accumulator registers 8q .. 8q+7 of dZt^T ARE the A operand;
//                        B: row x = lane i of the Xt tile staged in LDS as bf16 with its 32 samples PERMUTED so that
//                        those 8 samples are one 16-B read (xperm below)
constexpr int W0B_U = 56;                                      // bf16 per unit row of W0 in LDS: 48 input rows + pad (112-B stride)
constexpr int XS16 = 40;                                       // bf16 per staged Xt row: 32 samples + pad (80-B stride: 16 lanes' 16-B reads on distinct banks)
constexpr int WAVE_LDS16 = 32 * XS16 * 2 + (4 + 1) * XS * 4;   // bytes per wave: Xt rows 0..31 (bf16) | dzt rows 0..3 + Xt row 32 (f32)
constexpr int W0B_BYTES = THID * W0B_U * 2;

__device__ __forceinline__ float bf16r(float x) { return (float)(__bf16)x; }
__device__ __forceinline__ int xperm(int s)                    // position of sample s inside a staged Xt row
{
    const int r = (s & 3) + 4 * (s >> 3), hh = (s >> 2) & 1;   // s = acc_row(r, hh)
    return (2 * (r >> 3) + hh) * 8 + (r & 7);
}

// (two waves per SIMD since round 6: 252 registers, no accumulation-register moves -- 144 per tile before -- and no spills, once
// the file is built without the SLP vectoriser (esr_nerf_amd/build.py); with it the same bound gave 46 spills, 0.19 ms at C3)
__global__ void __launch_bounds__(256, 2) tone_wgrad16_t_kernel(ToneWgArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds16[];
    __bf16 *w0b = reinterpret_cast<__bf16 *>(lds16);                       // [192][W0B_U]: w0b[u * W0B_U + x] = bf16(W0[u][x])
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, ul = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char *mine = lds16 + W0B_BYTES + wv * WAVE_LDS16;
    __bf16 *xb = reinterpret_cast<__bf16 *>(mine);                         // [32][XS16]
    float *lz = reinterpret_cast<float *>(mine + 32 * XS16 * 2);           // dzt rows 0..3, [4][XS]
    float *lx32 = lz + 4 * XS;                                             // Xt row 32, [XS]
    for (int i = tid; i < THID * W0B_U; i += 256) {
        const int u = i / W0B_U, x = i % W0B_U;
        w0b[i] = (__bf16)(x < TIN ? A.W0[u * TIN + x] : 0.f);
    }
    __syncthreads();

    const int g = wv & 1;                                                   // this wave's hidden units [96 g, 96 g + 96)
    bf16x8 w1b[3];
    float b0r[3];
#pragma unroll
    for (int i3 = 0; i3 < 3; ++i3) {
        const int u = 96 * g + 32 * i3 + ul;
#pragma unroll
        for (int e = 0; e < 8; ++e) w1b[i3][e] = (__bf16)((h == 0 && e < TOUT) ? A.W1[e * THID + u] : 0.f);
        b0r[i3] = A.b0[u];
    }
    f32x16 dW0[3];
    zero_tiles<3>(dW0);
    float dW0c[3], dW1r[3][3], db0r[3], db1r[2] = {0.f, 0.f};
#pragma unroll
    for (int i3 = 0; i3 < 3; ++i3) {
        dW0c[i3] = 0.f; db0r[i3] = 0.f;
        dW1r[i3][0] = dW1r[i3][1] = dW1r[i3][2] = 0.f;
    }
    const int pair = blockIdx.x * 2 + (wv >> 1), npairs = gridDim.x * 2;
    // rows this lane feeds into the A operand of Ht^T: 16 q + 8 h + e (q = 0, 1) and 32 + 8 h (the 33rd input; rows
    // 33..47 of the tile are zero)
    float xn[17], zn[2];
    auto fetch = [&](int t) {
        const bool live = t < A.t1;
        const float *X = A.Xt + (size_t)(live ? t : A.t0) * XT_ROWS * 32 + ul;
        const float *Zt = A.dzt + (size_t)(live ? t : A.t0) * 4 * 32 + ul;
#pragma unroll
        for (int j = 0; j < 16; ++j) xn[j] = X[(16 * (j >> 3) + 8 * h + (j & 7)) * 32];
        xn[16] = X[(32 + 8 * h) * 32];
        zn[0] = Zt[h * 32];
        zn[1] = h == 0 ? Zt[2 * 32] : 0.f;
    };
    const int xp = xperm(ul);
    if (A.t0 + pair < A.t1) fetch(A.t0 + pair);
    for (int t = A.t0 + pair; t < A.t1; t += npairs) {
        float xa[17], za[2];
#pragma unroll
        for (int j = 0; j < 17; ++j) xa[j] = xn[j];
        za[0] = zn[0]; za[1] = zn[1];
        fetch(t + npairs);
        db1r[0] += za[0]; db1r[1] += za[1];
        // stage: Xt rows 0..31 as bf16 (permuted samples), dzt rows and Xt row 32 as f32
#pragma unroll
        for (int j = 0; j < 16; ++j) xb[(16 * (j >> 3) + 8 * h + (j & 7)) * XS16 + xp] = (__bf16)xa[j];
        lz[h * XS + ul] = za[0];
        if (h == 0) { lz[2 * XS + ul] = za[1]; lx32[ul] = xa[16]; }
        // ---- Ht^T[s][u]
        bf16x8 a8[3];
#pragma unroll
        for (int q = 0; q < 2; ++q)
            a8[q] = pack8(make_float4(xa[8 * q], xa[8 * q + 1], xa[8 * q + 2], xa[8 * q + 3]),
                          make_float4(xa[8 * q + 4], xa[8 * q + 5], xa[8 * q + 6], xa[8 * q + 7]));
        a8[2] = pack8(make_float4(xa[16], 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f));
        f32x16 ht[3];
#pragma unroll
        for (int i3 = 0; i3 < 3; ++i3) {
#pragma unroll
            for (int r = 0; r < 16; ++r) ht[i3][r] = b0r[i3];
            const __bf16 *wrow = w0b + (96 * g + 32 * i3 + ul) * W0B_U + 8 * h;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const bf16x8 wb = *reinterpret_cast<const bf16x8 *>(wrow + 16 * q);
                ht[i3] = mfma16(a8[q], wb, ht[i3]);
            }
        }
        // ReLU in C, NOT relu_tiles' inline asm: the compiler's hazard recogniser does not look inside an asm statement,
        // and here it schedules a tile's v_max right behind the tile's last (8-pass) MFMA -- the v_max then read the
        // accumulator before that MFMA had written it (seen as the 33rd input's contribution missing from some
        // units).  The f32 kernels' relu_tiles calls sit >= 80 vector instructions behind the MFMA that wrote their tile.
#pragma unroll
        for (int i3 = 0; i3 < 3; ++i3)
#pragma unroll
            for (int r = 0; r < 16; ++r) {                               // (integer form: one v_max_i32 per value, compiler-visible)
                const int bits = __float_as_int(ht[i3][r]);
                ht[i3][r] = __int_as_float(bits > 0 ? bits : 0);
            }
        // ---- dW1[c][u] += sum_s dzt[c][s] Ht[u][s]: fp32 sums of UNROUNDED values (the saved-tile path rounded both to
        // bf16 on their way to its MFMA; here they are vector operands, and rounding them would only add ~100 vector
        // instructions per tile to a kernel that is bound by its vector work)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 a = *reinterpret_cast<const float4 *>(lz + 0 * XS + 8 * q + 4 * h);
            const float4 b = *reinterpret_cast<const float4 *>(lz + 1 * XS + 8 * q + 4 * h);
            const float4 c = *reinterpret_cast<const float4 *>(lz + 2 * XS + 8 * q + 4 * h);
            const float za4[4] = {a.x, a.y, a.z, a.w}, zb4[4] = {b.x, b.y, b.z, b.w}, zc4[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
            for (int i3 = 0; i3 < 3; ++i3)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float hv = ht[i3][4 * q + i];
                    dW1r[i3][0] = fmaf(za4[i], hv, dW1r[i3][0]);
                    dW1r[i3][1] = fmaf(zb4[i], hv, dW1r[i3][1]);
                    dW1r[i3][2] = fmaf(zc4[i], hv, dW1r[i3][2]);
                }
        }
        __builtin_amdgcn_sched_barrier(0);
        float x32[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 d = *reinterpret_cast<const float4 *>(lx32 + 8 * q + 4 * h);
            x32[4 * q] = d.x; x32[4 * q + 1] = d.y; x32[4 * q + 2] = d.z; x32[4 * q + 3] = d.w;
        }
        // ---- dHt^T = dzt^T W1 masked by the recomputed activation; then dW0 += dZt Xt^T
        // slots 0..2 of half 0 = dzt rows 0..2 of this lane's sample (row 1 lives in the OTHER half's za[0])
        const float z1 = __shfl_xor(za[0], 32);
        const bf16x8 za8 = pack8(make_float4(h == 0 ? za[0] : 0.f, h == 0 ? z1 : 0.f, h == 0 ? za[1] : 0.f, 0.f),
                                 make_float4(0.f, 0.f, 0.f, 0.f));
#pragma unroll
        for (int i3 = 0; i3 < 3; ++i3) {
            f32x16 d;
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = 0.f;
            d = mfma16(za8, w1b[i3], d);
            float sb = 0.f, sc = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = ht[i3][r] > 0.f ? d[r] : 0.f;                // (rounded to bf16 only as the operand of dW0 below)
                d[r] = v;
                sb += v;
                sc = fmaf(v, x32[r], sc);
            }
            db0r[i3] += sb;
            dW0c[i3] += sc;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const bf16x8 aq = pack8(make_float4(d[8 * q], d[8 * q + 1], d[8 * q + 2], d[8 * q + 3]),
                                        make_float4(d[8 * q + 4], d[8 * q + 5], d[8 * q + 6], d[8 * q + 7]));
                const bf16x8 xq = *reinterpret_cast<const bf16x8 *>(xb + ul * XS16 + (2 * q + h) * 8);
                dW0[i3] = mfma16(aq, xq, dW0[i3]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- flush (same slab layout as the f32 kernel)
    float *S = A.slab + (size_t)pair * SLAB;
#pragma unroll
    for (int i3 = 0; i3 < 3; ++i3) {
        const int ub = 96 * g + 32 * i3;
#pragma unroll
        for (int r = 0; r < 16; ++r) S[(ub + acc_row(r, h)) * TIN + ul] = dW0[i3][r];
        const int u = ub + ul;
        const float c32 = dW0c[i3] + __shfl_xor(dW0c[i3], 32);
        const float b = db0r[i3] + __shfl_xor(db0r[i3], 32);
        float w1[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) w1[c] = dW1r[i3][c] + __shfl_xor(dW1r[i3][c], 32);
        if (h == 0) {
            S[u * TIN + 32] = c32;
            S[N_DW0 + N_DW1 + u] = b;
#pragma unroll
            for (int c = 0; c < 3; ++c) S[N_DW0 + c * THID + u] = w1[c];
        }
    }
    float d0 = db1r[0], d1 = db1r[1];
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) { d0 += __shfl_xor(d0, off); d1 += __shfl_xor(d1, off); }
    if (ul == 0 && g == 0) {
        S[N_DW0 + N_DW1 + THID + h] = d0;
        if (h == 0) { S[N_DW0 + N_DW1 + THID + 2] = d1; S[N_DW0 + N_DW1 + THID + 3] = 0.f; }
    }
}

// ---- split-fp16 twin (the f32 engine, round 4) ------------------------------------------------------------------------
// The bf16 twin's scheme with fp32-accurate products: every MFMA operand as two fp16 planes (x = hi + lo, hi = fp16(x),
// lo = fp16(x - hi)), a product as hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_f16 into the same fp32 accumulator
// (csrc/mlp_split.hip).  Scales (powers of two, exact): W0 and W1 planes hold 64 w (residuals of ~0.1-sized weights stay
// normal fp16 numbers), so the recomputed Ht is carried as 64 Ht -- the ReLU and the masks do not care, the dW1 sums are
// divided by 64 at the flush; the output gradient dzt (1e-3 .. 1e-9) is multiplied by S = 2^k with max |dzt| S in [16, 32)
// (the step's max |dzt|: `amax`, esr_absmax), so dZt is carried as 64 S dZt and dW0 / db0 are divided by 64 S at the flush.
// A tile whose gradients are far below the step's largest loses RELATIVE precision in its residual plane (2^-25 absolute of
// a scale where the largest value is ~1e3) -- in a sum over all tiles that is 2^-35 of the dominant terms.
typedef _Float16 tf16x8 __attribute__((ext_vector_type(8)));
constexpr float TW_SCALE = 64.f;
constexpr int WAVE_LDSS = (4 + 1) * XS * 4;                     // bytes per wave: dzt rows 0..3 + Xt row 32 (f32)
constexpr int W1S_BYTES = 2 * THID * 8 * 2;                     // 64 W1 as two fp16 planes [192][8] (B of dHt^T: k = output channel)
constexpr int W0S_BYTES = 2 * THID * W0B_U * 2;                 // W0 as two fp16 planes [192][W0B_U]

__device__ __forceinline__ void tsplit8(const float (&v)[8], tf16x8 &hi, tf16x8 &lo)
{
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const _Float16 hh = (_Float16)v[e];
        hi[e] = hh;
        lo[e] = (_Float16)(v[e] - (float)hh);
    }
}
__device__ __forceinline__ f32x16 tmfma(tf16x8 a, tf16x8 b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
// a.b with both operands as plane pairs: three MFMAs into one accumulator
__device__ __forceinline__ f32x16 tmfma3(tf16x8 ah, tf16x8 al, tf16x8 bh, tf16x8 bl, f32x16 c)
{
    c = tmfma(ah, bl, c);
    c = tmfma(ah, bh, c);
    return tmfma(al, bh, c);
}

struct ToneWgSplitArgs {
    ToneWgArgs a;
    const float *amax;                      // max |dzt| of the step (device)
};

// x -> (fp16(x), fp16(x - fp16(x))) for two values into slots I0, I0 + 1 of the plane registers: mlp_split.hip's put_pair /
// put_residual_pair (one v_cvt_pk + two v_fma_mix + one v_cvt_pk per pair; the C form -- tsplit8 -- costs ~5 instructions per
// value).  The operands must be VALU results or loaded values, never a fresh MFMA result (tests/test_isa.py).
typedef _Float16 tf16x2 __attribute__((ext_vector_type(2)));
template <int I0>
__device__ __forceinline__ void tsplit_pair(tf16x8 &hi, tf16x8 &lo, float v0, float v1)
{
    const tf16x2 hh = {(_Float16)v0, (_Float16)v1};
    const unsigned u = __builtin_bit_cast(unsigned, hh);
    unsigned r;
    float d0, d1;
    asm volatile("v_fma_mix_f32 %1, %3, -1.0, %4 op_sel_hi:[1,0,0]\n\t"
                 "v_fma_mix_f32 %2, %3, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                 "v_cvt_pk_f16_f32 %0, %1, %2"
                 : "=v"(r), "=&v"(d0), "=&v"(d1) : "v"(u), "v"(v0), "v"(v1));
    const tf16x2 ll = __builtin_bit_cast(tf16x2, r);
    hi[I0] = hh[0]; hi[I0 + 1] = hh[1];
    lo[I0] = ll[0]; lo[I0 + 1] = ll[1];
}
__device__ __forceinline__ void tsplit8f(const float (&v)[8], tf16x8 &hi, tf16x8 &lo)
{
    tsplit_pair<0>(hi, lo, v[0], v[1]);
    tsplit_pair<2>(hi, lo, v[2], v[3]);
    tsplit_pair<4>(hi, lo, v[4], v[5]);
    tsplit_pair<6>(hi, lo, v[6], v[7]);
}

// Round 6.  (i) The recomputed hidden layer is the FORWARD's, bit for bit: mlp_fwd_split_kernel<1> forms a pre-activation as
// fma(acc, 1/64, bias) with acc = the k-steps' products w1.x2, w1.x1, w2.x1 in that order from a zero accumulator; here the
// same products (A and B exchanged: the sample is on the lane's row) in the same order, the same fma, the same integer ReLU --
// so the branch this kernel takes at a unit is the branch the forward saved in its mask, and the step is self-consistent at
// kink units without reading the masks back (tests/test_gpu_split.py::test_tone_wgrad_takes_the_forwards_branches; until round
// 6 the accumulator started from 64 b and the products came in another order: a unit within summation noise of its kink could
// take the other branch, the one allowance the oracle comparison had to make).  (ii) The kernel's instruction count: the
// planes by mlp_split.hip's pair conversions, the B operand of dW0 (X rows over permuted samples) straight from global memory
// as two planes -- no transposition through LDS (32 two-byte writes + 12 reads per tile) --, everything inside 256 registers:
// no accumulation-register moves (176 per tile before) and two waves per SIMD.
__global__ void __launch_bounds__(256, 2) tone_wgrad_split_t_kernel(ToneWgSplitArgs AS)
{
    const ToneWgArgs &A = AS.a;
    extern __shared__ __attribute__((aligned(16))) unsigned char ldss[];
    _Float16 *w0h = reinterpret_cast<_Float16 *>(ldss), *w0l = w0h + THID * W0B_U;      // [192][W0B_U] each: 64 W0[u][x]
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, ul = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    _Float16 *w1hp = reinterpret_cast<_Float16 *>(ldss + W0S_BYTES), *w1lp = w1hp + THID * 8;   // [192][8] each: 64 W1[c][u], c < 3
    float *lz = reinterpret_cast<float *>(ldss + W0S_BYTES + W1S_BYTES) + wv * (5 * XS);          // this wave's dzt rows 0..3 and Xt row 32, [5][XS]
    float *lx32 = lz + 4 * XS;
    for (int i = tid; i < THID * W0B_U; i += 256) {
        const int u = i / W0B_U, x = i % W0B_U;
        const float w = x < TIN ? TW_SCALE * A.W0[u * TIN + x] : 0.f;
        const _Float16 hh = (_Float16)w;
        w0h[i] = hh;
        w0l[i] = (_Float16)(w - (float)hh);
    }
    for (int i = tid; i < THID * 8; i += 256) {
        const int u = i >> 3, c = i & 7;
        const float w = c < TOUT ? TW_SCALE * A.W1[c * THID + u] : 0.f;
        const _Float16 hh = (_Float16)w;
        w1hp[i] = hh;
        w1lp[i] = (_Float16)(w - (float)hh);
    }
    __syncthreads();
    // the step's gradient scale: 2^k with max |dzt| 2^k in [16, 32) (1 for an all-zero or non-finite maximum)
    float S = 1.f;
    {
        const float am = AS.amax ? *AS.amax : 0.f;
        const int ez = (__float_as_int(am) >> 23) & 0xff;
        if (am > 0.f && ez != 0 && ez != 0xff) {
            int k = 131 - ez;
            k = k < -100 ? -100 : (k > 100 ? 100 : k);
            S = __int_as_float((127 + k) << 23);
        }
    }

    const int g = wv & 1;                                                   // this wave's hidden units [96 g, 96 g + 96)
    float b0r[3];
#pragma unroll
    for (int i3 = 0; i3 < 3; ++i3) b0r[i3] = A.b0[96 * g + 32 * i3 + ul];
    f32x16 dW0[3];
    zero_tiles<3>(dW0);
    float dW0c[3], dW1r[3][3], db0r[3], db1r[2] = {0.f, 0.f};
#pragma unroll
    for (int i3 = 0; i3 < 3; ++i3) {
        dW0c[i3] = 0.f; db0r[i3] = 0.f;
        dW1r[i3][0] = dW1r[i3][1] = dW1r[i3][2] = 0.f;
    }
    const int pair = blockIdx.x * 2 + (wv >> 1), npairs = gridDim.x * 2;
    // a tile's rows as this lane needs them: A of Ht^T (sample ul: input rows 16 q + 8 h + e and 32), B of dW0 (input row ul:
    // the samples of k-slots (q, h, e) = acc_row(8 q + e, h): two runs of four), dzt (rows h and 2)
    float xn[17], zn[2];
    auto fetch = [&](int t) {
        const bool live = t < A.t1;
        const float *X = A.Xt + (size_t)(live ? t : A.t0) * XT_ROWS * 32;
        const float *Zt = A.dzt + (size_t)(live ? t : A.t0) * 4 * 32 + ul;
#pragma unroll
        for (int j = 0; j < 16; ++j) xn[j] = X[(16 * (j >> 3) + 8 * h + (j & 7)) * 32 + ul];
        xn[16] = X[(32 + 8 * h) * 32 + ul];
        zn[0] = Zt[h * 32];
        zn[1] = h == 0 ? Zt[2 * 32] : 0.f;
    };
    if (A.t0 + pair < A.t1) fetch(A.t0 + pair);
    for (int t = A.t0 + pair; t < A.t1; t += npairs) {
        // the B operand of dW0: this tile's row ul over the permuted samples (second read of the tile: L2), used a block from now
        float4 bn[4];
        {
            const float *X = A.Xt + (size_t)t * XT_ROWS * 32 + ul * 32 + 4 * h;
#pragma unroll
            for (int k = 0; k < 4; ++k) bn[k] = *reinterpret_cast<const float4 *>(X + 16 * (k >> 1) + 8 * (k & 1));
        }
        // this tile's A operands as planes, then the next tile's loads
        tf16x8 a8h[3], a8l[3], xqh[2], xql[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = xn[8 * q + e];
            tsplit8f(v, a8h[q], a8l[q]);
        }
        {
            const float v[8] = {xn[16], 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            tsplit8(v, a8h[2], a8l[2]);
        }
        const float za0 = zn[0], za1 = zn[1];
        lz[h * XS + ul] = za0;
        if (h == 0) { lz[2 * XS + ul] = za1; lx32[ul] = xn[16]; }
        db1r[0] += za0; db1r[1] += za1;
        // (S dzt)^T as the A operand of dHt^T: slots 0..2 of half 0 = dzt rows 0..2 of this lane's sample
        const float z1 = __shfl_xor(za0, 32);
        tf16x8 zah, zal;
        {
            const float v[8] = {h == 0 ? S * za0 : 0.f, h == 0 ? S * z1 : 0.f, h == 0 ? S * za1 : 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            tsplit8(v, zah, zal);
        }
        fetch(t + npairs);                              // (past the range: re-reads tile t0, never used)
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        // one 32-unit block at a time
#pragma unroll
        for (int i3 = 0; i3 < 3; ++i3) {
            // ---- the forward's pre-activation, bit for bit: acc = sum over the k-steps of w1.x2 + w1.x1 + w2.x1 (64 w planes)
            f32x16 acc = zero16;
            const int wrow = (96 * g + 32 * i3 + ul) * W0B_U + 8 * h;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const tf16x8 bh = *reinterpret_cast<const tf16x8 *>(w0h + wrow + 16 * q);
                const tf16x8 bl = *reinterpret_cast<const tf16x8 *>(w0l + wrow + 16 * q);
                acc = tmfma(a8l[q], bh, acc);
                acc = tmfma(a8h[q], bh, acc);
                acc = tmfma(a8h[q], bl, acc);
            }
            // value = acc / 64 + bias, ReLU as max((int) bits, 0): mlp_split.hip's epilogue (compiler-visible reads of the MFMA result)
            float ht[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int bits = __float_as_int(fmaf(acc[r], 1.f / TW_SCALE, b0r[i3]));
                ht[r] = __int_as_float(bits > 0 ? bits : 0);
            }
            // ---- dW1[c][u] += sum_s dzt[c][s] Ht[u][s]: fp32 vector sums
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 a = *reinterpret_cast<const float4 *>(lz + 0 * XS + 8 * q + 4 * h);
                const float4 b = *reinterpret_cast<const float4 *>(lz + 1 * XS + 8 * q + 4 * h);
                const float4 c = *reinterpret_cast<const float4 *>(lz + 2 * XS + 8 * q + 4 * h);
                const float za4[4] = {a.x, a.y, a.z, a.w}, zb4[4] = {b.x, b.y, b.z, b.w}, zc4[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float hv = ht[4 * q + i];
                    dW1r[i3][0] = fmaf(za4[i], hv, dW1r[i3][0]);
                    dW1r[i3][1] = fmaf(zb4[i], hv, dW1r[i3][1]);
                    dW1r[i3][2] = fmaf(zc4[i], hv, dW1r[i3][2]);
                }
            }
            // ---- 64 S dHt^T = (S dzt)^T (64 W1), masked by the forward's branch; then 64 S dW0 += (64 S dZt) Xt^T
            const int w1row = (96 * g + 32 * i3 + ul) * 8;
            const tf16x8 w1h = *reinterpret_cast<const tf16x8 *>(w1hp + w1row), w1l = *reinterpret_cast<const tf16x8 *>(w1lp + w1row);
            f32x16 d = tmfma3(zah, zal, w1h, w1l, zero16);      // (half 1's k-slots: zah / zal are zero there)
            float sb = 0.f, sc = 0.f, dv[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 x4 = *reinterpret_cast<const float4 *>(lx32 + 8 * q + 4 * h);       // Xt row 32 at this half-wave's samples
                const float x32[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 4 * q + i;
                    const float v = __float_as_int(ht[r]) > 0 ? d[r] : 0.f;
                    dv[r] = v;
                    sb += v;
                    sc = fmaf(v, x32[i], sc);
                }
            }
            db0r[i3] += sb;
            dW0c[i3] += sc;
            if (i3 == 0) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const float b[8] = {bn[2 * q].x, bn[2 * q].y, bn[2 * q].z, bn[2 * q].w, bn[2 * q + 1].x, bn[2 * q + 1].y, bn[2 * q + 1].z, bn[2 * q + 1].w};
                    tsplit8f(b, xqh[q], xql[q]);
                }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = dv[8 * q + e];
                tf16x8 aqh, aql;
                tsplit8f(v, aqh, aql);
                dW0[i3] = tmfma3(aqh, aql, xqh[q], xql[q], dW0[i3]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // ---- flush (same slab layout as the f32 kernel), scales removed
    const float ig = 1.f / (TW_SCALE * S);
    float *Sl = A.slab + (size_t)pair * SLAB;
#pragma unroll
    for (int i3 = 0; i3 < 3; ++i3) {
        const int ub = 96 * g + 32 * i3;
#pragma unroll
        for (int r = 0; r < 16; ++r) Sl[(ub + acc_row(r, h)) * TIN + ul] = dW0[i3][r] * ig;
        const int u = ub + ul;
        const float c32 = dW0c[i3] + __shfl_xor(dW0c[i3], 32);
        const float b = db0r[i3] + __shfl_xor(db0r[i3], 32);
        float w1[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) w1[c] = dW1r[i3][c] + __shfl_xor(dW1r[i3][c], 32);
        if (h == 0) {
            Sl[u * TIN + 32] = c32 * ig;
            Sl[N_DW0 + N_DW1 + u] = b * ig;
#pragma unroll
            for (int c = 0; c < 3; ++c) Sl[N_DW0 + c * THID + u] = w1[c];
        }
    }
    float d0 = db1r[0], d1 = db1r[1];
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) { d0 += __shfl_xor(d0, off); d1 += __shfl_xor(d1, off); }
    if (ul == 0 && g == 0) {
        Sl[N_DW0 + N_DW1 + THID + h] = d0;
        if (h == 0) { Sl[N_DW0 + N_DW1 + THID + 2] = d1; Sl[N_DW0 + N_DW1 + THID + 3] = 0.f; }
    }
}

// gw[e] += sum over the wave slabs, all four outputs in one launch
__global__ void __launch_bounds__(256) tone_wgrad_reduce_kernel(const float *__restrict__ slab, int n_slabs,
                                                                float *gw0, float *gw1, float *gb0, float *gb1)
{
    constexpr int PG = 32;
    const int groups = (n_slabs + PG - 1) / PG;
    const int total = (SLAB - 1) * groups;                              // the pad word of db1 is skipped
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int e = i % (SLAB - 1), g = i / (SLAB - 1);
        const int p1 = min((g + 1) * PG, n_slabs);
        float acc = 0.f;
        for (int p = g * PG; p < p1; ++p) acc += slab[(size_t)p * SLAB + e];
        float *dst = e < N_DW0 ? gw0 + e : e < N_DW0 + N_DW1 ? gw1 + (e - N_DW0)
                   : e < N_DW0 + N_DW1 + THID ? gb0 + (e - N_DW0 - N_DW1) : gb1 + (e - N_DW0 - N_DW1 - THID);
        atomicAdd(dst, acc);
    }
}

}  // namespace

ESR_API int64_t esr_tone_wgrad_scratch_floats(void) { return (int64_t)1024 * SLAB; }

ESR_API int esr_tone_wgrad_recompute(const float *Xt, const float *dzt, const float *W0, const float *b0, const float *W1,
                                     int32_t t0, int32_t t1, float *gw0, float *gb0, float *gw1, float *gb1,
                                     float *scratch, int64_t scratch_floats, void *stream)
{
    if (t0 < 0 || t1 < t0) return ESR_EINVAL;
    if (t1 == t0) return 0;
    if (!Xt || !dzt || !W0 || !b0 || !W1 || !gw0 || !gb0 || !gw1 || !gb1 || !scratch) return ESR_EINVAL;
    const int n_tiles = t1 - t0;
    int grid = (n_tiles + 1) / 2;                                      // a workgroup = two wave pairs = two tiles at a time
    if (grid > 256) grid = 256;                                        // one workgroup (4 waves, 1 per SIMD) per CU
    if ((int64_t)grid * 2 * SLAB > scratch_floats) return ESR_ECAP;
    constexpr size_t lds_bytes = (size_t)(W0T_FLOATS + 4 * WAVE_LDS) * sizeof(float);
    static std::atomic<uint64_t> optin{0};
    if (int rc = esr_lds_optin(reinterpret_cast<const void *>(&tone_wgrad_t_kernel), lds_bytes, optin)) return rc;
    ToneWgArgs A = {Xt, dzt, W0, b0, W1, t0, t1, scratch};
    hipStream_t s = esr_stream(stream);
    tone_wgrad_t_kernel<<<grid, 256, lds_bytes, s>>>(A);
    ESR_CHECK_LAUNCH();
    const int n_slabs = grid * 2;
    tone_wgrad_reduce_kernel<<<esr_grid_for((int64_t)(SLAB - 1) * ((n_slabs + 31) / 32), 256, 1024), 256, 0, s>>>(
        scratch, n_slabs, gw0, gw1, gb0, gb1);
    ESR_CHECK_LAUNCH();
    return 0;
}

// bf16-operand twin (the bf16 engine): same arguments; Xt / dzt / the weights stay fp32 in memory and are rounded to bf16
// where the bf16 engine's kernels round them.
ESR_API int esr_tone_wgrad_recompute_bf16(const float *Xt, const float *dzt, const float *W0, const float *b0, const float *W1,
                                          int32_t t0, int32_t t1, float *gw0, float *gb0, float *gw1, float *gb1,
                                          float *scratch, int64_t scratch_floats, void *stream)
{
    if (t0 < 0 || t1 < t0) return ESR_EINVAL;
    if (t1 == t0) return 0;
    if (!Xt || !dzt || !W0 || !b0 || !W1 || !gw0 || !gb0 || !gw1 || !gb1 || !scratch) return ESR_EINVAL;
    const int n_tiles = t1 - t0;
    int grid = (n_tiles + 1) / 2;
    if (grid > 512) grid = 512;                                        // two workgroups (8 waves) per CU
    if ((int64_t)grid * 2 * SLAB > scratch_floats) return ESR_ECAP;
    constexpr size_t lds_bytes = (size_t)W0B_BYTES + 4 * WAVE_LDS16;
    ToneWgArgs A = {Xt, dzt, W0, b0, W1, t0, t1, scratch};
    hipStream_t s = esr_stream(stream);
    tone_wgrad16_t_kernel<<<grid, 256, lds_bytes, s>>>(A);
    ESR_CHECK_LAUNCH();
    const int n_slabs = grid * 2;
    tone_wgrad_reduce_kernel<<<esr_grid_for((int64_t)(SLAB - 1) * ((n_slabs + 31) / 32), 256, 1024), 256, 0, s>>>(
        scratch, n_slabs, gw0, gw1, gb0, gb1);
    ESR_CHECK_LAUNCH();
    return 0;
}

// Split-fp16 twin (the f32 engine): same arguments + amax, a device pointer to max |dzt| over the step's tiles (esr_absmax);
// fp32 results of the f32 kernel's accuracy class with the products on the 16-bit matrix cores.
ESR_API int esr_tone_wgrad_recompute_split(const float *Xt, const float *dzt, const float *W0, const float *b0, const float *W1,
                                           const float *amax, int32_t t0, int32_t t1, float *gw0, float *gb0, float *gw1,
                                           float *gb1, float *scratch, int64_t scratch_floats, void *stream)
{
    if (t0 < 0 || t1 < t0) return ESR_EINVAL;
    if (t1 == t0) return 0;
    if (!Xt || !dzt || !W0 || !b0 || !W1 || !amax || !gw0 || !gb0 || !gw1 || !gb1 || !scratch) return ESR_EINVAL;
    const int n_tiles = t1 - t0;
    int grid = (n_tiles + 1) / 2;
    if (grid > 512) grid = 512;                                        // two workgroups (8 waves) per CU
    if ((int64_t)grid * 2 * SLAB > scratch_floats) return ESR_ECAP;
    constexpr size_t lds_bytes = (size_t)W0S_BYTES + W1S_BYTES + 4 * WAVE_LDSS;
    static std::atomic<uint64_t> optin{0};
    if (int rc = esr_lds_optin(reinterpret_cast<const void *>(&tone_wgrad_split_t_kernel), lds_bytes, optin)) return rc;
    ToneWgSplitArgs A = {{Xt, dzt, W0, b0, W1, t0, t1, scratch}, amax};
    hipStream_t s = esr_stream(stream);
    tone_wgrad_split_t_kernel<<<grid, 256, lds_bytes, s>>>(A);
    ESR_CHECK_LAUNCH();
    const int n_slabs = grid * 2;
    tone_wgrad_reduce_kernel<<<esr_grid_for((int64_t)(SLAB - 1) * ((n_slabs + 31) / 32), 256, 1024), 256, 0, s>>>(
        scratch, n_slabs, gw0, gw1, gb0, gb1);
    ESR_CHECK_LAUNCH();
    return 0;
}
