// The f32 engine's RADIANCE FORWARD on the 16-bit matrix cores, with fp32 results (round 4).
//
// On gfx950 v_mfma_f32_32x32x2_f32 runs at 1/16 of the 16-bit MFMA rate (157 vs 2500 TFLOP/s) and ON the vector lanes; the
// radiance forward -- 142.5 GFLOP per C2 step -- was the step's dominant launch at 1.06 ms (0.85 of the f32 matrix peak).
// Here every operand is split into TWO fp16 planes, x = x1 + x2 with x1 = fp16(x), x2 = fp16(x - x1) (the subtraction is
// exact in fp32), and a product is formed as
//     w . x  =  w1 . x1  +  w1 . x2  +  w2 . x1                   (+ w2 . x2, dropped: 2^-22 relative, below fp32's own rounding)
// on v_mfma_f32_32x32x16_f16 with fp32 accumulation, all three into ONE accumulator: three 16-bit MFMAs per k-step instead
// of eight f32 ones of a quarter of the depth -- 16/3 of the f32 matrix rate.  Two fp16 planes carry 22 mantissa bits as
// long as the residual x2 is a NORMAL fp16 number (|x| >= 0.125); below that it is rounded to 2^-25 ABSOLUTE (3e-8).  The
// weights (~0.07 in a 192-wide layer) are therefore stored times 64 (mlp_common.h: SPLIT_W_SCALE; the epilogue's bias
// multiply-add takes the 1/64), activations and inputs are O(1) and stay unscaled: their worst case is the 3e-8 absolute,
// i.e. the two-plane error of a value of 0.125.  Through the four layers of the net the result differs from a
// double-precision evaluation by 4e-7 .. 8e-7 of the layer's largest value, the same as torch's own fp32 chain (5e-7;
// tools/ubench/mfma_f32_shapes.hip for the rates, tests/test_gpu_split.py for the accuracy).  With bf16 planes the same
// three products give 6e-6 .. 9e-6: fp16's three extra mantissa bits per plane are what makes two planes enough; a |value|
// above 65504 (a weight above 1023) would overflow the first plane (the fp32 MFMA path stays available: ESR_SPLIT_FWD=0).
// (Until the middle of round 4 the residual planes were scaled by 2048 and summed in accumulators of their own: three
// accumulator sets whose every value had to be fetched from the accumulation registers and recombined in the epilogue --
// the epilogue, not the matrix pipe, bounded the kernel: tools/ubench/split_stamps.hip.)
//
// Everything OUTSIDE the products is the f32 engine's: fp32 input tile X, fp32 bias add, ReLU, the saved hidden tiles H
// (fp32, tile-major) and ReLU masks in mlp.hip's formats -- the f32 input-gradient and weight-gradient kernels consume them
// unchanged -- and the fp32 output rows.
//
// Structure: mlp_bf16.hip's shared-weights scheme, re-cut for the register budget.  A layer's input AND output live in
// registers as two fp16 planes each (2 x 48 + 2 x 48), beside two accumulator pairs (main / residual sums of the tile in
// flight and of the tile in its epilogue), the staged weights and the next group's inputs: ~400 registers, i.e. ONE wave
// per SIMD, four waves (= four 32-sample tiles) per workgroup and CU.  The weights (two planes: 148 KB per hidden layer)
// are staged through LDS in STEPS of one pair of output tiles (48 KB, double-buffered): a step's chunks are contiguous in
// the packed buffer, all 256 threads request the next step's 48 KB at the top of a step and write it to the other buffer at
// its end, one barrier per step.  Inside a step the tile order of mlp_bf16.hip's lds_layer16_tiled: a tile's epilogue
// (scale + bias, ReLU, fp32 stores, mask bits, the split into the next layer's planes) is issued behind the NEXT tile's
// MFMAs.
#include "mlp_common.h"

#include <type_traits>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// in-kernel time stamps for tools/ubench/split_stamps.hip (nothing in the product build)
#ifndef ESR_SPLIT_STAMP
#define ESR_SPLIT_STAMP(i)
#endif

namespace {

constexpr int SPW = 4;                                      // waves per workgroup = tiles per group
constexpr float SPLIT_RANGE = 60000.f;                      // hidden activations at or above this raise the range flag (fp16 ends at 65504)
#ifndef ESR_SPLIT_WRING
#define ESR_SPLIT_WRING 3
#endif
constexpr int WRING = ESR_SPLIT_WRING;
#ifndef ESR_NS_DEFAULT
#define ESR_NS_DEFAULT 0                                    // esr_mlp_split_variant's initial value
#endif                      // k-steps of weight operands in flight per wave (tools/ubench/split_stamps.hip: 3 / 4 / 5 / 6)

__device__ __forceinline__ f32x16 mfma_h(f16x8 a, f16x8 b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

template <int I, int N, typename F>
__device__ __forceinline__ void sfor(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, N>(f);
    }
}

// x -> (fp16(x), fp16(x - fp16(x))) for eight values
__device__ __forceinline__ void split8(const float (&v)[8], f16x8 &p1, f16x8 &p2)
{
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const _Float16 h = (_Float16)v[i];
        p1[i] = h;
        p2[i] = (_Float16)(v[i] - (float)h);
    }
}

// Two fp16 values into slots i0, i0 + 1 of a plane register, PINNED where they are computed: the planes of a layer are first
// read by the next layer's MFMAs, and LLVM sinks a computation towards its first use -- without the pin the conversions of
// all six tiles of a layer left their micro-slices and ran as one block of ~300 instructions behind the layer's last MFMA
// (tools/ubench/split_stamps.hip; the empty asm's operand is a VALU result: no MFMA hazard involved).
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
template <int I0>
__device__ __forceinline__ void put_pair(f16x8 &dst, float a, float b)
{
    const f16x2 hh = {(_Float16)a, (_Float16)b};
    unsigned u = __builtin_bit_cast(unsigned, hh);
    asm volatile("" : "+v"(u));
    const f16x2 pinned = __builtin_bit_cast(f16x2, u);
    dst[I0] = pinned[0];
    dst[I0 + 1] = pinned[1];
}

// second plane of a value pair: fp16(v - x1) with x1 read from the packed first-plane dword -- v_fma_mix_f32 takes the fp16
// half directly (one instruction per value instead of a conversion and a subtraction; the difference is exact either way).
// Operands are VALU results (phase 1's conversion, phase 0's values): nothing here reads an MFMA result.
template <int I0>
__device__ __forceinline__ void put_residual_pair(f16x8 &dst, const f16x8 &first, float v0, float v1)
{
    const unsigned u = __builtin_bit_cast(u32x4, first)[I0 >> 1];
    unsigned r;
    float d0, d1;
    // ONE asm statement: between two statements the compiler puts an `s_nop 0` when the second reads the first's result
    // (VALU -> VALU needs none on this hardware); volatile = the pin of put_pair
    asm volatile("v_fma_mix_f32 %1, %3, -1.0, %4 op_sel_hi:[1,0,0]\n\t"
                 "v_fma_mix_f32 %2, %3, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                 "v_cvt_pk_f16_f32 %0, %1, %2"
                 : "=v"(r), "=&v"(d0), "=&v"(d1) : "v"(u), "v"(v0), "v"(v1));
    const f16x2 pinned = __builtin_bit_cast(f16x2, r);
    dst[I0] = pinned[0];
    dst[I0 + 1] = pinned[1];
}

template <int KIND, bool BWD = false> struct SplitSteps {
    static constexpr SplitLayout L = BWD ? split_layout_t(KIND) : split_layout(KIND);
    static constexpr int BASE_CHUNK = BWD ? split_layout(KIND).total_chunks : 0;      // the transposed planes follow the forward ones
    static constexpr int NL = L.n_layers;
    static constexpr int n_steps()
    {
        int n = 0;
        for (int l = 0; l < NL; ++l) n += L.pairs[l];
        return n;
    }
    static constexpr int NS = n_steps();
    static constexpr int layer_of(int s)
    {
        int l = 0;
        while (s >= L.pairs[l]) { s -= L.pairs[l]; ++l; }
        return l;
    }
    static constexpr int pair_of(int s)
    {
        int l = 0;
        while (s >= L.pairs[l]) { s -= L.pairs[l]; ++l; }
        return s;
    }
    static constexpr int tiles_in(int s)                   // output tiles of the step (2, or 1 for an odd tail / the output layer)
    {
        const int l = layer_of(s), p = pair_of(s);
        return L.tiles_out[l] - 2 * p >= 2 ? 2 : 1;
    }
    static constexpr int chunks(int s) { return tiles_in(s) * 2 * L.ks[layer_of(s)]; }
    static constexpr int chunk0(int s) { return BASE_CHUNK + L.off_chunk[layer_of(s)] + pair_of(s) * 2 * 2 * L.ks[layer_of(s)]; }
    static constexpr int max_chunks()
    {
        int m = 0;
        for (int s = 0; s < NS; ++s) m = chunks(s) > m ? chunks(s) : m;
        return m;
    }
    static constexpr int BUF = max_chunks() * 1024;
    static constexpr int BIAS_FLOATS = 32 * MAX_HID_TILES;
    // a net whose planes fit 64 KB (the tone mapper: 60 chunks each way) keeps them RESIDENT in LDS for the kernel's whole
    // life: no per-step staging, no step barriers -- a two-layer net's steps are a handful of MFMAs each, far shorter than
    // the global -> register -> LDS round trip that used to sit between them
    static constexpr bool RES = L.total_chunks <= 64;
    static constexpr int WBYTES = RES ? L.total_chunks * 1024 : 2 * BUF;
    static constexpr int LDS_BYTES = WBYTES + NL * BIAS_FLOATS * 4;
    static constexpr int PRE = (max_chunks() * 64 + 64 * SPW - 1) / (64 * SPW);     // 16-byte pieces per thread and step
};

struct SplitSeg {
    const float *packed32;     // esr_mlp_pack buffer (biases)
    const _Float16 *planes;    // esr_mlp_pack_batch's split planes
    int t0, t1, save, crow;
    float *zout;
    int b0, nb;                // workgroups [b0, b0 + nb) of the launch
};
constexpr int MAX_SPLIT_SEG = 3;
struct SplitBatch {
    const float *X;
    float *H[3];
    unsigned *M[3];
    unsigned *range;           // optional: set to 1 when a hidden activation leaves fp16's range (esr_mlp_split_range_flag)
    int nseg;
    SplitSeg seg[MAX_SPLIT_SEG];
};

__device__ __forceinline__ void step_barrier()
{
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xc07f);                    // lgkmcnt(0): this wave's LDS reads / writes of the step are done
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// workgroups per CU the register budget is cut for: the 192-wide radiance net needs a whole SIMD's registers per wave; the
// 128-wide nets and the two-layer tone mapper fit two waves per SIMD (and 2 x 66 KB of LDS), which lets one wave's MFMAs run
// under the other's epilogue
constexpr int split_occ(int kind) { return kind == ESR_MLP_RADIANCE ? 1 : 2; }

template <int KIND>
__global__ void __launch_bounds__(64 * SPW, split_occ(KIND)) mlp_fwd_split_kernel(SplitBatch AB)
{
    using S = SplitSteps<KIND>;
    constexpr NetDesc D = net_desc(KIND);
    constexpr SplitLayout L = S::L;
    constexpr PackLayout L32 = pack_layout(KIND);
    constexpr int NL = S::NL, NHID = NL - 1, HT = D.hid_tiles, KS1 = L.ks[0], NS = S::NS;
    constexpr unsigned HBYTES = HT * 32 * 32 * 4, MBYTES = (HT / 2) * 256;
    static_assert((NL == 4 || NL == 2) && HT % 2 == 0, "the four-layer nets (radiance, BRDF, emission) and the tone mapper");
    // segment of this workgroup
    SplitSeg A = AB.seg[0];
#pragma unroll
    for (int k = 1; k < MAX_SPLIT_SEG; ++k)
        if (k < AB.nseg && (int)blockIdx.x >= AB.seg[k].b0) A = AB.seg[k];
    const int blk0 = A.b0, nblk = A.nb;
    extern __shared__ __attribute__((aligned(16))) unsigned char wl[];          // buffer 0 | buffer 1 | biases
    float *bias_l = reinterpret_cast<float *>(wl + S::WBYTES);
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, s_ = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntiles = A.t1 - A.t0, ngroups = (ntiles + SPW - 1) / SPW;
    for (int i = tid; i < NL * S::BIAS_FLOATS; i += 64 * SPW) {
        const int l = i / S::BIAS_FLOATS, k = i % S::BIAS_FLOATS;
        bias_l[i] = k < L32.tiles_out[l] * 32 ? A.packed32[L32.off_bf[l] + k] : 0.f;
    }
    const rsrc_t WP = make_rsrc(A.planes, (unsigned)(L.total_chunks * 1024));
    u32x4 pre[S::PRE];
    // request / write the chunks of step `st` (compile-time) into LDS buffer `dst`
    auto stage_load = [&](auto ST) __attribute__((always_inline)) {
        // (constexpr locals: as plain call arguments the step table's lookups -- loops over the layer list -- were evaluated at
        //  RUN time for the later steps, a chain of scalar loads per use: steps 5-8 took 10 k clocks instead of 2.5 k)
        constexpr int st = decltype(ST)::value, pieces = S::chunks(st) * 64, base = S::chunk0(st) * 1024;
#pragma unroll
        for (int k = 0; k < S::PRE; ++k)
            if (k * 64 * SPW < pieces) pre[k] = __builtin_amdgcn_raw_buffer_load_b128(WP, (tid + 64 * SPW * k) * 16, base, 0);
    };
    auto stage_store = [&](auto ST, unsigned char *dst) __attribute__((always_inline)) {
        constexpr int st = decltype(ST)::value, pieces = S::chunks(st) * 64;
#pragma unroll
        for (int k = 0; k < S::PRE; ++k)
            if (k * 64 * SPW < pieces && tid + 64 * SPW * k < pieces)
                *reinterpret_cast<u32x4 *>(dst + (size_t)(tid + 64 * SPW * k) * 16) = pre[k];
    };
    // one 16-byte piece per thread: the step's last tile issues these behind its MFMAs (the other LDS buffer is idle since
    // the previous step's barrier), instead of 12 writes + their wait between the last MFMA and the barrier
    auto stage_piece = [&](auto ST, auto KC, unsigned char *dst) __attribute__((always_inline)) {
        constexpr int st = decltype(ST)::value, k = decltype(KC)::value, pieces = S::chunks(st) * 64;
        if constexpr (k * 64 * SPW < pieces)
            if (tid + 64 * SPW * k < pieces) *reinterpret_cast<u32x4 *>(dst + (size_t)(tid + 64 * SPW * k) * 16) = pre[k];
    };
    if constexpr (S::RES) {
        for (int i = tid; i < L.total_chunks * 64; i += 64 * SPW)
            *reinterpret_cast<u32x4 *>(wl + (size_t)i * 16) = __builtin_amdgcn_raw_buffer_load_b128(WP, i * 16, S::BASE_CHUNK * 1024, 0);
    } else {
        stage_load(std::integral_constant<int, 0>{});
        stage_store(std::integral_constant<int, 0>{}, wl);
    }
    step_barrier();

    // the group's input rows: lane (h, s) needs rows 16 j + 8 h + i of its sample s (first layer's k order)
    float xn[KS1 * 8];
    auto fetch = [&](int tg) {
        const int tt = A.t0 + tg * SPW + wv;
        const int t = tt < A.t1 ? tt : A.t1 - 1;
        const rsrc_t RX = make_rsrc(AB.X + (size_t)t * D.xrows * 32, D.xrows * 32 * 4);
        const int xvoff = (h * 8 * 32 + s_) * 4, coff = A.crow * 128;
#pragma unroll
        for (int j = 0; j < KS1; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = 16 * j + 8 * h + i;
                xn[j * 8 + i] = bload1(RX, xvoff + (row < D.cw ? coff : 0), (16 * j + i) * 128);
            }
    };
    // one wave per SIMD: the next group's input rows are requested a group ahead (nobody else hides the load); two waves
    // per SIMD: the rows are loaded where they are needed -- 40 registers less, which is what makes the second wave fit
    constexpr bool PREFETCH_X = split_occ(KIND) == 1;
    constexpr int WR = PREFETCH_X ? WRING : 2;                  // (weight ring: one k-step ahead is enough beside a second wave)
    if (PREFETCH_X && (int)blockIdx.x - blk0 < ngroups) fetch((int)blockIdx.x - blk0);
    const int hvoff = tile_voff(lane);

    int rmax = 0;                                           // largest |input| / hidden activation of this wave, as bits (all >= 0)
    // LDS buffer of step st = (st + par) & 1: a net with an odd number of steps per group (the 128-wide nets: 7) starts every
    // other group in buffer 1
    for (int tg = (int)blockIdx.x - blk0, trip = 0; tg < ngroups; tg += nblk, ++trip) {
        const int par = (NS & 1) ? (trip & 1) : 0;
        const int tt = A.t0 + tg * SPW + wv;
        const bool live = tt < A.t1;                       // a wave past the range runs on the last tile, stores nothing
        const int t = live ? tt : A.t1 - 1;
        const bool save = A.save && live;
        // the lane's tile offset, opaque per group: as a loop invariant `hvoff + row offset` was hoisted out of the group loop
        // for all 16 rows (16 registers, spilled to accumulation registers, one v_accvgpr_read per store); inside the loop
        // the constant folds into the store's immediate offset
        int hv = hvoff;
        asm volatile("" : "+v"(hv));
        // planes: first layer's input (from X) | set A | set B; layer 0 writes A, 1 reads A writes B, 2 reads B writes A, 3 reads A
        f16x8 xi1[KS1], xi2[KS1], pa1[2 * HT], pa2[2 * HT], pb1[2 * HT], pb2[2 * HT];
        if constexpr (!PREFETCH_X) fetch(tg);
        float xmax = 0.f;                                  // largest |input| of the tile: its first plane is fp16 too
#pragma unroll
        for (int j = 0; j < KS1; ++j) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = xn[j * 8 + i];
#pragma unroll
            for (int i = 0; i < 8; i += 2) xmax = fmaxf(xmax, fmaxf(fabsf(v[i]), fabsf(v[i + 1])));      // (v_max3_f32 with |.| modifiers)
            split8(v, xi1[j], xi2[j]);
        }
        rmax = max(rmax, __float_as_int(xmax));            // (non-negative floats order like their bits; dead before the planes are live)
        ESR_SPLIT_STAMP(0);
        if constexpr (PREFETCH_X) fetch(tg + nblk < ngroups ? tg + nblk : tg);       // the next group's rows (past the end: this group again, never used)
        // one accumulator per tile (two tiles alternate: the one in flight and the one in its epilogue); bz: a tile's biases,
        // requested when its MFMAs start and used a tile later (a ds_read inside a micro-slice is a full LDS round trip in
        // front of one MFMA's worth of work: the first version of the slices waited ~100 clocks in each); ev: the pending
        // tile's finished values between the phases of its epilogue (vector registers: every touch of an accumulation
        // register costs a v_accvgpr_read / _write of its own)
        f32x16 am[2];
        float4 bz4[2][4];
        float ev[16];
        unsigned mword = 0;
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        // zero-record descriptors drop the stores of a pass that saves nothing (no branch inside the MFMA stream)
        const unsigned hrec = (save && A.save == 1) ? HBYTES : 0u, mrec = save ? MBYTES : 0u;

        // ---- the epilogue of a finished hidden tile, cut into 24 MICRO-SLICES (8 register pairs x 3 phases) ------------------
        // One wave per SIMD has nobody to overlap with, and the wave issues in order: independent vector instructions DO run
        // in the shadow of an MFMA's 32 clocks (tools/ubench/mfma_valu_overlap.hip: a group of one MFMA + V vector
        // instructions costs max(32, 4.75 V) + 4.5 clocks), but a tile's whole epilogue behind its MFMAs idles the pipe for its
        // length (the first version of this kernel: 0.98 ms at C2, 9 k clocks per step for 2.3 k of matrix work).  Each
        // micro-slice (5-8 vector instructions) is therefore issued right behind ONE MFMA of the FOLLOWING tile -- also across
        // a layer boundary: the last tile of layer l is finished inside the first tile of layer l + 1, whose k-steps 10 / 11
        // (the only ones that read that tile's planes) come after micro-slice 23.
        //   phase 0 (pair p): value = accumulator / 64 + bias (one fma), ReLU, the fp32 tile stores -> ev
        //   phase 1: mask bits, first plane (fp16 of the value)
        //   phase 2: second plane (fp16 of value - first plane)
        auto micro = [&](auto LC, auto IT, auto MS, f32x16 &accm, auto &o1, auto &o2) __attribute__((always_inline)) {
            constexpr int l = decltype(LC)::value, it = decltype(IT)::value, ms = decltype(MS)::value, p = ms / 3, q = ms % 3;
            constexpr int r0 = 2 * p, jj = r0 >> 3, i0 = r0 & 7;
            if constexpr (q == 0) {
                const float4 b4 = bz4[it & 1][p >> 1];
                const float bx = (p & 1) ? b4.z : b4.x, by = (p & 1) ? b4.w : b4.y;
                float v0 = fmaf(accm[r0], SPLIT_W_INV, bx), v1 = fmaf(accm[r0 + 1], SPLIT_W_INV, by);
                const int b0 = __float_as_int(v0), b1 = __float_as_int(v1);
                v0 = __int_as_float(b0 > 0 ? b0 : 0);
                v1 = __int_as_float(b1 > 0 ? b1 : 0);
                // range check: inf and +NaN order above every number as integers (volatile: left to itself the chain of maxima
                // sank to the kernel's end and kept every value alive; operands are the integer max's VALU results)
                asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(rmax) : "v"(__float_as_int(v0)), "v"(__float_as_int(v1)));
                const rsrc_t RH = make_rsrc(AB.H[l] + (size_t)t * (HBYTES / 4), hrec);      // fp32 tile, mlp.hip's store_tiles order
#if defined(ESR_SPLIT_H24)
                // timing variant (tools/ubench/split_stamps_h24: what would a 3-byte tile format cost / return in THIS kernel?):
                // the top 24 bits of a row pair's two values as one dword + one short store (wrong layout, right byte count)
                asm volatile("" : "+v"(hv));
                {
                    const unsigned u0 = __float_as_uint(v0), u1 = __float_as_uint(v1);
                    const unsigned hi2 = __builtin_amdgcn_perm(u1, u0, 0x07060302u), mid2 = __builtin_amdgcn_perm(u1, u0, 0x0c0c0501u);
                    bstore1_nt(RH, __uint_as_float(hi2), hv + tile_soff(0, r0), it * 4096);
                    __builtin_amdgcn_raw_buffer_store_b16((unsigned short)mid2, RH, hv + tile_soff(0, r0 + 1), it * 4096, ESR_NT_AUX);
                }
#elif !defined(ESR_SPLIT_NO_HSTORE)
                asm volatile("" : "+v"(hv));                       // (opaque per slice: a shared `hv + row offset` is kept in a
                bstore1_nt(RH, v0, hv + tile_soff(0, r0), it * 4096);       //  register of its own instead of the store's immediate)
                bstore1_nt(RH, v1, hv + tile_soff(0, r0 + 1), it * 4096);
#endif
                ev[r0] = v0; ev[r0 + 1] = v1;
            } else if constexpr (q == 1) {
                const float v0 = ev[r0], v1 = ev[r0 + 1];
                int one0, one1;                                    // (operands: phase 0's integer max -- VALU results, no MFMA hazard;
                asm volatile("v_med3_i32 %1, %3, 0, 1\n\t"         //  one statement: see put_residual_pair)
                             "v_med3_i32 %2, %4, 0, 1\n\t"
                             "v_lshl_or_b32 %0, %1, %5, %0\n\t"
                             "v_lshl_or_b32 %0, %2, %6, %0"
                             : "+v"(mword), "=&v"(one0), "=&v"(one1)
                             : "v"(__float_as_int(v0)), "v"(__float_as_int(v1)), "n"((it & 1) * 16 + r0), "n"((it & 1) * 16 + r0 + 1));
                put_pair<i0>(o1[2 * it + jj], v0, v1);
            } else {
                put_residual_pair<i0>(o2[2 * it + jj], o1[2 * it + jj], ev[r0], ev[r0 + 1]);
            }
        };
        // micro-slices of the pending tile that ride on MFMA slot u of the tile in flight: slice i on slot i % NAVAIL, where
        // NAVAIL = the slots before the tile in flight first READS the pending tile's planes (all of them for a tile of
        // the same layer; all but the last two k-steps when the pending tile is the previous layer's last tile)
        auto pending = [&](auto LC, auto IT, auto U, auto NAVAILC, f32x16 &accm, auto &o1, auto &o2) __attribute__((always_inline)) {
            constexpr int u = decltype(U)::value, navail = decltype(NAVAILC)::value;
            static_assert(navail >= 3 && navail % 3 == 0, "whole register pairs per pass");
            if constexpr (u < navail)
                sfor<0, (24 + navail - 1) / navail>([&](auto KC) {
                    constexpr int msi = u + decltype(KC)::value * navail;
                    if constexpr (msi < 24) micro(LC, IT, std::integral_constant<int, msi>{}, accm, o1, o2);
                });
            // behind the LAST micro-slice of an odd tile: the mask word of the tile pair (mlp_common.h: store_relu_mask's order)
            constexpr int l = decltype(LC)::value, it = decltype(IT)::value;
            if constexpr (u == (navail < 24 ? navail : 24) - 1 && (it & 1)) {
                __builtin_amdgcn_raw_buffer_store_b32(mword, make_rsrc(AB.M[l] + (size_t)t * (MBYTES / 4), mrec), lane * 4,
                                                      (it >> 1) * 256, 0);
                mword = 0;
            }
        };

        // one layer: its steps (pairs of output tiles); `in`: the layer's input planes (= the previous layer's output planes,
        // which the pending tile of that layer is still filling during tile 0), `o`: its output planes
        auto run_layer = [&](auto LC, auto &in1, auto &in2, auto &o1, auto &o2) __attribute__((always_inline)) {
            constexpr int l = decltype(LC)::value, KS = L.ks[l], NT = L.tiles_out[l], NP = L.pairs[l];
            constexpr int s0 = [] { int s = 0; for (int k = 0; k < l; ++k) s += L.pairs[k]; return s; }();
            constexpr bool LAST = l == NL - 1;
            f32x16 zm;                                             // (output layer: its single tile's sums)
            sfor<0, NP>([&](auto PC) {
                constexpr int p = decltype(PC)::value, st = s0 + p, tin = S::tiles_in(st), nxt_st = (st + 1) % NS;
                const unsigned char *wsrc = S::RES ? wl + (S::chunk0(st) - S::BASE_CHUNK) * 1024 : wl + ((st + par) & 1) * S::BUF;
                const u32x4 *mine = reinterpret_cast<const u32x4 *>(wsrc) + lane;
                if constexpr (!S::RES) stage_load(std::integral_constant<int, nxt_st>{});
                // flat k-step index n = tt_ * KS + j; chunk of (tile tt_, plane q, k-step j) = (tt_ * 2 + q) * KS + j
                constexpr int NTOT = tin * KS;
                // weight operands: a ring of three k-steps (requested two k-steps = ~190 clocks ahead)
                // weight operands: a ring of WR k-steps, requested WR - 1 k-steps ahead (the LDS serves four streaming
                // waves at ~91 B/clk: a read waits behind ~24 KB of its neighbours' requests)
                u32x4 wb[WR][2];
                sfor<0, (WR - 1 < NTOT ? WR - 1 : NTOT)>([&](auto NC) {
                    constexpr int n0 = decltype(NC)::value, t0_ = n0 / KS, j0_ = n0 % KS;
                    wb[n0][0] = mine[((t0_ * 2 + 0) * KS + j0_) * 64];
                    wb[n0][1] = mine[((t0_ * 2 + 1) * KS + j0_) * 64];
                });
                sfor<0, NTOT>([&](auto NC) {
                    constexpr int n = decltype(NC)::value, tt_ = n / KS, j = n % KS, it = 2 * p + tt_;
#ifndef ESR_SPLIT_NO_WREAD
                    if constexpr (n + WR - 1 < NTOT) {
                        constexpr int t2 = (n + WR - 1) / KS, j2 = (n + WR - 1) % KS;
                        wb[(n + WR - 1) % WR][0] = mine[((t2 * 2 + 0) * KS + j2) * 64];
                        wb[(n + WR - 1) % WR][1] = mine[((t2 * 2 + 1) * KS + j2) * 64];
                    }
#else
                    if constexpr (n == 0 && NTOT >= WR) { wb[WR - 1][0] = mine[(WR - 1) * 64]; wb[WR - 1][1] = mine[(KS + WR - 1) * 64]; }
#endif
                    if constexpr (j == 0 && !LAST && PREFETCH_X) { // this tile's biases, for its epilogue a tile from now
                        const float4 *bp = reinterpret_cast<const float4 *>(bias_l + l * S::BIAS_FLOATS + it * 32 + (lane >> 5) * 16);
#pragma unroll
                        for (int q = 0; q < 4; ++q) bz4[it & 1][q] = bp[q];
                    }
                    f32x16 &m = LAST ? zm : am[it & 1];
                    const f16x8 w1 = __builtin_bit_cast(f16x8, wb[n % WR][0]), w2 = __builtin_bit_cast(f16x8, wb[n % WR][1]);
                    // the pending tile: the previous tile of this layer, or the last tile of the previous layer
                    constexpr bool HAVE = it > 0 || l > 0;
                    constexpr int pl = it > 0 ? l : l - 1, pit = it > 0 ? it - 1 : (l > 0 ? L.tiles_out[l > 0 ? l - 1 : 0] - 1 : 0);
                    if constexpr (j == 0 && HAVE && !PREFETCH_X) { // two waves per SIMD: the PENDING tile's biases, right where its
                        const float4 *bp = reinterpret_cast<const float4 *>(bias_l + pl * S::BIAS_FLOATS + pit * 32 + (lane >> 5) * 16);
#pragma unroll                                                     // epilogue starts (one live set instead of two: 16 registers)
                        for (int q = 0; q < 4; ++q) bz4[pit & 1][q] = bp[q];
                    }
                    auto ride = [&](auto U) __attribute__((always_inline)) {
                        if constexpr (HAVE) {
                            if constexpr (it > 0) pending(std::integral_constant<int, pl>{}, std::integral_constant<int, pit>{}, U,
                                                          std::integral_constant<int, 3 * KS>{}, am[pit & 1], o1, o2);
                            else pending(std::integral_constant<int, pl>{}, std::integral_constant<int, pit>{}, U,
                                         std::integral_constant<int, 3 * (KS - 2)>{}, am[pit & 1], in1, in2);
                        }
                        if constexpr (tt_ == tin - 1 && !S::RES) { // the next step's weights: one piece per slot, last slots of the step
                            constexpr int u_ = decltype(U)::value, first = 3 * KS - S::PRE;
                            static_assert(first >= 0, "a tile has a slot for every staged piece");
                            if constexpr (u_ >= first) stage_piece(std::integral_constant<int, nxt_st>{}, std::integral_constant<int, u_ - first>{},
                                                                   wl + ((st + 1 + par) & 1) * S::BUF);
                        }
                        __builtin_amdgcn_sched_barrier(0);         // one MFMA + its micro-slice per scheduling region
                    };
#ifdef ESR_SPLIT_NO_MFMA                                           // (timing variants of tools/ubench/split_stamps.hip: wrong results)
                    if (j == 0) m = zero16;
                    m[j & 15] += (float)(w1[0] + in2[j][0]) + (float)(w2[0] + in1[j][0]);
                    ride(std::integral_constant<int, 3 * j + 0>{});
                    ride(std::integral_constant<int, 3 * j + 1>{});
                    ride(std::integral_constant<int, 3 * j + 2>{});
#else
                    // (three dependent MFMAs in a row: behind a micro-slice the predecessor has long finished; in the
                    //  slots without one the dependent issue costs a few clocks -- tools/ubench/mfma_valu_overlap.hip)
                    m = mfma_h(w1, in2[j], j == 0 ? zero16 : m);
                    ride(std::integral_constant<int, 3 * j + 0>{});
                    m = mfma_h(w1, in1[j], m);
                    ride(std::integral_constant<int, 3 * j + 1>{});
                    m = mfma_h(w2, in1[j], m);
                    ride(std::integral_constant<int, 3 * j + 2>{});
#endif
                });
                if constexpr (LAST) {
                    const float4 bz = *reinterpret_cast<const float4 *>(bias_l + l * S::BIAS_FLOATS + (lane >> 5) * 16);
                    const float bzv[4] = {bz.x, bz.y, bz.z, bz.w};
                    const rsrc_t RZ = make_rsrc(A.zout + (size_t)t * D.zrows * 32, live ? D.zrows * 32 * 4 : 0);
                    // rows 4 h + q of the output tile live in registers q = 0..3 of lane half h (acc_row(q, h)); the rows past
                    // out_dim are written as zeros, the rows past the tile (half 1 of a 4-row tile) are dropped by the descriptor
                    const int zvoff = (4 * h * 32 + s_) * 4;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        bstore1(RZ, 4 * h + q < D.out_dim ? fmaf(zm[q], SPLIT_W_INV, bzv[q]) : 0.f, zvoff, q * 128);
                }
                ESR_SPLIT_STAMP(1 + 3 * st);
                ESR_SPLIT_STAMP(2 + 3 * st);
                if constexpr (!S::RES) step_barrier();
                ESR_SPLIT_STAMP(3 + 3 * st);
            });
        };
        run_layer(std::integral_constant<int, 0>{}, xi1, xi2, pa1, pa2);
        if constexpr (NL == 4) {
            run_layer(std::integral_constant<int, 1>{}, pa1, pa2, pb1, pb2);
            run_layer(std::integral_constant<int, 2>{}, pb1, pb2, pa1, pa2);
        }
        run_layer(std::integral_constant<int, NL - 1>{}, pa1, pa2, pb1, pb2);      // output layer (pb: unused)
    }
    // a first plane holds |x| < 65504: an input or a hidden activation at or above SPLIT_RANGE (or inf; a +NaN activation) raises
    // the caller's sticky flag -- the host re-runs the step on the f32 MFMA kernels (fine_engine.py).  The OUTPUT layer's results
    // are fp32 sums that never become planes: nothing to check.  (A NaN input is NaN in both engines' results.)
    if (AB.range && rmax >= __float_as_int(SPLIT_RANGE)) atomicOr(AB.range, 1u);
}

// ---- the input-gradient chain on the same scheme ---------------------------------------------------------------------
// dZ[2] = mask ⊙ (W3ᵀ dz), dZ[1] = mask ⊙ (W2ᵀ dZ[2]), dZ[0] = mask ⊙ (W1ᵀ dZ[1]), dX = W0ᵀ dZ[0] (rows 0 .. 43: the rows that
// lead back to a grid) -- mlp.hip's mlp_dgrad_kernel<0> with the products on the 16-bit matrix cores.  Gradients are small
// (1e-3 .. 1e-9) where fp16's normal range ends at 6e-5, so a tile's chain runs SCALED: s = 2^k (the multiplications by s
// and 1 / s are exact, the ReLU masks do not care, and the chain is linear).  A layer can grow a value by at most the largest
// column sum of its |W|; the running product of those sums, G, comes with the planes (split_gain_kernel), and s puts
// G max |dz| of the tile into [2^14, 2^15): no plane of the chain can reach fp16's ceiling of 65504.  What counts for
// the consumers (weight gradients and grid scatters sum over samples) is the error relative to the tile's LARGEST
// gradients: 2^-22 of them, as in the forward; a sample whose gradient is 2^-20 of its tile's largest loses relative
// precision, as it does in any sum with the large ones.
struct DSplitSeg {
    const _Float16 *planes;    // the net's split planes (forward | transposed)
    int t0, t1;
    int b0, nb;
};
struct DSplitBatch {
    const float *dz;
    const unsigned *M[3];
    float *dZ[3];
    float *dX;
    float *amax;               // optional: max |dz| over the launch's tiles (atomic max of the tiles' own maxima)
    int nseg;
    DSplitSeg seg[2];
};

template <int KIND>
__global__ void __launch_bounds__(64 * SPW, split_occ(KIND)) mlp_dgrad_split_kernel(DSplitBatch AB)
{
    using S = SplitSteps<KIND, true>;
    constexpr NetDesc D = net_desc(KIND);
    constexpr SplitLayout L = S::L;
    constexpr int NL = S::NL, NHID = NL - 1, HT = D.hid_tiles, NS = S::NS;
    constexpr unsigned HBYTES = HT * 32 * 32 * 4, MBYTES = (HT / 2) * 256;
    static_assert((NL == 4 || NL == 2) && HT % 2 == 0 && L.ks[0] == 1 && L.tiles_out[NL - 1] == 2 && D.out_dim <= 8 && D.zrows <= 8,
                  "the four-layer nets and the tone mapper: outputs in one k-step, grid-fed input rows in two tiles");
    DSplitSeg A = AB.seg[0];
    if (AB.nseg > 1 && (int)blockIdx.x >= AB.seg[1].b0) A = AB.seg[1];
    const int blk0 = A.b0, nblk = A.nb;
    extern __shared__ __attribute__((aligned(16))) unsigned char wl[];          // buffer 0 | buffer 1
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, s_ = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntiles = A.t1 - A.t0, ngroups = (ntiles + SPW - 1) / SPW;
    const rsrc_t WP = make_rsrc(A.planes, (unsigned)((S::BASE_CHUNK + L.total_chunks) * 1024));
    u32x4 pre[S::PRE];
    auto stage_load = [&](auto ST) __attribute__((always_inline)) {
        constexpr int st = decltype(ST)::value, pieces = S::chunks(st) * 64, base = S::chunk0(st) * 1024;
#pragma unroll
        for (int k = 0; k < S::PRE; ++k)
            if (k * 64 * SPW < pieces) pre[k] = __builtin_amdgcn_raw_buffer_load_b128(WP, (tid + 64 * SPW * k) * 16, base, 0);
    };
    auto stage_store = [&](auto ST, unsigned char *dst) __attribute__((always_inline)) {
        constexpr int st = decltype(ST)::value, pieces = S::chunks(st) * 64;
#pragma unroll
        for (int k = 0; k < S::PRE; ++k)
            if (k * 64 * SPW < pieces && tid + 64 * SPW * k < pieces)
                *reinterpret_cast<u32x4 *>(dst + (size_t)(tid + 64 * SPW * k) * 16) = pre[k];
    };
    // one 16-byte piece per thread: the step's last tile issues these behind its MFMAs (the other LDS buffer is idle since
    // the previous step's barrier), instead of 12 writes + their wait between the last MFMA and the barrier
    auto stage_piece = [&](auto ST, auto KC, unsigned char *dst) __attribute__((always_inline)) {
        constexpr int st = decltype(ST)::value, k = decltype(KC)::value, pieces = S::chunks(st) * 64;
        if constexpr (k * 64 * SPW < pieces)
            if (tid + 64 * SPW * k < pieces) *reinterpret_cast<u32x4 *>(dst + (size_t)(tid + 64 * SPW * k) * 16) = pre[k];
    };
    if constexpr (S::RES) {
        for (int i = tid; i < L.total_chunks * 64; i += 64 * SPW)
            *reinterpret_cast<u32x4 *>(wl + (size_t)i * 16) = __builtin_amdgcn_raw_buffer_load_b128(WP, i * 16, S::BASE_CHUNK * 1024, 0);
    } else {
        stage_load(std::integral_constant<int, 0>{});
        stage_store(std::integral_constant<int, 0>{}, wl);
    }
    step_barrier();

    // the group's output gradients (rows 0..3 of the 4-row tile: half 0's slots 0..3, everything else of the k-step is zero)
    // and ReLU masks
    float zn[D.zrows];
    unsigned mn[NHID][HT / 2];
    auto fetch = [&](int tg) {
        const int tt = A.t0 + tg * SPW + wv;
        const int t = tt < A.t1 ? tt : A.t1 - 1;
        const rsrc_t RZ = make_rsrc(AB.dz + (size_t)t * D.zrows * 32, D.zrows * 32 * 4);
#pragma unroll
        for (int i = 0; i < D.zrows; ++i) zn[i] = bload1(RZ, s_ * 4, i * 128);
#pragma unroll
        for (int l = 0; l < NHID; ++l)
            load_relu_mask<HT>(make_rsrc(AB.M[l] + (size_t)t * (MBYTES / 4), MBYTES), mn[l], lane);
    };
    if ((int)blockIdx.x - blk0 < ngroups) fetch((int)blockIdx.x - blk0);
    const int hvoff = tile_voff(lane);

    float wmax = 0.f;                                       // largest |dz| of this wave's tiles (AB.amax)
    // the net's gradient gain bound G >= 1 (mlp.hip: split_gain_kernel, behind the planes): no hidden gradient of a tile exceeds
    // G max |dz|.  ge = ceil(log2 G)
    const float *gainp = reinterpret_cast<const float *>(A.planes + (size_t)(S::BASE_CHUNK + L.total_chunks) * 512);
    const int gbits = __builtin_amdgcn_readfirstlane(__float_as_int(*gainp));
    const int kbase = __builtin_amdgcn_readfirstlane(141 + 127 - ((gbits >> 23) & 0xff) - ((gbits & 0x7fffff) ? 1 : 0));   // 141 - ge (scalar)
    // LDS buffer of step st = (st + par) & 1: a net with an odd number of steps per group (the 128-wide nets: 7) starts every
    // other group in buffer 1
    for (int tg = (int)blockIdx.x - blk0, trip = 0; tg < ngroups; tg += nblk, ++trip) {
        const int par = (NS & 1) ? (trip & 1) : 0;
        const int tt = A.t0 + tg * SPW + wv;
        const bool live = tt < A.t1;
        const int t = live ? tt : A.t1 - 1;
        int hv = hvoff;                                                       // (opaque per group: see the forward)
        asm volatile("" : "+v"(hv));
        // the tile's scale: 2^k with G x (the largest |dz| of its 32 samples) in [2^14, 2^15) (exponent arithmetic; an all-zero
        // tile: 1): every plane of the chain stays below fp16's 65504 whatever the masks and signs do
        float zmax = 0.f;
#pragma unroll
        for (int i = 0; i < D.out_dim; ++i) zmax = fmaxf(zmax, fabsf(zn[i]));
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) zmax = fmaxf(zmax, __shfl_xor(zmax, o));
        if (live) wmax = fmaxf(wmax, zmax);                                   // (the launch's maximum: one atomic per wave, at the end)
        const int ez = (__float_as_int(zmax) >> 23) & 0xff;                   // biased exponent of the maximum
        const int ks = ez == 0 ? 0 : kbase - ez;                              // scale exponent: G max lands in [2^14, 2^15)
        const int kc = ks < -100 ? -100 : (ks > 100 ? 100 : ks);
        const float sc = __int_as_float((127 + kc) << 23), isc = __int_as_float((127 - kc) << 23);
        f16x8 xi1[1], xi2[1];
        {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = (h == 0 && i < D.out_dim) ? zn[i < D.out_dim ? i : 0] * sc : 0.f;
            split8(v, xi1[0], xi2[0]);
        }
        unsigned msk[NHID][HT / 2];
#pragma unroll
        for (int l = 0; l < NHID; ++l)
#pragma unroll
            for (int w = 0; w < HT / 2; ++w) msk[l][w] = mn[l][w];
        fetch(tg + nblk < ngroups ? tg + nblk : tg);
        f16x8 pa1[2 * HT], pa2[2 * HT], pb1[2 * HT], pb2[2 * HT];
        f32x16 am[2];
        float ev[16];
        const float wisc = SPLIT_W_INV * isc;                                 // accumulator (64 x the scaled gradient) -> the fp32 store
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

        // micro-slices of a finished tile of transposed layer q (8 register pairs x 3 phases, as in the forward):
        //   q < 3: phase 0 value (scaled), ReLU mask of the layer below, the unscaled fp32 dZ store; phases 1 / 2 the planes
        //   q = 3: phase 0 unscaled value -> dX rows (the descriptor ends at row 44: the rows above are not written)
        auto micro = [&](auto QC, auto IT, auto MS, f32x16 &accm, auto &o1, auto &o2) __attribute__((always_inline)) {
            constexpr int q = decltype(QC)::value, it = decltype(IT)::value, ms = decltype(MS)::value, p = ms / 3, ph = ms % 3;
            constexpr int r0 = 2 * p, jj = r0 >> 3, i0 = r0 & 7;
            if constexpr (q == NL - 1) {
                if constexpr (ph == 0) {
                    const float v0 = accm[r0] * wisc, v1 = accm[r0 + 1] * wisc;
                    const rsrc_t RX = make_rsrc(AB.dX + (size_t)t * 64 * 32, live ? dx_rows(KIND) / 4 * 4 * 128 + (dx_rows(KIND) % 4 ? 512 : 0) : 0);
                    asm volatile("" : "+v"(hv));
                    bstore1(RX, v0, hv + tile_soff(0, r0), it * 4096);           // (default policy: the scatter reads dX next)
                    bstore1(RX, v1, hv + tile_soff(0, r0 + 1), it * 4096);
                }
            } else {
                constexpr int d = NHID - 1 - q;                              // this tile is a tile of dZ[d]
                if constexpr (ph == 0) {
                    // mask bit -> 0 / ~0 with one v_bfe_i32 (in C, a shift pair or the bfe builtin became and + compare + select
                    // through vcc, with the wait states that go with vcc)
                    // (the operand is the mask word, loaded from memory a tile group ago: no MFMA result near this asm)
                    int k0, k1;
                    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(k0) : "v"(msk[d][it >> 1]), "n"((it & 1) * 16 + r0));
                    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(k1) : "v"(msk[d][it >> 1]), "n"((it & 1) * 16 + r0 + 1));
                    const int a0 = __float_as_int(accm[r0]) & k0, a1 = __float_as_int(accm[r0 + 1]) & k1;      // 64 x the masked value
                    const rsrc_t RD = make_rsrc(AB.dZ[d] + (size_t)t * (HBYTES / 4), (live && AB.dZ[d]) ? HBYTES : 0u);
                    asm volatile("" : "+v"(hv));
                    bstore1_nt(RD, __int_as_float(a0) * wisc, hv + tile_soff(0, r0), it * 4096);
                    bstore1_nt(RD, __int_as_float(a1) * wisc, hv + tile_soff(0, r0 + 1), it * 4096);
                    ev[r0] = __int_as_float(a0) * SPLIT_W_INV; ev[r0 + 1] = __int_as_float(a1) * SPLIT_W_INV;
                } else if constexpr (ph == 1) {
                    put_pair<i0>(o1[2 * it + jj], ev[r0], ev[r0 + 1]);
                } else {
                    put_residual_pair<i0>(o2[2 * it + jj], o1[2 * it + jj], ev[r0], ev[r0 + 1]);
                }
            }
        };
        // the pending tile's micro-slices u, u + navail, u + 2 navail, ... ride on MFMA slot u of the tile in flight (navail: as
        // in the forward)
        auto pending = [&](auto QC, auto IT, auto U, auto NAVAILC, f32x16 &accm, auto &o1, auto &o2) __attribute__((always_inline)) {
            constexpr int u = decltype(U)::value, navail = decltype(NAVAILC)::value;
            static_assert(navail >= 3 && navail % 3 == 0, "whole register pairs per pass");
            if constexpr (u < navail)
                sfor<0, (24 + navail - 1) / navail>([&](auto KC) {
                    constexpr int msi = u + decltype(KC)::value * navail;
                    if constexpr (msi < 24) micro(QC, IT, std::integral_constant<int, msi>{}, accm, o1, o2);
                });
        };
        auto run_layer = [&](auto QC, auto &in1, auto &in2, auto &o1, auto &o2) __attribute__((always_inline)) {
            constexpr int q = decltype(QC)::value, KS = L.ks[q], NT = L.tiles_out[q], NP = L.pairs[q];
            constexpr int s0 = [] { int s = 0; for (int k = 0; k < q; ++k) s += L.pairs[k]; return s; }();
            sfor<0, NP>([&](auto PC) {
                constexpr int p = decltype(PC)::value, st = s0 + p, tin = S::tiles_in(st), nxt_st = (st + 1) % NS;
                const unsigned char *wsrc = S::RES ? wl + (S::chunk0(st) - S::BASE_CHUNK) * 1024 : wl + ((st + par) & 1) * S::BUF;
                const u32x4 *mine = reinterpret_cast<const u32x4 *>(wsrc) + lane;
                if constexpr (!S::RES) stage_load(std::integral_constant<int, nxt_st>{});
                constexpr int NTOT = tin * KS;
                u32x4 wb[WRING][2];
                sfor<0, (WRING - 1 < NTOT ? WRING - 1 : NTOT)>([&](auto NC) {
                    constexpr int n0 = decltype(NC)::value, t0_ = n0 / KS, j0_ = n0 % KS;
                    wb[n0][0] = mine[((t0_ * 2 + 0) * KS + j0_) * 64];
                    wb[n0][1] = mine[((t0_ * 2 + 1) * KS + j0_) * 64];
                });
                sfor<0, NTOT>([&](auto NC) {
                    constexpr int n = decltype(NC)::value, tt_ = n / KS, j = n % KS, it = 2 * p + tt_;
                    if constexpr (n + WRING - 1 < NTOT) {
                        constexpr int t2 = (n + WRING - 1) / KS, j2 = (n + WRING - 1) % KS;
                        wb[(n + WRING - 1) % WRING][0] = mine[((t2 * 2 + 0) * KS + j2) * 64];
                        wb[(n + WRING - 1) % WRING][1] = mine[((t2 * 2 + 1) * KS + j2) * 64];
                    }
                    f32x16 &m = am[it & 1];
                    const f16x8 w1 = __builtin_bit_cast(f16x8, wb[n % WRING][0]), w2 = __builtin_bit_cast(f16x8, wb[n % WRING][1]);
                    constexpr bool HAVE = it > 0 || q > 0;
                    constexpr int pq = it > 0 ? q : q - 1, pit = it > 0 ? it - 1 : (q > 0 ? L.tiles_out[q > 0 ? q - 1 : 0] - 1 : 0);
                    auto ride = [&](auto U) __attribute__((always_inline)) {
                        if constexpr (HAVE) {
                            if constexpr (it > 0) pending(std::integral_constant<int, pq>{}, std::integral_constant<int, pit>{}, U,
                                                          std::integral_constant<int, 3 * KS>{}, am[pit & 1], o1, o2);
                            else pending(std::integral_constant<int, pq>{}, std::integral_constant<int, pit>{}, U,
                                         std::integral_constant<int, 3 * (KS - 2)>{}, am[pit & 1], in1, in2);
                        }
                        if constexpr (tt_ == tin - 1 && 3 * KS >= S::PRE && !S::RES) {
                            constexpr int u_ = decltype(U)::value, first = 3 * KS - S::PRE;
                            if constexpr (u_ >= first) stage_piece(std::integral_constant<int, nxt_st>{}, std::integral_constant<int, u_ - first>{},
                                                                   wl + ((st + 1 + par) & 1) * S::BUF);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    };
                    m = mfma_h(w1, in2[j], j == 0 ? zero16 : m);
                    ride(std::integral_constant<int, 3 * j + 0>{});
                    m = mfma_h(w1, in1[j], m);
                    ride(std::integral_constant<int, 3 * j + 1>{});
                    m = mfma_h(w2, in1[j], m);
                    ride(std::integral_constant<int, 3 * j + 2>{});
                });
                if constexpr (q == NL - 1 && p == NP - 1) {         // the very last tile (dX rows 32..63): nobody to ride on
                    sfor<0, 24>([&](auto MC) {
                        micro(QC, std::integral_constant<int, NT - 1>{}, MC, am[(NT - 1) & 1], o1, o2);
                    });
                }
                if constexpr (3 * KS < S::PRE && !S::RES)          // (the one-k-step first layer: too few slots, all pieces here)
                    stage_store(std::integral_constant<int, nxt_st>{}, wl + ((st + 1 + par) & 1) * S::BUF);
                if constexpr (!S::RES) step_barrier();
            });
        };
        run_layer(std::integral_constant<int, 0>{}, xi1, xi2, pa1, pa2);      // W3ᵀ dz -> dZ[2]   (tone mapper: W1ᵀ dz -> dZ[0])
        if constexpr (NL == 4) {
            run_layer(std::integral_constant<int, 1>{}, pa1, pa2, pb1, pb2);  // -> dZ[1]
            run_layer(std::integral_constant<int, 2>{}, pb1, pb2, pa1, pa2);  // -> dZ[0]
        }
        run_layer(std::integral_constant<int, NL - 1>{}, pa1, pa2, pb1, pb2);      // -> dX (pb unused)
    }
    // max |dz| of the launch: one atomic per wave, and only from a wave that would raise the value (non-negative floats order
    // like their bit patterns).  One atomic per TILE -- 16 384 on one address at C2 -- took 0.14 ms to drain: twice the tone
    // mapper's whole launch.
    // (what the weight-gradient kernels scale by: max |dz| x max(1, G / 16) -- their headroom above the scale source is >= 32x, so
    //  G max |dz|, the bound of every hidden gradient, fits their planes as well)
    wmax *= fmaxf(1.f, *gainp * 0.0625f);
    if (AB.amax && lane == 0 && wmax > *reinterpret_cast<volatile float *>(AB.amax))
        atomicMax(reinterpret_cast<unsigned *>(AB.amax), __float_as_uint(wmax));
}

// workgroups per segment proportional to its tile groups (every non-empty segment >= 1); returns the grid
int share_blocks_split(SplitSeg *seg, int nseg, int cap = 256)
{
    int groups[MAX_SPLIT_SEG], total = 0;
    for (int k = 0; k < nseg; ++k) { groups[k] = (seg[k].t1 - seg[k].t0 + SPW - 1) / SPW; total += groups[k]; }
    const int grid = total < cap ? total : cap;
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

// per-device sticky flag registered by the caller (esr_mlp_split_range_flag); NULL: no check
std::atomic<unsigned *> g_range_flag[16];

template <int KIND>
int launch_split_k(SplitBatch &B, hipStream_t s)
{
    using S = SplitSteps<KIND>;
    int dev = 0;
    B.range = (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 16) ? g_range_flag[dev].load() : nullptr;
    const int grid = share_blocks_split(B.seg, B.nseg, 256 * split_occ(KIND));       // resident workgroups: one or two per CU
    static std::atomic<uint64_t> optin{0};
    if (int rc = esr_lds_optin(reinterpret_cast<const void *>(&mlp_fwd_split_kernel<KIND>), S::LDS_BYTES, optin)) return rc;
    mlp_fwd_split_kernel<KIND><<<grid, 64 * SPW, S::LDS_BYTES, s>>>(B);
    ESR_CHECK_LAUNCH();
    return 0;
}
int launch_split(int kind, SplitBatch &B, hipStream_t s)
{
    switch (kind) {
    case ESR_MLP_RADIANCE: return launch_split_k<ESR_MLP_RADIANCE>(B, s);
    case ESR_MLP_TONEMAP:  return launch_split_k<ESR_MLP_TONEMAP>(B, s);
    case ESR_MLP_BRDF:     return launch_split_k<ESR_MLP_BRDF>(B, s);
    case ESR_MLP_EMIT:     return launch_split_k<ESR_MLP_EMIT>(B, s);
    default:               return ESR_EINVAL;
    }
}
bool split_kind_ok(int kind)
{
    return kind == ESR_MLP_RADIANCE || kind == ESR_MLP_TONEMAP || kind == ESR_MLP_BRDF || kind == ESR_MLP_EMIT;
}

bool crow_ok_split(int kind, int crow) { return crow == 0 || (kind != ESR_MLP_TONEMAP && (crow == 88 || crow == 96)); }

}  // namespace

// One radiance forward pass over tiles [t0, t1) (esr_mlp_fwd's contract: save 0 / 1 / 2, colour group color_row0), products
// on the 16-bit matrix cores from split fp16 planes.  packed32: esr_mlp_pack's buffer (biases); planes: the net's split
// planes (esr_mlp_pack_batch, esr_mlp_packed_split_elems values).  Radiance, tone mapper, BRDF and emission nets.
ESR_API int esr_mlp_fwd_split(int kind, const float *packed32, const void *planes, const float *X, int32_t t0, int32_t t1,
                              float *const *H, uint32_t *const *M, int save, int color_row0, float *zout, void *stream)
{
    if (!split_kind_ok(kind) || t0 < 0 || t1 < t0 || !crow_ok_split(kind, color_row0)) return ESR_EINVAL;
    if (t1 == t0) return 0;
    if (!packed32 || !planes || !X || !zout) return ESR_EINVAL;
    SplitBatch B = {};
    B.X = X;
    if (save) {
        if (!M || (save != 2 && !H)) return ESR_EINVAL;
        for (int l = 0; l < net_desc(kind).n_layers - 1; ++l) {
            if (!M[l] || (save != 2 && !H[l])) return ESR_EINVAL;
            B.H[l] = save != 2 ? H[l] : nullptr; B.M[l] = M[l];
        }
    }
    B.nseg = 1;
    B.seg[0] = SplitSeg{packed32, static_cast<const _Float16 *>(planes), t0, t1, save == 2 ? 2 : save ? 1 : 0, color_row0, zout, 0, 0};
    return launch_split(kind, B, esr_stream(stream));
}

unsigned *esr_split_range_flag_ptr()
{
    int dev = 0;
    return (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 16) ? g_range_flag[dev].load() : nullptr;
}

// Registers (NULL: removes) the CURRENT device's range flag: a device uint32 that every later split forward launch on this
// device ORs with 1 when a hidden activation reaches 60000 (or is inf / NaN): the products' first plane is fp16.
ESR_API int esr_mlp_split_range_flag(uint32_t *flag)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return ESR_EINVAL;
    g_range_flag[dev].store(flag);
    return 0;
}

// The fine stage's three radiance forward passes of a step as ONE launch (esr_mlp_fwd_fine's contract and argument meaning).
ESR_API int esr_mlp_fwd_fine_split(const float *packed32_off, const void *planes_off, const float *packed32_emo,
                                   const void *planes_emo, const float *X, int32_t t_on, int32_t t_all, float *const *H,
                                   uint32_t *const *M, int color_row_detached, float *z_off, float *z_emo, void *stream)
{
    if (t_on < 0 || t_all < t_on || !crow_ok_split(ESR_MLP_RADIANCE, color_row_detached)) return ESR_EINVAL;
    if (t_all == 0) return 0;
    if (!packed32_off || !planes_off || !packed32_emo || !planes_emo || !X || !H || !M || !z_off || !z_emo) return ESR_EINVAL;
    SplitBatch B = {};
    B.X = X;
    for (int l = 0; l < 3; ++l) {
        if (!H[l] || !M[l]) return ESR_EINVAL;
        B.H[l] = H[l]; B.M[l] = M[l];
    }
    const _Float16 *po = static_cast<const _Float16 *>(planes_off), *pe = static_cast<const _Float16 *>(planes_emo);
    int n = 0;
    if (t_on > 0) B.seg[n++] = SplitSeg{packed32_off, po, 0, t_on, 0, color_row_detached, z_off, 0, 0};
    if (t_all > t_on) B.seg[n++] = SplitSeg{packed32_off, po, t_on, t_all, 1, 0, z_off, 0, 0};
    if (t_on > 0) B.seg[n++] = SplitSeg{packed32_emo, pe, 0, t_on, 1, 0, z_emo, 0, 0};
    B.nseg = n;
    return launch_split(ESR_MLP_RADIANCE, B, esr_stream(stream));
}

namespace {
template <int KIND>
int launch_dsplit_k(DSplitBatch &B, hipStream_t s)
{
    using S = SplitSteps<KIND, true>;
    int groups[2], total = 0;
    for (int k = 0; k < B.nseg; ++k) { groups[k] = (B.seg[k].t1 - B.seg[k].t0 + SPW - 1) / SPW; total += groups[k]; }
    const int cap = 256 * split_occ(KIND);
    const int grid = total < cap ? total : cap;
    if (B.nseg == 1) { B.seg[0].b0 = 0; B.seg[0].nb = grid; }
    else {
        int n0 = (int)((int64_t)grid * groups[0] / total);
        if (n0 < 1) n0 = 1;
        if (n0 > groups[0]) n0 = groups[0];
        if (grid - n0 > groups[1]) n0 = grid - groups[1];
        if (grid - n0 < 1) n0 = grid - 1;
        B.seg[0].b0 = 0; B.seg[0].nb = n0; B.seg[1].b0 = n0; B.seg[1].nb = grid - n0;
    }
    static std::atomic<uint64_t> optin{0};
    if (int rc = esr_lds_optin(reinterpret_cast<const void *>(&mlp_dgrad_split_kernel<KIND>), S::WBYTES, optin)) return rc;
    mlp_dgrad_split_kernel<KIND><<<grid, 64 * SPW, S::WBYTES, s>>>(B);
    ESR_CHECK_LAUNCH();
    return 0;
}
int launch_dsplit(int kind, DSplitBatch &B, hipStream_t s)
{
    switch (kind) {
    case ESR_MLP_RADIANCE: return launch_dsplit_k<ESR_MLP_RADIANCE>(B, s);
    case ESR_MLP_TONEMAP:  return launch_dsplit_k<ESR_MLP_TONEMAP>(B, s);
    case ESR_MLP_BRDF:     return launch_dsplit_k<ESR_MLP_BRDF>(B, s);
    case ESR_MLP_EMIT:     return launch_dsplit_k<ESR_MLP_EMIT>(B, s);
    default:               return ESR_EINVAL;
    }
}
}  // namespace

// esr_mlp_dgrad's contract (radiance, tone mapper, BRDF and emission nets): input / hidden gradients over tiles [t0, t1) from the net's split planes.
ESR_API int esr_mlp_dgrad_split(int kind, const void *planes, const float *dz, int32_t t0, int32_t t1, const uint32_t *const *M,
                                float *const *dZ, float *dX, float *amax, void *stream)
{
    if (!split_kind_ok(kind) || t0 < 0 || t1 < t0) return ESR_EINVAL;
    if (t1 == t0) return 0;
    if (!planes || !dz || !M || !dZ || !dX) return ESR_EINVAL;
    DSplitBatch B = {};
    B.dz = dz; B.dX = dX; B.amax = amax;
    for (int l = 0; l < net_desc(kind).n_layers - 1; ++l) {
        if (!M[l]) return ESR_EINVAL;
        B.M[l] = M[l]; B.dZ[l] = dZ[l];                     // a NULL dZ[l] is computed but not stored
    }
    B.nseg = 1;
    B.seg[0] = DSplitSeg{static_cast<const _Float16 *>(planes), t0, t1, 0, 0};
    return launch_dsplit(kind, B, esr_stream(stream));
}

// esr_mlp_dgrad_fine's contract: the emissive net on tiles [0, t_on), the non-emissive net on [t_on, t_all), one launch.
ESR_API int esr_mlp_dgrad_fine_split(const void *planes_emo, const void *planes_off, const float *dz, int32_t t_on, int32_t t_all,
                                     const uint32_t *const *M, float *const *dZ, float *dX, float *amax, void *stream)
{
    if (t_on < 0 || t_all < t_on) return ESR_EINVAL;
    if (t_all == 0) return 0;
    if (!planes_emo || !planes_off || !dz || !M || !dZ || !dX) return ESR_EINVAL;
    DSplitBatch B = {};
    B.dz = dz; B.dX = dX; B.amax = amax;
    for (int l = 0; l < 3; ++l) {
        if (!M[l]) return ESR_EINVAL;
        B.M[l] = M[l]; B.dZ[l] = dZ[l];
    }
    int n = 0;
    if (t_on > 0) B.seg[n++] = DSplitSeg{static_cast<const _Float16 *>(planes_emo), 0, t_on, 0, 0};
    if (t_all > t_on) B.seg[n++] = DSplitSeg{static_cast<const _Float16 *>(planes_off), t_on, t_all, 0, 0};
    B.nseg = n;
    return launch_dsplit(ESR_MLP_RADIANCE, B, esr_stream(stream));
}
