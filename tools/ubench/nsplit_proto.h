// The 192-wide radiance net on the split-fp16 scheme (mlp_split.hip) at TWO WAVES PER SIMD, output tiles split (round 6).
//
// mlp_fwd_split_kernel<0> / mlp_dgrad_split_kernel<0> keep a layer's input AND output planes of a 32-sample tile in one
// wave's registers (~400): ONE wave per SIMD, and an in-order wave cannot issue its ~4600 vector / LDS / store instructions
// per tile in the shadow of its 576 MFMAs (one instruction per ~4.8 clocks: tools/ubench/issue_cost.hip) -- 44.6 k clocks per
// tile for 18.4 k of matrix work.  Here a PAIR of waves (w = 0, 1; waves wv and wv + 4 of an eight-wave workgroup) shares a
// tile and splits every layer's OUTPUT tiles: wave w owns output tiles it = 2 i + w.  Both waves hold the layer's whole input
// (two fp16 planes, 96 registers) and nothing else of the layer: the planes a wave's epilogue makes go to LDS (24 KB per pair,
// in place: a tile's slot is rewritten by the next layer's tile of the same index after both waves have fetched it), where
// both waves fetch the next layer's input from.  No partial sums cross waves (round 5's K-split pair kernels did: their
// exchange and its barrier ate what the second wave bought), a wave's weight reads per MFMA are those of the one-wave kernel,
// and at ~220 registers two waves fit a SIMD: one's epilogue runs beside the other's MFMAs.
//
// Weights: mlp_common.h's N-split order, UNITS of <= 24 KB through a double-buffered LDS window, one barrier per unit
// (17 units per tile group in the forward: 3 + 6 + 6 + 2).  A 12-k-step layer takes two units per tile (K halves): the
// first half reads only the planes of the previous layer's tiles 0..2, so the epilogue of that layer's LAST tiles (4, 5) rides
// on the first unit's MFMAs and is published by its barrier, in time for the second half.  The output layer (one tile) is
// wave 0's; wave 1 finishes its last epilogue beside it.
// Results: the same products in the same order as the one-wave kernels (k-steps 0..KS-1 into one accumulator): bit-identical
// H / masks / outputs; buffers and formats unchanged.
// PROTOTYPE (round 6), measured and NOT taken into the product: tools/ubench/nsplit_bench.hip runs it beside the product's
// one-wave kernels (bit-identical results) -- see that file's header and DESIGN.md section 4 for the numbers and why two waves per SIMD
// do not pay (the launch is bound by instruction issue per SIMD and by the power the chip may draw, not by exposed latency).
#pragma once
namespace {
// ---- split-fp16 planes in N-SPLIT order (this file: the 192-wide radiance net at two waves per SIMD) ------------------
// Two waves (w = 0, 1) share one 32-sample tile and split every layer's OUTPUT tiles: wave w multiplies output tiles
// it = 2 i + w (i = 0, 1, 2 of a 192-wide layer) over the layer's WHOLE K from planes it holds in registers; the planes a
// wave's epilogue makes go to LDS, where both waves of the pair fetch the next layer's input from.  The weights stream
// through LDS in UNITS of at most 24 chunks (24 KB, double-buffered), one barrier per unit:
//   a layer of KS <= 6 k-steps (the first layer: the 96 input rows): one unit per tile pair i, chunks [wave][plane][k-step];
//   a layer of KS = 12 k-steps: two units per tile pair, K halves hf = 0 / 1 (k-steps 6 hf .. 6 hf + 5) -- a layer's first
//     unit then needs only the planes of the previous layer's tiles 0..2, and those of tiles 4, 5 (whose epilogue rides on
//     that unit's MFMAs) one barrier later;
//   a layer of ONE output tile (the forward's output layer): wave 0 alone, one unit per K half, chunks [plane][k-step];
//   a layer of ONE k-step (the input-gradient chain's first: K = the output rows): a single unit, every wave multiplies its
//     three tiles, chunks [wave][tile of the wave][plane].
struct NsLayout {
    int n_layers, n_units;
    int ks[4], tiles[4];
    int u_layer[20], u_i[20], u_half[20], u_kb[20], u_ku[20], u_multi[20], u_waves[20], u_chunk0[20], u_chunks[20];
    int total_chunks, max_chunks;
};
__host__ __device__ constexpr NsLayout ns_layout_from(const SplitLayout S)
{
    NsLayout L = {};
    L.n_layers = S.n_layers;
    int u = 0, o = 0, mx = 0;
    for (int l = 0; l < S.n_layers; ++l) {
        const int KS = S.ks[l], NT = S.tiles_out[l];
        L.ks[l] = KS; L.tiles[l] = NT;
        const int multi = KS == 1 ? 1 : 0, waves = NT >= 2 ? 2 : 1;
        const int NP = multi ? 1 : (NT >= 2 ? NT / 2 : 1), NH = (!multi && KS > 6) ? 2 : 1, KU = KS / NH;
        for (int i = 0; i < NP; ++i)
            for (int hf = 0; hf < NH; ++hf, ++u) {
                L.u_layer[u] = l; L.u_i[u] = i; L.u_half[u] = hf; L.u_kb[u] = hf * KU; L.u_ku[u] = KU;
                L.u_multi[u] = multi; L.u_waves[u] = waves; L.u_chunk0[u] = o;
                L.u_chunks[u] = multi ? NT * 2 : waves * 2 * KU;
                o += L.u_chunks[u];
                mx = L.u_chunks[u] > mx ? L.u_chunks[u] : mx;
            }
    }
    L.n_units = u; L.total_chunks = o; L.max_chunks = mx;
    return L;
}
__host__ __device__ constexpr NsLayout ns_layout(int kind) { return ns_layout_from(split_layout(kind)); }
__host__ __device__ constexpr NsLayout ns_layout_t(int kind) { return ns_layout_from(split_layout_t(kind)); }
__host__ __device__ constexpr bool ns_kind(int kind) { return kind == ESR_MLP_RADIANCE; }
// the prototype's planes buffer: N-split forward planes | N-split transposed planes | the net's gain bound (one fp32, copied
// from the product's planes buffer)
__host__ __device__ constexpr int64_t ns_elems(int kind)
{
    return ns_kind(kind) ? ((int64_t)ns_layout(kind).total_chunks + ns_layout_t(kind).total_chunks) * 512 : 0;
}
// one element of the N-split-ordered planes; BWD: the transposed weights (packst_body's feature maps)
template <int KIND, bool BWD>
__device__ __forceinline__ void packn_body(const PackArgs &A, int64_t e)
{
    constexpr NsLayout L = BWD ? ns_layout_t(KIND) : ns_layout(KIND);
    constexpr SplitLayout S = BWD ? split_layout_t(KIND) : split_layout(KIND);
    constexpr int NL = L.n_layers;
    const int chunk = (int)(e >> 9), slot = (int)(e & 7), lane = (int)((e >> 3) & 63), h = lane >> 5;
    int u = 0;
#pragma unroll
    for (int k = 1; k < L.n_units; ++k)
        if (chunk >= L.u_chunk0[k]) u = k;
    const int l = L.u_layer[u], KU = L.u_ku[u];
    const int c = chunk - L.u_chunk0[u];
    int it, plane, j;
    if (L.u_multi[u]) {                                   // [wave][tile of the wave][plane], the single k-step
        const int ntw = L.tiles[l] / 2, w = c / (2 * ntw), r = c % (2 * ntw);
        it = 2 * (r >> 1) + w; plane = r & 1; j = 0;
    } else {                                             // [wave][plane][k-step of the unit]
        const int w = c / (2 * KU), r = c % (2 * KU);
        it = L.u_waves[u] == 2 ? 2 * L.u_i[u] + w : 0;
        plane = r / KU; j = L.u_kb[u] + r % KU;
    }
    float v = 0.f;
    if (!BWD) {
        const int row = 32 * it + (lane & 31);
        const int col = l == 0 ? in_colmap(KIND, 16 * j + 8 * h + slot) : kfeat16(j, h, slot);
        if (row < S.out_dim[l] && col >= 0 && col < S.in_dim[l]) v = A.w[l][(int64_t)row * S.in_dim[l] + col];
    } else {
        const int nl = NL - 1 - l;                        // network layer of transposed layer l
        const bool first = nl == 0, last = nl == NL - 1;
        const int orow = last ? (8 * h + slot) : kfeat16(j, h, slot);
        const int irow = 32 * it + (lane & 31);
        const int col = first ? in_colmap(KIND, irow) : irow;
        if (orow < S.out_dim[l] && col >= 0 && col < S.in_dim[l]) v = A.w[nl][(int64_t)orow * S.in_dim[l] + col];
    }
    v *= SPLIT_W_SCALE;
    const _Float16 w1 = (_Float16)v;
    const int64_t base = BWD ? (int64_t)ns_layout(KIND).total_chunks * 512 : 0;
    A.outs[base + e] = plane == 0 ? w1 : (_Float16)(v - (float)w1);
}


// the planes of one net in N-split order (forward | transposed), from the reference-layout tensors
struct NsPackArgs { const float *w[4]; _Float16 *outs; };
template <int KIND, bool BWD>
__device__ __forceinline__ void packn_proto_body(const NsPackArgs &P, int64_t e)
{
    PackArgs A = {};
    for (int l = 0; l < 4; ++l) A.w[l] = P.w[l];
    A.outs = P.outs;
    packn_body<KIND, BWD>(A, e);
}
template <int KIND>
__global__ void __launch_bounds__(256) ns_pack_kernel(NsPackArgs P)
{
    constexpr int64_t NF = (int64_t)ns_layout(KIND).total_chunks * 512, NB = (int64_t)ns_layout_t(KIND).total_chunks * 512;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < NF + NB; e += (int64_t)gridDim.x * blockDim.x) {
        if (e < NF) packn_proto_body<KIND, false>(P, e);
        else packn_proto_body<KIND, true>(P, e - NF);
    }
}


// in-kernel time stamps for tools/ubench/nsplit_bench.hip (nothing in the product build)
#ifndef ESR_NS_STAMP
#define ESR_NS_STAMP(u, i)
#endif
#ifndef ESR_NS_STAMP2
#define ESR_NS_STAMP2(u, i)
#endif

constexpr int NW = 8;                                       // waves per workgroup: four pairs = four tiles per group
#ifndef ESR_NS_WRING
#define ESR_NS_WRING 3
#endif
constexpr int NS_WR = ESR_NS_WRING;                         // k-steps of weight operands in flight per wave

template <int KIND, bool BWD> struct NsSteps {
    static constexpr NsLayout L = BWD ? ns_layout_t(KIND) : ns_layout(KIND);
    static constexpr int NL = L.n_layers, NU = L.n_units;
    static constexpr int BUF = L.max_chunks * 1024;
    static constexpr int PRE = (L.max_chunks * 64 + 64 * NW - 1) / (64 * NW);        // 16-byte pieces per thread and unit
    static constexpr int KSH = 2 * MAX_HID_TILES;                                     // k-steps of a hidden layer's planes
    static constexpr int PLANES0 = 2 * BUF;                                           // [pair][k-step][plane][64 lanes][16 B]
    static constexpr int PLANE_PAIR = KSH * 2 * 1024;
    static constexpr int BIAS0 = PLANES0 + (NW / 2) * PLANE_PAIR;
    static constexpr int BIAS_FLOATS = 32 * MAX_HID_TILES;
    static constexpr int LDS_BYTES = BIAS0 + (BWD ? 0 : NL * BIAS_FLOATS * 4);
    // element offset of the N-split planes inside the net's planes buffer
    static constexpr int64_t BASE = BWD ? (int64_t)ns_layout(KIND).total_chunks * 512 : 0;
    // running tile index (per wave: its tiles of all layers in order).  A unit starts tiles_started(u) tiles: one (K half 0),
    // none (K half 1), or the wave's three tiles of a one-k-step layer
    static constexpr int tiles_started(int u) { return L.u_multi[u] ? L.tiles[L.u_layer[u]] / 2 : (L.u_half[u] == 0 ? 1 : 0); }
    static constexpr int tile0_of(int u)                   // first (or only) running tile the unit works on
    {
        int c = 0;
        for (int k = 0; k < u; ++k) c += tiles_started(k);
        return tiles_started(u) ? c : c - 1;
    }
    static constexpr int unit_of(int c)                    // the unit that starts running tile c
    {
        for (int k = 0; k < NU; ++k)
            if (tiles_started(k) && c >= tile0_of(k) && c < tile0_of(k) + tiles_started(k)) return k;
        return 0;
    }
    static constexpr int layer_of(int c) { return L.u_layer[unit_of(c)]; }
    static constexpr int i_of(int c) { return L.u_multi[unit_of(c)] ? c - tile0_of(unit_of(c)) : L.u_i[unit_of(c)]; }
};

template <int KIND>
__global__ void __launch_bounds__(64 * NW, 2) mlp_fwd_ns_kernel(SplitBatch AB)
{
    using S = NsSteps<KIND, false>;
    constexpr NetDesc D = net_desc(KIND);
    constexpr NsLayout L = S::L;
    constexpr PackLayout L32 = pack_layout(KIND);
    constexpr int NL = S::NL, HT = D.hid_tiles, NU = S::NU, KS1 = L.ks[0];
    constexpr unsigned HBYTES = HT * 32 * 32 * 4, MBYTES = (HT / 2) * 256;
    static_assert(NL == 4 && HT == 6 && L.tiles[NL - 1] == 1 && L.ks[0] <= 6 && L.ks[1] == 2 * HT && L.ks[NL - 1] == 2 * HT &&
                  S::PRE * 64 * NW >= L.max_chunks * 64, "the 192-wide four-layer net");
    SplitSeg A = AB.seg[0];
#pragma unroll
    for (int k = 1; k < MAX_SPLIT_SEG; ++k)
        if (k < AB.nseg && (int)blockIdx.x >= AB.seg[k].b0) A = AB.seg[k];
    const int blk0 = A.b0, nblk = A.nb;
    extern __shared__ __attribute__((aligned(16))) unsigned char wl[];          // unit buffer 0 | 1 | the pairs' planes | biases
    float *bias_l = reinterpret_cast<float *>(wl + S::BIAS0);
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, s_ = lane & 31;
    __builtin_assume(tid < 64 * NW);
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pr = wv & 3, w = wv >> 2;                      // pair, wave of the pair
    const int ntiles = A.t1 - A.t0, ngroups = (ntiles + NW / 2 - 1) / (NW / 2);
    for (int i = tid; i < NL * S::BIAS_FLOATS; i += 64 * NW) {
        const int l = i / S::BIAS_FLOATS, k = i % S::BIAS_FLOATS;
        bias_l[i] = k < L32.tiles_out[l] * 32 ? A.packed32[L32.off_bf[l] + k] : 0.f;
    }
    const rsrc_t WP = make_rsrc(A.planes + S::BASE, (unsigned)(L.total_chunks * 1024));
    u32x4 pre[S::PRE];
    auto stage_load = [&](auto UC) __attribute__((always_inline)) {
        constexpr int u = decltype(UC)::value, pieces = L.u_chunks[u] * 64, base = L.u_chunk0[u] * 1024;
#pragma unroll
        for (int k = 0; k < S::PRE; ++k)
            if (k * 64 * NW < pieces) pre[k] = __builtin_amdgcn_raw_buffer_load_b128(WP, (tid + 64 * NW * k) * 16, base, 0);
    };
    auto stage_piece = [&](auto UC, auto KC, unsigned char *dst) __attribute__((always_inline)) {
        constexpr int u = decltype(UC)::value, k = decltype(KC)::value, pieces = L.u_chunks[u] * 64;
        if constexpr (k * 64 * NW < pieces)
            if (tid + 64 * NW * k < pieces) *reinterpret_cast<u32x4 *>(dst + (size_t)(tid + 64 * NW * k) * 16) = pre[k];
    };
    auto stage_store = [&](auto UC, unsigned char *dst) __attribute__((always_inline)) {
        sfor<0, S::PRE>([&](auto KC) { stage_piece(UC, KC, dst); });
    };
    stage_load(std::integral_constant<int, 0>{});
    stage_store(std::integral_constant<int, 0>{}, wl);
    step_barrier();

    // this pair's planes; slot (k-step j, plane q) of this lane: plv[(2 j + q) * 64]
    u32x4 *plv = reinterpret_cast<u32x4 *>(wl + S::PLANES0 + pr * S::PLANE_PAIR) + lane;
    u32x4 *plw = plv + w * 4 * 64;                           // ... of this wave's tile of a pair: k-steps 4 i + 2 w + jj
    const int hvoff = tile_voff(lane) + w * 4096;            // this wave's tile of a pair in a tile-major [6][32][32] block
    const float *bias_w = bias_l + w * 32 + h * 16;          // ... and its biases
    int rmax = 0;                                            // largest |input| / hidden activation of this wave, as bits

    for (int tg = (int)blockIdx.x - blk0, trip = 0; tg < ngroups; tg += nblk, ++trip) {
        const int par = (NU & 1) ? (trip & 1) : 0;           // LDS buffer of unit u: (u + par) & 1
        const int tt = A.t0 + tg * (NW / 2) + pr;
        const bool live = tt < A.t1;                         // a pair past the range runs on the last tile, stores nothing
        const int t = live ? tt : A.t1 - 1;
        const bool save = A.save && live;
        int hv = hvoff;
        asm volatile("" : "+v"(hv));
        // the layer's input planes (both waves: all of them)
        f16x8 in1[S::KSH], in2[S::KSH];
        {
            const rsrc_t RX = make_rsrc(AB.X + (size_t)t * D.xrows * 32, D.xrows * 32 * 4);
            const int xvoff = (h * 8 * 32 + s_) * 4, coff = A.crow * 128;
            float xmax = 0.f;
#pragma unroll
            for (int j = 0; j < KS1; ++j) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int row = 16 * j + 8 * h + i;
                    v[i] = bload1(RX, xvoff + (row < D.cw ? coff : 0), (16 * j + i) * 128);
                }
#pragma unroll
                for (int i = 0; i < 8; i += 2) xmax = fmaxf(xmax, fmaxf(fabsf(v[i]), fabsf(v[i + 1])));
                split8(v, in1[j], in2[j]);
            }
            rmax = max(rmax, __float_as_int(xmax));
        }
        f32x16 am[2];
        float4 bz4[4];
        float ev[16];
        f16x8 o1, o2;                                        // the planes of the k-step the pending tile's epilogue is making
        unsigned mword = 0;
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const unsigned hrec = (save && A.save == 1) ? HBYTES : 0u, mrec = save ? MBYTES : 0u;

        // ---- the epilogue of this wave's finished tile i of hidden layer l, in 24 micro-slices (mlp_split.hip) -------------
        //   phase 0 (register pair p): value = accumulator / 64 + bias, ReLU, the fp32 tile stores
        //   phase 1: mask bits, first plane      phase 2: second plane; behind a k-step's last pair: its planes -> LDS
        auto micro = [&](auto LC, auto IC, auto MS, f32x16 &accm) __attribute__((always_inline)) {
            constexpr int l = decltype(LC)::value, i = decltype(IC)::value, ms = decltype(MS)::value, p = ms / 3, q = ms % 3;
            constexpr int r0 = 2 * p, jj = r0 >> 3, i0 = r0 & 7;
            if constexpr (q == 0) {
                const float4 b4 = bz4[p >> 1];
                const float bx = (p & 1) ? b4.z : b4.x, by = (p & 1) ? b4.w : b4.y;
                float v0 = fmaf(accm[r0], SPLIT_W_INV, bx), v1 = fmaf(accm[r0 + 1], SPLIT_W_INV, by);
                const int b0 = __float_as_int(v0), b1 = __float_as_int(v1);
                v0 = __int_as_float(b0 > 0 ? b0 : 0);
                v1 = __int_as_float(b1 > 0 ? b1 : 0);
                asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(rmax) : "v"(__float_as_int(v0)), "v"(__float_as_int(v1)));
                const rsrc_t RH = make_rsrc(AB.H[l] + (size_t)t * (HBYTES / 4), hrec);
#ifndef ESR_NS_NO_HSTORE                                            // (timing variant of tools/ubench/nsplit_bench.hip: wrong results)
                asm volatile("" : "+v"(hv));
                bstore1_nt(RH, v0, hv + tile_soff(0, r0), i * 8192);
                bstore1_nt(RH, v1, hv + tile_soff(0, r0 + 1), i * 8192);
#endif
                ev[r0] = v0; ev[r0 + 1] = v1;
            } else if constexpr (q == 1) {
                const float v0 = ev[r0], v1 = ev[r0 + 1];
                int one0, one1;
                asm volatile("v_med3_i32 %1, %3, 0, 1\n\t"
                             "v_med3_i32 %2, %4, 0, 1\n\t"
                             "v_lshl_or_b32 %0, %1, %5, %0\n\t"
                             "v_lshl_or_b32 %0, %2, %6, %0"
                             : "+v"(mword), "=&v"(one0), "=&v"(one1)
                             : "v"(__float_as_int(v0)), "v"(__float_as_int(v1)), "n"(r0), "n"(r0 + 1));
                put_pair<i0>(o1, v0, v1);
            } else {
                put_residual_pair<i0>(o2, o1, ev[r0], ev[r0 + 1]);
                if constexpr (i0 == 6) {                             // the k-step's last pair: both planes to the pair's buffer
                    plw[((4 * i + jj) * 2 + 0) * 64] = __builtin_bit_cast(u32x4, o1);
                    plw[((4 * i + jj) * 2 + 1) * 64] = __builtin_bit_cast(u32x4, o2);
                }
                if constexpr (ms == 23) {                            // the wave's 16 mask bits of the tile pair's word
                    __builtin_amdgcn_raw_buffer_store_b16((unsigned short)mword, make_rsrc(AB.M[l] + (size_t)t * (MBYTES / 4), mrec),
                                                          lane * 4 + w * 2, i * 256, 0);
                    mword = 0;
                }
            }
        };
        // the pending tile's biases (its epilogue starts with the next slot)
        auto load_bias4 = [&](auto LC, auto IC) __attribute__((always_inline)) {
            constexpr int l = decltype(LC)::value, i = decltype(IC)::value;
            const float4 *bp = reinterpret_cast<const float4 *>(bias_w + l * S::BIAS_FLOATS + i * 64);
#pragma unroll
            for (int q = 0; q < 4; ++q) bz4[q] = bp[q];
        };
        // the pending tile's 24 micro-slices ride on the first `navail` slots of the tile in flight IN ORDER (o1 / o2 hold one
        // k-step's planes at a time): 24 / navail per slot, one more on the first 24 % navail slots
        auto pending = [&](auto LC, auto IC, auto V, auto NAVAILC, f32x16 &accm) __attribute__((always_inline)) {
            constexpr int v = decltype(V)::value, navail = decltype(NAVAILC)::value;
            constexpr int base = 24 / navail, extra = 24 % navail;
            if constexpr (v < navail) {
                constexpr int first = v * base + (v < extra ? v : extra), cnt = base + (v < extra ? 1 : 0);
                sfor<0, cnt>([&](auto KC) { micro(LC, IC, std::integral_constant<int, first + decltype(KC)::value>{}, accm); });
            }
        };

        // one unit: this wave's tile of (layer l, pair i) over the unit's k-steps
        auto run_unit = [&](auto UC) __attribute__((always_inline)) {
            constexpr int u = decltype(UC)::value, l = L.u_layer[u], i = L.u_i[u], hf = L.u_half[u], KU = L.u_ku[u], kb = L.u_kb[u];
            constexpr int KS = L.ks[l], c = S::tile0_of(u), nxt = (u + 1) % NU;
            constexpr bool LAST = l == NL - 1;
            // the pending tile: the previous tile of this wave (same layer, or the previous layer's last)
            constexpr bool HAVE = c > 0;
            constexpr int pl = S::layer_of(HAVE ? c - 1 : 0), pi = S::i_of(HAVE ? c - 1 : 0);
            // slots it may ride on: all of a tile of the same layer; the first unit only when it feeds this layer (its planes
            // are read by this layer's second K half, one barrier later)
            constexpr int navail = pl == l ? 3 * KS : 3 * KU;
            ESR_NS_STAMP(u, 0);
            unsigned char *wbuf = wl + ((u + par) & 1) * S::BUF, *nbuf = wl + ((u + 1 + par) & 1) * S::BUF;
#ifndef ESR_NS_NO_STAGE                                                // (timing variant: wrong results)
            stage_load(std::integral_constant<int, nxt>{});
#endif
            // this layer's input planes, as far as they are published: K half hf of the previous layer's planes
            if constexpr (l > 0 && i == 0) {
                if (!LAST || w == 0) {
#pragma unroll
                    for (int jj = 0; jj < KU; ++jj) {
                        in1[kb + jj] = __builtin_bit_cast(f16x8, plv[((kb + jj) * 2 + 0) * 64]);
                        in2[kb + jj] = __builtin_bit_cast(f16x8, plv[((kb + jj) * 2 + 1) * 64]);
                    }
                }
            }
            if constexpr (HAVE && hf == 0) load_bias4(std::integral_constant<int, pl>{}, std::integral_constant<int, pi>{});
            f32x16 &m = am[c & 1];
            auto ride = [&](auto U) __attribute__((always_inline)) {
                constexpr int u_ = decltype(U)::value, v = hf * 3 * KU + u_;
                if constexpr (HAVE)
                    pending(std::integral_constant<int, pl>{}, std::integral_constant<int, pi>{}, std::integral_constant<int, v>{},
                            std::integral_constant<int, navail>{}, am[(c - 1) & 1]);
                constexpr int first = 3 * KU - S::PRE;
                static_assert(first >= 0, "a unit has a slot for every staged piece");
#ifndef ESR_NS_NO_STAGE
                if constexpr (u_ >= first) stage_piece(std::integral_constant<int, nxt>{}, std::integral_constant<int, u_ - first>{}, nbuf);
#endif
                __builtin_amdgcn_sched_barrier(0);
            };
            if (!LAST || w == 0) {
                const u32x4 *mine = reinterpret_cast<const u32x4 *>(wbuf) + (LAST ? 0 : w * 2 * KU * 64) + lane;
                u32x4 wb[NS_WR][2];
                sfor<0, (NS_WR - 1 < KU ? NS_WR - 1 : KU)>([&](auto NC) {
                    constexpr int n0 = decltype(NC)::value;
                    wb[n0][0] = mine[n0 * 64];
                    wb[n0][1] = mine[(KU + n0) * 64];
                });
                sfor<0, KU>([&](auto JC) {
                    constexpr int jj = decltype(JC)::value, j = kb + jj;
                    if constexpr (jj == 1) { ESR_NS_STAMP2(u, 0); }
                    if constexpr (jj == KU - 1) { ESR_NS_STAMP2(u, 1); }
#ifndef ESR_NS_NO_WREAD
                    if constexpr (jj + NS_WR - 1 < KU) {
                        wb[(jj + NS_WR - 1) % NS_WR][0] = mine[(jj + NS_WR - 1) * 64];
                        wb[(jj + NS_WR - 1) % NS_WR][1] = mine[(KU + jj + NS_WR - 1) * 64];
                    }
                    const f16x8 w1 = __builtin_bit_cast(f16x8, wb[jj % NS_WR][0]), w2 = __builtin_bit_cast(f16x8, wb[jj % NS_WR][1]);
#else
                    const f16x8 w1 = __builtin_bit_cast(f16x8, wb[0][0]), w2 = __builtin_bit_cast(f16x8, wb[0][1]);
#endif
#ifdef ESR_NS_NO_MFMA                                               // (timing variant: wrong results)
                    if (j == 0) m = zero16;
                    m[j & 15] += (float)(w1[0] + in2[j][0]) + (float)(w2[0] + in1[j][0]);
                    ride(std::integral_constant<int, 3 * jj + 0>{});
                    ride(std::integral_constant<int, 3 * jj + 1>{});
                    ride(std::integral_constant<int, 3 * jj + 2>{});
#else
                    m = mfma_h(w1, in2[j], j == 0 ? zero16 : m);
                    ride(std::integral_constant<int, 3 * jj + 0>{});
                    m = mfma_h(w1, in1[j], m);
                    ride(std::integral_constant<int, 3 * jj + 1>{});
                    m = mfma_h(w2, in1[j], m);
                    ride(std::integral_constant<int, 3 * jj + 2>{});
#endif
                });
                if constexpr (LAST && hf == 1) {
                    const float4 bz = *reinterpret_cast<const float4 *>(bias_l + l * S::BIAS_FLOATS + h * 16);
                    const float bzv[4] = {bz.x, bz.y, bz.z, bz.w};
                    const rsrc_t RZ = make_rsrc(A.zout + (size_t)t * D.zrows * 32, live ? D.zrows * 32 * 4 : 0);
                    const int zvoff = (4 * h * 32 + s_) * 4;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        bstore1(RZ, 4 * h + q < D.out_dim ? fmaf(m[q], SPLIT_W_INV, bzv[q]) : 0.f, zvoff, q * 128);
                }
            } else {
                // wave 1 beside the output layer: its last hidden tile's epilogue (first unit), its share of the staging
                if constexpr (HAVE && hf == 0)
                    sfor<0, 24>([&](auto MC) { micro(std::integral_constant<int, pl>{}, std::integral_constant<int, pi>{}, MC, am[(c - 1) & 1]); });
                stage_store(std::integral_constant<int, nxt>{}, nbuf);
            }
            ESR_NS_STAMP(u, 1);
            step_barrier();
            ESR_NS_STAMP(u, 2);
        };
        sfor<0, NU>([&](auto UC) { run_unit(UC); });
    }
    if (AB.range && rmax >= __float_as_int(SPLIT_RANGE)) atomicOr(AB.range, 1u);
}

// ---- the input-gradient chain on the same scheme (mlp_dgrad_split_kernel<0>: the arithmetic and the tile's scale) ----------
// Transposed layer 0 (W3^T dz: ONE k-step, the 3 output rows): a single unit in which each wave multiplies its three tiles of
// dZ[2] one after the other (the epilogue of tile i rides on tile i + 1's three MFMAs); layers 1, 2 (-> dZ[1], dZ[0]) as the
// forward's hidden layers; layer 3 (-> dX, two tiles of 32 rows): wave w owns tile w, nothing becomes a plane.
template <int KIND>
__global__ void __launch_bounds__(64 * NW, 2) mlp_dgrad_ns_kernel(DSplitBatch AB)
{
    using S = NsSteps<KIND, true>;
    constexpr NetDesc D = net_desc(KIND);
    constexpr NsLayout L = S::L;
    constexpr int NL = S::NL, NHID = NL - 1, HT = D.hid_tiles, NU = S::NU;
    constexpr unsigned HBYTES = HT * 32 * 32 * 4, MBYTES = (HT / 2) * 256;
    static_assert(NL == 4 && HT == 6 && L.u_multi[0] && L.ks[0] == 1 && L.tiles[0] == HT && L.tiles[NL - 1] == 2 &&
                  L.ks[1] == 2 * HT && L.ks[NL - 1] == 2 * HT && D.out_dim <= 8 && D.zrows <= 8, "the 192-wide four-layer net");
    DSplitSeg A = AB.seg[0];
    if (AB.nseg > 1 && (int)blockIdx.x >= AB.seg[1].b0) A = AB.seg[1];
    const int blk0 = A.b0, nblk = A.nb;
    extern __shared__ __attribute__((aligned(16))) unsigned char wl[];          // unit buffer 0 | 1 | the pairs' planes
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, s_ = lane & 31;
    __builtin_assume(tid < 64 * NW);
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pr = wv & 3, w = wv >> 2;
    const int ntiles = A.t1 - A.t0, ngroups = (ntiles + NW / 2 - 1) / (NW / 2);
    const rsrc_t WP = make_rsrc(A.planes + S::BASE, (unsigned)(L.total_chunks * 1024));
    u32x4 pre[S::PRE];
    auto stage_load = [&](auto UC) __attribute__((always_inline)) {
        constexpr int u = decltype(UC)::value, pieces = L.u_chunks[u] * 64, base = L.u_chunk0[u] * 1024;
#pragma unroll
        for (int k = 0; k < S::PRE; ++k)
            if (k * 64 * NW < pieces) pre[k] = __builtin_amdgcn_raw_buffer_load_b128(WP, (tid + 64 * NW * k) * 16, base, 0);
    };
    auto stage_piece = [&](auto UC, auto KC, unsigned char *dst) __attribute__((always_inline)) {
        constexpr int u = decltype(UC)::value, k = decltype(KC)::value, pieces = L.u_chunks[u] * 64;
        if constexpr (k * 64 * NW < pieces)
            if (tid + 64 * NW * k < pieces) *reinterpret_cast<u32x4 *>(dst + (size_t)(tid + 64 * NW * k) * 16) = pre[k];
    };
    auto stage_store = [&](auto UC, unsigned char *dst) __attribute__((always_inline)) {
        sfor<0, S::PRE>([&](auto KC) { stage_piece(UC, KC, dst); });
    };
    stage_load(std::integral_constant<int, 0>{});
    stage_store(std::integral_constant<int, 0>{}, wl);
    step_barrier();

    u32x4 *plv = reinterpret_cast<u32x4 *>(wl + S::PLANES0 + pr * S::PLANE_PAIR) + lane;
    u32x4 *plw = plv + w * 4 * 64;
    const int hvoff = tile_voff(lane) + w * 4096;
    float wmax = 0.f;
    // the net's gradient gain bound (mlp.hip: split_gain_kernel), copied behind the prototype's planes: see mlp_dgrad_split_kernel
    const float *gainp = reinterpret_cast<const float *>(A.planes + ns_elems(KIND));
    const int gbits = __builtin_amdgcn_readfirstlane(__float_as_int(*gainp));
    const int kbase = __builtin_amdgcn_readfirstlane(141 + 127 - ((gbits >> 23) & 0xff) - ((gbits & 0x7fffff) ? 1 : 0));

    for (int tg = (int)blockIdx.x - blk0, trip = 0; tg < ngroups; tg += nblk, ++trip) {
        const int par = (NU & 1) ? (trip & 1) : 0;
        const int tt = A.t0 + tg * (NW / 2) + pr;
        const bool live = tt < A.t1;
        const int t = live ? tt : A.t1 - 1;
        int hv = hvoff;
        asm volatile("" : "+v"(hv));
        // the tile's output gradients (both waves of the pair: the same rows), this wave's ReLU mask bits, the tile's scale
        float zn[D.zrows];
        unsigned msk[NHID][HT / 2];
        {
            const rsrc_t RZ = make_rsrc(AB.dz + (size_t)t * D.zrows * 32, D.zrows * 32 * 4);
#pragma unroll
            for (int i = 0; i < D.zrows; ++i) zn[i] = bload1(RZ, s_ * 4, i * 128);
#pragma unroll
            for (int l = 0; l < NHID; ++l) {
                load_relu_mask<HT>(make_rsrc(AB.M[l] + (size_t)t * (MBYTES / 4), MBYTES), msk[l], lane);
#pragma unroll
                for (int k = 0; k < HT / 2; ++k) msk[l][k] >>= 16 * w;          // this wave's tile of the pair: bits 0..15
            }
        }
        float zmax = 0.f;
#pragma unroll
        for (int i = 0; i < D.out_dim; ++i) zmax = fmaxf(zmax, fabsf(zn[i]));
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) zmax = fmaxf(zmax, __shfl_xor(zmax, o));
        if (live) wmax = fmaxf(wmax, zmax);
        const int ez = (__float_as_int(zmax) >> 23) & 0xff;
        const int ks = ez == 0 ? 0 : kbase - ez;
        const int kc = ks < -100 ? -100 : (ks > 100 ? 100 : ks);
        const float sc = __int_as_float((127 + kc) << 23), isc = __int_as_float((127 - kc) << 23);
        f16x8 in1[S::KSH], in2[S::KSH];
        {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = (h == 0 && i < D.out_dim) ? zn[i < D.out_dim ? i : 0] * sc : 0.f;
            split8(v, in1[0], in2[0]);
        }
        f32x16 am[2];
        float ev[16];
        f16x8 o1, o2;
        const float wisc = SPLIT_W_INV * isc;                // accumulator (64 x the scaled gradient) -> the fp32 store
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

        // micro-slices of this wave's finished tile i of transposed layer q (mlp_dgrad_split_kernel's, planes to LDS)
        auto micro = [&](auto QC, auto IC, auto MS, f32x16 &accm) __attribute__((always_inline)) {
            constexpr int q = decltype(QC)::value, i = decltype(IC)::value, ms = decltype(MS)::value, p = ms / 3, ph = ms % 3;
            constexpr int r0 = 2 * p, jj = r0 >> 3, i0 = r0 & 7;
            if constexpr (q == NL - 1) {
                if constexpr (ph == 0) {
                    const float v0 = accm[r0] * wisc, v1 = accm[r0 + 1] * wisc;
                    const rsrc_t RX = make_rsrc(AB.dX + (size_t)t * 64 * 32, live ? dx_rows(KIND) / 4 * 4 * 128 + (dx_rows(KIND) % 4 ? 512 : 0) : 0);
                    asm volatile("" : "+v"(hv));
                    bstore1(RX, v0, hv + tile_soff(0, r0), i * 8192);
                    bstore1(RX, v1, hv + tile_soff(0, r0 + 1), i * 8192);
                }
            } else {
                constexpr int d = NHID - 1 - q;
                if constexpr (ph == 0) {
                    int k0, k1;
                    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(k0) : "v"(msk[d][i]), "n"(r0));
                    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(k1) : "v"(msk[d][i]), "n"(r0 + 1));
                    const int a0 = __float_as_int(accm[r0]) & k0, a1 = __float_as_int(accm[r0 + 1]) & k1;
                    const rsrc_t RD = make_rsrc(AB.dZ[d] + (size_t)t * (HBYTES / 4), (live && AB.dZ[d]) ? HBYTES : 0u);
#ifndef ESR_NS_NO_HSTORE
                    asm volatile("" : "+v"(hv));
                    bstore1_nt(RD, __int_as_float(a0) * wisc, hv + tile_soff(0, r0), i * 8192);
                    bstore1_nt(RD, __int_as_float(a1) * wisc, hv + tile_soff(0, r0 + 1), i * 8192);
#endif
                    ev[r0] = __int_as_float(a0) * SPLIT_W_INV; ev[r0 + 1] = __int_as_float(a1) * SPLIT_W_INV;
                } else if constexpr (ph == 1) {
                    put_pair<i0>(o1, ev[r0], ev[r0 + 1]);
                } else {
                    put_residual_pair<i0>(o2, o1, ev[r0], ev[r0 + 1]);
                    if constexpr (i0 == 6) {
                        plw[((4 * i + jj) * 2 + 0) * 64] = __builtin_bit_cast(u32x4, o1);
                        plw[((4 * i + jj) * 2 + 1) * 64] = __builtin_bit_cast(u32x4, o2);
                    }
                }
            }
        };
        auto pending = [&](auto QC, auto IC, auto V, auto NAVAILC, f32x16 &accm) __attribute__((always_inline)) {
            constexpr int v = decltype(V)::value, navail = decltype(NAVAILC)::value;
            constexpr int base = 24 / navail, extra = 24 % navail;
            if constexpr (v < navail) {
                constexpr int first = v * base + (v < extra ? v : extra), cnt = base + (v < extra ? 1 : 0);
                sfor<0, cnt>([&](auto KC) { micro(QC, IC, std::integral_constant<int, first + decltype(KC)::value>{}, accm); });
            }
        };

        auto run_unit = [&](auto UC) __attribute__((always_inline)) {
            constexpr int u = decltype(UC)::value, q = L.u_layer[u], hf = L.u_half[u], KU = L.u_ku[u], kb = L.u_kb[u];
            constexpr int KS = L.ks[q], c0 = S::tile0_of(u), nxt = (u + 1) % NU, NTU = L.u_multi[u] ? L.tiles[q] / 2 : 1;
            unsigned char *wbuf = wl + ((u + par) & 1) * S::BUF, *nbuf = wl + ((u + 1 + par) & 1) * S::BUF;
            ESR_NS_STAMP(u, 0);
            stage_load(std::integral_constant<int, nxt>{});
            if constexpr (q > 0 && L.u_i[u] == 0) {
#pragma unroll
                for (int jj = 0; jj < KU; ++jj) {
                    in1[kb + jj] = __builtin_bit_cast(f16x8, plv[((kb + jj) * 2 + 0) * 64]);
                    in2[kb + jj] = __builtin_bit_cast(f16x8, plv[((kb + jj) * 2 + 1) * 64]);
                }
            }
            // chunk (tile n of the unit, plane p, k-step jj of the unit) of this wave
            const u32x4 *mine = reinterpret_cast<const u32x4 *>(wbuf) + w * (L.u_chunks[u] / 2) * 64 + lane;
            sfor<0, NTU>([&](auto NC) {
                constexpr int n = decltype(NC)::value, c = c0 + n;
                constexpr bool HAVE = c > 0;
                constexpr int pq = S::layer_of(HAVE ? c - 1 : 0), pi = S::i_of(HAVE ? c - 1 : 0);
                constexpr int navail = pq == q ? 3 * KS : 3 * KU;
                constexpr int NSLOT = 3 * KU;
                f32x16 &m = am[c & 1];
                u32x4 wb[NS_WR][2];
                auto wread = [&](auto JJ, u32x4 (&dst)[2]) __attribute__((always_inline)) {
                    constexpr int jj = decltype(JJ)::value;
                    dst[0] = mine[((n * 2 + 0) * KU + jj) * 64];
                    dst[1] = mine[((n * 2 + 1) * KU + jj) * 64];
                };
                sfor<0, (NS_WR - 1 < KU ? NS_WR - 1 : KU)>([&](auto JC) { wread(JC, wb[decltype(JC)::value]); });
                sfor<0, KU>([&](auto JC) {
                    constexpr int jj = decltype(JC)::value, j = kb + jj;
                    if constexpr (jj + NS_WR - 1 < KU) wread(std::integral_constant<int, jj + NS_WR - 1>{}, wb[(jj + NS_WR - 1) % NS_WR]);
                    const f16x8 w1 = __builtin_bit_cast(f16x8, wb[jj % NS_WR][0]), w2 = __builtin_bit_cast(f16x8, wb[jj % NS_WR][1]);
                    auto ride = [&](auto U) __attribute__((always_inline)) {
                        constexpr int u_ = decltype(U)::value, v = hf * 3 * KU + u_;
                        if constexpr (HAVE)
                            pending(std::integral_constant<int, pq>{}, std::integral_constant<int, pi>{}, std::integral_constant<int, v>{},
                                    std::integral_constant<int, navail>{}, am[(c - 1) & 1]);
                        constexpr int first = NSLOT - S::PRE;
                        if constexpr (n == NTU - 1 && first >= 0 && u_ >= first)
                            stage_piece(std::integral_constant<int, nxt>{}, std::integral_constant<int, u_ - (first >= 0 ? first : 0)>{}, nbuf);
                        __builtin_amdgcn_sched_barrier(0);
                    };
                    m = mfma_h(w1, in2[j], j == 0 ? zero16 : m);
                    ride(std::integral_constant<int, 3 * jj + 0>{});
                    m = mfma_h(w1, in1[j], m);
                    ride(std::integral_constant<int, 3 * jj + 1>{});
                    m = mfma_h(w2, in1[j], m);
                    ride(std::integral_constant<int, 3 * jj + 2>{});
                });
                if constexpr (q == NL - 1 && hf == 1)       // the very last tile (dX rows 32 w ..): nobody to ride on
                    sfor<0, 24>([&](auto MC) { micro(std::integral_constant<int, q>{}, std::integral_constant<int, L.u_i[u]>{}, MC, m); });
            });
            ESR_NS_STAMP(u, 1);
            step_barrier();
            ESR_NS_STAMP(u, 2);
        };
        sfor<0, NU>([&](auto UC) { run_unit(UC); });
    }
    wmax *= fmaxf(1.f, *gainp * 0.0625f);
    if (AB.amax && lane == 0 && wmax > *reinterpret_cast<volatile float *>(AB.amax))
        atomicMax(reinterpret_cast<unsigned *>(AB.amax), __float_as_uint(wmax));
}

}  // namespace
