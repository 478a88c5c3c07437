// The 192-wide radiance net on the split-fp16 scheme (mlp_split.hip) at TWO WAVES PER SIMD (round 5).
//
// mlp_fwd_split_kernel<0> / mlp_dgrad_split_kernel<0> keep a layer's input AND output planes of a 32-sample tile in one
// wave's registers (~400): one wave per SIMD, whose in-order issue serialises its vector instructions (~25 k clocks per tile
// group), its LDS weight reads (~20 k) and its MFMAs (18.4 k) to 44.6 k clocks (tools/ubench/split_stamps.hip).  Here TWO
// waves share a tile and split every product along K by k-step parity:
//   wave A (w = 0) holds the planes of the EVEN k-steps of a layer's input, wave B (w = 1) those of the odd ones -- 24
//   registers per plane instead of 48 -- and multiplies its half of K into every output tile.  Of a tile's sixteen
//   accumulator registers a wave FINISHES eight (adds the partner's partial sums, bias, ReLU, the fp32 tile store, mask
//   bits, the two planes) and hands the other eight to its partner through LDS (2 KB each way, fp32).  B's weight rows are
//   rotated by 16 inside a tile (mlp_common.h: packp_body), so the eight registers a wave finishes are ALWAYS its registers
//   0..7 -- output rows 0..15 of the tile for A, 16..31 for B -- and those are exactly its own input planes of the next layer
//   (tile t, rows 0..15 = k-step 2 t; rows 16..31 = k-step 2 t + 1): nothing but partial sums ever moves between the waves,
//   and BOTH WAVES RUN THE SAME CODE with the same amount of work in every step (a first version that gave whole tiles to one
//   wave left its partner waiting at the barrier for half of every step: no faster than the one-wave kernel).
// ~200 registers per wave: eight waves per CU, the A and the B wave of a pair share a SIMD, one's vector work runs beside the
// other's MFMAs.  A step = ONE output tile (2 KS chunks for both waves, 24 KB; three-buffer ring staged two steps ahead, one
// barrier per step).  LDS traffic per tile and layer: 148 KB of weight reads as before + 48 KB of partial sums.
// Results: the same products summed in a different order (K parities first): equal to the one-wave kernels' to fp32
// rounding (tests/test_gpu_split.py); buffers, masks and planes formats unchanged.
#pragma once

// in-kernel time stamps for tools/ubench/pair_bench.hip (nothing in the product build)
#ifndef ESR_PAIR_STAMP
#define ESR_PAIR_STAMP(st, i)
#endif

// an MFMA pinned to its slot by an empty asm on its accumulator (run_step: why)
#ifdef ESR_PAIR_NO_PIN
#define ESR_PAIR_PIN(m)
#else
#define ESR_PAIR_PIN(m) asm volatile("" : "+v"(m))
#endif

constexpr int PW = 8;                                       // waves per workgroup: four pairs = four tiles per group
constexpr int PRING = 3;                                    // step buffers (staged two steps ahead)
constexpr int PAIR_XCH = 2048;                              // bytes of one wave's partial sums for its partner (8 fp32 per lane)

template <int KIND, bool BWD> struct PairSteps {
    static constexpr PairLayout L = BWD ? pair_layout_t(KIND) : pair_layout(KIND);
    static constexpr int NL = L.n_layers, NS = L.n_steps;
    static constexpr int BUF = L.max_chunks * 1024;
    static constexpr int PRE = (L.max_chunks * 64 + 64 * PW - 1) / (64 * PW);        // 16-byte pieces per thread and step
    static constexpr int BIAS_FLOATS = 32 * MAX_HID_TILES;
    static constexpr int XCH0 = PRING * BUF;                                         // partial sums: [pair][step parity][wave][2 KB]
    static constexpr int ZX0 = XCH0 + (PW / 2) * 2 * 2 * PAIR_XCH;                   // the output tile's: [pair][wave][1 KB]
    static constexpr int BIAS0 = ZX0 + (PW / 2) * 2 * 1024;
    static constexpr int LDS_BYTES = BIAS0 + (BWD ? 0 : NL * BIAS_FLOATS * 4);
    // element offset of the pair planes inside the net's planes buffer
    static constexpr int64_t BASE = pair_offset(KIND) + (BWD ? (int64_t)pair_layout(KIND).total_chunks * 512 : 0);
};

// the eight registers of a finished tile that a wave hands to its partner / takes from it
__device__ __forceinline__ void pair_send(unsigned char *slot, const f32x16 &m, int lane)
{
    f32x4 *px = reinterpret_cast<f32x4 *>(slot) + lane;
    px[0] = f32x4{m[8], m[9], m[10], m[11]};
    px[64] = f32x4{m[12], m[13], m[14], m[15]};
}
__device__ __forceinline__ void pair_recv(const unsigned char *slot, float (&part)[8], int lane)
{
    const f32x4 *px = reinterpret_cast<const f32x4 *>(slot) + lane;
    const f32x4 a = px[0], b = px[64];
    part[0] = a[0]; part[1] = a[1]; part[2] = a[2]; part[3] = a[3];
    part[4] = b[0]; part[5] = b[1]; part[6] = b[2]; part[7] = b[3];
}

template <int KIND>
__global__ void __launch_bounds__(64 * PW, 2) mlp_fwd_pair_kernel(SplitBatch AB)
{
    using S = PairSteps<KIND, false>;
    constexpr NetDesc D = net_desc(KIND);
    constexpr PairLayout L = S::L;
    constexpr PackLayout L32 = pack_layout(KIND);
    constexpr int NL = S::NL, HT = D.hid_tiles, NS = S::NS, KX = L.kw[0];
    constexpr unsigned HBYTES = HT * 32 * 32 * 4, MBYTES = (HT / 2) * 256;
    static_assert(NL == 4 && HT == 6 && L.tiles[NL - 1] == 1 && !L.rowsplit[0] && L.ks[0] % 2 == 0 && L.ks[1] == 2 * HT,
                  "the 192-wide four-layer net");
    SplitSeg A = AB.seg[0];
#pragma unroll
    for (int k = 1; k < MAX_SPLIT_SEG; ++k)
        if (k < AB.nseg && (int)blockIdx.x >= AB.seg[k].b0) A = AB.seg[k];
    const int blk0 = A.b0, nblk = A.nb;
    extern __shared__ __attribute__((aligned(16))) unsigned char wl[];          // ring of step buffers | partial sums | biases
    float *bias_l = reinterpret_cast<float *>(wl + S::BIAS0);
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, s_ = lane & 31;
    __builtin_assume(tid < 64 * PW);
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pr = wv & 3, w = wv >> 2;                      // pair, wave of the pair: the two share a SIMD
    const int ntiles = A.t1 - A.t0, ngroups = (ntiles + PW / 2 - 1) / (PW / 2);
    for (int i = tid; i < NL * S::BIAS_FLOATS; i += 64 * PW) {
        const int l = i / S::BIAS_FLOATS, k = i % S::BIAS_FLOATS;
        bias_l[i] = k < L32.tiles_out[l] * 32 ? A.packed32[L32.off_bf[l] + k] : 0.f;
    }
    const rsrc_t WP = make_rsrc(A.planes + S::BASE, (unsigned)(L.total_chunks * 1024));
    u32x4 pre[S::PRE];
    auto stage_load = [&](auto ST) __attribute__((always_inline)) {
        constexpr int st = decltype(ST)::value, pieces = L.chunks[st] * 64, base = L.chunk0[st] * 1024;
#pragma unroll
        for (int k = 0; k < S::PRE; ++k)
            if (k * 64 * PW < pieces) pre[k] = __builtin_amdgcn_raw_buffer_load_b128(WP, (tid + 64 * PW * k) * 16, base, 0);
    };
    auto stage_piece = [&](auto ST, auto KC, unsigned char *dst) __attribute__((always_inline)) {
        constexpr int st = decltype(ST)::value, k = decltype(KC)::value, pieces = L.chunks[st] * 64;
        if constexpr (k * 64 * PW < pieces)
            if (tid + 64 * PW * k < pieces) *reinterpret_cast<u32x4 *>(dst + (size_t)(tid + 64 * PW * k) * 16) = pre[k];
    };
    auto stage_store = [&](auto ST, unsigned char *dst) __attribute__((always_inline)) {
        sfor<0, S::PRE>([&](auto KC) { stage_piece(ST, KC, dst); });
    };
    // ring slot of the step about to run (steps are counted across tile groups: NS is not a multiple of the ring)
    int rb = 0, par = 0;
    stage_load(std::integral_constant<int, 0>{});
    stage_store(std::integral_constant<int, 0>{}, wl);
    stage_load(std::integral_constant<int, 1 % NS>{});
    stage_store(std::integral_constant<int, 1 % NS>{}, wl + S::BUF);
    step_barrier();

    unsigned char *xch = wl + S::XCH0 + pr * 4 * PAIR_XCH;   // this pair's partial sums: [step parity][written by wave]
    unsigned char *zx = wl + S::ZX0 + pr * 2 * 1024;
    const int hvoff = tile_voff(lane) + w * 16 * 128;        // this wave's rows of a tile-major tile: + 16 rows for B
    const int wsel = w * 2 * 1024;                           // this wave's chunks inside a step buffer, per k-step of the wave
    int rmax = 0;                                            // largest |input| / hidden activation of this wave, as bits

    for (int tg = (int)blockIdx.x - blk0; tg < ngroups; tg += nblk) {
        const int tt = A.t0 + tg * (PW / 2) + pr;
        const bool live = tt < A.t1;                         // a pair past the range runs on the last tile, stores nothing
        const int t = live ? tt : A.t1 - 1;
        const bool save = A.save && live;
        int hv = hvoff;
        asm volatile("" : "+v"(hv));
        // planes of this wave's k-steps: first layer's input | set A | set B (layer 0 writes A, 1 reads A writes B, ...)
        f16x8 xi1[KX], xi2[KX], pa1[HT], pa2[HT], pb1[HT], pb2[HT];
        {
            // k-step 2 jl + w of the input tile: rows 16 (2 jl + w) + 8 h + i (rows past the tile: the descriptor returns zeros)
            const rsrc_t RX = make_rsrc(AB.X + (size_t)t * D.xrows * 32, D.xrows * 32 * 4);
            const int xvoff = (h * 8 * 32 + s_) * 4 + w * 16 * 128;
            const int xv0 = xvoff + ((w == 0 && h == 0) ? A.crow * 128 : 0);       // the colour group (rows 0 .. cw-1) re-targeted
            static_assert(D.cw <= 8, "the colour group sits in the first eight rows");
            float xmax = 0.f;
#pragma unroll
            for (int j = 0; j < KX; ++j) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = bload1(RX, (j == 0 && i < D.cw) ? xv0 : xvoff, (32 * j + i) * 128);
#pragma unroll
                for (int i = 0; i < 8; i += 2) xmax = fmaxf(xmax, fmaxf(fabsf(v[i]), fabsf(v[i + 1])));
                split8(v, xi1[j], xi2[j]);
            }
            rmax = max(rmax, __float_as_int(xmax));
        }
        f32x16 am[2];
        float4 bz4[2];
        float ev[8], part[8];
        unsigned mword = 0;
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const unsigned hrec = (save && A.save == 1) ? HBYTES : 0u, mrec = save ? MBYTES : 0u;

        // the epilogue of this wave's half of hidden tile `it` of layer l: 4 register pairs x 3 phases = 12 micro-slices, each
        // issued behind one MFMA of the following tile (mlp_split.hip)
        //   phase 0 (pair p): value = (own + partner's sums) / 64 + bias, ReLU, the fp32 tile stores
        //   phase 1: mask bits, first plane      phase 2: second plane
        auto micro = [&](auto LC, auto IT, auto MS, f32x16 &accm, auto &o1, auto &o2) __attribute__((always_inline)) {
            constexpr int l = decltype(LC)::value, it = decltype(IT)::value, ms = decltype(MS)::value, p = ms / 3, q = ms % 3;
            constexpr int r0 = 2 * p;
            if constexpr (q == 0) {
                const float4 b4 = bz4[p >> 1];
                const float bx = (p & 1) ? b4.z : b4.x, by = (p & 1) ? b4.w : b4.y;
                // the partner's sums, PINNED to this slice (LDS-loaded values: no MFMA result near this asm): left free, the
                // additions / multiply-adds / maxima of a tile were vectorised into one block at the top of the step
                float p0 = part[r0], p1 = part[r0 + 1];
                asm volatile("" : "+v"(p0), "+v"(p1));
                float v0 = fmaf(accm[r0] + p0, SPLIT_W_INV, bx), v1 = fmaf(accm[r0 + 1] + p1, SPLIT_W_INV, by);
                const int b0 = __float_as_int(v0), b1 = __float_as_int(v1);
                v0 = __int_as_float(b0 > 0 ? b0 : 0);
                v1 = __int_as_float(b1 > 0 ? b1 : 0);
                asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(rmax) : "v"(__float_as_int(v0)), "v"(__float_as_int(v1)));
                const rsrc_t RH = make_rsrc(AB.H[l] + (size_t)t * (HBYTES / 4), hrec);
#ifndef ESR_PAIR_NO_HSTORE                                          // (timing variant: wrong results)
                asm volatile("" : "+v"(hv));
                bstore1_nt(RH, v0, hv + tile_soff(0, r0), it * 4096);
                bstore1_nt(RH, v1, hv + tile_soff(0, r0 + 1), it * 4096);
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
                put_pair<r0>(o1[it], v0, v1);
            } else {
                put_residual_pair<r0>(o2[it], o1[it], ev[r0], ev[r0 + 1]);
            }
        };
        auto pending = [&](auto LC, auto IT, auto U, auto NAVAILC, f32x16 &accm, auto &o1, auto &o2) __attribute__((always_inline)) {
            constexpr int u = decltype(U)::value, navail = decltype(NAVAILC)::value;
            static_assert(navail >= 3 && navail % 3 == 0, "whole register pairs per pass");
            if constexpr (u < navail)
                sfor<0, (12 + navail - 1) / navail>([&](auto KC) {
                    constexpr int msi = u + decltype(KC)::value * navail;
                    if constexpr (msi < 12) micro(LC, IT, std::integral_constant<int, msi>{}, accm, o1, o2);
                });
            // behind the LAST micro-slice: the wave's eight mask bits of the tile (one byte of the pair's mask word)
            constexpr int l = decltype(LC)::value, it = decltype(IT)::value;
            if constexpr (u == (navail < 12 ? navail : 12) - 1) {
                __builtin_amdgcn_raw_buffer_store_b8((unsigned char)mword, make_rsrc(AB.M[l] + (size_t)t * (MBYTES / 4), mrec),
                                                     lane * 4 + (it & 1) * 2 + w, (it >> 1) * 256, 0);
                mword = 0;
            }
        };

        // one step = output tile `it` of layer l over this wave's k-steps
        auto run_step = [&](auto LC, auto IT, auto &in1, auto &in2, auto &o1, auto &o2) __attribute__((always_inline)) {
            constexpr int l = decltype(LC)::value, it = decltype(IT)::value;
            constexpr int KW = L.kw[l], st = L.step0[l] + it;
            constexpr bool LAST = l == NL - 1;
            // the tile both waves finish during this step: the previous step's
            constexpr bool HAVE = it > 0 || l > 0;
            constexpr int pl = it > 0 ? l : l - 1, pit = it > 0 ? it - 1 : (l > 0 ? L.tiles[l > 0 ? l - 1 : 0] - 1 : 0);
            constexpr int nxt2 = (st + 2) % NS;
            ESR_PAIR_STAMP(st, 0);
            const u32x4 *mine = reinterpret_cast<const u32x4 *>(wl + rb * S::BUF + wsel * KW) + lane;
            stage_load(std::integral_constant<int, nxt2>{});
            if constexpr (HAVE) {
                // the pending tile's biases (this wave's eight rows)
                const float4 *bp = reinterpret_cast<const float4 *>(bias_l + pl * S::BIAS_FLOATS + pit * 32 + h * 16 + w * 8);
                bz4[0] = bp[0]; bz4[1] = bp[1];
            }
            // THE BARRIER THAT CLOSES THE PREVIOUS STEP SITS INSIDE THIS ONE, behind its first JB k-steps: this step's weights
            // were staged two steps ago (complete since the barrier before last), so a wave that finishes a step early runs
            // on into the next tile's first MFMAs instead of waiting for the slowest of eight waves 19 times per tile group
            // (stamps: 0.3-0.7 k of a 2.4 k-clock step at its end).  Behind it: the partner's partial sums of the pending tile
            // (sent at the end of the previous step) and the ring slot this step writes in its last slots (read two steps ago).
            constexpr int WR = 2;
            u32x4 wb[WR][2];
            wb[0][0] = mine[0];
            wb[0][1] = mine[KW * 64];
            f32x16 &m = am[it & 1];
            sfor<0, KW>([&](auto JC) {
                constexpr int j = decltype(JC)::value;
#ifndef ESR_PAIR_NO_WREAD                                       // (timing variant of tools/ubench/pair_bench.hip: wrong results)
                if constexpr (j + 1 < KW) {
                    wb[(j + 1) % WR][0] = mine[(j + 1) * 64];
                    wb[(j + 1) % WR][1] = mine[(KW + j + 1) * 64];
                }
                const f16x8 w1 = __builtin_bit_cast(f16x8, wb[j % WR][0]), w2 = __builtin_bit_cast(f16x8, wb[j % WR][1]);
#else
                const f16x8 w1 = __builtin_bit_cast(f16x8, wb[0][0]), w2 = __builtin_bit_cast(f16x8, wb[0][1]);
#endif
                auto ride = [&](auto U) __attribute__((always_inline)) {
                    constexpr int u_ = decltype(U)::value, first = 3 * KW - S::PRE;
                    constexpr int JB = 1, ub = 3 * JB;               // the previous step's barrier: behind slot ub - 1
                    static_assert(first >= ub, "the ring slot is written behind the barrier");
                    if constexpr (HAVE && u_ == ub - 1) {
                        step_barrier();
                        pair_recv(xch + ((par ^ 1) * 2 + (w ^ 1)) * PAIR_XCH, part, lane);
                    }
#ifdef ESR_PAIR_NO_EPI                                              // (timing variant: wrong results)
                    if constexpr (false) {
#else
                    if constexpr (HAVE && u_ >= ub) {
#endif
                        // a tile of the same layer: every slot behind the barrier; the previous layer's last tile: the slots
                        // before this step first READS its planes (the wave's last k-step)
                        constexpr auto UU = std::integral_constant<int, u_ - ub>{};
                        if constexpr (it > 0) pending(std::integral_constant<int, pl>{}, std::integral_constant<int, pit>{}, UU,
                                                      std::integral_constant<int, 3 * KW - ub>{}, am[pit & 1], o1, o2);
                        else pending(std::integral_constant<int, pl>{}, std::integral_constant<int, pit>{}, UU,
                                     std::integral_constant<int, 3 * (KW - 1) - ub>{}, am[pit & 1], in1, in2);
                    }
                    static_assert(first >= 0, "a step has a slot for every staged piece");
                    if constexpr (u_ >= first)
                        stage_piece(std::integral_constant<int, nxt2>{}, std::integral_constant<int, u_ - first>{},
                                    wl + ((rb + 2) % PRING) * S::BUF);
                    __builtin_amdgcn_sched_barrier(0);
                };
                // (each MFMA PINNED to its slot by an empty asm on the accumulator: the instruction selector is free to sink an
                //  MFMA -- a pure operation -- towards its use, and did: all but the first five of a step ran in a row behind
                //  the epilogue.  No instruction is emitted, the accumulator stays where it is, and its next reader is the
                //  next MFMA of the chain: same destination and srcC, which needs no wait state)
                m = mfma_h(w1, in2[j], j == 0 ? zero16 : m);
                ESR_PAIR_PIN(m);
                ride(std::integral_constant<int, 3 * j + 0>{});
                m = mfma_h(w1, in1[j], m);
                ESR_PAIR_PIN(m);
                ride(std::integral_constant<int, 3 * j + 1>{});
                m = mfma_h(w2, in1[j], m);
                ESR_PAIR_PIN(m);
                ride(std::integral_constant<int, 3 * j + 2>{});
            });
            ESR_PAIR_STAMP(st, 1);
            // hand the partner its half of the sums
            if constexpr (LAST) {
                f32x4 *px = reinterpret_cast<f32x4 *>(zx + w * 1024) + lane;         // (rows 0..7 of the output tile: B's registers 8..11)
                px[0] = f32x4{m[8], m[9], m[10], m[11]};
            } else {
#ifndef ESR_PAIR_NO_XCH                                              // (timing variant: wrong results)
                pair_send(xch + (par * 2 + w) * PAIR_XCH, m, lane);
#endif
            }
            ESR_PAIR_STAMP(st, 2);
            if constexpr (LAST) step_barrier();                   // (the group's last step: closed here, the next group's first has no barrier)
            ESR_PAIR_STAMP(st, 3);
            rb = rb + 1 == PRING ? 0 : rb + 1;
            par ^= 1;
            if constexpr (LAST) {
                if (w == 0) {
                    // the output rows: A's sums + B's (rows 4 h + q in registers q = 0..3)
                    const f32x4 pz = (reinterpret_cast<const f32x4 *>(zx + 1024) + lane)[0];
                    const float4 bz = *reinterpret_cast<const float4 *>(bias_l + l * S::BIAS_FLOATS + h * 16);
                    const float bzv[4] = {bz.x, bz.y, bz.z, bz.w};
                    const rsrc_t RZ = make_rsrc(A.zout + (size_t)t * D.zrows * 32, live ? D.zrows * 32 * 4 : 0);
                    const int zvoff = (4 * h * 32 + s_) * 4;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        bstore1(RZ, 4 * h + q < D.out_dim ? fmaf(m[q] + pz[q], SPLIT_W_INV, bzv[q]) : 0.f, zvoff, q * 128);
                }
            }
        };
        constexpr auto C0 = std::integral_constant<int, 0>{};
        constexpr auto C1 = std::integral_constant<int, 1>{};
        constexpr auto C2 = std::integral_constant<int, 2>{};
        constexpr auto C3 = std::integral_constant<int, 3>{};
        sfor<0, HT>([&](auto IT) { run_step(C0, IT, xi1, xi2, pa1, pa2); });
        sfor<0, HT>([&](auto IT) { run_step(C1, IT, pa1, pa2, pb1, pb2); });
        sfor<0, HT>([&](auto IT) { run_step(C2, IT, pb1, pb2, pa1, pa2); });
        run_step(C3, C0, pa1, pa2, pb1, pb2);                 // output layer (pb: unused)
    }
    if (AB.range && rmax >= __float_as_int(SPLIT_RANGE)) atomicOr(AB.range, 1u);
}
