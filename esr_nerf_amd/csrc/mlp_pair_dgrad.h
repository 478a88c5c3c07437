// The input-gradient chain of the 192-wide radiance net on wave pairs (mlp_pair.h: the scheme; mlp_dgrad_split_kernel<0>: the
// arithmetic -- a tile's chain runs scaled by a power of two chosen from max |dz| and the net's gain bound).
// Transposed layer 0 (W3^T dz: ONE k-step, the 3 output rows) cannot be split along K: it is a single step in which BOTH
// waves multiply all six tiles of dZ[2] from the same dz planes (B with its rows rotated) and each finishes its eight
// registers of every tile -- no partial sums, 18 MFMAs more per wave and tile group.  Layers 1, 2 (-> dZ[1], dZ[0]) and 3
// (-> dX, two tiles of 32 rows; nothing becomes a plane) as in the forward.
#pragma once

template <int KIND>
__global__ void __launch_bounds__(64 * PW, 2) mlp_dgrad_pair_kernel(DSplitBatch AB)
{
    using S = PairSteps<KIND, true>;
    constexpr NetDesc D = net_desc(KIND);
    constexpr PairLayout L = S::L;
    constexpr int NL = S::NL, NHID = NL - 1, HT = D.hid_tiles, NS = S::NS;
    constexpr unsigned HBYTES = HT * 32 * 32 * 4, MBYTES = (HT / 2) * 256;
    static_assert(NL == 4 && HT == 6 && L.rowsplit[0] && L.ks[0] == 1 && L.tiles[0] == HT && L.tiles[NL - 1] == 2 &&
                  !L.rowsplit[1] && !L.rowsplit[2] && !L.rowsplit[3] && L.ks[1] == 2 * HT && D.out_dim <= 8 && D.zrows <= 8,
                  "the 192-wide four-layer net");
    DSplitSeg A = AB.seg[0];
    if (AB.nseg > 1 && (int)blockIdx.x >= AB.seg[1].b0) A = AB.seg[1];
    const int blk0 = A.b0, nblk = A.nb;
    extern __shared__ __attribute__((aligned(16))) unsigned char wl[];          // ring of step buffers | partial sums
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, s_ = lane & 31;
    __builtin_assume(tid < 64 * PW);
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pr = wv & 3, w = wv >> 2;
    const int ntiles = A.t1 - A.t0, ngroups = (ntiles + PW / 2 - 1) / (PW / 2);
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
    int rb = 0, par = 0;
    stage_load(std::integral_constant<int, 0>{});
    stage_store(std::integral_constant<int, 0>{}, wl);
    stage_load(std::integral_constant<int, 1>{});
    stage_store(std::integral_constant<int, 1>{}, wl + S::BUF);
    step_barrier();

    unsigned char *xch = wl + S::XCH0 + pr * 4 * PAIR_XCH;   // this pair's partial sums: [step parity][written by wave]
    const int hvoff = tile_voff(lane) + w * 16 * 128;        // this wave's rows of a tile-major tile: + 16 rows for B
    const int wsel = w * 2 * 1024;
    float wmax = 0.f;
    // the net's gradient gain bound (mlp.hip: split_gain_kernel, behind the one-wave planes): see mlp_dgrad_split_kernel
    const float *gainp = reinterpret_cast<const float *>(A.planes + split_gain_offset(KIND));
    const int gbits = __builtin_amdgcn_readfirstlane(__float_as_int(*gainp));
    const int kbase = __builtin_amdgcn_readfirstlane(141 + 127 - ((gbits >> 23) & 0xff) - ((gbits & 0x7fffff) ? 1 : 0));

    for (int tg = (int)blockIdx.x - blk0; tg < ngroups; tg += nblk) {
        const int tt = A.t0 + tg * (PW / 2) + pr;
        const bool live = tt < A.t1;
        const int t = live ? tt : A.t1 - 1;
        int hv = hvoff;
        asm volatile("" : "+v"(hv));
        // the tile's output gradients (both waves of the pair: the same rows), its ReLU masks, its scale
        float zn[D.zrows];
        unsigned msk[NHID][HT / 2];
        {
            const rsrc_t RZ = make_rsrc(AB.dz + (size_t)t * D.zrows * 32, D.zrows * 32 * 4);
#pragma unroll
            for (int i = 0; i < D.zrows; ++i) zn[i] = bload1(RZ, s_ * 4, i * 128);
#pragma unroll
            for (int l = 0; l < NHID; ++l) {
                load_relu_mask<HT>(make_rsrc(AB.M[l] + (size_t)t * (MBYTES / 4), MBYTES), msk[l], lane);
                // this wave's rows of a tile: bits 8 w .. 8 w + 7 of its half-word -> bits 0 .. 7
#pragma unroll
                for (int k = 0; k < HT / 2; ++k) msk[l][k] >>= 8 * w;
            }
        }
        float zmax = 0.f;
#pragma unroll
        for (int i = 0; i < D.out_dim; ++i) zmax = fmaxf(zmax, fabsf(zn[i]));
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) zmax = fmaxf(zmax, __shfl_xor(zmax, o));
        if (live && w == 0) wmax = fmaxf(wmax, zmax);
        const int ez = (__float_as_int(zmax) >> 23) & 0xff;
        const int ks = ez == 0 ? 0 : kbase - ez;
        const int kc = ks < -100 ? -100 : (ks > 100 ? 100 : ks);
        const float sc = __int_as_float((127 + kc) << 23), isc = __int_as_float((127 - kc) << 23);
        f16x8 xi1[1], xi2[1];
        {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = (h == 0 && i < D.out_dim) ? zn[i < D.out_dim ? i : 0] * sc : 0.f;
            split8(v, xi1[0], xi2[0]);
        }
        f16x8 pa1[HT], pa2[HT], pb1[HT], pb2[HT];
        f32x16 am[2];
        float ev[8], part[8];
        const float wisc = SPLIT_W_INV * isc;
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

        // micro-slices of this wave's half of a finished tile of transposed layer q (4 register pairs x 3 phases; ADD: the
        // partner's partial sums come with it)
        auto micro = [&](auto QC, auto IT, auto ADDC, auto MS, f32x16 &accm, auto &o1, auto &o2) __attribute__((always_inline)) {
            constexpr int q = decltype(QC)::value, it = decltype(IT)::value, ms = decltype(MS)::value, p = ms / 3, ph = ms % 3;
            constexpr bool ADD = decltype(ADDC)::value != 0;
            constexpr int r0 = 2 * p;
            if constexpr (ph == 0) {
                float t0 = accm[r0], t1 = accm[r0 + 1];
                if constexpr (ADD) {
                    float p0 = part[r0], p1 = part[r0 + 1];
                    asm volatile("" : "+v"(p0), "+v"(p1));                   // (pinned to the slice: see the forward)
                    t0 += p0; t1 += p1;
                }
                if constexpr (q == NL - 1) {
                    const rsrc_t RX = make_rsrc(AB.dX + (size_t)t * 64 * 32, live ? dx_rows(KIND) / 4 * 4 * 128 + (dx_rows(KIND) % 4 ? 512 : 0) : 0);
                    asm volatile("" : "+v"(hv));
                    bstore1(RX, t0 * wisc, hv + tile_soff(0, r0), it * 4096);           // (default policy: the scatter reads dX next)
                    bstore1(RX, t1 * wisc, hv + tile_soff(0, r0 + 1), it * 4096);
                } else {
                    constexpr int d = NHID - 1 - q;
                    int k0, k1;                                              // (the mask word: loaded a tile group ago, no MFMA result)
                    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(k0) : "v"(msk[d][it >> 1]), "n"((it & 1) * 16 + r0));
                    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(k1) : "v"(msk[d][it >> 1]), "n"((it & 1) * 16 + r0 + 1));
                    const int a0 = __float_as_int(t0) & k0, a1 = __float_as_int(t1) & k1;
                    const rsrc_t RD = make_rsrc(AB.dZ[d] + (size_t)t * (HBYTES / 4), (live && AB.dZ[d]) ? HBYTES : 0u);
                    asm volatile("" : "+v"(hv));
                    bstore1_nt(RD, __int_as_float(a0) * wisc, hv + tile_soff(0, r0), it * 4096);
                    bstore1_nt(RD, __int_as_float(a1) * wisc, hv + tile_soff(0, r0 + 1), it * 4096);
                    ev[r0] = __int_as_float(a0) * SPLIT_W_INV; ev[r0 + 1] = __int_as_float(a1) * SPLIT_W_INV;
                }
            } else if constexpr (q < NL - 1) {
                if constexpr (ph == 1) put_pair<r0>(o1[it], ev[r0], ev[r0 + 1]);
                else put_residual_pair<r0>(o2[it], o1[it], ev[r0], ev[r0 + 1]);
            }
        };
        auto pending = [&](auto QC, auto IT, auto ADDC, auto U, auto NAVAILC, f32x16 &accm, auto &o1, auto &o2) __attribute__((always_inline)) {
            constexpr int u = decltype(U)::value, navail = decltype(NAVAILC)::value;
            static_assert(navail >= 3 && navail % 3 == 0, "whole register pairs per pass");
            if constexpr (u < navail)
                sfor<0, (12 + navail - 1) / navail>([&](auto KC) {
                    constexpr int msi = u + decltype(KC)::value * navail;
                    if constexpr (msi < 12) micro(QC, IT, ADDC, std::integral_constant<int, msi>{}, accm, o1, o2);
                });
        };
        // (the barrier that closes a step sits inside the NEXT step, behind its first k-step: mlp_pair.h, run_step)
        auto end_step = [&]() __attribute__((always_inline)) {
            rb = rb + 1 == PRING ? 0 : rb + 1;
            par ^= 1;
        };
        constexpr auto C0 = std::integral_constant<int, 0>{};
        constexpr auto C1 = std::integral_constant<int, 1>{};
        constexpr auto C2 = std::integral_constant<int, 2>{};
        constexpr auto C3 = std::integral_constant<int, 3>{};

        // transposed layer 0: one step, all six tiles of dZ[2] over the single k-step; this wave finishes its eight registers of each
        {
            constexpr int st = 0, nxt2 = (st + 2) % NS;
            const u32x4 *mine = reinterpret_cast<const u32x4 *>(wl + rb * S::BUF + wsel * HT) + lane;      // [wave][tile][plane]
            stage_load(std::integral_constant<int, nxt2>{});
            sfor<0, HT>([&](auto ITC) {
                constexpr int it = decltype(ITC)::value;
                f32x16 &m = am[it & 1];
                const f16x8 w1 = __builtin_bit_cast(f16x8, mine[(2 * it) * 64]), w2 = __builtin_bit_cast(f16x8, mine[(2 * it + 1) * 64]);
                auto ride = [&](auto U) __attribute__((always_inline)) {
                    if constexpr (it > 0) pending(C0, std::integral_constant<int, it - 1>{}, C0, U, std::integral_constant<int, 3>{},
                                                  am[(it - 1) & 1], pa1, pa2);
                    __builtin_amdgcn_sched_barrier(0);
                };
                // (NO pins here: behind a pinned MFMA the compiler no longer provides the wait states a vector instruction needs
                //  before it reads the accumulator, and this step reads a tile's sums a few instructions after its last MFMA --
                //  a third of the tiles came out wrong, run to run different ones.  The K-split steps read a finished tile only
                //  after the exchange writes and the step barrier)
                m = mfma_h(w1, xi2[0], zero16);
                ride(std::integral_constant<int, 0>{});
                m = mfma_h(w1, xi1[0], m);
                ride(std::integral_constant<int, 1>{});
                m = mfma_h(w2, xi1[0], m);
                ride(std::integral_constant<int, 2>{});
            });
            stage_store(std::integral_constant<int, nxt2>{}, wl + ((rb + 2) % PRING) * S::BUF);
            end_step();
        }
        // one K-split step: tile `it` of transposed layer q over this wave's k-steps
        auto run_step = [&](auto QC, auto IT, auto &in1, auto &in2, auto &o1, auto &o2) __attribute__((always_inline)) {
            constexpr int q = decltype(QC)::value, it = decltype(IT)::value;
            constexpr int KW = L.kw[q], st = L.step0[q] + it;
            // the tile both waves finish during this step: the previous step's (the single-k-step layer's last: no partner sums)
            constexpr int pq = it > 0 ? q : q - 1, pit = it > 0 ? it - 1 : L.tiles[q - 1] - 1;
            constexpr bool PROWS = L.rowsplit[pq] != 0;
            constexpr int nxt2 = (st + 2) % NS;
            static_assert((pit & 1) != (it & 1), "the pending tile and the tile in flight use different accumulators");
            const u32x4 *mine = reinterpret_cast<const u32x4 *>(wl + rb * S::BUF + wsel * KW) + lane;
            stage_load(std::integral_constant<int, nxt2>{});
            constexpr int WR = 2;
            u32x4 wb[WR][2];
            wb[0][0] = mine[0];
            wb[0][1] = mine[KW * 64];
            f32x16 &m = am[it & 1];
            sfor<0, KW>([&](auto JC) {
                constexpr int j = decltype(JC)::value;
                if constexpr (j + 1 < KW) {
                    wb[(j + 1) % WR][0] = mine[(j + 1) * 64];
                    wb[(j + 1) % WR][1] = mine[(KW + j + 1) * 64];
                }
                const f16x8 w1 = __builtin_bit_cast(f16x8, wb[j % WR][0]), w2 = __builtin_bit_cast(f16x8, wb[j % WR][1]);
                auto ride = [&](auto U) __attribute__((always_inline)) {
                    constexpr auto ADDC = std::integral_constant<int, PROWS ? 0 : 1>{};
                    constexpr int u_ = decltype(U)::value, first = 3 * KW - S::PRE;
                    constexpr int JB = 1, ub = 3 * JB;               // the previous step's barrier: behind slot ub - 1
                    static_assert(first >= ub, "the ring slot is written behind the barrier");
                    if constexpr (u_ == ub - 1) {
                        step_barrier();
                        if constexpr (!PROWS) pair_recv(xch + ((par ^ 1) * 2 + (w ^ 1)) * PAIR_XCH, part, lane);
                    }
                    if constexpr (u_ >= ub) {
                        constexpr auto UU = std::integral_constant<int, u_ - ub>{};
                        if constexpr (it > 0) pending(std::integral_constant<int, pq>{}, std::integral_constant<int, pit>{}, ADDC, UU,
                                                      std::integral_constant<int, 3 * KW - ub>{}, am[pit & 1], o1, o2);
                        else pending(std::integral_constant<int, pq>{}, std::integral_constant<int, pit>{}, ADDC, UU,
                                     std::integral_constant<int, 3 * (KW - 1) - ub>{}, am[pit & 1], in1, in2);
                    }
                    if constexpr (u_ >= first)
                        stage_piece(std::integral_constant<int, nxt2>{}, std::integral_constant<int, u_ - first>{},
                                    wl + ((rb + 2) % PRING) * S::BUF);
                    __builtin_amdgcn_sched_barrier(0);
                };
                m = mfma_h(w1, in2[j], j == 0 ? zero16 : m);
                ESR_PAIR_PIN(m);                                  // (pinned to its slot: see the forward)
                ride(std::integral_constant<int, 3 * j + 0>{});
                m = mfma_h(w1, in1[j], m);
                ESR_PAIR_PIN(m);
                ride(std::integral_constant<int, 3 * j + 1>{});
                m = mfma_h(w2, in1[j], m);
                ESR_PAIR_PIN(m);
                ride(std::integral_constant<int, 3 * j + 2>{});
            });
            pair_send(xch + (par * 2 + w) * PAIR_XCH, m, lane);
            if constexpr (q == NL - 1 && it == L.tiles[q] - 1) step_barrier();      // (the group's last step is closed here)
            end_step();
            if constexpr (q == NL - 1 && it == L.tiles[q] - 1) {
                // the group's very last tile (dX rows 32..63): nobody to ride on
                pair_recv(xch + ((par ^ 1) * 2 + (w ^ 1)) * PAIR_XCH, part, lane);
                sfor<0, 4>([&](auto PC) { micro(QC, IT, C1, std::integral_constant<int, 3 * decltype(PC)::value>{}, m, o1, o2); });
            }
        };
        sfor<0, HT>([&](auto IT) { run_step(C1, IT, pa1, pa2, pb1, pb2); });                 // -> dZ[1]
        sfor<0, HT>([&](auto IT) { run_step(C2, IT, pb1, pb2, pa1, pa2); });                 // -> dZ[0]
        sfor<0, 2>([&](auto IT) { run_step(C3, IT, pa1, pa2, pb1, pb2); });                  // -> dX (pb unused)
    }
    // (what the weight-gradient kernels scale by: max |dz| x max(1, G / 16): see mlp_dgrad_split_kernel)
    wmax *= fmaxf(1.f, *gainp * 0.0625f);
    if (AB.amax && lane == 0 && w == 0 && wmax > *reinterpret_cast<volatile float *>(AB.amax))
        atomicMax(reinterpret_cast<unsigned *>(AB.amax), __float_as_uint(wmax));
}
