// The N-split radiance kernels (csrc/mlp_nsplit.h: two waves per SIMD, a pair of waves splits a tile's OUTPUT tiles) beside
// the one-wave-per-SIMD split kernels: same packed weights, same inputs; differences of outputs / saved tiles / masks (the
// two sum the same products in the same order: expected 0) and launch times at C2's shape (8192 on-tiles + 8192 off-tiles).
//   gpurun -- './tools/ubench/nsplit_bench'
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <vector>
// stamps of waves 0 (w = 0) and 4 (w = 1) of every workgroup, first three tile groups: [wg][w][trip][unit][4]
constexpr int NST_ = 17;
__device__ unsigned long long g_stamps[256 * 2 * 3 * NST_ * 4];
#define ESR_NS_STAMP(u_, i_)                                                                                      \
    do {                                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        if (lane == 0 && pr == 0 && trip < 3 && blockIdx.x < 256 && (u_) < NST_)                                  \
        {                                                                                                         \
            g_stamps[(((blockIdx.x * 2 + w) * 3 + trip) * NST_ + (u_)) * 4 + (i_)] = __builtin_amdgcn_s_memtime(); \
            if ((i_) == 0) g_stamps[(((blockIdx.x * 2 + w) * 3 + trip) * NST_ + (u_)) * 4 + 3] = __builtin_amdgcn_s_memrealtime(); \
        }                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)
__device__ unsigned long long g_stamps2[256 * 2 * 3 * NST_ * 2];
#define ESR_NS_STAMP2(u_, i_)                                                                                     \
    do {                                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        if (lane == 0 && pr == 0 && trip < 3 && blockIdx.x < 256 && (u_) < NST_)                                  \
            g_stamps2[(((blockIdx.x * 2 + w) * 3 + trip) * NST_ + (u_)) * 2 + (i_)] = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)
// the product's one-wave forward: shader ticks and wall ticks at the start of each tile group (wave 0), for its clock
__device__ unsigned long long g_ow[256 * 4 * 2];
#define ESR_SPLIT_STAMP(i)                                                                                        \
    do {                                                                                                          \
        if ((i) == 0 && tid == 0 && blockIdx.x < 256) {                                                           \
            const int trip_ = (tg - ((int)blockIdx.x - blk0)) / nblk;                                             \
            if (trip_ < 4) {                                                                                      \
                g_ow[(blockIdx.x * 4 + trip_) * 2] = __builtin_amdgcn_s_memtime();                                \
                g_ow[(blockIdx.x * 4 + trip_) * 2 + 1] = __builtin_amdgcn_s_memrealtime();                        \
            }                                                                                                     \
        }                                                                                                         \
    } while (0)
#include "../../esr_nerf_amd/csrc/mlp_split.hip"
#include "../../esr_nerf_amd/csrc/mlp.hip"
#include "nsplit_proto.h"
// the prototype's launches: esr_mlp_fwd_split / esr_mlp_fwd_fine_split / esr_mlp_dgrad_fine_split on the N-split kernels
static int g_variant = 0;
static int esr_mlp_split_variant(int v) { const int p = g_variant; g_variant = v; return p; }
static const _Float16 *g_nsplanes = nullptr;               // N-split forward | transposed | gain
static int ns_launch_fwd(SplitBatch &B)
{
    using P = NsSteps<ESR_MLP_RADIANCE, false>;
    for (int k = 0; k < B.nseg; ++k) B.seg[k].planes = g_nsplanes;         // (one net in this harness)
    const int grid = share_blocks_split(B.seg, B.nseg, 256);
    static std::atomic<uint64_t> optin{0};
    if (int rc = esr_lds_optin(reinterpret_cast<const void *>(&mlp_fwd_ns_kernel<ESR_MLP_RADIANCE>), P::LDS_BYTES, optin)) return rc;
    mlp_fwd_ns_kernel<ESR_MLP_RADIANCE><<<grid, 64 * NW, P::LDS_BYTES, 0>>>(B);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
static int fwd_split(int kind, const float *packed32, const void *planes, const float *X, int32_t t0, int32_t t1, float *const *H,
                     uint32_t *const *M, int save, int crow, float *zout)
{
    if (!g_variant) return esr_mlp_fwd_split(kind, packed32, planes, X, t0, t1, H, M, save, crow, zout, nullptr);
    SplitBatch B = {};
    B.X = X;
    for (int l = 0; l < 3; ++l) { B.H[l] = save == 1 ? H[l] : nullptr; B.M[l] = M[l]; }
    B.nseg = 1;
    B.seg[0] = SplitSeg{packed32, nullptr, t0, t1, save == 2 ? 2 : save ? 1 : 0, crow, zout, 0, 0};
    return ns_launch_fwd(B);
}
static int fwd_fine_split(const float *p32, const void *planes, const float *X, int32_t t_on, int32_t t_all, float *const *H,
                          uint32_t *const *M, int crow, float *z_off, float *z_emo)
{
    if (!g_variant) return esr_mlp_fwd_fine_split(p32, planes, p32, planes, X, t_on, t_all, H, M, crow, z_off, z_emo, nullptr);
    SplitBatch B = {};
    B.X = X;
    for (int l = 0; l < 3; ++l) { B.H[l] = H[l]; B.M[l] = M[l]; }
    int n = 0;
    if (t_on > 0) B.seg[n++] = SplitSeg{p32, nullptr, 0, t_on, 0, crow, z_off, 0, 0};
    if (t_all > t_on) B.seg[n++] = SplitSeg{p32, nullptr, t_on, t_all, 1, 0, z_off, 0, 0};
    if (t_on > 0) B.seg[n++] = SplitSeg{p32, nullptr, 0, t_on, 1, 0, z_emo, 0, 0};
    B.nseg = n;
    return ns_launch_fwd(B);
}
static int dgrad_fine_split(const void *planes, const float *dz, int32_t t_on, int32_t t_all, const uint32_t *const *M, float *const *dZ,
                            float *dX, float *amax)
{
    if (!g_variant) return esr_mlp_dgrad_fine_split(planes, planes, dz, t_on, t_all, M, dZ, dX, amax, nullptr);
    using P = NsSteps<ESR_MLP_RADIANCE, true>;
    DSplitBatch B = {};
    B.dz = dz; B.dX = dX; B.amax = amax;
    for (int l = 0; l < 3; ++l) { B.M[l] = M[l]; B.dZ[l] = dZ[l]; }
    int n = 0;
    if (t_on > 0) B.seg[n++] = DSplitSeg{g_nsplanes, 0, t_on, 0, 0};
    if (t_all > t_on) B.seg[n++] = DSplitSeg{g_nsplanes, t_on, t_all, 0, 0};
    B.nseg = n;
    int groups[2], total = 0;
    for (int k = 0; k < n; ++k) { groups[k] = (B.seg[k].t1 - B.seg[k].t0 + 3) / 4; total += groups[k]; }
    const int grid = total < 256 ? total : 256;
    if (n == 1) { B.seg[0].b0 = 0; B.seg[0].nb = grid; }
    else {
        int n0 = (int)((int64_t)grid * groups[0] / total);
        n0 = n0 < 1 ? 1 : n0;
        B.seg[0].b0 = 0; B.seg[0].nb = n0; B.seg[1].b0 = n0; B.seg[1].nb = grid - n0;
    }
    static std::atomic<uint64_t> optin{0};
    if (int rc = esr_lds_optin(reinterpret_cast<const void *>(&mlp_dgrad_ns_kernel<ESR_MLP_RADIANCE>), P::LDS_BYTES, optin)) return rc;
    mlp_dgrad_ns_kernel<ESR_MLP_RADIANCE><<<grid, 64 * NW, P::LDS_BYTES, 0>>>(B);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}


template <typename T> static T *dalloc(size_t n) { T *p; (void)hipMalloc(&p, n * sizeof(T)); return p; }
static double maxrel(const std::vector<float> &a, const std::vector<float> &b)
{
    double d = 0, m = 0;
    for (size_t i = 0; i < a.size(); ++i) { d = std::max(d, (double)std::fabs(a[i] - b[i])); m = std::max(m, (double)std::fabs(b[i])); }
    return d / (m + 1e-300);
}
int main(int argc, char **argv)
{
    const int T = argc > 1 ? atoi(argv[1]) : 16384, Ton = T / 2, kind = ESR_MLP_RADIANCE;
    const int dims[5] = {85, 192, 192, 192, 3};
    srand(1);
    esr_mlp_weights_t w = {};
    std::vector<std::vector<float>> hw(4), hb(4);
    for (int l = 0; l < 4; ++l) {
        hw[l].resize((size_t)dims[l + 1] * dims[l]); hb[l].resize(dims[l + 1]);
        const float sc = 1.7f / std::sqrt((float)dims[l]);
        for (auto &v : hw[l]) v = (rand() / (float)RAND_MAX - 0.5f) * 2.f * sc;
        for (auto &v : hb[l]) v = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
        float *dw = dalloc<float>(hw[l].size()), *db = dalloc<float>(hb[l].size());
        (void)hipMemcpy(dw, hw[l].data(), hw[l].size() * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(db, hb[l].data(), hb[l].size() * 4, hipMemcpyHostToDevice);
        w.w[l] = dw; w.b[l] = db;
    }
    float *packed = dalloc<float>(esr_mlp_packed_floats(kind));
    _Float16 *planes = dalloc<_Float16>(esr_mlp_packed_split_elems(kind));
    const int32_t kinds[1] = {kind};
    const esr_mlp_weights_t *ws[1] = {&w};
    float *p32[1] = {packed};
    void *psp[1] = {planes};
    if (esr_mlp_pack_batch(1, kinds, ws, p32, nullptr, psp, nullptr)) { printf("pack failed\n"); return 1; }
    {
        _Float16 *nsp = dalloc<_Float16>(ns_elems(kind) + 8);
        NsPackArgs P = {};
        for (int l = 0; l < 4; ++l) P.w[l] = w.w[l];
        P.outs = nsp;
        ns_pack_kernel<ESR_MLP_RADIANCE><<<256, 256>>>(P);
        (void)hipMemcpy(nsp + ns_elems(kind), planes + esr_mlp_split_gain_offset(kind), 4, hipMemcpyDeviceToDevice);
        (void)hipDeviceSynchronize();
        g_nsplanes = nsp;
    }
    std::vector<float> hx((size_t)T * 104 * 32);
    for (auto &v : hx) v = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
    float *X = dalloc<float>(hx.size());
    (void)hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    float *H[2][3], *z[2], *ze[2]; uint32_t *M[2][3];
    for (int v = 0; v < 2; ++v) {
        for (int l = 0; l < 3; ++l) { H[v][l] = dalloc<float>((size_t)T * 192 * 32); M[v][l] = dalloc<uint32_t>((size_t)T * 3 * 64);
                                      (void)hipMemset(H[v][l], 0xff, (size_t)T * 192 * 32 * 4); (void)hipMemset(M[v][l], 0xff, (size_t)T * 3 * 64 * 4); }
        z[v] = dalloc<float>((size_t)T * 4 * 32); ze[v] = dalloc<float>((size_t)T * 4 * 32);
    }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int v = 0; v < 2; ++v) {
        esr_mlp_split_variant(v);
        // single pass, ragged range, colour rows 88
        if (int rc = fwd_split(kind, packed, planes, X, 3, std::min(T, 1003), H[v], M[v], 1, 88, z[v])) { printf("rc %d\n", rc); return 1; }
        (void)hipDeviceSynchronize();
        hipError_t err = hipGetLastError();
        if (err != hipSuccess) { printf("variant %d: %s\n", v, hipGetErrorString(err)); return 1; }
    }
    {
        const size_t nz = (size_t)std::min(T, 1003) * 128, nh = (size_t)std::min(T, 1003) * 192 * 32, nm = (size_t)std::min(T, 1003) * 192;
        std::vector<float> a(nz), b(nz);
        (void)hipMemcpy(a.data(), z[1], nz * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(b.data(), z[0], nz * 4, hipMemcpyDeviceToHost);
        printf("single pass [3, 1003): z N-split vs one-wave %.2e\n", maxrel(std::vector<float>(a.begin() + 3 * 128, a.end()), std::vector<float>(b.begin() + 3 * 128, b.end())));
        for (int l = 0; l < 3; ++l) {
            std::vector<float> ha(nh), hb_(nh);
            (void)hipMemcpy(ha.data(), H[1][l], nh * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(hb_.data(), H[0][l], nh * 4, hipMemcpyDeviceToHost);
            std::vector<uint32_t> ma(nm), mb(nm);
            (void)hipMemcpy(ma.data(), M[1][l], nm * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(mb.data(), M[0][l], nm * 4, hipMemcpyDeviceToHost);
            size_t nd = 0, untouched_ok = 1;
            for (size_t i = 3 * 192; i < nm; ++i) nd += __builtin_popcount(ma[i] ^ mb[i]);
            for (size_t i = 0; i < 3 * 192; ++i) untouched_ok &= (ma[i] == 0xffffffffu);
            printf("  H[%d] %.2e   mask bits that differ %zu of %zu   tiles before t0 untouched: %s\n", l,
                   maxrel(std::vector<float>(ha.begin() + 3 * 6144, ha.end()), std::vector<float>(hb_.begin() + 3 * 6144, hb_.end())), nd, (nm - 576) * 32,
                   untouched_ok ? "yes" : "NO");
        }
    }
    for (int round = 0; round < 3; ++round)           // (alternating, after a long warm-up: the clocks ramp for ~100 ms)
    for (int v = 0; v < 2; ++v) {
        esr_mlp_split_variant(v);
        for (int rep = 0; rep < (round ? 20 : 150); ++rep)
            fwd_fine_split(packed, planes, X, Ton, T, H[v], M[v], 88, z[v], ze[v]);
        (void)hipEventRecord(e0);
        for (int rep = 0; rep < 20; ++rep)
            fwd_fine_split(packed, planes, X, Ton, T, H[v], M[v], 88, z[v], ze[v]);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("merged fine launch (%d on + %d off tiles), variant %d (%s): %.3f ms\n", Ton, T - Ton, v, v ? "N-split, 2 waves/SIMD" : "1 wave/SIMD", ms / 20);
    }
    {
        std::vector<unsigned long long> st(256 * 2 * 3 * NST_ * 4);
        (void)hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * 8);
        auto med = [&](int w_, int s_, int a, int b, int s2 = -1) {
            std::vector<long long> d;
            for (int wg = 0; wg < 256; ++wg)
                for (int trip = 1; trip < 3; ++trip) {
                    const unsigned long long *p = &st[((wg * 2 + w_) * 3 + trip) * NST_ * 4];
                    d.push_back((long long)(p[(s2 < 0 ? s_ : s2) * 4 + b] - p[s_ * 4 + a]));
                }
            std::sort(d.begin(), d.end());
            return d[d.size() / 2];
        };
        std::vector<unsigned long long> st2(256 * 2 * 3 * NST_ * 2);
        (void)hipMemcpyFromSymbol(st2.data(), HIP_SYMBOL(g_stamps2), st2.size() * 8);
        // unit start -> second k-step | second -> last k-step | last k-step -> unit end (before the barrier)
        auto med2 = [&](int w_, int s_, int which) {
            std::vector<long long> d;
            for (int wg = 0; wg < 256; ++wg)
                for (int trip = 1; trip < 3; ++trip) {
                    const unsigned long long *p = &st[((wg * 2 + w_) * 3 + trip) * NST_ * 4 + s_ * 4];
                    const unsigned long long *q = &st2[((wg * 2 + w_) * 3 + trip) * NST_ * 2 + s_ * 2];
                    d.push_back(which == 0 ? (long long)(q[0] - p[0]) : which == 1 ? (long long)(q[1] - q[0]) : (long long)(p[1] - q[1]));
                }
            std::sort(d.begin(), d.end());
            return d[d.size() / 2];
        };
        printf("inside a unit (start -> k-step 1 | k-step 1 -> last k-step | last k-step -> end):\n");
        for (int s_ = 0; s_ < NST_; ++s_)
            printf("  unit %2d: w0 %5lld %5lld %5lld | w1 %5lld %5lld %5lld\n", s_, med2(0, s_, 0), med2(0, s_, 1), med2(0, s_, 2), med2(1, s_, 0), med2(1, s_, 1), med2(1, s_, 2));
        printf("stamps (s_memtime ticks, median over workgroups x 2 groups): unit: w0 work / barrier | w1 work / barrier\n");
        for (int s_ = 0; s_ < NST_; ++s_)
            printf("  unit %2d: w0 %5lld %5lld | w1 %5lld %5lld\n", s_, med(0, s_, 0, 1), med(0, s_, 1, 2), med(1, s_, 0, 1), med(1, s_, 1, 2));
        printf("  group total: w0 %lld  w1 %lld ticks\n", med(0, 0, 0, 2, NST_ - 1), med(1, 0, 0, 2, NST_ - 1));
        {
            std::vector<unsigned long long> ow(256 * 4 * 2);
            (void)hipMemcpyFromSymbol(ow.data(), HIP_SYMBOL(g_ow), ow.size() * 8);
            std::vector<long long> dt, dr;
            for (int wg = 0; wg < 256; ++wg)
                for (int trip = 1; trip < 3; ++trip) {
                    dt.push_back((long long)(ow[(wg * 4 + trip + 1) * 2] - ow[(wg * 4 + trip) * 2]));
                    dr.push_back((long long)(ow[(wg * 4 + trip + 1) * 2 + 1] - ow[(wg * 4 + trip) * 2 + 1]));
                }
            std::sort(dt.begin(), dt.end()); std::sort(dr.begin(), dr.end());
            printf("  one-wave kernel, one tile group: %lld ticks in %lld x 10 ns: %.0f MHz\n", dt[dt.size() / 2], dr[dr.size() / 2],
                   100.0 * dt[dt.size() / 2] / dr[dr.size() / 2]);
        }
        printf("  units 0 -> 16 start: %lld ticks in %lld x 10 ns: %.0f MHz\n", med(0, 0, 0, 0, NST_ - 1), med(0, 0, 3, 3, NST_ - 1),
               100.0 * med(0, 0, 0, 0, NST_ - 1) / med(0, 0, 3, 3, NST_ - 1));
    }
    {
        const size_t nz = (size_t)T * 128;
        std::vector<float> a(nz), b(nz);
        (void)hipMemcpy(a.data(), z[1], nz * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(b.data(), z[0], nz * 4, hipMemcpyDeviceToHost);
        printf("merged: z_off N-split vs one-wave %.2e\n", maxrel(a, b));
        (void)hipMemcpy(a.data(), ze[1], nz / 2 * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(b.data(), ze[0], nz / 2 * 4, hipMemcpyDeviceToHost);
        a.resize(nz / 2); b.resize(nz / 2);
        printf("merged: z_emo N-split vs one-wave %.2e\n", maxrel(a, b));
    }
    // ---- the input-gradient chain: both variants on the masks the forward left, per-sample gradients over five decades
    {
        std::vector<float> hz((size_t)T * 128);
        for (size_t i = 0; i < hz.size(); ++i) {
            const size_t smp = i % 32 + (i / 128) * 32;
            const float mag = std::pow(10.f, -5.f * ((smp * 2654435761u) % 1000) / 1000.f) * 1e-3f;
            hz[i] = ((i / 32) % 4 == 3) ? 0.f : (rand() / (float)RAND_MAX - 0.5f) * mag;
        }
        float *dz = dalloc<float>(hz.size());
        (void)hipMemcpy(dz, hz.data(), hz.size() * 4, hipMemcpyHostToDevice);
        float *dZ[2][3], *dX[2], *amax = dalloc<float>(2);
        for (int v = 0; v < 2; ++v) {
            for (int l = 0; l < 3; ++l) { dZ[v][l] = dalloc<float>((size_t)T * 192 * 32); (void)hipMemset(dZ[v][l], 0xff, (size_t)T * 192 * 32 * 4); }
            dX[v] = dalloc<float>((size_t)T * 64 * 32); (void)hipMemset(dX[v], 0, (size_t)T * 64 * 32 * 4);
        }
        (void)hipMemset(amax, 0, 8);
        for (int v = 0; v < 2; ++v) {
            esr_mlp_split_variant(v);
            if (int rc = dgrad_fine_split(planes, dz, Ton, T, M[0], dZ[v], dX[v], amax + v)) { printf("dgrad rc %d\n", rc); return 1; }
            (void)hipDeviceSynchronize();
            hipError_t err = hipGetLastError();
            if (err != hipSuccess) { printf("dgrad variant %d: %s\n", v, hipGetErrorString(err)); return 1; }
        }
        float ha[2];
        (void)hipMemcpy(ha, amax, 8, hipMemcpyDeviceToHost);
        printf("dgrad: amax one-wave %.6e N-split %.6e\n", ha[0], ha[1]);
        auto tile_err = [&](float *a_, float *b_, size_t rows, size_t use_rows) {       // worst per-tile max-norm error
            std::vector<float> a((size_t)T * rows * 32), b(a.size());
            (void)hipMemcpy(a.data(), a_, a.size() * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(b.data(), b_, b.size() * 4, hipMemcpyDeviceToHost);
            double worst = 0;
            for (int t = 0; t < T; ++t) {
                double d = 0, m = 0;
                for (size_t i = 0; i < use_rows * 32; ++i) {
                    const size_t k = (size_t)t * rows * 32 + i;
                    d = std::max(d, (double)std::fabs(a[k] - b[k])); m = std::max(m, (double)std::fabs(b[k]));
                }
                worst = std::max(worst, d / (m + 1e-300));
            }
            return worst;
        };
        for (int l = 0; l < 3; ++l) printf("  dZ[%d] N-split vs one-wave (worst tile, max-norm) %.2e\n", l, tile_err(dZ[1][l], dZ[0][l], 192, 192));
        {
            std::vector<float> a((size_t)T * 6144), b(a.size());
            (void)hipMemcpy(a.data(), dZ[1][1], a.size() * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(b.data(), dZ[0][1], b.size() * 4, hipMemcpyDeviceToHost);
            int nbad = 0;
            printf("  bad tiles of dZ[1] (tile: first bad 16-row block):");
            for (int t = 0; t < T; ++t) {
                int badblk = -1;
                for (int blk = 0; blk < 12 && badblk < 0; ++blk) {
                    double d = 0, m = 0;
                    for (int i = 0; i < 512; ++i) { const size_t k = (size_t)t * 6144 + blk * 512 + i; d = std::max(d, (double)std::fabs(a[k] - b[k])); m = std::max(m, (double)std::fabs(b[k])); }
                    if (d > 1e-4 * m + 1e-30) badblk = blk;
                }
                if (badblk >= 0) {
                    if (nbad < 3) {
                        printf("\n    tile %d:", t);
                        for (int blk = 0; blk < 12; ++blk) {
                            double d = 0, m = 0;
                            for (int i = 0; i < 512; ++i) { const size_t k = (size_t)t * 6144 + blk * 512 + i; d = std::max(d, (double)std::fabs(a[k] - b[k])); m = std::max(m, (double)std::fabs(b[k])); }
                            printf(" %.1e", d / (m + 1e-300));
                        }
                        printf("\n      pair rows 0..5 sample 0..1:"); for (int r = 0; r < 6; ++r) printf(" % .3e % .3e |", a[(size_t)t * 6144 + r * 32], a[(size_t)t * 6144 + r * 32 + 1]);
                        printf("\n      ref  rows 0..5 sample 0..1:"); for (int r = 0; r < 6; ++r) printf(" % .3e % .3e |", b[(size_t)t * 6144 + r * 32], b[(size_t)t * 6144 + r * 32 + 1]);
                        printf("\n");
                    } else if (nbad < 30) printf(" %d:%d", t, badblk);
                    ++nbad;
                }
            }
            printf("  ... %d of %d\n", nbad, T);
        }
        if (argc > 2) {      // diagnostics: sample tile 5, per output tile and half of dZ[1] and dZ[0]
            for (int l = 1; l >= 0; --l) {
                std::vector<float> a(192 * 32), b(192 * 32);
                (void)hipMemcpy(a.data(), dZ[1][l] + (size_t)5 * 6144, 6144 * 4, hipMemcpyDeviceToHost);
                (void)hipMemcpy(b.data(), dZ[0][l] + (size_t)5 * 6144, 6144 * 4, hipMemcpyDeviceToHost);
                printf("    dZ[%d] sample 0, rows 0..7 : pair", l); for (int r = 0; r < 8; ++r) printf(" % .3e", a[r * 32]); printf("\n");
                printf("    dZ[%d] sample 0, rows 0..7 : ref ", l); for (int r = 0; r < 8; ++r) printf(" % .3e", b[r * 32]); printf("\n");
                printf("    dZ[%d] sample 0, rows 16..23: pair", l); for (int r = 16; r < 24; ++r) printf(" % .3e", a[r * 32]); printf("\n");
                printf("    dZ[%d] sample 0, rows 16..23: ref ", l); for (int r = 16; r < 24; ++r) printf(" % .3e", b[r * 32]); printf("\n");
                for (int hf = 0; hf < 2; ++hf) {
                    double d = 0, m = 0, ma = 0;
                    for (int i = 0; i < 16 * 32; ++i) { d = std::max(d, (double)std::fabs(a[hf * 512 + i] - b[hf * 512 + i])); m = std::max(m, (double)std::fabs(b[hf * 512 + i])); ma = std::max(ma, (double)std::fabs(a[hf * 512 + i])); }
                    printf("    dZ[%d] tile %d rows %2d..: diff %.2e ref max %.2e pair max %.2e\n", l, hf / 2, (hf & 1) * 16, d, m, ma);
                }
            }
        }
        printf("  dX    N-split vs one-wave (rows 0..43)               %.2e\n", tile_err(dX[1], dX[0], 64, 44));
        for (int round = 0; round < 2; ++round)
        for (int v = 0; v < 2; ++v) {
            esr_mlp_split_variant(v);
            for (int rep = 0; rep < 20; ++rep) dgrad_fine_split(planes, dz, Ton, T, M[0], dZ[v], dX[v], amax + v);
            (void)hipEventRecord(e0);
            for (int rep = 0; rep < 20; ++rep) dgrad_fine_split(planes, dz, Ton, T, M[0], dZ[v], dX[v], amax + v);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("merged dgrad launch (%d tiles), variant %d: %.3f ms\n", T, v, ms / 20);
        }
    }
    return 0;
}
