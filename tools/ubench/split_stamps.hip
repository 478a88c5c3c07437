// Where does a wave of the split-fp16 forward (csrc/mlp_split.hip) spend its time?  The PRODUCT kernel with s_memtime stamps of
// wave 0 of every workgroup: stamp 0 = inputs split, then per step s (10 per tile group): 1+3s MFMA block done, 2+3s next
// step's weights written to LDS (since the pieces ride in the last tile's slots: nothing), 3+3s barrier passed.
//   gpurun -- './tools/ubench/split_stamps'
// Timing variants (wrong results, same instruction stream minus one ingredient): build with -DESR_SPLIT_NO_MFMA (everything
// but the MFMAs: 3.4 k of a hidden step's 4.0 k clocks -- the step is bound by the in-order issue of its ~450 vector /
// scalar instructions, 48 LDS reads and 32 stores, not by the matrix pipe's 2.3 k), -DESR_SPLIT_NO_HSTORE (-0.35 k per
// step), -DESR_SPLIT_NO_WREAD (no weight reads from LDS: -0.8 k per step).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
constexpr int NS_ = 31;
__device__ unsigned long long g_stamps[256 * 4 * NS_];
#define ESR_SPLIT_STAMP(i)                                                                                        \
    do {                                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        if (tid == 0) {                                                                                           \
            const int trip_ = (tg - ((int)blockIdx.x - blk0)) / nblk;                                             \
            if (trip_ < 4) g_stamps[(blockIdx.x * 4 + trip_) * NS_ + (i)] = __builtin_amdgcn_s_memtime();         \
        }                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)
#include "../../esr_nerf_amd/csrc/mlp_split.hip"
#include "../../esr_nerf_amd/csrc/mlp.hip"

int main()
{
    const int T = 8192, kind = ESR_MLP_RADIANCE;
    const int64_t np = esr_mlp_packed_floats(kind), ns = esr_mlp_packed_split_elems(kind);
    float *packed, *X, *H[3], *z; uint32_t *M[3]; void *pl;
    (void)hipMalloc(&packed, np * 4); (void)hipMalloc(&pl, ns * 2);
    std::vector<float> h(np);
    for (auto &v : h) v = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
    (void)hipMemcpy(packed, h.data(), np * 4, hipMemcpyHostToDevice);
    (void)hipMemset(pl, 0x2c, ns * 2);
    (void)hipMalloc(&X, (size_t)T * 104 * 32 * 4);
    std::vector<float> hx((size_t)T * 104 * 32);
    for (auto &v : hx) v = rand() / (float)RAND_MAX - 0.5f;
    (void)hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    for (int l = 0; l < 3; ++l) { (void)hipMalloc(&H[l], (size_t)T * 192 * 32 * 4); (void)hipMalloc(&M[l], (size_t)T * 3 * 64 * 4); }
    (void)hipMalloc(&z, (size_t)T * 4 * 32 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int save = 1; save >= 0; --save) {
        for (int rep = 0; rep < 2; ++rep) esr_mlp_fwd_split(kind, packed, pl, X, 0, T, H, M, save, 0, z, nullptr);
        (void)hipEventRecord(e0);
        esr_mlp_fwd_split(kind, packed, pl, X, 0, T, H, M, save, 0, z, nullptr);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> st(256 * 4 * NS_);
        (void)hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * 8);
        printf("save=%d: %.3f ms for %d tiles (f32 kernel for comparison below); median over 256 workgroups x 3 groups, counter ticks\n", save, ms, T);
        auto med = [&](int a, int b) {
            std::vector<long long> d;
            for (int wg = 0; wg < 256; ++wg)
                for (int trip = 0; trip < 3; ++trip) {
                    const unsigned long long *s = &st[(wg * 4 + trip) * NS_];
                    d.push_back((long long)(s[b] - s[a]));
                }
            std::sort(d.begin(), d.end());
            return d[d.size() / 2];
        };
        for (int s = 0; s < 10; ++s)
            printf("  step %d: MFMA block %6lld   stage_store %5lld   barrier %5lld\n", s, med(3 * s, 1 + 3 * s), med(1 + 3 * s, 2 + 3 * s),
                   med(2 + 3 * s, 3 + 3 * s));
        printf("  group total (stamp 0 -> last barrier) %lld\n", med(0, 30));
        (void)hipEventRecord(e0);
        esr_mlp_fwd(kind, packed, X, 0, T, H, M, save, 0, z, nullptr);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("  f32 MFMA kernel, same tiles: %.3f ms\n", ms);
    }
    return 0;
}
