// Where does a wave of the recomputing tone-mapper weight-gradient kernel spend its time?  Builds the PRODUCT kernel
// (esr_nerf_amd/csrc/tone_wgrad.hip included as is) with ESR_TSTAMP recording s_memtime at the section seams of the first
// wave of every workgroup, runs it on C2-sized synthetic tiles (16 384) and prints the median segment lengths.
//   stamps: 0 tile start | 1 operands copied, next tile requested, X / dzt staged in LDS | 2 Ht^T MFMAs done |
//           3 ReLU + dW1 sums done | 4 dHt MFMAs, masks, db0 / column-32 sums done | 5 dW0 MFMAs issued (tile end)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
__device__ unsigned long long g_stamps[256 * 16 * 8];     // [workgroup][trip][stamp]
#define ESR_TSTAMP(i)                                                                                             \
    do {                                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        if (tid == 0) {                                                                                           \
            const int trip_ = (t - A.t0 - pair) / npairs;                                                         \
            if (trip_ < 16) g_stamps[(blockIdx.x * 16 + trip_) * 8 + (i)] = __builtin_amdgcn_s_memtime();         \
        }                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)
#include "../../esr_nerf_amd/csrc/tone_wgrad.hip"

int main()
{
    const int T = 16384;
    float *Xt, *dzt, *W0, *b0, *W1, *g[4], *scratch;
    std::vector<float> h((size_t)T * 48 * 32);
    for (auto &v : h) v = rand() / (float)RAND_MAX - 0.5f;
    hipMalloc(&Xt, h.size() * 4); hipMemcpy(Xt, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&dzt, (size_t)T * 4 * 32 * 4); hipMemcpy(dzt, h.data(), (size_t)T * 4 * 32 * 4, hipMemcpyHostToDevice);
    hipMalloc(&W0, 192 * 33 * 4); hipMemcpy(W0, h.data(), 192 * 33 * 4, hipMemcpyHostToDevice);
    hipMalloc(&b0, 192 * 4); hipMemcpy(b0, h.data(), 192 * 4, hipMemcpyHostToDevice);
    hipMalloc(&W1, 3 * 192 * 4); hipMemcpy(W1, h.data(), 3 * 192 * 4, hipMemcpyHostToDevice);
    const size_t sizes[4] = {192 * 33, 192, 3 * 192, 4};
    for (int i = 0; i < 4; ++i) { hipMalloc(&g[i], sizes[i] * 4); hipMemset(g[i], 0, sizes[i] * 4); }
    const int64_t ns = esr_tone_wgrad_scratch_floats();
    hipMalloc(&scratch, ns * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) esr_tone_wgrad_recompute(Xt, dzt, W0, b0, W1, 0, T, g[0], g[1], g[2], g[3], scratch, ns, nullptr);
    hipEventRecord(e0);
    for (int rep = 0; rep < 10; ++rep) esr_tone_wgrad_recompute(Xt, dzt, W0, b0, W1, 0, T, g[0], g[1], g[2], g[3], scratch, ns, nullptr);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%.1f us per call (kernel + reduce, with the stamps in)\n", ms * 100.f);
    std::vector<unsigned long long> st(256 * 16 * 8);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * 8);
    const char *names[6] = {"copy operands, request next tile, stage X / dzt in LDS", "Ht^T: 51 MFMAs (+ B operands from LDS)",
                            "ReLU + dW1 per-lane sums", "dHt MFMAs + masks + db0 / column sums", "dW0: 48 MFMAs (+ Xt rows from LDS)",
                            "tile total (stamp 0 -> 5)"};
    for (int seg = 0; seg < 6; ++seg) {
        std::vector<long long> d;
        for (int wg = 0; wg < 256; ++wg)
            for (int trip = 1; trip < 8; ++trip) {
                const unsigned long long *s = &st[(wg * 16 + trip) * 8];
                d.push_back(seg < 5 ? (long long)(s[seg + 1] - s[seg]) : (long long)(s[5] - s[0]));
            }
        std::sort(d.begin(), d.end());
        printf("  %-58s %8lld   (p10 %lld, p90 %lld)\n", names[seg], d[d.size() / 2], d[d.size() / 10], d[d.size() * 9 / 10]);
    }
    // gap between consecutive tiles of a wave (stamp 5 of trip k -> stamp 0 of trip k + 1): loop overhead
    std::vector<long long> gaps;
    for (int wg = 0; wg < 256; ++wg)
        for (int trip = 1; trip < 7; ++trip) gaps.push_back((long long)(st[(wg * 16 + trip + 1) * 8] - st[(wg * 16 + trip) * 8 + 5]));
    std::sort(gaps.begin(), gaps.end());
    printf("  %-58s %8lld\n", "between tiles", gaps[gaps.size() / 2]);
    return 0;
}
