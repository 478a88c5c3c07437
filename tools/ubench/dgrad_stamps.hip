// Where does an input-gradient wave spend its time?  The PRODUCT kernel (esr_nerf_amd/csrc/mlp.hip included as is) with
// ESR_DSTAMP recording s_memtime at the layer seams of the first wave of every workgroup, on C2-sized synthetic tiles.
//   stamps: 0 tile start | 1 output layer^T | 2 mask + dZ2 stores | 3 layer-3^T MFMAs | 4 epilogue | 5 layer-2^T MFMAs |
//           6 epilogue | 7 dX rows 0-31 (32x32 tile) | 8 its stores | 9 dX rows 32-43 (4x4x1 passes) | 10 their stores
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
constexpr int NS = 12;
__device__ unsigned long long g_stamps[512 * 16 * NS];     // [workgroup][trip][stamp]
#define ESR_DSTAMP(i)                                                                                             \
    do {                                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        if ((threadIdx.x >> 6) == 0 && lane == 0) {                                                               \
            const int trip_ = (t - A.t0 - wave) / nwaves;                                                         \
            if (trip_ < 16) g_stamps[(blockIdx.x * 16 + trip_) * NS + (i)] = __builtin_amdgcn_s_memtime();        \
        }                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)
#include "../../esr_nerf_amd/csrc/mlp.hip"

int main()
{
    const int T = 8192, kind = ESR_MLP_RADIANCE;
    const int64_t np = esr_mlp_packed_floats(kind);
    float *packed, *dz, *dZ[3], *dX; uint32_t *M[3];
    (void)hipMalloc(&packed, np * 4);
    std::vector<float> h(np);
    for (auto &v : h) v = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
    (void)hipMemcpy(packed, h.data(), np * 4, hipMemcpyHostToDevice);
    (void)hipMalloc(&dz, (size_t)T * 4 * 32 * 4);
    std::vector<float> hz((size_t)T * 4 * 32);
    for (auto &v : hz) v = rand() / (float)RAND_MAX - 0.5f;
    (void)hipMemcpy(dz, hz.data(), hz.size() * 4, hipMemcpyHostToDevice);
    for (int l = 0; l < 3; ++l) {
        (void)hipMalloc(&dZ[l], (size_t)T * 192 * 32 * 4);
        (void)hipMalloc(&M[l], (size_t)T * 3 * 64 * 4);
        (void)hipMemset(M[l], 0x5a, (size_t)T * 3 * 64 * 4);
    }
    (void)hipMalloc(&dX, (size_t)T * 64 * 32 * 4);
    for (int rep = 0; rep < 3; ++rep) esr_mlp_dgrad(kind, packed, dz, 0, T, M, dZ, dX, nullptr);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> st(512 * 16 * NS);
    (void)hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * 8);
    const char *names[11] = {"dz + masks + output layer^T", "mask + dZ2 stores", "layer-3^T MFMAs", "epilogue", "layer-2^T MFMAs",
                             "epilogue", "dX rows 0-31 (32x32)", "their stores", "dX rows 32-43 (4x4x1)", "their stores",
                             "tile total"};
    printf("(median over 512 traced waves x 4 tiles, shader cycles)\n");
    for (int seg = 0; seg < 11; ++seg) {
        std::vector<long long> d;
        for (int wg = 0; wg < 512; ++wg)
            for (int trip = 0; trip < 4; ++trip) {
                const unsigned long long *s = &st[(wg * 16 + trip) * NS];
                d.push_back(seg < 10 ? (long long)(s[seg + 1] - s[seg]) : (long long)(s[10] - s[0]));
            }
        std::sort(d.begin(), d.end());
        printf("  %-30s %8lld   (p10 %lld, p90 %lld)\n", names[seg], d[d.size() / 2], d[d.size() / 10], d[d.size() * 9 / 10]);
    }
    return 0;
}
