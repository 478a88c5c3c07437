// Where does a forward wave spend its time?  Builds the PRODUCT kernel (esr_nerf_amd/csrc/mlp.hip is included as is)
// with ESR_STAMP recording s_memtime (shader cycles) at the layer seams of the first wave of every workgroup, runs it on
// C2-sized synthetic tiles (8192 tiles, saves on) and prints the median segment lengths.
//   stamps: 0 tile start | 1 end of layer-1 MFMAs | 2 end of epilogue 1 | 3 end of layer-2 MFMAs | 4 end of epilogue 2 |
//           5 end of layer-3 MFMAs | 6 end of epilogue 3 | 7 end of the output layer
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
__device__ unsigned long long g_stamps[512 * 16 * 8];     // [workgroup][trip][stamp]
#define ESR_STAMP(i)                                                                                              \
    do {                                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        if ((threadIdx.x >> 6) == 0 && lane == 0) {                                                               \
            const int trip_ = (t - A.t0 - wave) / nwaves;                                                         \
            if (trip_ < 16) g_stamps[(blockIdx.x * 16 + trip_) * 8 + (i)] = __builtin_amdgcn_s_memtime();         \
        }                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)
#include "../../esr_nerf_amd/csrc/mlp.hip"

int main()
{
    const int T = 8192, kind = ESR_MLP_RADIANCE;
    const int64_t np = esr_mlp_packed_floats(kind);
    float *packed, *X, *H[3], *z; uint32_t *M[3];
    hipMalloc(&packed, np * 4);
    hipMalloc(&X, (size_t)T * 104 * 32 * 4);
    std::vector<float> h(np);
    for (auto &v : h) v = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
    hipMemcpy(packed, h.data(), np * 4, hipMemcpyHostToDevice);
    std::vector<float> hx((size_t)T * 104 * 32);
    for (auto &v : hx) v = rand() / (float)RAND_MAX - 0.5f;
    hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    for (int l = 0; l < 3; ++l) { hipMalloc(&H[l], (size_t)T * 192 * 32 * 4); hipMalloc(&M[l], (size_t)T * 3 * 64 * 4); }
    hipMalloc(&z, (size_t)T * 4 * 32 * 4);
    for (int save = 1; save >= 0; --save) {
        for (int rep = 0; rep < 3; ++rep) esr_mlp_fwd(kind, packed, X, 0, T, H, M, save, 0, z, nullptr);
        hipDeviceSynchronize();
        std::vector<unsigned long long> st(512 * 16 * 8);
        hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * 8);
        const char *names[8] = {"X loads + bias + layer-1 MFMAs", "epilogue 1 (+ requests for layer 2)", "layer-2 MFMAs",
                                "epilogue 2", "layer-3 MFMAs", "epilogue 3", "output layer", "tile total"};
        printf("save=%d  (median over 512 traced waves x 4 tiles, shader cycles)\n", save);
        for (int seg = 0; seg < 8; ++seg) {
            std::vector<long long> d;
            for (int wg = 0; wg < 512; ++wg)
                for (int trip = 0; trip < 4; ++trip) {
                    const unsigned long long *s = &st[(wg * 16 + trip) * 8];
                    d.push_back(seg < 7 ? (long long)(s[seg + 1] - s[seg]) : (long long)(s[7] - s[0]));
                }
            std::sort(d.begin(), d.end());
            printf("  %-38s %8lld   (p10 %lld, p90 %lld)\n", names[seg], d[d.size() / 2], d[d.size() / 10], d[d.size() * 9 / 10]);
        }
    }
    return 0;
}
