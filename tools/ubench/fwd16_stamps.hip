// Where does a wave of the shared-weights bf16 forward spend its time?  The PRODUCT kernel (esr_nerf_amd/csrc/mlp_bf16.hip
// included as is) with s_memtime stamps of wave 0 of every workgroup, C2-sized synthetic tiles, saves on and off.
//   0 group start | 1 X loads + layer-0 products | 2 epilogue 0 | 3 barrier | 4 layer-1 products | 5 epilogue 1 | 6 barrier |
//   7 layer-2 products | 8 epilogue 2 | 9 barrier | 10 output layer | 11 output epilogue | 12 barrier
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
constexpr int NS = 13;
__device__ unsigned long long g_stamps[256 * 8 * NS];     // [workgroup][trip][stamp]
#define ESR_STAMP16(i)                                                                                            \
    do {                                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        if (tid == 0) {                                                                                           \
            const int trip_ = (tg - (int)blockIdx.x) / (int)gridDim.x;                                            \
            if (trip_ < 8) g_stamps[(blockIdx.x * 8 + trip_) * NS + (i)] = __builtin_amdgcn_s_memtime();          \
        }                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
    } while (0)
#include "../../esr_nerf_amd/csrc/mlp_bf16.hip"
#include "../../esr_nerf_amd/csrc/mlp.hip"

int main()
{
    const int T = 8192, kind = ESR_MLP_RADIANCE;
    const int64_t np = esr_mlp_packed_floats(kind), np16 = esr_mlp_packed_bf16_elems(kind);
    float *packed, *X, *H[3], *z; uint32_t *M[3]; void *p16;
    (void)hipMalloc(&packed, np * 4); (void)hipMalloc(&p16, np16 * 2);
    std::vector<float> h(np);
    for (auto &v : h) v = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
    (void)hipMemcpy(packed, h.data(), np * 4, hipMemcpyHostToDevice);
    (void)hipMemset(p16, 0x3c, np16 * 2);
    (void)hipMalloc(&X, (size_t)T * 104 * 32 * 4);
    std::vector<float> hx((size_t)T * 104 * 32);
    for (auto &v : hx) v = rand() / (float)RAND_MAX - 0.5f;
    (void)hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    for (int l = 0; l < 3; ++l) { (void)hipMalloc(&H[l], (size_t)T * 192 * 32 * 4); (void)hipMalloc(&M[l], (size_t)T * 3 * 64 * 4); }
    (void)hipMalloc(&z, (size_t)T * 4 * 32 * 4);
    const char *names[NS] = {"X loads + layer 0", "epilogue 0", "barrier", "layer 1", "epilogue 1", "barrier", "layer 2", "epilogue 2",
                             "barrier", "output layer", "output epilogue", "barrier", "group total"};
    for (int save = 1; save >= 0; --save) {
        for (int rep = 0; rep < 3; ++rep) esr_mlp_fwd_bf16(kind, packed, p16, X, 0, T, H, M, save, 0, z, nullptr);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> st(256 * 8 * NS);
        (void)hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * 8);
        printf("save=%d  (median over 256 workgroups x 3 tile groups, shader cycles)\n", save);
        for (int seg = 0; seg < NS; ++seg) {
            std::vector<long long> d;
            for (int wg = 0; wg < 256; ++wg)
                for (int trip = 0; trip < 3; ++trip) {
                    const unsigned long long *s = &st[(wg * 8 + trip) * NS];
                    d.push_back(seg < NS - 1 ? (long long)(s[seg + 1] - s[seg]) : (long long)(s[NS - 1] - s[0]));
                }
            std::sort(d.begin(), d.end());
            printf("  %-22s %8lld   (p10 %lld, p90 %lld)\n", names[seg], d[d.size() / 2], d[d.size() / 10], d[d.size() * 9 / 10]);
        }
    }
    return 0;
}
