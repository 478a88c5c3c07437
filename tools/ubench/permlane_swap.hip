// Micro-benchmark: cost of v_permlane32_swap_b32 (+ add) vs ds_bpermute (__shfl_xor 32), dependent chain, one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void __launch_bounds__(256, 1) k(float *out, int iters, float seed)
{
    float x = seed * threadIdx.x, y = seed + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) {
                const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
                x = __uint_as_float(r[0]) + __uint_as_float(r[1]) * 0.5f;
            } else if (MODE == 1) {
                x = x + __shfl_xor(x, 32) * 0.5f;
            } else {                          // 8 independent values per iteration
                const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(y + i), __float_as_uint(y + i), false, false);
                x += __uint_as_float(r[0]) + __uint_as_float(r[1]);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
template <int MODE> void run(const char *name)
{
    float *out; (void)hipMalloc(&out, 256 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<256, 256>>>(out, 10, 0.001f);
    (void)hipEventRecord(e0);
    k<MODE><<<256, 256>>>(out, 20000, 0.001f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %.1f cycles per exchange @2.4GHz\n", name, ms * 2.4e6 / (20000.0 * 8));
}
int main()
{
    run<0>("permlane32_swap + fma, dependent");
    run<1>("ds_bpermute (__shfl_xor 32) + fma, dependent");
    run<2>("permlane32_swap, independent");
    return 0;
}
