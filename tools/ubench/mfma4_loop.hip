// Micro-benchmark: v_mfma_f32_4x4x1_16B_f32 cost per instruction with NACC independent accumulators, issued round-robin,
// one or two waves per SIMD.  (How many accumulators does out4_layer / dx4_layer need to run at the 8-cycle issue rate?)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC, int WAVES>
__global__ void __launch_bounds__(256 * WAVES, 1) k(float *out, int iters, float seed)
{
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a[4] = {seed, seed + 1, seed + 2, seed + 3}, b[4] = {seed * 2, 1, 2, 3};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 32 / NACC; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[(r + i) & 3], b[i & 3], acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, int WAVES> void run()
{
    float *out; hipMalloc(&out, 256 * 512 * 4);
    const int iters = 20000, grid = 256, nt = 256 * WAVES;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC, WAVES><<<grid, nt>>>(out, 50, 0.001f);
    hipEventRecord(e0);
    k<NACC, WAVES><<<grid, nt>>>(out, iters, 0.001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)iters * 32 * WAVES;
    printf("NACC %2d, %d wave(s)/SIMD: %.1f cycles per mfma4 per SIMD @2.4GHz\n", NACC, WAVES, ms * 1e6 / per_simd * 2.4);
    hipFree(out);
}
int main()
{
    run<1, 1>(); run<2, 1>(); run<4, 1>(); run<8, 1>(); run<16, 1>();
    run<1, 2>(); run<2, 2>(); run<4, 2>(); run<8, 2>(); run<16, 2>();
    return 0;
}
