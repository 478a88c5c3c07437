// Micro-benchmark: does a wave issuing v_mfma_f32_4x4x1 share the SIMD's matrix pipe efficiently with a second wave
// issuing v_mfma_f32_32x32x2?  Two waves per SIMD; wave A runs N32 32x32x2 products, wave B N4 4x4x1 products.
//   (a) A alone, (b) B alone, (c) both: if (c) ~ (a) + (b) the pipe is shared without loss.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(512, 1) k(float *out, int n32, int n4, float seed)
{
    const int wave = threadIdx.x >> 6;
    float s = 0.f;
    if (wave < 4) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < n32; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(seed + i, seed, acc[i & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    } else {
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < n4; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f32_4x4x1f32(seed + i, seed, acc[i & 7], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
static float run(float *out, int n32, int n4)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<<<256, 512>>>(out, 10, 10, 0.001f);
    (void)hipEventRecord(e0);
    k<<<256, 512>>>(out, n32, n4, 0.001f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main()
{
    float *out; (void)hipMalloc(&out, 256 * 512 * 4);
    const int n32 = 4000, n4 = 32000;          // 64 k x 64 cycles  vs  512 k x 8 cycles: equal matrix-pipe time
    const float a = run(out, n32, 0), b = run(out, 0, n4), c = run(out, n32, n4);
    printf("32x32x2 alone %.3f ms (%.1f cyc each), 4x4x1 alone %.3f ms (%.1f cyc each), both %.3f ms (sum %.3f)\n",
           a, a * 2.4e6 / (n32 * 16.0), b, b * 2.4e6 / (n4 * 16.0), c, a + b);
    const float c2 = run(out, n32, n4 / 8);
    printf("32x32x2 + 1/8 of the 4x4x1: %.3f ms (sum %.3f)\n", c2, a + b / 8);
    return 0;
}
