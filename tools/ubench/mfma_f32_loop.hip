// Micro-benchmark: v_mfma_f32_32x32x2_f32 issue rate of ONE wave per SIMD (the wgrad kernel's situation).
//   MODE 0: 9 accumulators, 4 dependent MFMAs in a row per accumulator (the wgrad source order)
//   MODE 1: same MFMAs, consecutive MFMAs on different accumulators
//   MODE 2: MODE 1 + six ds_read_b128 per 36 MFMAs (register double-buffered)
//   MODE 3: MODE 0 with 2 waves per SIMD (512 threads)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ void __launch_bounds__(MODE == 3 ? 512 : 256, 1) k(float *out, int iters, float seed)
{
    __shared__ __attribute__((aligned(16))) float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = seed * i;
    __syncthreads();
    f32x16 acc[9];
    for (int i = 0; i < 9; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float4 a[3], b[3];
    for (int i = 0; i < 3; ++i) { a[i] = make_float4(seed, seed + 1, seed + 2, seed + 3); b[i] = make_float4(seed * 2, 1, 2, 3); }
    const float *L = lds + (threadIdx.x & 63) * 4;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                a[i] = *reinterpret_cast<const float4 *>(L + ((it + i) & 7) * 256);
                b[i] = *reinterpret_cast<const float4 *>(L + ((it + i + 3) & 7) * 256 + 2048);
            }
        }
        if (MODE == 0 || MODE == 3) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    acc[3 * i + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[3 * i + j], 0, 0, 0);
                    acc[3 * i + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[3 * i + j], 0, 0, 0);
                    acc[3 * i + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[3 * i + j], 0, 0, 0);
                    acc[3 * i + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[3 * i + j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const float av = c == 0 ? a[i].x : c == 1 ? a[i].y : c == 2 ? a[i].z : a[i].w;
                        const float bv = c == 0 ? b[j].x : c == 1 ? b[j].y : c == 2 ? b[j].z : b[j].w;
                        acc[3 * i + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[3 * i + j], 0, 0, 0);
                    }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 9; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char *name)
{
    float *out; hipMalloc(&out, 256 * 512 * 4);
    const int iters = 4000, grid = 256, nt = MODE == 3 ? 512 : 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<grid, nt>>>(out, 50, 0.001f);
    hipEventRecord(e0);
    k<MODE><<<grid, nt>>>(out, iters, 0.001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * 36 * (MODE == 3 ? 2 : 1);
    const double tf = mfma_per_simd * 1024 * 4096 / (ms * 1e-3) / 1e12;
    printf("%-28s %.3f ms  %.1f ns per MFMA per SIMD (%.1f cycles @2.4GHz)  %.1f TFLOP/s\n", name, ms, ms * 1e6 / mfma_per_simd,
           ms * 1e6 / mfma_per_simd * 2.4, tf);
    hipFree(out);
}
int main()
{
    run<0>("chain4, 1 wave/SIMD");
    run<1>("interleaved, 1 wave/SIMD");
    run<2>("interleaved + ds_read");
    run<3>("chain4, 2 waves/SIMD");
    return 0;
}
