// Do VALU instructions of the SAME wave issue in the shadow of its MFMAs?  One wave per SIMD (the split-fp16 kernels'
// residency), per loop trip G groups of { one v_mfma_f32_32x32x16_f16 (8 passes = 32 cycles of matrix pipe) + V independent
// v_fma_f32 (4 cycles of VALU issue each) }, accumulators rotating over 4 registers sets, VALU chains over 8 registers:
// nothing depends on anything recent.  If the two pipes overlap, a group costs max(32, 4 V) cycles; if they add, 32 + 4 V.
//   gpurun -- './tools/ubench/mfma_valu_overlap'
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int V, bool MFMA>
__global__ void __launch_bounds__(256, 1) k(float *out, unsigned long long *cyc, int iters, float seed)
{
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + threadIdx.x * 1e-3f); b[i] = (_Float16)(seed * 2.f); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed * (i + 1);
    const float m = 1.0001f, c = seed;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (MFMA) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[g & 3]) : "v"(a), "v"(b));
#pragma unroll
            for (int j = 0; j < V; ++j)
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(g * V + j) & 7]) : "v"(m), "v"(c));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V, bool MFMA> void run(float *out, unsigned long long *cyc)
{
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<V, MFMA><<<256, 256>>>(out, cyc, 200, 0.001f);
    (void)hipEventRecord(e0);
    k<V, MFMA><<<256, 256>>>(out, cyc, iters, 0.001f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(256);
    (void)hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto x : h) mean += (double)x; mean /= 256;
    // s_memtime counts at a fixed 100 MHz on this part: cycles from the event time at the clock the MFMA-only run implies
    printf("%-12s V=%2d  %8.3f ms  %7.2f ns per group  (memtime ticks per group %.3f)\n", MFMA ? "mfma+valu" : "valu only", V, ms,
           ms * 1e6 / (iters * 8.0), mean / (iters * 8.0));
}

int main()
{
    float *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 256 * 8);
    run<0, true>(out, cyc);
    run<2, true>(out, cyc);
    run<4, true>(out, cyc);
    run<6, true>(out, cyc);
    run<8, true>(out, cyc);
    run<12, true>(out, cyc);
    run<16, true>(out, cyc);
    run<24, true>(out, cyc);
    run<4, false>(out, cyc);
    run<8, false>(out, cyc);
    run<16, false>(out, cyc);
    run<24, false>(out, cyc);
    return 0;
}
