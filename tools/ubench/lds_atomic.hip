// Micro-benchmark: LDS atomic-add rate by operand type on gfx950 (which form should feat_bwd's window use?).
// One workgroup of 256 threads per CU; every lane adds into its own word (no address conflicts), stride-1.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((address_space(3))) float lds_f;
typedef __attribute__((address_space(3))) unsigned lds_u;
typedef __attribute__((address_space(3))) unsigned long long lds_u64;
typedef __attribute__((address_space(3))) double lds_d;
template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int iters, int spread)
{
    __shared__ __attribute__((aligned(16))) unsigned long long buf[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) buf[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x;
    // spread = 1: lane i -> word i; spread = 0: pseudo-random word in a 1024-word window
    int a = spread ? lane : (lane * 37 + 11) & 1023;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) __hip_atomic_fetch_add((lds_f *)buf + a, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 1) __hip_atomic_fetch_add((lds_u *)buf + a, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 2) __hip_atomic_fetch_add((lds_u64 *)buf + a, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 4) __hip_atomic_fetch_add((lds_d *)buf + a, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 5) {   // float -> scaled int64 -> ds_add_u64 (the conversion cost included)
            const float v = (float)(it + lane) * 1.37f;
            const long long q = __float2ll_rn(v * 1048576.f);
            __hip_atomic_fetch_add((lds_u64 *)buf + a, (unsigned long long)q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (MODE == 6) {   // float -> double -> ds_add_f64
            const float v = (float)(it + lane) * 1.37f;
            __hip_atomic_fetch_add((lds_d *)buf + a, (double)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (MODE == 3) ((volatile lds_f *)buf)[a] = (float)it;                   // plain store for reference
        a = spread ? ((a + 256) & 2047) : ((a * 5 + 1) & 1023);
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (float)buf[1];
}
template <int MODE> double run(const char *name, int spread)
{
    float *out; hipMalloc(&out, 4096);
    const int iters = 20000, grid = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<grid, 256>>>(out, 100, spread);
    hipEventRecord(e0);
    k<MODE><<<grid, 256>>>(out, iters, spread);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double inst = (double)iters * 4;            // wave-instructions per CU
    printf("%-10s spread=%d  %.3f ms  %.1f ns per wave-instruction per CU  (%.2f lanes/clk/CU @2.4GHz)\n", name, spread, ms,
           ms * 1e6 / inst, 64.0 / (ms * 1e6 / inst * 2.4));
    hipFree(out);
    return ms;
}
int main()
{
    for (int s = 1; s >= 0; --s) {
        run<0>("add_f32", s); run<1>("add_u32", s); run<2>("add_u64", s); run<3>("store_b32", s); run<4>("add_f64", s); run<5>("cvt+u64", s); run<6>("cvt+f64", s);
    }
    return 0;
}
