// Micro-benchmark: what one VMEM instruction per four f32 MFMAs costs the issuing wave (the forward / dgrad kernels'
// weight stream), one or two waves per SIMD.  The loads hit a 16-KB window (L1 / L2 resident): issue cost, not memory.
//   MODE 0: MFMAs only    MODE 1: + one buffer_load_dwordx4 per 4 MFMAs (consumed by the next group's MFMAs)
//   MODE 2: MODE 1 + one dword store per 6 MFMAs (the saved-activation stores)   MODE 3: + ds_read_b128 instead of the load
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
template <int MODE, int WPS>
__global__ void __launch_bounds__(256 * WPS, 1) k(const float *w, float *out, float *sink, int iters)
{
    __shared__ __attribute__((aligned(16))) float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = w[i];
    __syncthreads();
    rsrc_t R = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(w), 0, 16384, 0x00020000);
    rsrc_t S = __builtin_amdgcn_make_buffer_rsrc(sink + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) / 64 * 4096, 0, 16384, 0x00020000);
    const int lane = threadIdx.x & 63;
    f32x16 acc[6];
    for (int i = 0; i < 6; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float4 a = make_float4(1.f, 2.f, 3.f, 4.f), nxt = a;
    float b[4] = {0.5f, 0.25f, 0.125f, 0.75f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 24; ++g) {               // 24 groups of 4 MFMAs = 96 MFMAs per iteration
            if (MODE == 1 || MODE == 2 || MODE == 4 || MODE == 5) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(R, lane * 16, ((it * 24 + g) * 1024) & 16383, 0);
                nxt = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
            }
            if (MODE == 3) nxt = *reinterpret_cast<const float4 *>(lds + ((lane * 4 + (it * 24 + g) * 256) & 4095));
            acc[g % 6] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[0], acc[g % 6], 0, 0, 0);
            acc[g % 6] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[1], acc[g % 6], 0, 0, 0);
            acc[g % 6] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[2], acc[g % 6], 0, 0, 0);
            acc[g % 6] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[3], acc[g % 6], 0, 0, 0);
            if (MODE == 2 && (g % 3) == 0) {
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[(g + 3) % 6][g % 16]), S, lane * 4, (g * 256) & 16383, 2);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[(g + 4) % 6][g % 16]), S, lane * 4, (g * 256 + 128) & 16383, 2);
            }
            if (MODE == 4 && (g % 6) == 0) {     // the same bytes as MODE 2 in one 16-B-per-lane store per 24 MFMAs
                u32x4 v = {__float_as_uint(acc[(g + 3) % 6][0]), __float_as_uint(acc[(g + 3) % 6][1]),
                           __float_as_uint(acc[(g + 4) % 6][2]), __float_as_uint(acc[(g + 4) % 6][3])};
                __builtin_amdgcn_raw_buffer_store_b128(v, S, lane * 16, (g * 1024) & 16383, 2);
            }
            if (MODE == 5 && (g % 6) == 0) {     // MODE 4 through an LDS transpose: 4 ds_write_b32 + 1 ds_read_b128 + the store
                lds[lane + 0] = acc[(g + 3) % 6][0]; lds[lane + 64] = acc[(g + 3) % 6][1];
                lds[lane + 128] = acc[(g + 4) % 6][2]; lds[lane + 192] = acc[(g + 4) % 6][3];
                const float4 t = *reinterpret_cast<const float4 *>(lds + lane * 4);
                u32x4 v = {__float_as_uint(t.x), __float_as_uint(t.y), __float_as_uint(t.z), __float_as_uint(t.w)};
                __builtin_amdgcn_raw_buffer_store_b128(v, S, lane * 16, (g * 1024) & 16383, 2);
            }
            a = nxt;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 6; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE, int WPS> void run(const char *name)
{
    float *w, *out, *sink;
    hipMalloc(&w, 65536); hipMemset(w, 0, 65536); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&sink, (size_t)256 * 8 * 16384);
    const int iters = 1500, grid = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, WPS><<<grid, 256 * WPS>>>(w, out, sink, 20);
    hipEventRecord(e0);
    k<MODE, WPS><<<grid, 256 * WPS>>>(w, out, sink, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)iters * 96 * WPS;
    printf("%-34s %d wave(s)/SIMD  %.3f ms  %.1f cycles per MFMA per SIMD @2.4GHz  (%.1f TFLOP/s)\n", name, WPS, ms,
           ms * 1e6 / per_simd * 2.4, per_simd * 1024 * 4096 / (ms * 1e-3) / 1e12);
    hipFree(w); hipFree(out); hipFree(sink);
}
int main()
{
    run<0, 1>("MFMA only"); run<1, 1>("+ 1 dwordx4 load / 4 MFMA"); run<2, 1>("+ loads + 2 dword stores / 12 MFMA"); run<3, 1>("+ 1 ds_read_b128 / 4 MFMA"); run<4, 1>("+ loads + 1 dwordx4 store / 24 MFMA"); run<5, 1>("+ same via LDS transpose");
    run<0, 2>("MFMA only"); run<1, 2>("+ 1 dwordx4 load / 4 MFMA"); run<2, 2>("+ loads + 2 dword stores / 12 MFMA"); run<3, 2>("+ 1 ds_read_b128 / 4 MFMA"); run<4, 2>("+ loads + 1 dwordx4 store / 24 MFMA"); run<5, 2>("+ same via LDS transpose");
    return 0;
}
