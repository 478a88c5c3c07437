// Reproducer for the hazard of rounds 1-3's ReLU: an inline-asm vector instruction that reads an MFMA result.  hipcc's
// hazard recogniser inserts the wait states a VALU instruction needs behind the MFMA that wrote its operand (up to 18 for
// a 16-pass MFMA), but it does not look inside asm statements.  One wave, one v_mfma_f32_32x32x2_f32 (16 passes) or one
// v_mfma_f32_32x32x16_bf16 (8 passes) into a zeroed accumulator, followed IMMEDIATELY (sched_barrier on both sides) by
//   A  asm("v_max_f32 %0, %1, 0")        -- the old relu_tiles
//   B  max((int)bits, 0) in C             -- the compiler-visible form used since round 4 (mlp_common.h)
//   C  asm behind an explicit s_nop 7 ; s_nop 7 ; s_nop 3 (19 wait states)
// Expected value of every element: a * b * K (positive), so ReLU must leave it unchanged; a hazard shows as the STALE
// accumulator content (0) or a partial sum.  Prints mismatches per variant.
//   gpurun -- './tools/ubench/asm_behind_mfma'
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int NT = 10;                 // accumulator tiles: enough register pressure that the MFMAs write VGPRs, as in the product kernels
template <int MODE, bool BF>
__global__ void __launch_bounds__(512, 1) probe(float *out, float a, float b)
{
    f32x16 acc[NT];
    for (int t = 0; t < NT; ++t)
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    bf16x8 a8, b8;
    for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)a; b8[i] = (__bf16)b; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        // each tile: ONE MFMA, then its ReLU right behind it (the schedule round 3's bf16 tone-mapper kernel got)
        if (BF) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[t], 0, 0, 0);
        else acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
        if (MODE == 2) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3");
        if (MODE == 1) {
            for (int r = 0; r < 16; ++r) { const int v = __float_as_int(acc[t][r]); acc[t][r] = __int_as_float(v > 0 ? v : 0); }
        } else {
            for (int r = 0; r < 16; ++r) asm volatile("v_max_f32 %0, %1, 0" : "=v"(acc[t][r]) : "v"(acc[t][r]));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (threadIdx.x < 64)
        for (int t = 0; t < NT; ++t)
            for (int r = 0; r < 16; ++r) out[(t * 16 + r) * 64 + threadIdx.x] = acc[t][r];
}

template <int MODE, bool BF>
int run(const char *name, float *dout)
{
    const float a = 1.5f, b = 2.0f, want = a * b * (BF ? 16.f : 2.f);
    (void)hipMemset(dout, 0xff, NT * 16 * 64 * 4);
    hipLaunchKernelGGL((probe<MODE, BF>), dim3(1), dim3(512), 0, 0, dout, a, b);
    (void)hipDeviceSynchronize();
    std::vector<float> got(NT * 16 * 64);
    (void)hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (size_t i = 0; i < got.size(); ++i)
        if (got[i] != want) { if (bad < 3) printf("    tile %zu register %zu lane %zu: got %g want %g\n", i / 1024, (i / 64) % 16, i % 64, got[i], want); ++bad; }
    printf("%-74s %4d of %d values wrong\n", name, bad, NT * 1024);
    return bad;
}

int main()
{
    float *dout;
    (void)hipMalloc(&dout, NT * 16 * 64 * 4);
    run<0, false>("f32 32x32x2 (16 passes)  A: asm v_max_f32 right behind the MFMA", dout);
    run<1, false>("f32 32x32x2 (16 passes)  B: integer max in C (compiler-visible)", dout);
    run<2, false>("f32 32x32x2 (16 passes)  C: asm behind 19 explicit wait states", dout);
    run<0, true>("bf16 32x32x16 (8 passes)  A: asm v_max_f32 right behind the MFMA", dout);
    run<1, true>("bf16 32x32x16 (8 passes)  B: integer max in C (compiler-visible)", dout);
    run<2, true>("bf16 32x32x16 (8 passes)  C: asm behind 19 explicit wait states", dout);
    return 0;
}
