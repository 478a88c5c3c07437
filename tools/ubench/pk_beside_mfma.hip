// Do packed-fp32 instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) of one wave return wrong results while ANOTHER kernel
// runs beside it on the same SIMDs?  Round 6: esr_expgrad_fwd, whose interpolation weights the SLP vectoriser had packed, returned
// wrong rows in lanes 48-63 of single waves whenever the light-transport step ran on another stream (tools/debug/two_process_lanes.py).
// Victim (stream A): every lane computes, R times, a packed result and the same two results with scalar instructions, and counts
// mismatches per lane quarter.  Aggressor (stream B): one kind of load, running throughout.
//   gpurun -- './tools/ubench/pk_beside_mfma'
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// ---- victims -------------------------------------------------------------------------------------------------------------------
template <int V>
__global__ void __launch_bounds__(256) victim(const float *in, unsigned *bad, int n, int reps)
{
    const int lane = threadIdx.x & 63;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        f32x2 x = {in[2 * i], in[2 * i + 1]};
        for (int r = 0; r < reps; ++r) {
            f32x2 fl = {__builtin_floorf(x[0]), __builtin_floorf(x[1])};
            f32x2 got, want;
            if (V == 0) {            // plain: fl + x
                asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(got) : "v"(fl), "v"(x));
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(want[0]) : "v"(fl[0]), "v"(x[0]));
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(want[1]) : "v"(fl[1]), "v"(x[1]));
            } else if (V == 1) {     // negated second operand: x - fl
                asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(got) : "v"(x), "v"(fl));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(want[0]) : "v"(x[0]), "v"(fl[0]));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(want[1]) : "v"(x[1]), "v"(fl[1]));
            } else if (V == 2) {     // crossed halves, as the compiler wrote them in expgrad_kernel: (x.lo - fl.hi, x.hi - fl.lo)
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(got) : "v"(x), "v"(fl));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(want[0]) : "v"(x[0]), "v"(fl[1]));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(want[1]) : "v"(x[1]), "v"(fl[0]));
            } else if (V == 3) {     // broadcast low half of the second operand
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(got) : "v"(x), "v"(fl));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(want[0]) : "v"(x[0]), "v"(fl[0]));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(want[1]) : "v"(x[1]), "v"(fl[0]));
            } else if (V == 4) {     // fused multiply-add
                got = x;  want = x;
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(got) : "v"(fl), "v"(x));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(want[0]) : "v"(fl[0]), "v"(x[0]));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(want[1]) : "v"(fl[1]), "v"(x[1]));
            } else if (V == 6) {     // crossed halves on the multiply
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(got) : "v"(x), "v"(fl));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(want[0]) : "v"(x[0]), "v"(fl[1]));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(want[1]) : "v"(x[1]), "v"(fl[0]));
            } else if (V == 7) {     // crossed halves without the negation
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(got) : "v"(x), "v"(fl));
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(want[0]) : "v"(x[0]), "v"(fl[1]));
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(want[1]) : "v"(x[1]), "v"(fl[0]));
            } else if (V == 8) {     // the split-fp16 kernels' own op_sel instruction as the victim: fp32 minus the HIGH fp16 half of a register
                const unsigned u = (__float_as_uint(x[0]) & 0xffff0000u) >> 0 | 0x3c00u;      // hi half: some fp16 pattern of x; lo half: 1.0
                asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(got[0]) : "v"(u), "v"(x[1]));
                float hf;
                asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(hf) : "v"(u >> 16));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(want[0]) : "v"(x[1]), "v"(hf));
                got[1] = want[1] = 0.f;
            } else if (V == 9) {     // ... and the LOW half (op_sel 0)
                const unsigned u = (__float_as_uint(x[0]) >> 16) | 0x3c000000u;
                asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(got[0]) : "v"(u), "v"(x[1]));
                float hf;
                asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(hf) : "v"(u & 0xffffu));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(want[0]) : "v"(x[1]), "v"(hf));
                got[1] = want[1] = 0.f;
            } else if (V == 10) {    // mlp.hip's form: the HIGH fp16 half as the ADDEND (third operand)
                const unsigned u = (__float_as_uint(x[0]) & 0xffff0000u) | 0x3c00u;
                asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(got[0]) : "v"(x[1]), "v"(fl[1]), "v"(u));
                float hf;
                asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(hf) : "v"(u >> 16));
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(want[0]) : "v"(x[1]), "v"(fl[1]), "v"(hf));
                got[1] = want[1] = 0.f;
            } else if (V == 11) {    // v_fma_mixlo_f16 / v_fma_mixhi_f16 (fp16 results into the two halves of one register), against v_cvt_f16_f32 of the fp32 fma
                const unsigned u = 0x3c003c00u;        // (1.0, 1.0): x * 1.0 + fl, rounded once to fp16 -- the fp32 fma below is exact in fp32 only
                unsigned p = 0;                        // for small integers, so compare the packed register of two identical computations instead
                unsigned q = 0;
                asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %0, %4, %2, %3 op_sel_hi:[0,0,1]" : "+v"(p) : "v"(x[0]), "v"(fl[1]), "v"(u), "v"(x[1]));
                asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[0,0,1]" : "+v"(q) : "v"(x[0]), "v"(fl[1]), "v"(u));
                unsigned q2 = 0;
                asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[0,0,1]" : "+v"(q2) : "v"(x[1]), "v"(fl[1]), "v"(u));
                got[0] = __uint_as_float(p);  want[0] = __uint_as_float((q & 0xffffu) | (q2 << 16));
                got[1] = want[1] = 0.f;
            } else if (V == 12) {    // v_cvt_pk_f16_f32 against two v_cvt_f16_f32
                unsigned p, a, b;
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p) : "v"(x[0]), "v"(x[1]));
                asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(a) : "v"(x[0]));
                asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(b) : "v"(x[1]));
                got[0] = __uint_as_float(p);  want[0] = __uint_as_float((a & 0xffffu) | (b << 16));
                got[1] = want[1] = 0.f;
            } else if (V == 13) {    // v_pk_min_u16 (mlp_bf16.hip) against v_min_u32 on the halves
                const unsigned a = __float_as_uint(x[0]), b = __float_as_uint(x[1]);
                unsigned p;
                asm volatile("v_pk_min_u16 %0, %1, %2" : "=v"(p) : "v"(a), "v"(b));
                const unsigned lo = min(a & 0xffffu, b & 0xffffu), hi = min(a >> 16, b >> 16);
                got[0] = __uint_as_float(p);  want[0] = __uint_as_float(lo | (hi << 16));
                got[1] = want[1] = 0.f;
            } else if (V == 14) {    // v_pk_min_u16 with crossed halves
                const unsigned a = __float_as_uint(x[0]), b = __float_as_uint(x[1]);
                unsigned p;
                asm volatile("v_pk_min_u16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(p) : "v"(a), "v"(b));
                const unsigned lo = min(a & 0xffffu, b >> 16), hi = min(a >> 16, b & 0xffffu);
                got[0] = __uint_as_float(p);  want[0] = __uint_as_float(lo | (hi << 16));
                got[1] = want[1] = 0.f;
            } else {                 // control: no packed instruction at all
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(got[0]) : "v"(x[0]), "v"(fl[0]));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(got[1]) : "v"(x[1]), "v"(fl[1]));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(want[0]) : "v"(x[0]), "v"(fl[0]));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(want[1]) : "v"(x[1]), "v"(fl[1]));
            }
            if (__float_as_uint(got[0]) != __float_as_uint(want[0]) || __float_as_uint(got[1]) != __float_as_uint(want[1]))
                atomicAdd(&bad[lane >> 4], 1u);
            x[0] = x[0] * 1.0001f + 0.37f;  x[1] = x[1] * 0.9997f + 1.91f;
            if (!(fabsf(x[0]) < 1e4f)) x[0] = 1.5f;
            if (!(fabsf(x[1]) < 1e4f)) x[1] = 2.5f;
        }
    }
}

// ---- aggressors ----------------------------------------------------------------------------------------------------------------
template <int A>
__global__ void __launch_bounds__(256, 1) aggressor(float *out, int iters, float seed)
{
    __shared__ float lds[4096];
    const int t = threadIdx.x;
    float acc_s = seed + t;
    if (A == 1) {              // v_mfma_f32_32x32x16_f16, four independent accumulators (the split-fp16 kernels' instruction)
        f32x16 acc[4] = {};
        f16x8 a, b;
        for (int k = 0; k < 8; ++k) { a[k] = (_Float16)(seed + k * 0.01f + t * 1e-3f); b[k] = (_Float16)(0.5f - k * 0.02f); }
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[q], 0, 0, 0);
        for (int q = 0; q < 4; ++q) for (int k = 0; k < 16; ++k) acc_s += acc[q][k];
    } else if (A == 2) {       // v_mfma_f32_32x32x2_f32 (the weight-gradient kernels')
        f32x16 acc[4] = {};
        const float a = seed + t * 1e-3f, b = 0.25f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
        for (int q = 0; q < 4; ++q) for (int k = 0; k < 16; ++k) acc_s += acc[q][k];
    } else if (A == 3) {       // v_mfma_f32_4x4x1_16B_f32 (the narrow output layers')
        f32x4 acc[4] = {};
        const float a = seed + t * 1e-3f, b = 0.25f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[q], 0, 0, 0);
        for (int q = 0; q < 4; ++q) for (int k = 0; k < 4; ++k) acc_s += acc[q][k];
    } else if (A == 4) {       // plain vector arithmetic
        float r[8];
        for (int k = 0; k < 8; ++k) r[k] = seed + k + t;
        for (int it = 0; it < iters * 4; ++it)
#pragma unroll
            for (int k = 0; k < 8; ++k) r[k] = fmaf(r[k], 0.999f, 0.25f);
        for (int k = 0; k < 8; ++k) acc_s += r[k];
    } else if (A == 5) {       // LDS traffic
        for (int k = t; k < 4096; k += 256) lds[k] = seed + k;
        __syncthreads();
        for (int it = 0; it < iters * 4; ++it) {
            acc_s += lds[(t * 17 + it) & 4095];
            lds[(t * 33 + it) & 4095] = acc_s;
        }
    } else if (A == 6) {       // packed fp32 itself
        f32x2 r[4];
        for (int k = 0; k < 4; ++k) r[k] = f32x2{seed + k, seed - k};
        const f32x2 m = {0.999f, 1.001f}, c = {0.25f, -0.25f};
        for (int it = 0; it < iters * 4; ++it)
#pragma unroll
            for (int k = 0; k < 4; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(r[k]) : "v"(m), "v"(c));
        for (int k = 0; k < 4; ++k) acc_s += r[k][0] + r[k][1];
    } else if (A == 7) {       // v_fma_mix_f32 with op_sel (the split-fp16 kernels' residual: fp32 value minus the hi half of a packed fp16 pair)
        float r[4];
        unsigned u = 0x3c003800u + t;
        for (int k = 0; k < 4; ++k) r[k] = seed + k;
        for (int it = 0; it < iters * 4; ++it)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                asm volatile("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r[k]) : "v"(u));
        for (int k = 0; k < 4; ++k) acc_s += r[k];
    } else if (A == 8) {       // v_cvt_pk_f16_f32
        float r[4];
        unsigned p[4] = {};
        for (int k = 0; k < 4; ++k) r[k] = seed + k;
        for (int it = 0; it < iters * 4; ++it)
#pragma unroll
            for (int k = 0; k < 4; ++k) asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(p[k]) : "v"(r[k]));
        for (int k = 0; k < 4; ++k) acc_s += (float)p[k];
    } else if (A == 9) {       // MFMA and the op_sel instructions of the same wave, alternating
        f32x16 acc[2] = {};
        f16x8 a, b;
        float r[4];
        unsigned u = 0x3c003800u + t;
        for (int k = 0; k < 4; ++k) r[k] = seed + k;
        for (int k = 0; k < 8; ++k) { a[k] = (_Float16)(seed + k * 0.01f + t * 1e-3f); b[k] = (_Float16)(0.5f - k * 0.02f); }
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[q], 0, 0, 0);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    asm volatile("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r[k]) : "v"(u));
            }
        for (int q = 0; q < 2; ++q) for (int k = 0; k < 16; ++k) acc_s += acc[q][k];
        for (int k = 0; k < 4; ++k) acc_s += r[k];
    }
    out[blockIdx.x * 256 + t] = acc_s;
}

template <int A> static void launch_aggressor(float *out, int iters, hipStream_t s) { aggressor<A><<<1024, 256, 0, s>>>(out, iters, 0.5f); }
template <int V> static void launch_victim(const float *in, unsigned *bad, int n, int reps, hipStream_t s)
{
    victim<V><<<(n + 255) / 256, 256, 0, s>>>(in, bad, n, reps);
}

int main(int argc, char **argv)
{
    const int only_a = argc > 1 ? atoi(argv[1]) : -1;      // one aggressor only (0 = none: run it beside another process)
    const int n = 8192, reps = 64, launches = 3000;
    std::vector<float> h(2 * n);
    for (int i = 0; i < 2 * n; ++i) h[i] = (float)((i * 2654435761u) % 100000) * 1e-3f - 37.f;
    float *in, *out;
    unsigned *bad;
    CK(hipMalloc(&in, 2 * n * 4));  CK(hipMalloc(&out, 1024 * 256 * 4));  CK(hipMalloc(&bad, 16));
    CK(hipMemcpy(in, h.data(), 2 * n * 4, hipMemcpyHostToDevice));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));  CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    const char *an[] = {"nothing", "v_mfma_f32_32x32x16_f16", "v_mfma_f32_32x32x2_f32", "v_mfma_f32_4x4x1_f32", "v_fma_f32", "LDS reads/writes", "v_pk_fma_f32", "v_fma_mix_f32 op_sel", "v_cvt_pk_f16_f32", "MFMA + v_fma_mix op_sel"};
    const char *vn[] = {"v_pk_add_f32", "v_pk_add_f32 neg", "v_pk_add_f32 op_sel neg", "v_pk_mul_f32 op_sel_hi", "v_pk_fma_f32", "scalar only", "v_pk_mul_f32 op_sel", "v_pk_add_f32 op_sel", "v_fma_mix_f32 hi half", "v_fma_mix_f32 lo half",
                        "v_fma_mix_f32 hi addend", "v_fma_mixlo/hi_f16", "v_cvt_pk_f16_f32", "v_pk_min_u16", "v_pk_min_u16 op_sel"};
    printf("victim: %d launches of %d lanes x %d repetitions per cell; wrong results by lane quarter [0-15, 16-31, 32-47, 48-63]\n", launches, n, reps);
    for (int a = (only_a >= 0 ? only_a : 0); a < (only_a >= 0 ? only_a + 1 : 10); ++a)
        for (int v = 0; v < 15; ++v) {
            CK(hipMemset(bad, 0, 16));
            CK(hipDeviceSynchronize());
            for (int l = 0; l < launches; ++l) {
                if (l % 50 == 0 && a) {      // ~50 victim launches per aggressor launch (a few ms of load each)
                    const int it = 40000;
                    switch (a) {
                    case 1: launch_aggressor<1>(out, it, sb); break;   case 2: launch_aggressor<2>(out, it, sb); break;
                    case 3: launch_aggressor<3>(out, it, sb); break;   case 4: launch_aggressor<4>(out, it, sb); break;
                    case 5: launch_aggressor<5>(out, it, sb); break;   case 6: launch_aggressor<6>(out, it, sb); break;
                    case 7: launch_aggressor<7>(out, it, sb); break;   case 8: launch_aggressor<8>(out, it, sb); break;
                    case 9: launch_aggressor<9>(out, it, sb); break;
                    }
                }
                switch (v) {
                case 0: launch_victim<0>(in, bad, n, reps, sa); break;   case 1: launch_victim<1>(in, bad, n, reps, sa); break;
                case 2: launch_victim<2>(in, bad, n, reps, sa); break;   case 3: launch_victim<3>(in, bad, n, reps, sa); break;
                case 4: launch_victim<4>(in, bad, n, reps, sa); break;   case 5: launch_victim<5>(in, bad, n, reps, sa); break;
                case 6: launch_victim<6>(in, bad, n, reps, sa); break;   case 7: launch_victim<7>(in, bad, n, reps, sa); break;
                case 8: launch_victim<8>(in, bad, n, reps, sa); break;   case 9: launch_victim<9>(in, bad, n, reps, sa); break;
                case 10: launch_victim<10>(in, bad, n, reps, sa); break; case 11: launch_victim<11>(in, bad, n, reps, sa); break;
                case 12: launch_victim<12>(in, bad, n, reps, sa); break; case 13: launch_victim<13>(in, bad, n, reps, sa); break;
                case 14: launch_victim<14>(in, bad, n, reps, sa); break;
                }
                if (l % 50 == 49) CK(hipStreamSynchronize(sa));
            }
            CK(hipDeviceSynchronize());
            unsigned hb[4];
            CK(hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost));
            printf("beside %-26s %-26s wrong: [%u, %u, %u, %u]\n", an[a], vn[v], hb[0], hb[1], hb[2], hb[3]);
            fflush(stdout);
        }
    return 0;
}
