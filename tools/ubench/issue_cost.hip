// Issue cost (clocks per instruction) of the instructions that make up the split-fp16 kernels' epilogues, one wave per
// SIMD (their residency), 64 independent instructions per loop trip (8 destination registers in rotation), s_memtime
// around 2000 trips.  What the micro-slice budget is counted in: an MFMA leaves 32 clocks for independent instructions.
//   gpurun -- './tools/ubench/issue_cost'
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define OPS(X)                                                                                                        \
    X(0, "v_fma_f32", "v_fma_f32 %0, %1, %2, %0", "+v"(r[i & 7]), "v"(a), "v"(b))                                      \
    X(1, "v_add_f32", "v_add_f32 %0, %1, %0", "+v"(r[i & 7]), "v"(a), "v"(b))                                          \
    X(2, "v_pk_add_f32", "v_pk_add_f32 %0, %1, %0", "+v"(d[i & 3]), "v"(da), "v"(b))                                   \
    X(3, "v_pk_fma_f32", "v_pk_fma_f32 %0, %1, %1, %0", "+v"(d[i & 3]), "v"(da), "v"(b))                               \
    X(4, "v_max_i32", "v_max_i32 %0, %1, %0", "+v"(r[i & 7]), "v"(a), "v"(b))                                          \
    X(5, "v_med3_i32", "v_med3_i32 %0, %0, 0, 1", "+v"(r[i & 7]), "v"(a), "v"(b))                                      \
    X(6, "v_lshlrev_b32", "v_lshlrev_b32 %0, 3, %0", "+v"(r[i & 7]), "v"(a), "v"(b))                                   \
    X(7, "v_or3_b32", "v_or3_b32 %0, %0, %1, %2", "+v"(r[i & 7]), "v"(a), "v"(b))                                      \
    X(8, "v_lshl_or_b32", "v_lshl_or_b32 %0, %1, 5, %0", "+v"(r[i & 7]), "v"(a), "v"(b))                               \
    X(9, "v_cvt_pk_f16_f32", "v_cvt_pk_f16_f32 %0, %1, %2", "+v"(r[i & 7]), "v"(a), "v"(b))                            \
    X(10, "v_cvt_pkrtz_f16_f32", "v_cvt_pkrtz_f16_f32 %0, %1, %2", "+v"(r[i & 7]), "v"(a), "v"(b))                     \
    X(11, "v_cvt_f32_f16", "v_cvt_f32_f16 %0, %1", "+v"(r[i & 7]), "v"(a), "v"(b))                                     \
    X(12, "v_cvt_f32_f16 sdwa", "v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1", "+v"(r[i & 7]), "v"(a), "v"(b)) \
    X(13, "v_cvt_f16_f32", "v_cvt_f16_f32 %0, %1", "+v"(r[i & 7]), "v"(a), "v"(b))                                     \
    X(14, "v_accvgpr_read", "v_accvgpr_read_b32 %0, a7", "+v"(r[i & 7]), "v"(a), "v"(b))                               \
    X(15, "v_accvgpr_write", "v_accvgpr_write_b32 a7, %1", "+v"(r[i & 7]), "v"(a), "v"(b))                             \
    X(16, "v_mov_b32", "v_mov_b32 %0, %1", "+v"(r[i & 7]), "v"(a), "v"(b))                                             \
    X(17, "v_fma_mix_f32", "v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]", "+v"(r[i & 7]), "v"(a), "v"(b))           \
    X(18, "s_movk_i32", "s_movk_i32 s20, 0x1234", "+v"(r[i & 7]), "v"(a), "v"(b))                                      \
    X(19, "s_nop 0", "s_nop 0", "+v"(r[i & 7]), "v"(a), "v"(b))                                                        \
    X(20, "v_and_b32", "v_and_b32 %0, %1, %0", "+v"(r[i & 7]), "v"(a), "v"(b))                                         \
    X(21, "v_perm_b32", "v_perm_b32 %0, %1, %2, %0", "+v"(r[i & 7]), "v"(a), "v"(b))                                   \
    X(22, "v_pk_mul_f16", "v_pk_mul_f16 %0, %1, %0", "+v"(r[i & 7]), "v"(a), "v"(b))                                   \
    X(23, "v_pk_max_f16", "v_pk_max_f16 %0, %1, %0", "+v"(r[i & 7]), "v"(a), "v"(b))

template <int OP>
__global__ void __launch_bounds__(256, 1) k(float *out, unsigned long long *cyc, int iters, float seed)
{
    float r[8];
    double d[4];
    for (int i = 0; i < 8; ++i) r[i] = seed * (i + 1);
    for (int i = 0; i < 4; ++i) d[i] = seed * (i + 1);
    float a = seed + threadIdx.x * 1e-3f, b = 1.0001f;
    double da = a;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 64; ++i) {
#define X(N, NAME, ASM, OUT, IN1, IN2) if (OP == N) { if (N == 14 || N == 15 || N == 18) asm volatile(ASM : OUT : IN1, IN2 : "s20", "a7"); else asm volatile(ASM : OUT : IN1, IN2); }
            OPS(X)
#undef X
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += r[i];
    for (int i = 0; i < 4; ++i) s += (float)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// what slows the issue down: a register clobber list on the same 64 x v_mov_b32 loop
template <int CL>
__global__ void __launch_bounds__(256, 1) kc(float *out, unsigned long long *cyc, int iters, float seed)
{
    float r[8];
    for (int i = 0; i < 8; ++i) r[i] = seed * (i + 1);
    float a = seed + threadIdx.x * 1e-3f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            if (CL == 0) asm volatile("v_mov_b32 %0, %1" : "+v"(r[i & 7]) : "v"(a));
            if (CL == 1) asm volatile("v_mov_b32 %0, %1" : "+v"(r[i & 7]) : "v"(a) : "a7");
            if (CL == 2) asm volatile("v_mov_b32 %0, %1" : "+v"(r[i & 7]) : "v"(a) : "s20");
            if (CL == 3) asm volatile("v_mov_b32 %0, %1" : "+v"(r[i & 7]) : "v"(a) : "v200");
            if (CL == 4) asm volatile("v_mov_b32 %0, %1" : "+v"(r[i & 7]) : "v"(a) : "a200");
            if (CL == 5) asm volatile("v_mov_b32 %0, %1" : "+v"(r[i & 7]) : "v"(a) : "v255", "a255");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int CL> void runc(const char *name, float *out, unsigned long long *cyc)
{
    kc<CL><<<256, 256>>>(out, cyc, 50, 0.001f);
    kc<CL><<<256, 256>>>(out, cyc, 2000, 0.001f);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(256);
    (void)hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto x : h) mean += (double)x; mean /= 256;
    printf("v_mov_b32, clobber %-12s %6.2f clocks per instruction\n", name, mean / (2000 * 64.0));
}

// memory instructions: LDS reads (lane-linear 16 B), dword stores through a zero-record descriptor (dropped by the range check)
__global__ void __launch_bounds__(256, 1) kmem(float *out, unsigned long long *cyc, int iters, int which)
{
    __shared__ f32x4 sh[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) sh[i] = f32x4{1.f, 2.f, 3.f, 4.f};
    __syncthreads();
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0, 0x00020000);      // zero records: every store is dropped
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (which == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { const f32x4 v = sh[(threadIdx.x + 64 * i) & 1023]; acc += v; }
        } else if (which == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[0]), rsrc, threadIdx.x * 4, i * 128, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[0]), rsrc, threadIdx.x * 4, 0x1000 + i * 128, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP> void run(const char *name, float *out, unsigned long long *cyc)
{
    const int iters = 2000;
    k<OP><<<256, 256>>>(out, cyc, 50, 0.001f);
    k<OP><<<256, 256>>>(out, cyc, iters, 0.001f);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(256);
    (void)hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto x : h) mean += (double)x; mean /= 256;
    printf("%-22s %6.2f clocks per instruction\n", name, mean / (iters * 64.0));
}

int main()
{
    float *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 256 * 8);
#define X(N, NAME, ASM, OUT, IN1, IN2) run<N>(NAME, out, cyc);
    OPS(X)
#undef X
    runc<0>("none", out, cyc); runc<1>("a7", out, cyc); runc<2>("s20", out, cyc); runc<3>("v200", out, cyc); runc<4>("a200", out, cyc); runc<5>("v255+a255", out, cyc);
    const char *names[3] = {"ds_read_b128", "buffer_store_dword (imm offset)", "buffer_store_dword (s offset)"};
    for (int w = 0; w < 3; ++w) {
        kmem<<<256, 256>>>(out, cyc, 50, w);
        kmem<<<256, 256>>>(out, cyc, 2000, w);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(256);
        (void)hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
        double mean = 0; for (auto x : h) mean += (double)x; mean /= 256;
        printf("%-32s %6.2f clocks per instruction (4 waves per CU issuing)\n", names[w], mean / (2000 * 16.0));
    }
    return 0;
}
