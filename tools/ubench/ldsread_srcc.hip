// Reproducer attempt for the anomaly recorded in csrc/mlp_bf16.hip (round 2): accumulators initialised by ds_read_b128
// STRAIGHT into the registers an MFMA then uses as srcC gave wrong values in two registers of the last tile of a 192-wide
// layer (rows 11 / 15 / 16 / 20 of units 160-191), deterministically, with bias and weights in LDS verified correct.
// The product kernels zero-initialise and add the bias after the layer.  This harness runs ONE hidden layer of the product
// (lds_layer16<12, 6, 0>: 72 MFMAs, weights from LDS, double-buffered ds_read_b128) three ways on the same inputs:
//   A  accumulators <- bias by ds_read_b128 (the form that failed), then the layer
//   B  accumulators <- 0, the layer, then += bias from LDS (the product form)
//   C  as A, with an explicit s_waitcnt lgkmcnt(0) + s_nop 4 between the bias reads and the first MFMA
// and compares every register of every lane with a host fp32 evaluation (bf16 operands exactly representable, so the sums
// are exact).  Prints the number of mismatching registers per variant and the first few (tile, register, lane).
//   gpurun -- './tools/ubench/ldsread_srcc'
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define ESR_STAMP16(i)
#include "../../esr_nerf_amd/csrc/mlp_bf16.hip"

namespace {
constexpr int HT = 6, KS = 12, CH = KS * HT;                 // 72 one-KB chunks = one 192x192 bf16 layer
template <int MODE>
__global__ void __launch_bounds__(64) probe(const __bf16 *w, const float *bias, const __bf16 *x, float *out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char wl[];
    float *bl = reinterpret_cast<float *>(wl + CH * 1024);
    const int lane = threadIdx.x;
    for (int i = lane; i < CH * 64; i += 64) reinterpret_cast<u32x4 *>(wl)[i] = reinterpret_cast<const u32x4 *>(w)[i];
    for (int i = lane; i < HT * 32; i += 64) bl[i] = bias[i];
    __syncthreads();
    bf16x8 hb[KS];
    for (int j = 0; j < KS; ++j) hb[j] = reinterpret_cast<const bf16x8 *>(x)[j * 64 + lane];
    f32x16 acc[HT];
    if (MODE == 1) zero_tiles<HT>(acc);
    else {
        const float4 *b4 = reinterpret_cast<const float4 *>(bl + (lane >> 5) * 16);
#pragma unroll
        for (int it = 0; it < HT; ++it)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = b4[it * 8 + q];
                acc[it][4 * q + 0] = v.x; acc[it][4 * q + 1] = v.y; acc[it][4 * q + 2] = v.z; acc[it][4 * q + 3] = v.w;
            }
        if (MODE == 2) { __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 4"); __builtin_amdgcn_sched_barrier(0); }
    }
    lds_layer16<KS, HT, 0>(wl, [&](int j) { return hb[j]; }, acc, lane, lane, make_rsrc(w, CH * 1024), 0, wl);
    if (MODE == 1) lds_bias_add<HT>(bl, acc, lane);
    for (int it = 0; it < HT; ++it)
        for (int r = 0; r < 16; ++r) out[(it * 16 + r) * 64 + lane] = acc[it][r];
}
}  // namespace

int main()
{
    std::vector<float> wf(CH * 512), xf(KS * 512), bf(HT * 32);
    srand(7);
    for (auto &v : wf) v = (float)((rand() % 17) - 8) / 8.f;          // exactly representable in bf16
    for (auto &v : xf) v = (float)((rand() % 9) - 4) / 4.f;
    for (auto &v : bf) v = (float)((rand() % 33) - 16) / 16.f;
    std::vector<__bf16> w16(wf.size()), x16(xf.size());
    for (size_t i = 0; i < wf.size(); ++i) w16[i] = (__bf16)wf[i];
    for (size_t i = 0; i < xf.size(); ++i) x16[i] = (__bf16)xf[i];
    // host evaluation in the packed layouts: chunk n = j * HT + it holds A[lane][8] = rows (lane & 31), k block (lane >> 5);
    // B operand hb[j][lane][8] = column (lane & 31), k block (lane >> 5); accumulator register r of lane: row acc_row(r, lane >> 5),
    // column lane & 31; bias in accumulator order bl[it * 32 + (lane >> 5) * 16 + r]
    std::vector<float> want(HT * 16 * 64);
    for (int it = 0; it < HT; ++it)
        for (int r = 0; r < 16; ++r)
            for (int lane = 0; lane < 64; ++lane) {
                const int h = lane >> 5, col = lane & 31, row = acc_row(r, h);
                double s = bf[it * 32 + h * 16 + r];
                for (int j = 0; j < KS; ++j)
                    for (int kb = 0; kb < 2; ++kb)
                        for (int i = 0; i < 8; ++i)
                            s += (double)wf[((j * HT + it) * 64 + kb * 32 + row) * 8 + i] * xf[(j * 64 + kb * 32 + col) * 8 + i];
                want[(it * 16 + r) * 64 + lane] = (float)s;
            }
    __bf16 *dw, *dx; float *db, *dout;
    (void)hipMalloc(&dw, w16.size() * 2); (void)hipMalloc(&dx, x16.size() * 2); (void)hipMalloc(&db, bf.size() * 4);
    (void)hipMalloc(&dout, want.size() * 4);
    (void)hipMemcpy(dw, w16.data(), w16.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(dx, x16.data(), x16.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(db, bf.data(), bf.size() * 4, hipMemcpyHostToDevice);
    const size_t lds = CH * 1024 + HT * 32 * 4;
    const char *names[3] = {"A: acc <- ds_read_b128 (bias), then MFMAs", "B: acc <- 0, MFMAs, += bias (product form)",
                            "C: as A + lgkmcnt(0) + s_nop 4 before the first MFMA"};
    for (int mode = 0; mode < 3; ++mode) {
        (void)hipMemset(dout, 0xff, want.size() * 4);
        void (*k)(const __bf16 *, const float *, const __bf16 *, float *) = mode == 0 ? probe<0> : mode == 1 ? probe<1> : probe<2>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), lds, 0, dw, db, dx, dout);
        if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", names[mode]); continue; }
        std::vector<float> got(want.size());
        (void)hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (size_t i = 0; i < got.size(); ++i)
            if (got[i] != want[i]) {
                if (bad < 6) printf("    tile %zu register %zu lane %zu: got %g want %g\n", i / 1024, (i / 64) % 16, i % 64, got[i], want[i]);
                ++bad;
            }
        printf("%s: %d of %zu accumulator values differ from the host sums\n", names[mode], bad, got.size());
    }
    return 0;
}
