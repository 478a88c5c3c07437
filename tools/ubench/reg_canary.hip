// Does anything write the vector registers of a wave that is only WAITING?  (Round 6: esr_expgrad_fwd, 69 registers, returned wrong
// rows in lanes 48-63 whenever the split-fp16 kernels -- ~440 registers, one wave per SIMD: 512 - 440 = 72 registers left, exactly one
// expgrad wave -- ran beside it.)  Every wave of the canary fills NR registers with known values, sleeps ~20 us, and checks them.
//   ./tools/ubench/reg_canary [launches]        (run it beside a loop of ./tools/ubench/split_stamps, or alone)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NR>
__global__ void __launch_bounds__(256) canary(const float *in, unsigned *bad, int spins)
{
    const int lane = threadIdx.x & 63;
    const float base = in[blockIdx.x * 256 + threadIdx.x];
    float r[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) { r[k] = base + (float)k; asm volatile("" : "+v"(r[k])); }
    for (int s = 0; s < spins; ++s) {
        __builtin_amdgcn_s_sleep(32);
#pragma unroll
        for (int k = 0; k < NR; ++k) asm volatile("" : "+v"(r[k]));
    }
    const float again = *(const volatile float *)&in[blockIdx.x * 256 + threadIdx.x];      // (re-read: the expected values do not stay in registers)
#pragma unroll
    for (int k = 0; k < NR; ++k)
        if (r[k] != again + (float)k) atomicAdd(&bad[k * 4 + (lane >> 4)], 1u);
}

int main(int argc, char **argv)
{
    const int launches = argc > 1 ? atoi(argv[1]) : 20000, spins = argc > 2 ? atoi(argv[2]) : 8;
    constexpr int NR = 60;
    const int wgs = 64;
    std::vector<float> h(wgs * 256);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(i % 977) * 0.5f;
    float *in;
    unsigned *bad;
    CK(hipMalloc(&in, h.size() * 4));  CK(hipMalloc(&bad, NR * 16));
    CK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(bad, 0, NR * 16));
    for (int l = 0; l < launches; ++l) {
        canary<NR><<<wgs, 256>>>(in, bad, spins);
        if (l % 64 == 63) CK(hipDeviceSynchronize());
    }
    CK(hipDeviceSynchronize());
    std::vector<unsigned> hb(NR * 4);
    CK(hipMemcpy(hb.data(), bad, NR * 16, hipMemcpyDeviceToHost));
    unsigned tot = 0;
    for (int k = 0; k < NR; ++k) {
        const unsigned s = hb[4 * k] + hb[4 * k + 1] + hb[4 * k + 2] + hb[4 * k + 3];
        tot += s;
        if (s) printf("value %2d changed: lanes 0-15 %u, 16-31 %u, 32-47 %u, 48-63 %u\n", k, hb[4 * k], hb[4 * k + 1], hb[4 * k + 2], hb[4 * k + 3]);
    }
    printf("%d launches of %d waves: %u changed values in all\n", launches, wgs * 4, tot);
    return 0;
}
