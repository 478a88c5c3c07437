// Does an LDS-DMA (buffer_load_dwordx4 ... lds) reach LDS addresses beyond 64 KiB on gfx950 (M0 wider than 16 bits)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef __attribute__((address_space(3))) void lds_void;
__global__ void __launch_bounds__(64) k(const float *g, float *out, int lds_floats_off)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 160 * 256 - 256; i += 64) lds[i] = -1.f;
    __syncthreads();
    rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(g), 0, 64 * 16, 0x00020000);
    lds_void *dst = (lds_void *)(lds + lds_floats_off);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 16, threadIdx.x * 16, 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0f70 & ~0xf);   // vmcnt(0)
    __syncthreads();
    for (int i = threadIdx.x; i < 160 * 256 - 256; i += 64) out[i] = lds[i];
}
int main()
{
    const int NF = 160 * 256 - 256;      // floats of LDS used (just under 160 KiB)
    std::vector<float> h(256), o(NF);
    for (int i = 0; i < 256; ++i) h[i] = 1000.f + i;
    float *g, *out;
    hipMalloc(&g, 1024); hipMalloc(&out, NF * 4);
    hipMemcpy(g, h.data(), 1024, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, NF * 4);
    for (int off : {0, 12288, 16384 + 256, 30000 / 4 * 4, 36864 - 256}) {
        k<<<1, 64, NF * 4>>>(g, out, off);
        hipMemcpy(o.data(), out, NF * 4, hipMemcpyDeviceToHost);
        int first = -1, cnt = 0;
        for (int i = 0; i < NF; ++i) if (o[i] != -1.f) { if (first < 0) first = i; ++cnt; }
        printf("dst float offset %6d (byte %7d): landed at float %6d, %d values, first value %.0f  %s\n", off, off * 4, first, cnt,
               first >= 0 ? o[first] : 0.f, first == off && cnt == 256 ? "OK" : "MISMATCH");
    }
    return 0;
}
