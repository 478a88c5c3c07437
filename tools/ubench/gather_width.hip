// Gathers of z-adjacent grid cells: N separate dword loads vs one dwordx2 / dwordx4 at 4-byte alignment.
// Question 1: does gfx950 serve a dwordx2 / dwordx4 global load whose address is only 4-byte aligned (the z-adjacent corners of
// a trilinear tap start at an arbitrary float)?  Question 2: what does a wave-instruction of each width cost when every lane
// gathers from a different place of a grid larger than L2 (the feature kernels' access pattern)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

template <int MODE>   // 0: 4 dword loads, 1: 2 dwordx2, 2: 1 dwordx4 (unaligned)
__global__ void gather(const float *__restrict__ g, const int *__restrict__ idx, int n, int reps, float *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    for (int r = 0; r < reps; ++r) {
        const int b = idx[(i + r * 977) % n];
        if (MODE == 0) {        // (opaque offsets: the compiler must not merge the four loads into one dwordx4 -- which it does by itself
                                // when it can see that they are adjacent: unaligned wide loads are legal on this target)
            int o1 = 1, o2 = 2, o3 = 3;
            asm volatile("" : "+v"(o1), "+v"(o2), "+v"(o3));
            acc += g[b] + g[b + o1] + g[b + o2] + g[b + o3];
        }
        if (MODE == 1) {
            int o2 = 2;
            asm volatile("" : "+v"(o2));
            const f2u a = *reinterpret_cast<const f2u *>(g + b), c = *reinterpret_cast<const f2u *>(g + b + o2);
            acc += a[0] + a[1] + c[0] + c[1];
        }
        if (MODE == 2) { const f4u a = *reinterpret_cast<const f4u *>(g + b); acc += a[0] + a[1] + a[2] + a[3]; }
    }
    out[i] = acc;
}

int main(int argc, char **argv)
{
    const int N = 64 << 20;                 // 256 MB grid
    const int M = 1 << 22;                  // gather indices
    float *g, *out; int *idx;
    hipMalloc(&g, (size_t)N * 4 + 64); hipMalloc(&out, (size_t)M * 4); hipMalloc(&idx, (size_t)M * 4);
    std::vector<float> h(N);
    for (int i = 0; i < N; ++i) h[i] = (float)(i % 1000) * 0.001f;
    hipMemcpy(g, h.data(), (size_t)N * 4, hipMemcpyHostToDevice);
    std::vector<int> hi(M);
    unsigned s = 12345;
    // argv[1] == "ray": the feature kernels' pattern -- the 64 lanes of a wave are consecutive samples of a ray, half a
    // cell apart along the fastest axis (two lanes per cell, all in one or two cache lines); default: every lane elsewhere
    const bool ray = argc > 1 && argv[1][0] == 'r';
    for (int i = 0; i < M; ++i) {
        if (!ray || (i & 63) == 0) { s = s * 1664525u + 1013904223u; }
        hi[i] = ray ? (int)(s % (unsigned)(N - 64)) + (i & 63) / 2 : (int)(s % (unsigned)(N - 8));
        if (!ray) s = s * 1664525u + 1013904223u;
    }
    hipMemcpy(idx, hi.data(), (size_t)M * 4, hipMemcpyHostToDevice);
    std::vector<float> o0(M), o(M);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 16, blocks = M / 256;
    for (int mode = 0; mode < 3; ++mode) {
        for (int it = 0; it < 3; ++it) {
            hipEventRecord(e0);
            if (mode == 0) gather<0><<<blocks, 256>>>(g, idx, M, reps, out);
            if (mode == 1) gather<1><<<blocks, 256>>>(g, idx, M, reps, out);
            if (mode == 2) gather<2><<<blocks, 256>>>(g, idx, M, reps, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipError_t err = hipGetLastError();
        hipMemcpy(o.data(), out, (size_t)M * 4, hipMemcpyDeviceToHost);
        if (mode == 0) o0 = o;
        double maxd = 0; for (int i = 0; i < M; ++i) maxd = fmax(maxd, fabs((double)o[i] - o0[i]));
        printf("mode %d (%s): %.3f ms for %d x %d gathers of 16 B, err %d, max diff vs dword loads %.3g\n", mode,
               mode == 0 ? "4 x dword" : mode == 1 ? "2 x dwordx2 @4B" : "1 x dwordx4 @4B", ms, M, reps, (int)err, maxd);
    }
    return 0;
}
