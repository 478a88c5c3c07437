// Shared device helpers for libesr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/esr_hip.h"

#define ESR_API extern "C" __attribute__((visibility("default")))

#define ESR_CHECK_LAUNCH()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return (int)e__;              \
    } while (0)

#define ESR_WAVE 64

static inline hipStream_t esr_stream(void *s) { return (hipStream_t)s; }

// Opt-in for more than 64 KB of dynamic LDS (hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of
// a kernel): set once per (kernel, device).  `done` = one static bit mask per kernel instantiation; the only state the
// library keeps is this idempotent "attribute already set" memo.
static inline int esr_lds_optin(const void *kernel, size_t bytes, std::atomic<uint64_t> &done)
{
    if (bytes <= 64 * 1024) return 0;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return 0;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return (int)e;
    done.fetch_or(bit, std::memory_order_release);
    return 0;
}

static inline int esr_grid_for(int64_t n, int block, int cap = 256 * 8)
{
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

// ---------------------------------------------------------------------------
// Ray / box geometry.  Every operation here is a separately rounded binary32
// op (contraction off) so sample positions and the integer outputs derived
// from them are bit-identical to oracle/esr_oracle.c.
// ---------------------------------------------------------------------------
struct RayGeom {
    float start[3];
    float dir[3];
    float t_min, t_max;
    int n_steps;
};

__device__ __forceinline__ float esr_ray_norm(const float d[3])
{
#pragma clang fp contract(off)
    float s = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    return sqrtf(s);   // correctly rounded (NOT __fsqrt_rn: that lowers to bare v_sqrt_f32, ~1 ulp)
}

__device__ __forceinline__ void esr_ray_trange(const float o[3], const float d[3],
                                               const float bmin[3], const float bmax[3],
                                               float near_, float far_, float &tmin, float &tmax)
{
#pragma clang fp contract(off)
    float lo = 0.f, hi = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float v = (d[a] == 0.0f) ? (float)1e-6 : d[a];
        float ta = __fdiv_rn(bmax[a] - o[a], v);
        float tb = __fdiv_rn(bmin[a] - o[a], v);
        float mn = fminf(ta, tb), mx = fmaxf(ta, tb);
        if (a == 0) { lo = mn; hi = mx; }
        else        { lo = fmaxf(lo, mn); hi = fminf(hi, mx); }
    }
    tmin = fmaxf(fminf(lo, far_), near_);
    tmax = fmaxf(fminf(hi, far_), near_);
}

__device__ __forceinline__ int64_t esr_ray_nsteps(float tmin, float tmax, float nrm, float stepdist)
{
#pragma clang fp contract(off)
    float len = __fdiv_rn((tmax - tmin) * nrm, stepdist);
    double c = (double)ceilf(len);
    return (int64_t)(c > 1.0 ? c : 1.0);
}

__device__ __forceinline__ void esr_ray_start_dir(const float o[3], const float d[3], float tmin,
                                                  float nrm, float start[3], float dir[3])
{
#pragma clang fp contract(off)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        start[a] = o[a] + d[a] * tmin;
        dir[a] = __fdiv_rn(d[a], nrm);
    }
}

__device__ __forceinline__ void esr_ray_point(const float start[3], const float dir[3],
                                              float stepdist, int step, float p[3])
{
#pragma clang fp contract(off)
    float dist = stepdist * (float)step;
#pragma unroll
    for (int a = 0; a < 3; ++a) p[a] = start[a] + dir[a] * dist;
}

__device__ __forceinline__ bool esr_out_of_box(const float p[3], const float bmin[3], const float bmax[3])
{
    return (bmin[0] > p[0]) | (bmin[1] > p[1]) | (bmin[2] > p[2]) |
           (bmax[0] < p[0]) | (bmax[1] < p[1]) | (bmax[2] < p[2]);
}

__device__ __forceinline__ RayGeom esr_ray_geom(const float *rays_o, const float *rays_d, int r,
                                                const float bmin[3], const float bmax[3],
                                                float near_, float far_, float stepdist)
{
    RayGeom g;
    float o[3] = {rays_o[3 * r], rays_o[3 * r + 1], rays_o[3 * r + 2]};
    float d[3] = {rays_d[3 * r], rays_d[3 * r + 1], rays_d[3 * r + 2]};
    esr_ray_trange(o, d, bmin, bmax, near_, far_, g.t_min, g.t_max);
    float nrm = esr_ray_norm(d);
    g.n_steps = (int)esr_ray_nsteps(g.t_min, g.t_max, nrm, stepdist);
    esr_ray_start_dir(o, d, g.t_min, nrm, g.start, g.dir);
    return g;
}

// ---------------------------------------------------------------------------
// Trilinear lookup with the arithmetic of F.grid_sample(mode="bilinear",
// align_corners=True, padding_mode="zeros") on a [X,Y,Z] volume, as the
// reference calls it (app/utils/base/module.py:24-35): world x -> slowest axis.
// ---------------------------------------------------------------------------
// sin and cos of one argument with ONE range reduction (ocml's sincosf: the same reduction and polynomials as its sinf /
// cosf, so the values are theirs bit for bit); the positional encodings evaluate 18-30 such pairs per sample and the
// two separate calls were most of the feature kernel's vector instructions
__device__ __forceinline__ void esr_sincos(float a, float &s, float &c)
{
#ifdef ESR_EXP_SEPARATE_SINCOS
    s = sinf(a); c = cosf(a);
#else
    sincosf(a, &s, &c);
#endif
}

struct Tri {
    int i0[3];      // floor index per GRID axis (0 = X slowest, 2 = Z fastest)
    float f[3];     // fractional part per grid axis
};

// world point -> continuous grid index per grid axis, replicating
// ((p-min)/(max-min))*2-1 followed by ((n+1)/2)*(size-1)
__device__ __forceinline__ void esr_world_to_index(const float p[3], const float bmin[3],
                                                   const float bmax[3], const int dims[3], float idx[3])
{
#pragma clang fp contract(off)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float u = __fdiv_rn(p[a] - bmin[a], bmax[a] - bmin[a]);
        float n = u * 2.0f - 1.0f;
        idx[a] = __fdiv_rn(n + 1.0f, 2.0f) * (float)(dims[a] - 1);
    }
}

__device__ __forceinline__ Tri esr_tri_setup(const float idx[3])
{
    Tri t;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float fl = floorf(idx[a]);
        t.i0[a] = (int)fl;
        t.f[a] = idx[a] - fl;
    }
    return t;
}

// corner weight in ATen's spelling: (x1 - ix) for the low corner, (ix - x0) for the high one
__device__ __forceinline__ float esr_corner_w(const Tri &t, const float idx[3], int cx, int cy, int cz)
{
    float wx = cx ? (idx[0] - (float)t.i0[0]) : ((float)(t.i0[0] + 1) - idx[0]);
    float wy = cy ? (idx[1] - (float)t.i0[1]) : ((float)(t.i0[1] + 1) - idx[1]);
    float wz = cz ? (idx[2] - (float)t.i0[2]) : ((float)(t.i0[2] + 1) - idx[2]);
    // ATen multiplies fastest-axis weight first: (wz * wy) * wx
    return (wz * wy) * wx;
}

// Conditional gathers.  A load the compiler may not speculate -- `if (inb) acc += g[i] * w` or `inb ? g[i] : 0` -- becomes an
// exec-masked region that ends in its own s_waitcnt vmcnt(0): the 8 corners of a trilinear fetch were 8 SERIAL memory
// round trips (feat_fwd: ~140 of them per sample, 72 % of its wave time waiting).  Here the address of an out-of-grid
// corner is replaced by element 0 and the value by 0: the loads of a fetch issue back to back, the sum is unchanged
// (x + 0 * w == x).
__device__ __forceinline__ float esr_ld_or0(const float *__restrict__ g, int64_t i, bool inb)
{
    const float v = g[inb ? i : 0];
    return inb ? v : 0.f;
}

// 1-channel fetch with zero padding
__device__ __forceinline__ float esr_tri_fetch1(const float *__restrict__ g, const int dims[3],
                                                const float idx[3])
{
    Tri t = esr_tri_setup(idx);
    float acc = 0.f;
#pragma unroll
    for (int cx = 0; cx < 2; ++cx)
#pragma unroll
        for (int cy = 0; cy < 2; ++cy)
#pragma unroll
            for (int cz = 0; cz < 2; ++cz) {
                int x = t.i0[0] + cx, y = t.i0[1] + cy, z = t.i0[2] + cz;
                bool inb = (x >= 0) & (x < dims[0]) & (y >= 0) & (y < dims[1]) & (z >= 0) & (z < dims[2]);
                float w = esr_corner_w(t, idx, cx, cy, cz);
                // (explicit fma: what `if (inb) acc += g[i] * w` contracted to)
                acc = __builtin_fmaf(esr_ld_or0(g, ((int64_t)x * dims[1] + y) * dims[2] + z, inb), w, acc);
            }
    return acc;
}

// scatter-add of one value through the same 8 corners
__device__ __forceinline__ void esr_tri_scatter1(float *__restrict__ g, const int dims[3],
                                                 const float idx[3], float v)
{
    Tri t = esr_tri_setup(idx);
#pragma unroll
    for (int cx = 0; cx < 2; ++cx)
#pragma unroll
        for (int cy = 0; cy < 2; ++cy)
#pragma unroll
            for (int cz = 0; cz < 2; ++cz) {
                int x = t.i0[0] + cx, y = t.i0[1] + cy, z = t.i0[2] + cz;
                bool inb = (x >= 0) & (x < dims[0]) & (y >= 0) & (y < dims[1]) & (z >= 0) & (z < dims[2]);
                float w = esr_corner_w(t, idx, cx, cy, cz);
                if (inb && w != 0.f) atomicAdd(&g[((int64_t)x * dims[1] + y) * dims[2] + z], v * w);
            }
}

// Continuous index of one clamped stencil tap and the clamped coordinate along
// its axis.  Replicates ind + offset -> clamp -> /(size-1)*2-1 -> grid_sample's
// ((n+1)/2)*(size-1) so the tap lands on the same float as the reference's.
__device__ __forceinline__ float tap_index(const float ind[3], const int dims[3], int axis, float disp,
                                           float ix[3])
{
#pragma clang fp contract(off)
    float along = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float t = ind[a] + ((a == axis) ? disp : 0.f);
        t = fminf(fmaxf(t, 0.f), (float)(dims[a] - 1));
        if (a == axis) along = t;
        float n = __fdiv_rn(t, (float)(dims[a] - 1)) * 2.0f - 1.0f;
        ix[a] = __fdiv_rn(n + 1.0f, 2.0f) * (float)(dims[a] - 1);
    }
    return along;
}

__device__ __forceinline__ float esr_softplus(float x)
{
    // F.softplus(beta=1, threshold=20)
    return x > 20.f ? x : log1pf(expf(x));
}
__device__ __forceinline__ float esr_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

__device__ __forceinline__ int esr_lane() { return (int)(threadIdx.x & 63); }

// value of `x` in lane `i` for a wave-uniform i (v_readlane_b32; __shfl would go through the LDS permute network)
static __device__ __forceinline__ float esr_readlane(float x, int i)
{
    return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(x), i));
}

#ifndef ESR_NT_AUX
#define ESR_NT_AUX 2          // gfx940+ buffer cache-policy bits: 1 = sc0, 2 = nt, 16 = sc1
#endif
// Buffer addressing for every hot kernel's tile traffic.  Besides the addressing economy described below, buffer STORES
// retire much faster than plain global stores here: a wave's later loads wait for its earlier stores (vmcnt retires in
// order and counts stores), and 6 `global_store_dword` per tile at the end of the MLP input-gradient kernel cost 31 us
// per launch against ~0 for the same rows through a descriptor (profiles/r02_n_*).
// Buffer addressing (SGPR descriptor + per-lane 32-bit offset + scalar constant offset):
// with plain pointers hipcc materialises ~100 loop-invariant 64-bit addresses per kernel
// (one per store/load slot) and spills kilobytes per lane.  Out-of-range accesses are
// dropped by the hardware range check instead of faulting.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

static __device__ __forceinline__ rsrc_t make_rsrc(const void *p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}
static __device__ __forceinline__ float4 bload4(rsrc_t r, int voff, int soff)
{
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
static __device__ __forceinline__ float bload1(rsrc_t r, int voff, int soff)
{
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
static __device__ __forceinline__ void bstore1(rsrc_t r, float v, int voff, int soff)
{
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 0);
}
// streaming store (nt): saved activations / gradients are written once and read by a later kernel; keeping
// them out of the L2's working set leaves it to the packed weights that every wave re-reads
static __device__ __forceinline__ void bstore1_nt(rsrc_t r, float v, int voff, int soff)
{
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, ESR_NT_AUX);
}

// ---- LDS-DMA (buffer_load ... lds) and hand-counted waits: used by the weight-gradient kernels (mlp.hip) and the
// shared-weights bf16 forward / input-gradient kernels (mlp_bf16.hip) ----
static constexpr int wait_vm_lgkm0(int n) { return (n & 15) | 0x70 | ((n >> 4) << 14); }           // vmcnt(n) lgkmcnt(0)
static constexpr int wait_vm(int n) { return (n & 15) | 0x70 | (15 << 8) | ((n >> 4) << 14); }     // vmcnt(n)

// The DMA is issued from inline asm on purpose: hipcc orders every later ds_read behind a compiler-visible
// LDS-DMA with `s_waitcnt vmcnt(0)` (it cannot tell the buffers apart), which drains the prefetch every step.
// An asm load is absent from its bookkeeping; its completion is counted by hand below.  M0 (the LDS
// destination base, wide enough for all 160 KB: tools/ubench/lds_dma_m0.hip) is compiler-reserved, so it is
// saved and restored inside the statement that uses it.
static __device__ __forceinline__ u32x4 raw_rsrc(const void *p, unsigned bytes)
{
    const uint64_t a = (uint64_t)(uintptr_t)p;
    return u32x4{(unsigned)a, (unsigned)(a >> 32) & 0xffffu, bytes, 0x00020000u};
}
static __device__ __forceinline__ void lds_dma16(u32x4 rsrc, unsigned lds_byte, int voff)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_byte), "s"(rsrc) : "memory");
}


// (mlp_split.hip) the current device's registered range flag of the split-fp16 forward kernels, or NULL
unsigned *esr_split_range_flag_ptr();
