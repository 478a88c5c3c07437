// Brick-sparse view of the dense-grid gradient buffer, for the data-parallel gradient exchange.
//
// The reference is single-process; under ray sharding every rank must sum its dense-grid
// gradients (sdf + colour grids: 218 MB fp32 at C2, > 99 % of the gradient payload) with the other
// ranks' once per step.  A ray batch touches a fraction of the grid (C2: the 2x2 colour columns of
// 4096 rays = 25 % of the colour cells), so the flat buffer is cut into 512-B BRICKS (128 floats; a
// z-column of the channels-last colour grid is a whole number of bricks at the usual sizes) and only
// bricks that are non-zero on SOME rank travel:
//   esr_brick_flags   one streaming read of the buffer -> one byte per brick (any non-zero value)
//   esr_brick_pack    gather the listed bricks into a dense buffer (what the all-reduce sees); a NEGATIVE index is
//                     an unused slot of a fixed-capacity list (grad_sync.py sizes the list on the host from the
//                     previous step's count, without waiting for this step's): it packs as zeros
//   esr_brick_unpack  scatter the reduced bricks back (negative indices are skipped)
// All three are pure HBM streams (16-B accesses, one 32-lane group per brick); the union of the flags
// over ranks and the brick list come from torch.distributed / torch (esr_nerf_amd/grad_sync.py).
#include "esr_common.h"

namespace {

constexpr int BRICK = 128;           // floats per brick (32 lanes x float4)

__global__ void __launch_bounds__(256) brick_flags_kernel(const float *__restrict__ g, int64_t n, int64_t n_bricks,
                                                          uint8_t *__restrict__ flags)
{
    const int sub = threadIdx.x & 31;                                   // lane inside its 32-lane group
    const int64_t grp0 = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 5;
    const int64_t ngrp = ((int64_t)gridDim.x * blockDim.x) >> 5;
    for (int64_t b = grp0; b < n_bricks; b += ngrp) {
        const int64_t e = b * BRICK + sub * 4;
        bool nz = false;
        if (e + 4 <= n) {
            const float4 v = *reinterpret_cast<const float4 *>(g + e);
            nz = (v.x != 0.f) | (v.y != 0.f) | (v.z != 0.f) | (v.w != 0.f);
        } else {
            for (int64_t i = e; i < n; ++i) nz |= g[i] != 0.f;
        }
        const unsigned long long m = __ballot(nz);
        const unsigned half = (threadIdx.x & 32) ? (unsigned)(m >> 32) : (unsigned)m;
        if (sub == 0) flags[b] = half ? 1 : 0;
    }
}

template <bool UNPACK>
__global__ void __launch_bounds__(256) brick_move_kernel(float *__restrict__ g, int64_t n,
                                                         const int64_t *__restrict__ idx, int64_t n_idx,
                                                         float *__restrict__ packed)
{
    const int sub = threadIdx.x & 31;
    const int64_t grp0 = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 5;
    const int64_t ngrp = ((int64_t)gridDim.x * blockDim.x) >> 5;
    for (int64_t k = grp0; k < n_idx; k += ngrp) {
        const int64_t b = idx[k];
        float4 *pk = reinterpret_cast<float4 *>(packed + k * BRICK + sub * 4);
        if (b < 0) {                                   // unused slot of a fixed-capacity list: packs as zeros, unpacks nowhere
            if (!UNPACK) *pk = make_float4(0.f, 0.f, 0.f, 0.f);
            continue;
        }
        const int64_t e = b * BRICK + sub * 4;
        if (e + 4 <= n) {
            float4 *gp = reinterpret_cast<float4 *>(g + e);
            if (UNPACK) *gp = *pk;
            else *pk = *gp;
        } else {                                                       // the ragged last brick of the buffer
            float t[4] = {0.f, 0.f, 0.f, 0.f};
            if (UNPACK) {
                const float4 v = *pk;
                t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
                for (int i = 0; i < 4; ++i)
                    if (e + i < n) g[e + i] = t[i];
            } else {
                for (int i = 0; i < 4; ++i)
                    if (e + i < n) t[i] = g[e + i];
                *pk = make_float4(t[0], t[1], t[2], t[3]);
            }
        }
    }
}

// ---- the union's brick list, built on the device (one call instead of six torch launches: to(int64), cumsum, where,
// full, arange, scatter_ over ~4e5 - 2e6 bricks inside the step) ------------------------------------------------------
// LIST_BLOCKS workgroups own contiguous chunks of the flags.  Pass 1: flagged bricks per chunk.  Pass 2: every
// workgroup sums the chunk counts in front of it (<= 256 ints), ranks its own flags with a block scan and writes
// idx[rank] = brick for rank < cap; slots [total, cap) get -1; workgroup 0 publishes the total.
constexpr int LIST_BLOCKS = 256, LIST_THREADS = 256;

__device__ __forceinline__ int block_scan_excl(int v, int *tmp, int &total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int u = __shfl_up(inc, off);
        if (lane >= off) inc += u;
    }
    if (lane == 63) tmp[w] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < LIST_THREADS / 64; ++k) {
        const int t = tmp[k];
        if (k < w) base += t;
        tot += t;
    }
    __syncthreads();
    total = tot;
    return base + inc - v;
}

__global__ void __launch_bounds__(LIST_THREADS) brick_count_kernel(const uint8_t *__restrict__ flags, int64_t nb,
                                                                   int *__restrict__ counts)
{
    __shared__ int tmp[LIST_THREADS / 64];
    const int64_t chunk = (nb + LIST_BLOCKS - 1) / LIST_BLOCKS, per = (chunk + LIST_THREADS - 1) / LIST_THREADS;
    const int64_t b0 = blockIdx.x * chunk, e0 = min(b0 + chunk, nb);
    const int64_t lo = min(b0 + threadIdx.x * per, e0), hi = min(lo + per, e0);
    int c = 0;
    for (int64_t i = lo; i < hi; ++i) c += flags[i] != 0;
    int total;
    block_scan_excl(c, tmp, total);
    if (threadIdx.x == 0) counts[blockIdx.x] = total;
}

__global__ void __launch_bounds__(LIST_THREADS) brick_list_kernel(const uint8_t *__restrict__ flags, int64_t nb, int64_t cap,
                                                                  const int *__restrict__ counts, int64_t *__restrict__ idx,
                                                                  int64_t *__restrict__ count_out)
{
    __shared__ int tmp[LIST_THREADS / 64];
    // chunk counts in front of this workgroup, and all of them
    int mine = 0, all = 0;
    {
        const int c = counts[threadIdx.x];                  // LIST_BLOCKS == LIST_THREADS
        int tot;
        const int ex = block_scan_excl(c, tmp, tot);
        __shared__ int base_s;
        if (threadIdx.x == blockIdx.x) base_s = ex;
        __syncthreads();
        mine = base_s;
        all = tot;
    }
    const int64_t chunk = (nb + LIST_BLOCKS - 1) / LIST_BLOCKS, per = (chunk + LIST_THREADS - 1) / LIST_THREADS;
    const int64_t b0 = blockIdx.x * chunk, e0 = min(b0 + chunk, nb);
    const int64_t lo = min(b0 + threadIdx.x * per, e0), hi = min(lo + per, e0);
    int c = 0;
    for (int64_t i = lo; i < hi; ++i) c += flags[i] != 0;
    int total;
    int64_t r = mine + block_scan_excl(c, tmp, total);
    for (int64_t i = lo; i < hi; ++i)
        if (flags[i]) {
            if (r < cap) idx[r] = i;
            ++r;
        }
    // unused slots of the fixed-capacity list
    for (int64_t k = all + blockIdx.x * (int64_t)LIST_THREADS + threadIdx.x; k < cap; k += (int64_t)LIST_BLOCKS * LIST_THREADS)
        idx[k] = -1;
    if (blockIdx.x == 0 && threadIdx.x == 0) *count_out = all;
}

}  // namespace

ESR_API int esr_brick_floats(void) { return BRICK; }

ESR_API int64_t esr_brick_list_scratch_ints(void) { return LIST_BLOCKS; }

// idx[0 .. cap): the first `cap` flagged bricks in ascending order, then -1; *count = number of flagged bricks (may exceed
// cap: the caller sends the overflow in a second pass).  scratch: esr_brick_list_scratch_ints() ints.
ESR_API int esr_brick_list(const uint8_t *flags, int64_t n_bricks, int64_t cap, int64_t *idx, int64_t *count,
                           int32_t *scratch, void *stream)
{
    if (n_bricks < 0 || cap < 0) return ESR_EINVAL;
    if (!count || (n_bricks && !flags) || (cap && !idx) || !scratch) return ESR_EINVAL;
    hipStream_t s = esr_stream(stream);
    brick_count_kernel<<<LIST_BLOCKS, LIST_THREADS, 0, s>>>(flags, n_bricks, scratch);
    ESR_CHECK_LAUNCH();
    brick_list_kernel<<<LIST_BLOCKS, LIST_THREADS, 0, s>>>(flags, n_bricks, cap, scratch, idx, count);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_brick_flags(const float *buf, int64_t n, uint8_t *flags, void *stream)
{
    if (n < 0) return ESR_EINVAL;
    if (n == 0) return 0;
    if (!buf || !flags || ((uintptr_t)buf & 15u)) return ESR_EINVAL;
    const int64_t nb = (n + BRICK - 1) / BRICK;
    brick_flags_kernel<<<esr_grid_for(nb * 32, 256, 256 * 16), 256, 0, esr_stream(stream)>>>(buf, n, nb, flags);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_brick_pack(const float *buf, int64_t n, const int64_t *brick_idx, int64_t n_idx, float *packed,
                           void *stream)
{
    if (n < 0 || n_idx < 0) return ESR_EINVAL;
    if (n_idx == 0) return 0;
    if (!buf || !brick_idx || !packed || (((uintptr_t)buf | (uintptr_t)packed) & 15u)) return ESR_EINVAL;
    brick_move_kernel<false><<<esr_grid_for(n_idx * 32, 256, 256 * 16), 256, 0, esr_stream(stream)>>>(
        const_cast<float *>(buf), n, brick_idx, n_idx, packed);
    ESR_CHECK_LAUNCH();
    return 0;
}

ESR_API int esr_brick_unpack(const float *packed, const int64_t *brick_idx, int64_t n_idx, float *buf, int64_t n,
                             void *stream)
{
    if (n < 0 || n_idx < 0) return ESR_EINVAL;
    if (n_idx == 0) return 0;
    if (!buf || !brick_idx || !packed || (((uintptr_t)buf | (uintptr_t)packed) & 15u)) return ESR_EINVAL;
    brick_move_kernel<true><<<esr_grid_for(n_idx * 32, 256, 256 * 16), 256, 0, esr_stream(stream)>>>(
        buf, n, brick_idx, n_idx, const_cast<float *>(packed));
    ESR_CHECK_LAUNCH();
    return 0;
}
