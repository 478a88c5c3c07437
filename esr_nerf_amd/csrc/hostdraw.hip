// Host-side helper (no device code): the surface-point draw of the LTS / PDRA renderer,
//   idx = np.random.choice(M3, num_ltspts, replace=False)          (app/fine/model/esrnerf.py:792, :470)
// which numpy's legacy RandomState evaluates as permutation(M3)[:k]: a full Fisher-Yates shuffle of arange(M3) driven
// by MT19937 through random_interval (masked rejection on 32-bit outputs).  That is 1.6 ms of host time at C4's 80 k
// surviving samples -- GPU idle, because the light-transport pass cannot be enqueued before the indices exist.  This
// is the same algorithm on the same generator state, bit for bit, callable without the GIL so that it runs on a
// worker thread beside the enqueueing of the primary pass (the count M3 is known right after the plan sync).
// The caller checks numpy's global state out (np.random.get_state) and back in (set_state) around the call.
#include <cstdint>
#include <vector>

#include "esr_common.h"

namespace {

struct Mt19937 {
    uint32_t *key;   // 624 words
    int pos;
    void refill()
    {
        constexpr int N = 624, M = 397;
        constexpr uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MATRIX_A = 0x9908b0dfu;
        int i;
        uint32_t y;
        for (i = 0; i < N - M; ++i) {
            y = (key[i] & UPPER) | (key[i + 1] & LOWER);
            key[i] = key[i + M] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
        }
        for (; i < N - 1; ++i) {
            y = (key[i] & UPPER) | (key[i + 1] & LOWER);
            key[i] = key[i + (M - N)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
        }
        y = (key[N - 1] & UPPER) | (key[0] & LOWER);
        key[N - 1] = key[M - 1] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
        pos = 0;
    }
    uint32_t next32()
    {
        if (pos == 624) refill();
        uint32_t y = key[pos++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
    uint64_t next64() { const uint64_t hi = next32(); return (hi << 32) | next32(); }
    // numpy/random/src/distributions/distributions.c::random_interval
    uint64_t interval(uint64_t max)
    {
        if (max == 0) return 0;
        uint64_t mask = max, value;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
        if (max <= 0xffffffffull) {
            while ((value = (next32() & mask)) > max) {}
        } else {
            while ((value = (next64() & mask)) > max) {}
        }
        return value;
    }
};

}  // namespace

// key: the 624 state words (updated in place), *pos: position in the block (0..624, updated).
// out[0..k) = np.random.RandomState(state).choice(n, k, replace=False).
ESR_API int esr_host_choice_noreplace(uint32_t *key, int32_t *pos, int64_t n, int64_t k, int64_t *out)
{
    if (!key || !pos || !out || n < 0 || k < 0 || k > n || *pos < 0 || *pos > 624) return ESR_EINVAL;
    Mt19937 g{key, *pos};
    std::vector<int64_t> a((size_t)n);
    for (int64_t i = 0; i < n; ++i) a[(size_t)i] = i;
    for (int64_t i = n - 1; i >= 1; --i) {               // mtrand.pyx::_shuffle_raw
        const int64_t j = (int64_t)g.interval((uint64_t)i);
        const int64_t t = a[(size_t)j];
        a[(size_t)j] = a[(size_t)i];
        a[(size_t)i] = t;
    }
    for (int64_t i = 0; i < k; ++i) out[i] = a[(size_t)i];
    *pos = g.pos;
    return 0;
}
