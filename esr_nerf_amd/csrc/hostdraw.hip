// Host-side helper (no device code): the surface-point draw of the LTS / PDRA renderer,
//   idx = np.random.choice(M3, num_ltspts, replace=False)          (app/fine/model/esrnerf.py:792, :470)
// which numpy's legacy RandomState evaluates as permutation(M3)[:k]: a full Fisher-Yates shuffle of arange(M3) driven
// by MT19937 through random_interval (masked rejection on 32-bit outputs).  That is 1.6 ms of host time at C4's 80 k
// surviving samples -- GPU idle, because the light-transport pass cannot be enqueued before the indices exist.  This
// is the same algorithm on the same generator state, bit for bit, callable without the GIL so that it runs on a
// worker thread beside the enqueueing of the primary pass (the count M3 is known right after the plan sync).
// The caller checks numpy's global state out (np.random.get_state) and back in (set_state) around the call.
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <mutex>
#include <pthread.h>
#include <thread>
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
//
// Two stages (round 3: as the device side of the primary pass got faster the draw -- 1.2 ms for C5's 157 k survivors -- became
// the step's critical path: the GPU sat idle for 0.46 of 4.4 ms waiting for the indices).  Stage 1 turns generator output
// into the ACCEPTED draws j_i, i = n-1 .. 1: a block of 624 outputs is tempered in one vectorisable loop, the
// rejection test is branch-free (store the candidate, advance only if it was accepted; the mask is constant while i
// stays between two powers of two) -- the one-at-a-time form mispredicted its rejection branch on ~40 % of the draws.
// Stage 2 applies the swaps on a 32-bit array (half the cache footprint), the addresses known in advance.  The generator
// is consumed exactly as numpy does: one 32-bit output per candidate.
ESR_API int esr_host_choice_noreplace(uint32_t *key, int32_t *pos, int64_t n, int64_t k, int64_t *out)
{
    if (!key || !pos || !out || n < 0 || k < 0 || k > n || *pos < 0 || *pos > 624) return ESR_EINVAL;
    Mt19937 g{key, *pos};
    if (n - 1 > 0x7fffffffLL) {                          // (64-bit draws: the plain form)
        std::vector<int64_t> a((size_t)n);
        for (int64_t i = 0; i < n; ++i) a[(size_t)i] = i;
        for (int64_t i = n - 1; i >= 1; --i) {           // mtrand.pyx::_shuffle_raw
            const int64_t j = (int64_t)g.interval((uint64_t)i);
            const int64_t t = a[(size_t)j];
            a[(size_t)j] = a[(size_t)i];
            a[(size_t)i] = t;
        }
        for (int64_t i = 0; i < k; ++i) out[i] = a[(size_t)i];
        *pos = g.pos;
        return 0;
    }
    std::vector<uint32_t> js((size_t)n + 1), a((size_t)n);
    uint32_t buf[624];
    int64_t i = n - 1;
    while (i >= 1) {
        if (g.pos == 624) g.refill();
        const int cnt = 624 - g.pos;
        for (int q = 0; q < cnt; ++q) {
            uint32_t y = key[g.pos + q];
            y ^= (y >> 11);
            y ^= (y << 7) & 0x9d2c5680u;
            y ^= (y << 15) & 0xefc60000u;
            y ^= (y >> 18);
            buf[q] = y;
        }
        int q = 0;
        while (q < cnt && i >= 1) {
            uint32_t mask = (uint32_t)i;                 // smallest 2^b - 1 >= i: constant while i > mask >> 1
            mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
            const int64_t lo = (int64_t)(mask >> 1);
            for (; q < cnt && i > lo; ++q) {
                const uint32_t v = buf[q] & mask;
                js[(size_t)i] = v;                       // (a rejected candidate is overwritten by the next one)
                i -= (v <= (uint32_t)i) ? 1 : 0;
            }
        }
        g.pos += q;
    }
    for (int64_t t = 0; t < n; ++t) a[(size_t)t] = (uint32_t)t;
    for (int64_t t = n - 1; t >= 1; --t) {               // mtrand.pyx::_shuffle_raw
        const uint32_t j = js[(size_t)t], x = a[j];
        a[j] = a[(size_t)t];
        a[(size_t)t] = x;
    }
    for (int64_t t = 0; t < k; ++t) out[t] = (int64_t)a[(size_t)t];
    *pos = g.pos;
    return 0;
}

// ---- the same draw on a worker thread of this library ------------------------------------------------------------------
// The Python side used to hand the call above to a concurrent.futures worker: submitting it, and the worker taking the
// interpreter lock to start the call, cost the enqueueing thread ~0.1 ms right after the plan read-back -- in the one
// segment of the LTS step where the device waits for the host (tools/host_gaps.py).  Here: one detached thread that sleeps
// on a condition variable; start = push a job and notify (microseconds, no interpreter involved), wait = block until that
// job is done (ctypes releases the interpreter lock around it).  Jobs run in submission order.
namespace {
struct DrawJob {
    uint32_t *key;
    int32_t *pos;
    int64_t n, k;
    int64_t *out;
    int rc = 0;
    bool done = false;
};
struct DrawQueue {
    std::mutex m;
    std::condition_variable cv_job, cv_done;
    std::deque<DrawJob *> q;
    bool started = false;
    void loop()
    {
        for (;;) {
            DrawJob *j;
            {
                std::unique_lock<std::mutex> l(m);
                cv_job.wait(l, [&] { return !q.empty(); });
                j = q.front();
                q.pop_front();
            }
            const int rc = esr_host_choice_noreplace(j->key, j->pos, j->n, j->k, j->out);
            {
                std::lock_guard<std::mutex> l(m);
                j->rc = rc;
                j->done = true;
            }
            cv_done.notify_all();
        }
    }
};
DrawQueue *g_draw_queue = nullptr;              // never destroyed: the detached worker may outlive static destruction
std::once_flag g_draw_once;
DrawQueue &draw_queue()
{
    std::call_once(g_draw_once, [] {
        g_draw_queue = new DrawQueue;
        // a fork()ed child has this library's state but not its worker thread (and possibly a locked mutex): fresh queue,
        // the worker starts again with the child's first job
        pthread_atfork(nullptr, nullptr, [] { g_draw_queue = new DrawQueue; });
    });
    return *g_draw_queue;
}
}  // namespace

ESR_API int esr_host_choice_start(uint32_t *key, int32_t *pos, int64_t n, int64_t k, int64_t *out, void **job)
{
    if (!job) return ESR_EINVAL;
    *job = nullptr;
    if (!key || !pos || !out || n < 0 || k < 0 || k > n || *pos < 0 || *pos > 624) return ESR_EINVAL;
    DrawJob *j = new DrawJob{key, pos, n, k, out};
    DrawQueue &Q = draw_queue();
    {
        std::lock_guard<std::mutex> l(Q.m);
        if (!Q.started) {
            std::thread([&Q] { Q.loop(); }).detach();
            Q.started = true;
        }
        Q.q.push_back(j);
    }
    Q.cv_job.notify_one();
    *job = j;
    return 0;
}

ESR_API int esr_host_choice_wait(void *job)
{
    if (!job) return ESR_EINVAL;
    DrawJob *j = static_cast<DrawJob *>(job);
    DrawQueue &Q = draw_queue();
    int rc;
    {
        std::unique_lock<std::mutex> l(Q.m);
        Q.cv_done.wait(l, [&] { return j->done; });
        rc = j->rc;
    }
    delete j;
    return rc;
}
