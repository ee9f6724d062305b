// dh_parallel.h -- a tiny persistent thread pool for the host-side loops between kernel launches
// (per-read funnels, record sorts).  dh_parallel_for(n, grain, fn) calls fn(lo, hi) on disjoint
// chunks of [0, n) from up to DH_HOST_THREADS (default min(64, cores / ranks of the node)) threads, the caller included,
// and returns when all chunks are done.  fn must not throw.
#pragma once
#include <atomic>
#include <chrono>
#include <climits>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>

class DhPool {
public:
    static DhPool &get()
    {
        static DhPool p;
        return p;
    }
    void run(int64_t n, int64_t grain, const std::function<void(int64_t, int64_t)> &fn)
    {
        if (n <= 0) return;
        if (grain < 1) grain = 1;
        while ((n + grain - 1) / grain > (1ll << 31)) grain *= 2;  // (a chunk index is the low half of the ticket)
        const int64_t nchunks = (n + grain - 1) / grain;
        if (nthreads_ <= 1 || nchunks <= 1) {
            fn(0, n);
            return;
        }
        std::unique_lock<std::mutex> serial(serial_);  // one parallel region at a time
        // the ticket is CLOSED (generation g, no chunk left) before the region's fields change: a worker still looking at
        // the last region's ticket -- all of its chunks claimed -- must not take the new region's chunk count for the old
        // one's and claim a chunk beyond the old count (its swap fails now: the word has moved on)
        const uint32_t g = (uint32_t)(ticket_.load(std::memory_order_relaxed) >> 32) + 1u;
        ticket_.store(((uint64_t)g << 32) | 0xFFFFFFFFull, std::memory_order_seq_cst);
        fn_.store(&fn, std::memory_order_relaxed);
        n_.store(n, std::memory_order_relaxed);
        grain_.store(grain, std::memory_order_relaxed);
        done_.store(0, std::memory_order_relaxed);
        nchunks_.store(nchunks, std::memory_order_release);  // (who sees this count sees the closed ticket, or a later one)
        ticket_.store((uint64_t)g << 32, std::memory_order_release);  // generation g, next chunk 0
        gen_.store(g);  // (seq_cst against the sleepers' count: a worker either sees the new value or is counted)
        // (one call wakes all sleepers: 45 us for 63 of them before the caller's first chunk.  Waking four and letting every
        // thread that claims a chunk wake two more was slower on the 256-core hosts -- empty regions 57-82 against 46-56 us,
        // the sharded plan 4.9-5.4 against 3.6-4.4 ms)
        if (sleepers_.load() > 0) wake(INT_MAX);
        work(g);
        // the region is over when its CHUNKS are done, not when every worker has shown up: a worker that wakes late (or not
        // before the next region) finds nothing to claim and is waited for by nobody.  The caller has nothing else to do:
        // it spins for the chunks still running, then yields
        for (int spins = 0; done_.load(std::memory_order_acquire) != nchunks; spins++) {
            if (spins < 4096)
                cpu_relax();
            else
                std::this_thread::yield();
        }
        fn_.store(nullptr, std::memory_order_relaxed);
    }

private:
    // Workers sleep on the generation word itself (futex): a wake-up is one system call and the woken threads meet at no
    // mutex -- with a condition variable the 63 workers of a 64-thread pool queued up at its mutex twice per region, once
    // to sleep and once woken (130-150 us per region on the 256-core hosts of the pool whatever its body,
    // scripts/dev/pool_probe.cpp).  Chunks are claimed by compare-and-swap on (generation, next chunk) in ONE word: a
    // claim that succeeds is a chunk of the generation the worker read the region's fields for -- the caller changes them
    // only when every chunk of the generation is done and after it has closed the ticket of the next generation, so the
    // word has moved on and the swap of a worker that read a field of the next region fails.
    void wake(int count) { syscall(SYS_futex, (uint32_t *)&gen_, FUTEX_WAKE_PRIVATE, count, nullptr, nullptr, 0); }
    static uint64_t ticks()
    {
#if defined(__x86_64__) || defined(__i386__)
        return __builtin_ia32_rdtsc();
#else
        return (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count() * 5 / 2;
#endif
    }
    static void cpu_relax()
    {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
    }
    DhPool()
    {
        // the cores of the box divided among the ranks of this node (LOCAL_WORLD_SIZE, as torchrun sets it), at most 64:
        // measured on configs[2] with 256 cores -- collect stage 12.0 / 9.5 / 7.5 ms with 16 / 32 / 64 threads
        int want = (int)std::thread::hardware_concurrency();
        int local = 1;
        if (const char *e = getenv("LOCAL_WORLD_SIZE")) local = atoi(e) > 0 ? atoi(e) : 1;
        want /= local;
        if (want > 64) want = 64;
        if (const char *e = getenv("DH_HOST_THREADS")) want = atoi(e);
        if (want < 1) want = 1;
        nthreads_ = want;
        // DH_POOL_SPIN_US=<n>: a worker that has finished a region keeps looking for the next one for n us before it goes
        // to sleep.  Off by default: back-to-back regions cost 8 instead of 45 us on the 256-core hosts of the pool and
        // the plan of the sharded collector 4.4 instead of 5.1 ms, but workers that spin, give up and are woken again were
        // late by milliseconds now and then on these virtual machines (one region of the 8-rank emulation 6.8 ms instead
        // of 0.3), and on an 8-CPU container with 8 threads every region took 1.4 ms.
        spin_us_ = 0;
        if (const char *e = getenv("DH_POOL_SPIN_US")) spin_us_ = atoi(e);
        for (int i = 1; i < want; i++) workers_.emplace_back([this] { loop(); });
    }
    ~DhPool()
    {
        stop_.store(true);
        gen_.fetch_add(1);
        wake(INT_MAX);
        for (auto &t : workers_) t.join();
    }
    void work(uint32_t g)
    {
        for (;;) {
            uint64_t t = ticket_.load(std::memory_order_acquire);
            if ((uint32_t)(t >> 32) != g) return;  // (the region this thread was woken for is over)
            const int64_t c = (int64_t)(uint32_t)t;
            if (c >= nchunks_.load(std::memory_order_acquire)) return;
            if (!ticket_.compare_exchange_weak(t, t + 1, std::memory_order_acq_rel, std::memory_order_acquire)) continue;
            const int64_t n = n_.load(std::memory_order_relaxed), grain = grain_.load(std::memory_order_relaxed);
            const int64_t lo = c * grain;
            (*fn_.load(std::memory_order_relaxed))(lo, lo + grain < n ? lo + grain : n);
            done_.fetch_add(1, std::memory_order_release);
        }
    }
    void loop()
    {
        uint32_t seen = 0;
        for (;;) {
            if (spin_us_ > 0) {
                // (timed by the cycle counter, taken as 2.5 GHz: no clock calls from sixty threads at once)
                const uint64_t t0 = ticks(), limit = (uint64_t)spin_us_ * 2500u;
                for (int i = 0; gen_.load(std::memory_order_acquire) == seen; i++) {
                    cpu_relax();
                    if ((i & 15) == 15 && ticks() - t0 > limit) break;
                }
            }
            while (gen_.load() == seen) {
                sleepers_.fetch_add(1);
                syscall(SYS_futex, (uint32_t *)&gen_, FUTEX_WAIT_PRIVATE, seen, nullptr, nullptr, 0);  // (returns at once if the word has moved on)
                sleepers_.fetch_sub(1);
            }
            seen = gen_.load(std::memory_order_acquire);
            if (stop_.load()) return;
            work(seen);
        }
    }
    std::vector<std::thread> workers_;
    std::mutex serial_;
    int nthreads_ = 1, spin_us_ = 0;
    // the region's fields (written by the caller before the ticket of the generation is published)
    std::atomic<const std::function<void(int64_t, int64_t)> *> fn_{nullptr};
    std::atomic<int64_t> n_{0}, grain_{1}, nchunks_{0};
    // (the words the threads meet at, each on a cache line of its own)
    alignas(64) std::atomic<uint32_t> gen_{0};
    alignas(64) std::atomic<uint64_t> ticket_{0};  // generation << 32 | next chunk
    alignas(64) std::atomic<int64_t> done_{0};     // chunks of the generation that have returned
    alignas(64) std::atomic<int> sleepers_{0};
    alignas(64) std::atomic<bool> stop_{false};
};
static_assert(sizeof(std::atomic<uint32_t>) == sizeof(uint32_t), "the generation word is the futex word");

template <class F>
inline void dh_parallel_for(int64_t n, int64_t grain, F &&fn)
{
    const std::function<void(int64_t, int64_t)> f = std::forward<F>(fn);
    DhPool::get().run(n, grain, f);
}

// Starts of the runs of equal key(i) in [0, n), plus n at the end -- found by the host threads over slices of
// the index range and concatenated in order (a serial pass over a million records is a millisecond nobody hides).
template <class K>
inline std::vector<int64_t> dh_run_starts(int64_t n, K &&key)
{
    const int64_t grain = 1 << 15, nchunks = (n + grain - 1) / grain;
    std::vector<std::vector<int64_t>> part((size_t)(nchunks > 0 ? nchunks : 1));
    dh_parallel_for(nchunks, 1, [&](int64_t clo, int64_t chi) {
        for (int64_t c = clo; c < chi; c++) {
            std::vector<int64_t> &v = part[(size_t)c];
            const int64_t i1 = (c + 1) * grain < n ? (c + 1) * grain : n;
            for (int64_t i = c * grain; i < i1; i++)
                if (i == 0 || key(i) != key(i - 1)) v.push_back(i);
        }
    });
    std::vector<int64_t> out;
    size_t total = 1;
    for (const auto &v : part) total += v.size();
    out.reserve(total);
    for (const auto &v : part) out.insert(out.end(), v.begin(), v.end());
    out.push_back(n);
    return out;
}
