// tests/native/pool_host.cpp -- the host thread pool of dentist_amd/csrc/dh_parallel.h on its own (test infrastructure):
// many parallel regions of all shapes from one caller and from several callers at once, every index visited exactly once.
#include "../../dentist_amd/csrc/dh_parallel.h"

#include <cstdint>
#include <thread>
#include <vector>

// regions of n = 0 .. nmax (step `nstep`) units with several grains; returns the number of regions whose visit counts were
// wrong (0 = fine); *regions = how many were run
extern "C" int32_t dh_pool_host_regions(int32_t nmax, int32_t nstep, int64_t *regions)
{
    int32_t bad = 0;
    int64_t done = 0;
    std::vector<uint8_t> seen;
    for (int32_t n = 0; n <= nmax; n += nstep)
        for (int64_t grain : {1ll, 3ll, 64ll, 1ll << 20}) {
            seen.assign((size_t)n, 0);
            std::atomic<int64_t> calls{0};
            dh_parallel_for(n, grain, [&](int64_t lo, int64_t hi) {
                calls++;
                if (lo < 0 || hi > n || lo >= hi) {
                    calls += 1 << 20;
                    return;
                }
                for (int64_t i = lo; i < hi; i++) seen[(size_t)i]++;  // (disjoint chunks: no two threads on one index)
            });
            bool ok = calls.load() < (1 << 20);
            for (int32_t i = 0; i < n; i++) ok = ok && seen[(size_t)i] == 1;
            bad += ok ? 0 : 1;
            done++;
        }
    if (regions) *regions = done;
    return bad;
}

// `callers` threads run `reps` regions each at the same time (the pool serves one region at a time): sum of 0 .. n-1 per region
extern "C" int32_t dh_pool_host_concurrent(int32_t callers, int32_t reps, int32_t n)
{
    std::atomic<int32_t> bad{0};
    std::vector<std::thread> th;
    for (int32_t c = 0; c < callers; c++)
        th.emplace_back([&, c] {
            for (int32_t r = 0; r < reps; r++) {
                std::atomic<int64_t> sum{0};
                dh_parallel_for(n + c, 7, [&](int64_t lo, int64_t hi) {
                    int64_t s = 0;
                    for (int64_t i = lo; i < hi; i++) s += i;
                    sum += s;
                });
                const int64_t m = n + c;
                if (sum.load() != m * (m - 1) / 2) bad++;
            }
        });
    for (auto &t : th) t.join();
    return bad.load();
}

// regions of few and of many chunks in turn, empty bodies but a visit count per index: a worker that is still looking at the
// last region when the next one is set up must not take a chunk of it under the old region's terms (the claim race of the
// first compare-and-swap version: a chunk index beyond the old count, executed twice).  Returns the number of bad regions.
extern "C" int32_t dh_pool_host_alternate(int32_t reps, int32_t nsmall, int32_t nlarge)
{
    int32_t bad = 0;
    std::vector<std::atomic<int32_t>> seen((size_t)nlarge);
    for (int32_t r = 0; r < reps; r++) {
        const int32_t n = (r & 1) ? nlarge : nsmall;
        for (int32_t i = 0; i < n; i++) seen[(size_t)i].store(0, std::memory_order_relaxed);
        dh_parallel_for(n, 1, [&](int64_t lo, int64_t hi) {
            for (int64_t i = lo; i < hi; i++) seen[(size_t)i].fetch_add(1, std::memory_order_relaxed);
        });
        bool ok = true;
        for (int32_t i = 0; i < n; i++) ok = ok && seen[(size_t)i].load(std::memory_order_relaxed) == 1;
        bad += ok ? 0 : 1;
    }
    return bad;
}
