// dh_parallel.h -- a tiny persistent thread pool for the host-side loops between kernel launches
// (per-read funnels, record sorts).  dh_parallel_for(n, grain, fn) calls fn(lo, hi) on disjoint
// chunks of [0, n) from up to DH_HOST_THREADS (default min(64, cores / ranks of the node)) threads, the caller included,
// and returns when all chunks are done.  fn must not throw.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

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
        const int64_t nchunks = (n + grain - 1) / grain;
        if (nthreads_ <= 1 || nchunks <= 1) {
            fn(0, n);
            return;
        }
        std::unique_lock<std::mutex> serial(serial_);  // one parallel region at a time
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = &fn;
            n_ = n;
            grain_ = grain;
            next_.store(0);
            pending_ = (int)workers_.size();
            gen_++;
        }
        cv_.notify_all();
        work();
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [&] { return pending_ == 0; });
        fn_ = nullptr;
    }

private:
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
        for (int i = 1; i < want; i++) workers_.emplace_back([this] { loop(); });
    }
    ~DhPool()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
            gen_++;
        }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }
    void work()
    {
        for (;;) {
            const int64_t c = next_.fetch_add(1);
            const int64_t lo = c * grain_;
            if (lo >= n_) break;
            (*fn_)(lo, lo + grain_ < n_ ? lo + grain_ : n_);
        }
    }
    void loop()
    {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
            }
            work();
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (--pending_ == 0) done_.notify_all();
            }
        }
    }
    std::vector<std::thread> workers_;
    std::mutex mu_, serial_;
    std::condition_variable cv_, done_;
    const std::function<void(int64_t, int64_t)> *fn_ = nullptr;
    std::atomic<int64_t> next_{0};
    int64_t n_ = 0, grain_ = 1;
    int pending_ = 0, nthreads_ = 1;
    uint64_t gen_ = 0;
    bool stop_ = false;
};

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
