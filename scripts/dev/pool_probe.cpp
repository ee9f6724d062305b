// dev: what one dh_parallel_for region costs on this host (empty bodies; 64 chunks), and with 200 us of serial work between regions
// g++ -O2 -std=c++17 -pthread -I dentist_amd/csrc scripts/dev/pool_probe.cpp -o /tmp/pool_probe
#include "dh_parallel.h"
#include <chrono>
#include <cstdio>
int main()
{
    using clk = std::chrono::steady_clock;
    std::atomic<int64_t> sink{0};
    for (int gap_us : {0, 200, 2000}) {
        dh_parallel_for(64, 1, [&](int64_t lo, int64_t hi) { sink += hi - lo; });
        double in_region = 0;
        const int reps = 200;
        for (int i = 0; i < reps; i++) {
            const auto t0 = clk::now();
            dh_parallel_for(64, 1, [&](int64_t lo, int64_t hi) { sink += hi - lo; });
            in_region += std::chrono::duration<double, std::micro>(clk::now() - t0).count();
            const auto t1 = clk::now();
            while (std::chrono::duration<double, std::micro>(clk::now() - t1).count() < gap_us) {}
        }
        printf("gap %4d us between regions: %.1f us per empty region of 64 chunks\n", gap_us, in_region / reps);
    }
    return (int)(sink.load() & 1);
}
