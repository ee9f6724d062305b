// rand_access_probe.cpp -- random 64-byte line reads per second against the working-set size
// (the access pattern of the seed kernel's directory / entry lookups).  Dev tool, not product.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/rand_probe scripts/rand_access_probe.cpp && /tmp/rand_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x)                                                                     \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
            return 1;                                                              \
        }                                                                          \
    } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// every thread issues `per_thread` independent loads of LOADB bytes at random 64-byte lines
template <int UNROLL>
__global__ void __launch_bounds__(256) k_probe(const uint64_t *__restrict__ tab, uint64_t nlines, int per_thread,
                                                uint64_t *__restrict__ sink, int window_lines)
{
    uint64_t s = mix(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 1);
    uint64_t acc = 0;
    for (int i = 0; i < per_thread; i += UNROLL) {
        uint64_t v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            uint64_t line = (s >> 20) % nlines;
            if (window_lines) line = (line / window_lines) * window_lines + ((s >> 50) % window_lines);
            v[u] = tab[line * 8 + (threadIdx.x & 7)];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc += v[u];
    }
    if (acc == 0x1234567) sink[0] = acc;
}

// the same with 16-byte loads (ulonglong2: what k_seed reads from its fat directory)
template <int UNROLL>
__global__ void __launch_bounds__(256) k_probe16(const ulonglong2 *__restrict__ tab, uint64_t nlines, int per_thread,
                                                  uint64_t *__restrict__ sink)
{
    uint64_t s = mix(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 1);
    uint64_t acc = 0;
    for (int i = 0; i < per_thread; i += UNROLL) {
        ulonglong2 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            const uint64_t line = (s >> 20) % nlines;
            v[u] = tab[line * 4 + (threadIdx.x & 3)];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc += v[u].x + v[u].y;
    }
    if (acc == 0x1234567) sink[0] = acc;
}

// usage: rand_probe [8 | 16 (bytes per load)] [working set in MB: one size only]
// Under `rocprofv3 --pmc FETCH_SIZE` the per-launch FETCH_SIZE against lines * 64 B (printed) calibrates the counter
// for this access pattern.
int main(int argc, char **argv)
{
    const int loadb = argc > 1 ? atoi(argv[1]) : 8;
    const size_t only_mb = argc > 2 ? (size_t)atoll(argv[2]) : 0;
    const size_t sizes_mb[] = {32, 64, 128, 256, 512, 1024, 2048, 4096, 16384};
    uint64_t *sink;
    CHK(hipMalloc(&sink, 64));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    for (size_t mb : sizes_mb) {
        if (only_mb && mb != only_mb) continue;
        uint64_t *tab;
        const size_t bytes = mb << 20;
        CHK(hipMalloc(&tab, bytes));
        CHK(hipMemset(tab, 1, bytes));
        const uint64_t nlines = bytes / 64;
        const int per_thread = 2048, blocks = 256 * 16;
        for (int rep = 0; rep < 2; rep++) {
            CHK(hipEventRecord(e0));
            if (loadb == 16)
                hipLaunchKernelGGL(k_probe16<8>, dim3(blocks), dim3(256), 0, 0, (const ulonglong2 *)tab, nlines, per_thread, sink);
            else
                hipLaunchKernelGGL(k_probe<8>, dim3(blocks), dim3(256), 0, 0, tab, nlines, per_thread, sink, 0);
            CHK(hipEventRecord(e1));
            CHK(hipEventSynchronize(e1));
        }
        float ms;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        const double n = (double)blocks * 256 * per_thread;
        printf("working set %6zu MB, %d-byte loads: %8.2f G lines/s = %7.1f GB/s of 64 B lines (%.2f ms); per launch %.0f loads = %.0f KB of lines\n",
               mb, loadb, n / ms / 1e6, n * 64 / ms / 1e6, ms, n, n * 64 / 1024);
        CHK(hipFree(tab));
    }
    return 0;
}
