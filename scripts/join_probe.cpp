// join_probe.cpp -- design probe for the partitioned k-mer join of the mapping seeds (round 5).  Dev tool, not product.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/join_probe scripts/join_probe.cpp && /tmp/join_probe
// Questions:
//   E1  random 16-byte loads inside a window that fits one XCD's L2 (every block uses the window of the XCD it runs on)
//       against the same loads over the union of the windows (no XCD affinity) and over 2 GB (the directory today)
//   E2  streaming 8-byte query entries from HBM through a presence bitmap held in LDS (128 KB, one block per CU):
//       entries per second, with a fraction `pass` of them followed by a 16-byte directory load (window as in E1)
//   E3  LDS-staged partition scatter: 8-byte entries binned into P partitions per tile through an LDS counting sort
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

__host__ __device__ __forceinline__ uint64_t mix(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ uint32_t xcc_id()
{
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}

// E1: mode 0 = window of the block's XCD, 1 = any of the 8 windows, 2 = whole table
template <int UNROLL>
__global__ void __launch_bounds__(256) k_e1(const ulonglong2 *__restrict__ tab, uint64_t win16, uint64_t tab16, int mode,
                                             int per_thread, uint64_t *__restrict__ sink)
{
    uint64_t s = mix(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 1);
    const uint32_t xcc = xcc_id();
    uint64_t acc = 0;
    for (int i = 0; i < per_thread; i += UNROLL) {
        ulonglong2 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            uint64_t idx;
            if (mode == 0)
                idx = xcc * win16 + (s >> 20) % win16;
            else if (mode == 1)
                idx = (s >> 20) % (8 * win16);
            else
                idx = (s >> 20) % tab16;
            v[u] = tab[idx];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc += v[u].x + v[u].y;
    }
    if (acc == 0x1234567) sink[0] = acc;
}

// E2: one block of 1024 threads per CU, bitmap of BM_WORDS words in LDS; partitions are pulled in rounds: partition p
// is taken by the blocks with (blockIdx % 8) == (p % 8) -- blocks of one XCD share the directory window of p
#define BM_WORDS 32768
__global__ void __launch_bounds__(1024) k_e2(const uint64_t *__restrict__ ent, uint64_t per_part, int nparts,
                                              const uint32_t *__restrict__ bitmaps, const ulonglong2 *__restrict__ dir,
                                              uint64_t win16, int do_lookup, uint32_t pass_mask,
                                              unsigned long long *__restrict__ out)
{
    __shared__ uint32_t bm[BM_WORDS];
    const int xs = blockIdx.x & 7, bi = blockIdx.x >> 3, nb = gridDim.x >> 3;
    unsigned long long found = 0;
    for (int p = xs; p < nparts; p += 8) {
        __syncthreads();
        for (int i = threadIdx.x; i < BM_WORDS / 4; i += 1024)
            ((uint4 *)bm)[i] = ((const uint4 *)(bitmaps + (uint64_t)p * BM_WORDS))[i];
        __syncthreads();
        const uint64_t lo = per_part * bi / nb, hi = per_part * (bi + 1) / nb;
        const uint64_t *e = ent + (uint64_t)p * per_part;
        for (uint64_t i = lo + threadIdx.x * 2; i < hi; i += 2048) {
            const ulonglong2 q = *(const ulonglong2 *)(e + i);
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const uint64_t x = h ? q.y : q.x;
                const uint32_t b = (uint32_t)(x >> 40) & (BM_WORDS * 32 - 1);
                // the bitmap decides `pass`; pass_mask thins the passing fraction to what the real filter lets through
                const bool pass = ((bm[b >> 5] >> (b & 31)) & 1u) && ((uint32_t)x & pass_mask) == 0;
                if (pass) {
                    if (do_lookup) {
                        const ulonglong2 d = dir[(uint64_t)p * win16 + (x >> 8) % win16];
                        found += (d.x == x);
                    }
                    found++;
                }
            }
        }
    }
    if (found) atomicAdd(out, found);
}

// E3: tile of TILE entries per block round: generate (stand-in for rolling), count per partition in LDS, scan, scatter
// into LDS, write the tile out sorted by partition + the segment offsets
template <int P, int TILE>
__global__ void __launch_bounds__(512) k_e3(const uint8_t *__restrict__ bases, uint64_t nbases, int mod,
                                             uint64_t *__restrict__ out, uint16_t *__restrict__ segoff, uint32_t *__restrict__ tile_n,
                                             int ntiles, int bases_per_tile)
{
    __shared__ uint64_t buf[TILE];
    __shared__ uint32_t cnt[P];
    __shared__ uint32_t s_w[8];
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        for (int i = threadIdx.x; i < P; i += 512) cnt[i] = 0;
        __syncthreads();
        // every thread rolls bases_per_tile / 512 positions (k = 20), sampled canonical k-mers are kept in registers
        const int per = bases_per_tile / 512;
        const uint64_t b0 = (uint64_t)t * bases_per_tile + (uint64_t)threadIdx.x * per;
        const uint64_t mask = (1ull << 40) - 1;
        uint64_t km = 0, rc = 0;
        constexpr int QMAX = 16;
        uint64_t q[QMAX];
        int nq = 0;
        uint64_t w = 0;
        for (int x = 0; x < per + 19; x++) {
            const uint64_t pp = b0 + x;
            if ((x & 7) == 0) w = pp + 8 <= nbases ? *(const uint64_t *)(bases + pp) : 0ull;
            const uint8_t c = (uint8_t)w & 3;
            w >>= 8;
            km = ((km << 2) | c) & mask;
            rc = (rc >> 2) | ((uint64_t)(3 - c) << 38);
            const uint64_t canon = km < rc ? km : rc;
            if (x >= 19 && (uint32_t)(mix(canon) >> 40) % (uint32_t)mod == 0 && nq < QMAX) {
                const uint64_t e = (canon << 24) | (uint32_t)((threadIdx.x * per + x) & 0xFFFFF);
#pragma unroll
                for (int u = 0; u < QMAX; u++)
                    if (u == nq) q[u] = e;
                nq++;
                atomicAdd(&cnt[(uint32_t)(canon >> 30) & (P - 1)], 1u);
            }
        }
        __syncthreads();
        // exclusive scan of cnt (P / 512 per thread)
        uint32_t c4[P / 512 > 0 ? P / 512 : 1], sum = 0;
#pragma unroll
        for (int u = 0; u < P / 512; u++) {
            c4[u] = cnt[threadIdx.x * (P / 512) + u];
            sum += c4[u];
        }
        uint32_t incl = sum;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off, 64);
            if ((threadIdx.x & 63) >= off) incl += up;
        }
        if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint32_t base = incl - sum;
        for (int w = 0; w < (int)(threadIdx.x >> 6); w++) base += s_w[w];
        uint32_t tot = 0;
        for (int w = 0; w < 8; w++) tot += s_w[w];
#pragma unroll
        for (int u = 0; u < P / 512; u++) {
            cnt[threadIdx.x * (P / 512) + u] = base;
            segoff[(uint64_t)t * P + threadIdx.x * (P / 512) + u] = (uint16_t)base;
            base += c4[u];
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < QMAX; u++)
            if (u < nq) {
                const uint32_t slot = atomicAdd(&cnt[(uint32_t)(q[u] >> 54) & (P - 1)], 1u);
                if (slot < TILE) buf[slot] = q[u];
            }
        __syncthreads();
        if (tot > TILE) tot = TILE;
        for (uint32_t i = threadIdx.x; i < tot; i += 512) out[(uint64_t)t * TILE + i] = buf[i];
        if (threadIdx.x == 0) tile_n[t] = tot;
        __syncthreads();
    }
}

int main(int argc, char **argv)
{
    uint64_t *sink;
    CHK(hipMalloc(&sink, 64));
    CHK(hipMemset(sink, 0, 64));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    float ms;
    const size_t tab_bytes = 2048ull << 20;
    ulonglong2 *tab;
    CHK(hipMalloc(&tab, tab_bytes));
    CHK(hipMemset(tab, 1, tab_bytes));
    // ---- E1
    for (size_t win_kb : {512, 1024, 2048, 3072, 4096}) {
        for (int mode = 0; mode < 3; mode++) {
            const int per_thread = 2048, blocks = 256 * 8;
            for (int rep = 0; rep < 2; rep++) {
                CHK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_e1<8>, dim3(blocks), dim3(256), 0, 0, tab, (uint64_t)win_kb * 1024 / 16, (uint64_t)tab_bytes / 16,
                                   mode, per_thread, sink);
                CHK(hipEventRecord(e1));
                CHK(hipEventSynchronize(e1));
            }
            CHK(hipEventElapsedTime(&ms, e0, e1));
            const double n = (double)blocks * 256 * per_thread;
            printf("E1 window %4zu KB per XCD, mode %d (%s): %8.2f G loads/s (%.2f ms)\n", win_kb, mode,
                   mode == 0 ? "own XCD's window" : (mode == 1 ? "any of the 8 windows" : "2 GB table"), n / ms / 1e6, ms);
            if (mode == 2 && win_kb != 512) break;
        }
    }
    // ---- E2
    {
        const int nparts = 256;
        const uint64_t per_part = 8ull << 20;  // entries per partition: 64 MB, 16 GB in all
        uint64_t *ent;
        uint32_t *bitmaps;
        unsigned long long *out;
        CHK(hipMalloc(&ent, nparts * per_part * 8));
        CHK(hipMalloc(&bitmaps, (size_t)nparts * BM_WORDS * 4));
        CHK(hipMalloc(&out, 8));
        // entries: random 64-bit words; bitmaps: every word 0x11111111 (a quarter of the bits) -- thinned by pass_mask
        std::vector<uint64_t> h(1 << 20);
        uint64_t s = 12345;
        for (auto &x : h) x = (s = mix(s + 0x9E3779B97F4A7C15ull));
        for (uint64_t o = 0; o < nparts * per_part; o += h.size())
            CHK(hipMemcpy(ent + o, h.data(), h.size() * 8, hipMemcpyHostToDevice));
        CHK(hipMemset(bitmaps, 0x11, (size_t)nparts * BM_WORDS * 4));
        for (int do_lookup = 0; do_lookup < 2; do_lookup++)
            for (uint32_t pm : {0u, 1u, 3u}) {  // pass = 25 %, 12.5 %, 6.25 %
                for (size_t win_kb : {2048, 8192}) {
                    if (!do_lookup && win_kb != 2048) continue;
                    for (int rep = 0; rep < 2; rep++) {
                        CHK(hipMemset(out, 0, 8));
                        CHK(hipEventRecord(e0));
                        hipLaunchKernelGGL(k_e2, dim3(256), dim3(1024), 0, 0, ent, per_part, nparts, bitmaps, tab,
                                           (uint64_t)win_kb * 1024 / 16, do_lookup, pm, out);
                        CHK(hipEventRecord(e1));
                        CHK(hipEventSynchronize(e1));
                    }
                    CHK(hipEventElapsedTime(&ms, e0, e1));
                    unsigned long long f;
                    CHK(hipMemcpy(&f, out, 8, hipMemcpyDeviceToHost));
                    const double n = (double)nparts * per_part;
                    printf("E2 lookup %d pass %.4f dir window %zu KB: %8.2f G entries/s = %7.1f GB/s of entries (%.2f ms)\n",
                           do_lookup, (double)f / n / (do_lookup ? 1.0 : 1.0), win_kb, n / ms / 1e6, n * 8 / ms / 1e6, ms);
                }
            }
        CHK(hipFree(ent));
        CHK(hipFree(bitmaps));
    }
    // ---- E3
    {
        const uint64_t nbases = 4ull << 30;
        uint8_t *bases;
        CHK(hipMalloc(&bases, nbases));
        std::vector<uint8_t> h(1 << 24);
        uint64_t s = 777;
        for (auto &x : h) x = (uint8_t)((s = mix(s + 1)) & 3);
        for (uint64_t o = 0; o < nbases; o += h.size()) CHK(hipMemcpy(bases + o, h.data(), h.size(), hipMemcpyHostToDevice));
        for (int mod : {8, 1}) {
            constexpr int TILE = 12288;
            const int bases_per_tile = mod == 8 ? 65536 : 8192;  // ~8192 sampled k-mers per tile
            const int ntiles = (int)(nbases / bases_per_tile);
            uint64_t *out;
            uint16_t *segoff;
            uint32_t *tile_n;
            CHK(hipMalloc(&out, (size_t)ntiles * TILE * 8));
            CHK(hipMalloc(&segoff, (size_t)ntiles * 1024 * 2));
            CHK(hipMalloc(&tile_n, (size_t)ntiles * 4));
            for (int rep = 0; rep < 2; rep++) {
                CHK(hipEventRecord(e0));
                hipLaunchKernelGGL((k_e3<1024, TILE>), dim3(256 * 3), dim3(512), 0, 0, bases, nbases, mod, out, segoff, tile_n, ntiles,
                                   bases_per_tile);
                CHK(hipEventRecord(e1));
                CHK(hipEventSynchronize(e1));
            }
            CHK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<uint32_t> tn(ntiles);
            CHK(hipMemcpy(tn.data(), tile_n, (size_t)ntiles * 4, hipMemcpyDeviceToHost));
            double tot = 0;
            for (auto x : tn) tot += x;
            printf("E3 mod %d: %d tiles of %d bases, %.0f entries (%.1f per tile): %.2f ms = %.1f G bases/s, %.2f G entries/s\n", mod, ntiles,
                   bases_per_tile, tot, tot / ntiles, ms, nbases / ms / 1e6, tot / ms / 1e6);
            CHK(hipFree(out));
            CHK(hipFree(segoff));
            CHK(hipFree(tile_n));
        }
    }
    return 0;
}
