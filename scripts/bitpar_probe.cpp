// bitpar_probe.cpp -- how fast could a banded bit-parallel recurrence run on this part?
//
// Round-1 verdict, task 3: "evaluate a banded bit-parallel kernel (Myers/Hyyro: 64 DP cells per 64-bit
// op per lane) against the same oracle contract".  This is the evaluation of the *recurrence*, not a
// replacement of k_wave2: one global alignment per LANE inside a diagonal band of 64 (one machine word)
// or 128 cells (two words, carries chained), Hyyro's diagonal-band variant of Myers' algorithm
// (H. Hyyro, "A bit-vector algorithm for computing Levenshtein and Damerau edit distances", 2003, Fig. 8;
// G. Myers, JACM 46(3), 1999), sequences 2-bit packed, 32 bases per 8-byte load.  It reports
//   * columns (= aligned A bases) per second and DP cells per second of the kernel,
//   * a check of the band scores against a plain banded DP on the host,
//   * how far the optimal path of read-read pairs at the pile-up stage's divergence strays from the main
//     diagonal (what band a fixed-band kernel would need; an adaptive band needs a shift rule on top).
// Build and run on the GPU box:  hipcc -O3 --offload-arch=gfx950 -o bitpar_probe scripts/bitpar_probe.cpp && ./bitpar_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CHK(x)                                                                 \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));     \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

// equality mask of the 64 pattern bases in `w` (2-bit codes, low bits first, 128 bits in lo / hi) with
// the base c: bit r = (base r == c)
__device__ __forceinline__ uint64_t eq_mask(uint64_t lo, uint64_t hi, uint32_t c)
{
    const uint64_t rep = 0x5555555555555555ull * c;  // c in every 2-bit field
    uint64_t xl = lo ^ rep, xh = hi ^ rep;           // field == 0 <=> equal
    xl = ~(xl | (xl >> 1)) & 0x5555555555555555ull;  // bit 2r set <=> equal
    xh = ~(xh | (xh >> 1)) & 0x5555555555555555ull;
    // compress the even bits of xl (32 fields) and xh (32 fields) into 64 bits
    auto squeeze = [](uint64_t x) {
        x = (x | (x >> 1)) & 0x3333333333333333ull;
        x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0Full;
        x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
        x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
        x = (x | (x >> 16)) & 0x00000000FFFFFFFFull;
        return x;
    };
    return squeeze(xl) | (squeeze(xh) << 32);
}

// 128 bits of the packed sequence starting at base position p (p may be negative / past the end:
// the caller pads the buffers)
__device__ __forceinline__ void window128(const uint8_t *pk, int64_t p, uint64_t &lo, uint64_t &hi)
{
    const int64_t byte = p >> 2;
    const int sh = (int)(p & 3) << 1;
    uint64_t a, b, c;
    __builtin_memcpy(&a, pk + byte, 8);
    __builtin_memcpy(&b, pk + byte + 8, 8);
    __builtin_memcpy(&c, pk + byte + 16, 8);
    lo = sh ? (a >> sh) | (b << (64 - sh)) : a;
    hi = sh ? (b >> sh) | (c << (64 - sh)) : b;
}

// One alignment per lane.  Band of W = 64 * NW cells around the main diagonal: column j holds the rows
// j - W/2 .. j + W/2 - 1 of B.  Returns D[min(n, m) path end] tracked along the bottom cell of the band --
// enough for a throughput probe; scores are checked against the host for the bottom-cell convention.
template <int NW>
__global__ void __launch_bounds__(256)
k_bitpar(const uint8_t *__restrict__ apk, const uint8_t *__restrict__ bpk, const int64_t *__restrict__ aoff,
         const int64_t *__restrict__ boff, const int32_t *__restrict__ alen, int32_t npairs, int32_t *__restrict__ score)
{
    const int32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npairs) return;
    const uint8_t *a = apk, *b = bpk;
    const int64_t ao = aoff[p], bo = boff[p];
    const int32_t n = alen[p];
    constexpr int W = 64 * NW;
    uint64_t VP[NW], VN[NW];
#pragma unroll
    for (int w = 0; w < NW; w++) {
        VP[w] = ~0ull;  // first column: D[i][0] = i inside the band (rows above 0 are padding)
        VN[w] = 0;
    }
    int32_t sc = W - 1;  // D of the bottom cell of the band before the first column (the band starts as 0 .. W - 1 from its top)
    uint64_t acur = 0;
    for (int32_t j = 0; j < n; j++) {
        if ((j & 31) == 0) __builtin_memcpy(&acur, a + ((ao + j) >> 2), 8);  // ao is a multiple of 32
        const uint32_t c = (uint32_t)(acur >> (2 * (j & 31))) & 3u;
        uint64_t carry = 0, d0_top = 0;
        uint64_t HPs[NW], HNs[NW], D0s[NW];
#pragma unroll
        for (int w = 0; w < NW; w++) {
            uint64_t lo, hi;
            window128(b, bo + (int64_t)j - W / 2 + 64 * w, lo, hi);
            const uint64_t Eq = eq_mask(lo, hi, c);
            // D0 = (((Eq & VP) + VP) ^ VP) | Eq | VN, the addition chained over the words
            const uint64_t x = Eq & VP[w];
            const uint64_t s1 = x + VP[w];
            const uint64_t s2 = s1 + carry;
            carry = (s1 < x) | (s2 < s1);
            const uint64_t D0 = (s2 ^ VP[w]) | Eq | VN[w];
            D0s[w] = D0;
            HPs[w] = VN[w] | ~(D0 | VP[w]);
            HNs[w] = VP[w] & D0;
            if (w == NW - 1) d0_top = D0 >> 63;
        }
        // the band moves one row down: shift the vertical vectors (D0 >> 1 across the words)
#pragma unroll
        for (int w = 0; w < NW; w++) {
            const uint64_t X = (D0s[w] >> 1) | (w + 1 < NW ? D0s[w + 1] << 63 : 0ull);
            VN[w] = X & HPs[w];
            VP[w] = HNs[w] | ~(X | HPs[w]);
        }
        sc += d0_top ? 0 : 1;  // diagonal step of the tracked cell: + 1 unless D0 says "no increase"
    }
    score[p] = sc;
}

// The same recurrence with the match vectors kept in registers: one 64 * NW-bit match mask per letter,
// shifted by one row per column with the base that enters the band at its bottom (no window loads, no
// bit squeeze per column) -- the form a production kernel would take.
template <int NW>
__global__ void __launch_bounds__(256)
k_bitpar_rolling(const uint8_t *__restrict__ apk, const uint8_t *__restrict__ bpk, const int64_t *__restrict__ aoff,
                 const int64_t *__restrict__ boff, const int32_t *__restrict__ alen, int32_t npairs,
                 int32_t *__restrict__ score)
{
    const int32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npairs) return;
    const int64_t ao = aoff[p], bo = boff[p];
    const int32_t n = alen[p];
    constexpr int W = 64 * NW;
    uint64_t VP[NW], VN[NW], M[4][NW];
#pragma unroll
    for (int w = 0; w < NW; w++) {
        VP[w] = ~0ull;
        VN[w] = 0;
        // match masks of the rows -W/2 .. W/2 - 1 (column 0 minus one shift): built once
        uint64_t lo, hi;
        window128(bpk, bo - W / 2 - 1 + 64 * w, lo, hi);
#pragma unroll
        for (uint32_t c = 0; c < 4; c++) M[c][w] = eq_mask(lo, hi, c);
    }
    int32_t sc = W - 1;
    uint64_t acur = 0, bcur = 0;
    const int64_t bin0 = bo + W / 2 - 1;  // base entering at the bottom of the band in column 0
    for (int32_t j = 0; j < n; j++) {
        if ((j & 31) == 0) __builtin_memcpy(&acur, apk + ((ao + j) >> 2), 8);
        const int64_t bp = bin0 + j;
        if ((j & 31) == 0 || (bp & 31) == 0) __builtin_memcpy(&bcur, bpk + ((bp >> 5) << 3), 8);
        const uint32_t c = (uint32_t)(acur >> (2 * (j & 31))) & 3u;
        const uint32_t nb = (uint32_t)(bcur >> (2 * (bp & 31))) & 3u;
        // the band moves one row down: shift the match masks, the new base enters at the top bit
#pragma unroll
        for (uint32_t l = 0; l < 4; l++) {
#pragma unroll
            for (int w = 0; w < NW; w++)
                M[l][w] = (M[l][w] >> 1) | (w + 1 < NW ? M[l][w + 1] << 63 : (uint64_t)(nb == l) << 63);
        }
        uint64_t carry = 0, d0_top = 0, HPs[NW], HNs[NW], D0s[NW];
#pragma unroll
        for (int w = 0; w < NW; w++) {
            const uint64_t Eq = c == 0 ? M[0][w] : c == 1 ? M[1][w] : c == 2 ? M[2][w] : M[3][w];
            const uint64_t x = Eq & VP[w];
            const uint64_t s1 = x + VP[w];
            const uint64_t s2 = s1 + carry;
            carry = (s1 < x) | (s2 < s1);
            const uint64_t D0 = (s2 ^ VP[w]) | Eq | VN[w];
            D0s[w] = D0;
            HPs[w] = VN[w] | ~(D0 | VP[w]);
            HNs[w] = VP[w] & D0;
            if (w == NW - 1) d0_top = D0 >> 63;
        }
#pragma unroll
        for (int w = 0; w < NW; w++) {
            const uint64_t X = (D0s[w] >> 1) | (w + 1 < NW ? D0s[w + 1] << 63 : 0ull);
            VN[w] = X & HPs[w];
            VP[w] = HNs[w] | ~(X | HPs[w]);
        }
        sc += d0_top ? 0 : 1;
    }
    score[p] = sc;
}

// host: plain DP restricted to the same band, same boundary convention, value of the tracked cell
static int host_band(const std::vector<uint8_t> &A, const std::vector<uint8_t> &B, int W)
{
    const int n = (int)A.size();
    const int INF = 1 << 28;
    // cell (i, j): i row of B (1-based, 0 = boundary), column j of A; band rows of column j: j - W/2 + r, r = 0..W-1
    auto at = [&](std::vector<int> &col, int r) -> int & { return col[(size_t)r + 1]; };
    std::vector<int> prev((size_t)W + 2, INF), cur((size_t)W + 2, INF);
    for (int r = 0; r < W; r++) {
        const int i = 0 - W / 2 + r + 1;  // column 0 holds rows shifted like the kernel: row index i = r - W/2 + 1
        at(prev, r) = i >= 0 ? i : -i;    // padding rows mirror (the kernel starts with all-ones VP)
    }
    // the kernel's initial state is "VP all ones": D increases by one per row from the top of the band;
    // reproduce exactly that: D(top) = 0 .. D(bottom) = W - 1, tracked cell starts at W/2 by its own convention
    for (int r = 0; r < W; r++) at(prev, r) = r;
    int track = W / 2;
    (void)track;
    for (int j = 0; j < n; j++) {
        for (int r = 0; r < W; r++) {
            const long long i = (long long)j - W / 2 + r;  // base of B compared in this cell
            const int eq = (i >= 0 && i < (long long)B.size() && B[(size_t)i] == A[(size_t)j]) ? 0 : 1;
            // band moved down by one: (i-1, j-1) is prev[r], (i, j-1) is prev[r+1], (i-1, j) is cur[r-1]
            int v = at(prev, r) + eq;
            if (r + 1 < W) v = std::min(v, at(prev, r + 1) + 1);
            if (r > 0) v = std::min(v, at(cur, r - 1) + 1);
            at(cur, r) = v;
        }
        prev.swap(cur);
    }
    return at(prev, W - 1);
}

int main()
{
    std::mt19937_64 rng(12345);
    const int npairs = 1 << 16, L = 4096;
    auto make_pairs = [&](double sub, double indel, std::vector<std::vector<uint8_t>> &As, std::vector<std::vector<uint8_t>> &Bs,
                          int count, std::vector<int> *drift) {
        std::uniform_real_distribution<double> U(0, 1);
        for (int p = 0; p < count; p++) {
            std::vector<uint8_t> A((size_t)L), B;
            for (auto &x : A) x = (uint8_t)(rng() & 3);
            int d = 0, dmax = 0;
            for (int i = 0; i < L; i++) {
                const double u = U(rng);
                if (u < indel) {  // deletion in B
                    d--;
                } else if (u < 2 * indel) {  // insertion in B
                    B.push_back((uint8_t)(rng() & 3));
                    B.push_back(A[(size_t)i]);
                    d++;
                } else if (u < 2 * indel + sub)
                    B.push_back((uint8_t)((A[(size_t)i] + 1 + rng() % 3) & 3));
                else
                    B.push_back(A[(size_t)i]);
                dmax = std::max(dmax, std::abs(d));
            }
            if (drift) drift->push_back(dmax);
            As.push_back(A);
            Bs.push_back(B);
        }
    };
    // ---- drift of read-read overlaps at the pile-up stage's divergence (13 % per read: ~8 % indels + ~5 % subs each)
    {
        std::vector<std::vector<uint8_t>> As, Bs;
        std::vector<int> drift;
        make_pairs(0.10, 0.085, As, Bs, 2000, &drift);
        std::sort(drift.begin(), drift.end());
        printf("drift of the true path from the main diagonal over %d columns (26 %% divergence, 17 %% indels): median %d, 90 %% %d, 99 %% %d, max %d\n",
               L, drift[drift.size() / 2], drift[drift.size() * 9 / 10], drift[drift.size() * 99 / 100], drift.back());
    }
    // ---- throughput + check
    std::vector<std::vector<uint8_t>> As, Bs;
    make_pairs(0.20, 0.01, As, Bs, npairs, nullptr);
    const int64_t stride = ((L + 256 + 31) / 32) * 32;  // bases per slot, multiple of 32, padded
    std::vector<uint8_t> apk((size_t)(npairs * stride / 4 + 64), 0), bpk((size_t)(npairs * (2 * stride) / 4 + 64), 0);
    std::vector<int64_t> aoff((size_t)npairs), boff((size_t)npairs);
    std::vector<int32_t> alen((size_t)npairs);
    auto put = [](std::vector<uint8_t> &pk, int64_t pos, uint8_t c) { pk[(size_t)(pos >> 2)] |= (uint8_t)(c << (2 * (pos & 3))); };
    for (int p = 0; p < npairs; p++) {
        aoff[(size_t)p] = (int64_t)p * stride;
        boff[(size_t)p] = (int64_t)p * 2 * stride + 128;  // room for the band's negative rows
        alen[(size_t)p] = L;
        for (int i = 0; i < L; i++) put(apk, aoff[(size_t)p] + i, As[(size_t)p][(size_t)i]);
        for (size_t i = 0; i < Bs[(size_t)p].size(); i++) put(bpk, boff[(size_t)p] + (int64_t)i, Bs[(size_t)p][i]);
    }
    uint8_t *d_a, *d_b;
    int64_t *d_ao, *d_bo;
    int32_t *d_al, *d_sc;
    CHK(hipMalloc(&d_a, apk.size()));
    CHK(hipMalloc(&d_b, bpk.size()));
    CHK(hipMalloc(&d_ao, 8 * (size_t)npairs));
    CHK(hipMalloc(&d_bo, 8 * (size_t)npairs));
    CHK(hipMalloc(&d_al, 4 * (size_t)npairs));
    CHK(hipMalloc(&d_sc, 4 * (size_t)npairs));
    CHK(hipMemcpy(d_a, apk.data(), apk.size(), hipMemcpyHostToDevice));
    CHK(hipMemcpy(d_b, bpk.data(), bpk.size(), hipMemcpyHostToDevice));
    CHK(hipMemcpy(d_ao, aoff.data(), 8 * (size_t)npairs, hipMemcpyHostToDevice));
    CHK(hipMemcpy(d_bo, boff.data(), 8 * (size_t)npairs, hipMemcpyHostToDevice));
    CHK(hipMemcpy(d_al, alen.data(), 4 * (size_t)npairs, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    for (int nw = 1; nw <= 2; nw++) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; rep++) {
            CHK(hipEventRecord(e0));
            if (nw == 1)
                hipLaunchKernelGGL(k_bitpar<1>, dim3((npairs + 255) / 256), dim3(256), 0, 0, d_a, d_b, d_ao, d_bo, d_al, npairs, d_sc);
            else
                hipLaunchKernelGGL(k_bitpar<2>, dim3((npairs + 255) / 256), dim3(256), 0, 0, d_a, d_b, d_ao, d_bo, d_al, npairs, d_sc);
            CHK(hipEventRecord(e1));
            CHK(hipEventSynchronize(e1));
            float ms;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms);
        }
        std::vector<int32_t> sc((size_t)npairs);
        CHK(hipMemcpy(sc.data(), d_sc, 4 * (size_t)npairs, hipMemcpyDeviceToHost));
        // the device treats the row above the band like Myers' free-start row (horizontal delta 0), the host
        // DP forbids it: scores may differ by a few units through that edge, never by more
        int maxdiff = 0;
        for (int p = 0; p < 64; p++) maxdiff = std::max(maxdiff, std::abs(host_band(As[(size_t)p], Bs[(size_t)p], 64 * nw) - sc[(size_t)p]));
        const double cols = (double)npairs * L;
        printf("band %3d, window reloaded per column: %.2f ms for %d alignments x %d columns -> %.3g columns/s, %.3g DP cells/s; |device - host band DP| <= %d on 64 pairs\n",
               64 * nw, best, npairs, L, cols / (best * 1e-3), cols * 64 * nw / (best * 1e-3), maxdiff);
        float best2 = 1e30f;
        for (int rep = 0; rep < 4; rep++) {
            CHK(hipEventRecord(e0));
            if (nw == 1)
                hipLaunchKernelGGL(k_bitpar_rolling<1>, dim3((npairs + 255) / 256), dim3(256), 0, 0, d_a, d_b, d_ao, d_bo, d_al, npairs, d_sc);
            else
                hipLaunchKernelGGL(k_bitpar_rolling<2>, dim3((npairs + 255) / 256), dim3(256), 0, 0, d_a, d_b, d_ao, d_bo, d_al, npairs, d_sc);
            CHK(hipEventRecord(e1));
            CHK(hipEventSynchronize(e1));
            float ms;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            best2 = std::min(best2, ms);
        }
        std::vector<int32_t> sc2((size_t)npairs);
        CHK(hipMemcpy(sc2.data(), d_sc, 4 * (size_t)npairs, hipMemcpyDeviceToHost));
        int neq = 0;
        for (int p = 0; p < npairs; p++) neq += sc2[(size_t)p] != sc[(size_t)p];
        printf("band %3d, rolling match masks:        %.2f ms -> %.3g columns/s, %.3g DP cells/s; %d of %d scores differ from the first kernel\n",
               64 * nw, best2, cols / (best2 * 1e-3), cols * 64 * nw / (best2 * 1e-3), neq, npairs);
    }
    return 0;
}
