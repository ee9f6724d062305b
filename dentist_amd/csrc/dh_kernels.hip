// dh_kernels.hip -- gfx950 (CDNA4, wave64) kernels of the alignment pass.
//
// K1  k_revcomp        reverse-complement copy of a DB                     (HBM stream)
// K2  k_kmer_pass      k-mer extraction of A: count pass and fill pass     (HBM stream + atomics)
//     k_scan*          exclusive scan of the bucket directory
// K4  k_seed           per (B read, strand): k-mer lookups, LDS-staged hit buffer, in-LDS
//                      bitonic sort by (diagonal, position), band-pair coverage filter, seeds
// K5  k_wave           per (B read, strand): O(ND) furthest-reaching wave, one 64-lane wavefront
//                      per alignment (lane == diagonal), trace points every tspace A-bases
//     k_gather_trace   compaction of the per-slot trace vectors
//
// The arithmetic specification these kernels implement is written down in DESIGN.md
// ("Algorithm DH-1"); reference call sites: source/dentist/dazzler.d:6121-6170.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <algorithm>
#include <cstdlib>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

#include "dh_device.h"
#include "dh_join.h"

#define LANES 64

#include "dh_kmer.h"

// ------------------------------------------------------------------------------------ K1

// 8 bases per thread and step: unaligned 8-byte load of the mirrored window, byte swap,
// complement of the codes 0..3 (c ^ 3; other codes are kept), unaligned 8-byte store.
__global__ void __launch_bounds__(256)
k_revcomp(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, const int64_t *__restrict__ off,
          int32_t n)
{
    // blockIdx.y = sequence, grid-stride over its 8-base words
    const int32_t s = blockIdx.y;
    if (s >= n) return;
    const int64_t o = off[s], len = off[s + 1] - o;
    const int64_t nw = len >> 3;
    for (int64_t wd = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; wd < nw;
         wd += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = wd << 3;
        uint64_t x;
        __builtin_memcpy(&x, src + o + len - 8 - i, 8);
        x = __builtin_bswap64(x);
        const uint64_t hi = x & 0xFCFCFCFCFCFCFCFCull;  // bytes >= 4 are not bases
        const uint64_t nz = (((hi & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | hi) & 0x8080808080808080ull;
        const uint64_t keep = (nz >> 7) * 0xFFull;
        x ^= 0x0303030303030303ull & ~keep;
        __builtin_memcpy(dst + o + i, &x, 8);
    }
    // tail (len % 8 bases): the first threads of block x == 0
    if (blockIdx.x == 0) {
        const int64_t i = (nw << 3) + threadIdx.x;
        if (i < len) {
            const uint8_t c = src[o + len - 1 - i];
            dst[o + i] = c < 4 ? (uint8_t)(3 - c) : c;
        }
    }
}

// 2-bit packed copy of a base array for the wave kernel: base g sits in byte g >> 2 at bits
// 2 * (g & 3), so a little-endian 8-byte load holds 32 consecutive bases, low bits first.
// One thread per 8-byte word; flag is raised when a code outside 0..3 is met (such DBs are
// aligned from the byte arrays instead).  The source is readable 63 bytes past `total` (DB_PAD).
// PLANES: the word is stored plane-packed for k_tile (dh_tile.h: PlanePair -- low bits of the 32 bases in the low
// half, high bits in the high half) instead of being converted by a pass of its own over the copy
__device__ __forceinline__ uint32_t squeeze_even64(uint64_t x)
{
    x &= 0x5555555555555555ull;
    x = (x | (x >> 1)) & 0x3333333333333333ull;
    x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
    x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
    x = (x | (x >> 16)) & 0x00000000FFFFFFFFull;
    return (uint32_t)x;
}
__device__ __forceinline__ uint64_t pk_to_planes(uint64_t x) { return (uint64_t)squeeze_even64(x) | ((uint64_t)squeeze_even64(x >> 1) << 32); }
template <bool PLANES>
__global__ void __launch_bounds__(256)
k_pack2(const uint8_t *__restrict__ src, int64_t total, uint64_t *__restrict__ dst,
        int32_t *__restrict__ flag)
{
    const int64_t wd = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nw = (total + 31) >> 5;
    if (wd >= nw) return;
    uint64_t out = 0;
    bool bad = false;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint64_t x = load8(src + (wd << 5) + 8 * q);
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t c = (uint32_t)(x >> (8 * u)) & 0xFFu;
            const int64_t g = (wd << 5) + 8 * q + u;
            if (c > 3u && g < total) bad = true;
            out |= (uint64_t)(c & 3u) << (2 * (8 * q + u));
        }
    }
    dst[wd] = PLANES ? pk_to_planes(out) : out;
    if (bad) atomicOr(flag, 1);
}
template __global__ void k_pack2<false>(const uint8_t *, int64_t, uint64_t *, int32_t *);
template __global__ void k_pack2<true>(const uint8_t *, int64_t, uint64_t *, int32_t *);

// 2-bit packed reverse complements straight from the forward bytes: sequence s occupies the same base
// range [off[s], off[s+1]) in the packed copy, mirrored inside it.  One thread per 16-base word of the
// destination that the sequence touches: whole words are stored, the (at most two) words a sequence
// shares with its neighbours are ORed into the zeroed buffer.  `a0` (multiple of 32) = base position of
// dst word 0.  Codes outside 0..3 pack as (c ^ 3) & 3; such DBs are not aligned from the packed copies.
__device__ __forceinline__ uint32_t pack8_rc(uint64_t y)
{
    uint64_t t = (y ^ 0x0303030303030303ull) & 0x0303030303030303ull;
    t = (t | (t >> 6)) & 0x000F000F000F000Full;
    t = (t | (t >> 12)) & 0x000000FF000000FFull;
    t = (t | (t >> 24)) & 0xFFFFull;
    return (uint32_t)t;
}
__global__ void __launch_bounds__(256)
k_pack2_rc(const uint8_t *__restrict__ src, const int64_t *__restrict__ off, int32_t n, int64_t a0,
           uint32_t *__restrict__ dst)
{
    const int32_t s = blockIdx.y;
    if (s >= n) return;
    const int64_t o = off[s], len = off[s + 1] - o;
    if (len <= 0) return;
    const int64_t w0 = (o - a0) >> 4, w1 = (o + len - 1 - a0) >> 4;  // first / last destination word
    const int64_t sbase = 2 * o + len - 1;                           // source of base g is src[sbase - g]
    for (int64_t w = w0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w <= w1; w += (int64_t)gridDim.x * blockDim.x) {
        const int64_t gw = a0 + (w << 4);
        if (gw >= o && gw + 16 <= o + len) {
            const uint8_t *A = src + (sbase - gw - 15);
            uint64_t x0, x1;
            __builtin_memcpy(&x0, A, 8);      // positions 15 .. 8
            __builtin_memcpy(&x1, A + 8, 8);  // positions 7 .. 0
            dst[w] = pack8_rc(__builtin_bswap64(x1)) | (pack8_rc(__builtin_bswap64(x0)) << 16);
        } else {
            const int64_t g0 = gw > o ? gw : o, g1 = gw + 16 < o + len ? gw + 16 : o + len;
            uint32_t out = 0;
            for (int64_t g = g0; g < g1; g++) out |= (uint32_t)((src[sbase - g] ^ 3u) & 3u) << (2 * (int)(g - gw));
            atomicOr(&dst[w], out);
        }
    }
}

// the same copy plane-packed, one thread per 32-base word (= one PlanePair); the words a sequence shares with its
// neighbours are ORed in (the plane form is a permutation of the packed word's bits, so the parts combine the same way)
__global__ void __launch_bounds__(256)
k_pack2_rc_planes(const uint8_t *__restrict__ src, const int64_t *__restrict__ off, int32_t n, int64_t a0,
                  unsigned long long *__restrict__ dst)
{
    const int32_t s = blockIdx.y;
    if (s >= n) return;
    const int64_t o = off[s], len = off[s + 1] - o;
    if (len <= 0) return;
    const int64_t w0 = (o - a0) >> 5, w1 = (o + len - 1 - a0) >> 5;  // first / last destination word
    const int64_t sbase = 2 * o + len - 1;                           // source of base g is src[sbase - g]
    for (int64_t w = w0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w <= w1; w += (int64_t)gridDim.x * blockDim.x) {
        const int64_t gw = a0 + (w << 5);
        if (gw >= o && gw + 32 <= o + len) {
            const uint8_t *A = src + (sbase - gw - 31);
            uint64_t x0, x1, x2, x3;
            __builtin_memcpy(&x0, A, 8);       // positions 31 .. 24
            __builtin_memcpy(&x1, A + 8, 8);   // positions 23 .. 16
            __builtin_memcpy(&x2, A + 16, 8);  // positions 15 .. 8
            __builtin_memcpy(&x3, A + 24, 8);  // positions 7 .. 0
            const uint64_t lo = pack8_rc(__builtin_bswap64(x3)) | (pack8_rc(__builtin_bswap64(x2)) << 16);
            const uint64_t hi = pack8_rc(__builtin_bswap64(x1)) | (pack8_rc(__builtin_bswap64(x0)) << 16);
            dst[w] = pk_to_planes(lo | (hi << 32));
        } else {
            const int64_t g0 = gw > o ? gw : o, g1 = gw + 32 < o + len ? gw + 32 : o + len;
            uint64_t out = 0;
            for (int64_t g = g0; g < g1; g++) out |= (uint64_t)((src[sbase - g] ^ 3u) & 3u) << (2 * (int)(g - gw));
            atomicOr(&dst[w], (unsigned long long)pk_to_planes(out));
        }
    }
}
// the plane-packed reverse complements from the plane-packed FORWARD copy (made just before by k_pack2<true>) instead of
// from the bytes again: 8 bytes read per 32 bases instead of 32 -- base g of the reverse complement of sequence s is the
// complement of forward base sbase - g, so the 32 bases of a destination word are a run of 32 forward bases in reverse
// order: two funnel shifts over two forward words per plane, a bit reversal, a complement.  Words shared with a
// neighbouring sequence are ORed in, as in k_pack2_rc_planes.  fwd / dst: word 0 = base a0 (a multiple of 32), readable /
// zeroed PK_PAD bytes beyond both ends.
__global__ void __launch_bounds__(256)
k_planes_rc(const unsigned long long *__restrict__ fwd, const int64_t *__restrict__ off, int32_t n, int64_t a0,
            unsigned long long *__restrict__ dst)
{
    const int32_t s = blockIdx.y;
    if (s >= n) return;
    const int64_t o = off[s], len = off[s + 1] - o;
    if (len <= 0) return;
    const int64_t w0 = (o - a0) >> 5, w1 = (o + len - 1 - a0) >> 5;  // first / last destination word
    const int64_t sbase = 2 * o + len - 1;                           // source of base g is forward base sbase - g
    for (int64_t w = w0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w <= w1; w += (int64_t)gridDim.x * blockDim.x) {
        const int64_t gw = a0 + (w << 5);
        const int64_t p = sbase - gw - 31 - a0;  // first forward base of the run, relative to word 0 (may lie in the padding)
        const int64_t ws = p >> 5;               // (arithmetic shift: floor)
        const uint32_t sh = (uint32_t)(p & 31);
        const unsigned long long f0 = fwd[ws], f1 = fwd[ws + 1];
        const uint32_t lo = __builtin_amdgcn_alignbit((uint32_t)f1, (uint32_t)f0, sh);
        const uint32_t hi = __builtin_amdgcn_alignbit((uint32_t)(f1 >> 32), (uint32_t)(f0 >> 32), sh);
        // (run position i = forward base p + i = destination base 31 - i; complement = both planes inverted)
        unsigned long long out = (unsigned long long)(~__brev(lo)) | ((unsigned long long)(~__brev(hi)) << 32);
        if (gw >= o && gw + 32 <= o + len)
            dst[w] = out;
        else {
            const int64_t g0 = gw > o ? gw : o, g1 = gw + 32 < o + len ? gw + 32 : o + len;
            const uint32_t m = (uint32_t)(((g1 - g0) >= 32 ? ~0ull : ((1ull << (g1 - g0)) - 1)) << (g0 - gw));
            out &= (unsigned long long)m | ((unsigned long long)m << 32);
            atomicOr(&dst[w], out);
        }
    }
}
__global__ void __launch_bounds__(256)
k_pack2_rc_bounds32(const int64_t *__restrict__ off, int32_t n, int64_t a0, uint64_t *__restrict__ dst)
{
    const int32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const int64_t o = off[s], len = off[s + 1] - o;
    if (len <= 0) return;
    const int64_t w0 = (o - a0) >> 5, w1 = (o + len - 1 - a0) >> 5;
    const int64_t g0 = a0 + (w0 << 5), g1 = a0 + (w1 << 5);
    if (!(g0 >= o && g0 + 32 <= o + len)) dst[w0] = 0;
    if (!(g1 >= o && g1 + 32 <= o + len)) dst[w1] = 0;
}

// ------------------------------------------------------------------------------------ K2

// tiles: (sequence, start) pairs, KM_TILE positions each; 256 threads x 16 positions.
template <bool FILL>
__global__ void __launch_bounds__(256)
k_kmer_pass(DbView A, const int2 *__restrict__ tiles, int32_t ntiles, int32_t k, int32_t kmer_mod,
            int32_t shift,
            uint32_t *__restrict__ dir, ulonglong2 *__restrict__ ent, const int64_t *__restrict__ goff)
{
    const int32_t t = blockIdx.x;
    if (t >= ntiles) return;
    const int32_t s = tiles[t].x;
    const int64_t o = A.off[s];
    const int32_t len = (int32_t)(A.off[s + 1] - o);
    const uint64_t grp = A.group ? (uint64_t)A.group[s] : 0ull;
    const uint64_t mask = (1ull << (2 * k)) - 1;
    const int32_t p0 = tiles[t].y + threadIdx.x * (KM_TILE / 256);
    const KmerSampler smp = kmer_sampler(kmer_mod, k);
    uint64_t km = 0, rc = 0;
    int32_t valid = 0;
    const uint8_t *a = A.bases + o;
    const int rcsh = 2 * (k - 1);
    for (int32_t x = 0; x < KM_TILE / 256 + k - 1; x++) {
        const int32_t p = p0 + x;
        if (p >= len) break;
        const uint8_t c = a[p];
        if (c < 4) {
            km = ((km << 2) | c) & mask;
            rc = (rc >> 2) | ((uint64_t)(3 - c) << rcsh);
            valid++;
        } else {
            km = 0;
            rc = 0;
            valid = 0;
        }
        // the index is keyed by the canonical k-mer; bit 63 of the stored key says that the k-mer of A
        // is the reverse complement of its key (the lookup tells the strands apart with it)
        const uint64_t canon = km < rc ? km : rc;
        if (x >= k - 1 && valid >= k && kmer_sampled(canon, smp) &&
            !(A.mask_bits && mask_touch(A.mask_bits, o + p - k + 1, k))) {
            const uint64_t key = (grp << (2 * k)) | canon;
            const uint32_t b = (uint32_t)(key >> shift);
            if (FILL) {
                const uint32_t slot = atomicAdd(&dir[b], 1u);
                ent[slot] = make_ulonglong2(key | (km != canon ? 1ull << 63 : 0ull),
                                            ((uint64_t)s << 40) | (uint64_t)(goff[s] + (p - k + 1)));
            } else
                atomicAdd(&dir[b], 1u);
        }
    }
}
// The same two passes for a GROUPED DB (the pile-up stage: group = pile-up) without global atomics.  Keys carry the
// group in their top bits, so a group owns the contiguous bucket range [g * nbg, (g + 1) * nbg), nbg = 4^k >> shift.
// A block takes (group, slice of `slice` <= GI_SLICE buckets), rolls every k-mer of the group -- its tiles are
// tiles[gtile[g] .. gtile[g + 1]) -- and counts (FILL: places) those of its slice with LDS atomics; the counts / the
// advanced cursors go to dir[] in one coalesced pass.  170 M device-scope atomics on random counters took 19 ms per
// step (configs[2], 1 000 pile-ups); the price here is that a group's k-mers are rolled once per slice (8 times at
// k = 14 with 2^27 buckets), which is VALU work of about a millisecond.
#define GI_SLICE 32768
#define GI_THREADS 1024
// KT = the rolling k-mers' word: uint32_t when k <= 16 (the pile-up stage's k = 14: half the instructions of the
// 64-bit roll), uint64_t otherwise.  The bucket of a k-mer relative to the slice needs no group bits: the group's
// first bucket is ((g << 2k) >> shift) exactly (shift <= 2k), so rel = (canon >> shift) - sl * slice.
template <bool FILL, typename KT>
__global__ void __launch_bounds__(GI_THREADS)
k_group_index(DbView A, const int2 *__restrict__ tiles, const int32_t *__restrict__ gtile, int32_t slices_per_group,
              int32_t slice, int32_t k, int32_t kmer_mod, int32_t shift, uint32_t *__restrict__ dir,
              ulonglong2 *__restrict__ ent, const int64_t *__restrict__ goff)
{
    __shared__ uint32_t cnt[GI_SLICE];
    const int32_t g = blockIdx.x / slices_per_group, sl = blockIdx.x % slices_per_group;
    const int tid = threadIdx.x;
    const uint32_t sl0 = (uint32_t)sl * (uint32_t)slice;
    const uint32_t b0 = (uint32_t)((((uint64_t)g) << (2 * k)) >> shift) + sl0;
    for (int32_t i = tid; i < slice; i += GI_THREADS) cnt[i] = FILL ? dir[b0 + i] : 0u;
    __syncthreads();
    const uint64_t grp = (uint64_t)g;
    const KT mask = (KT)(((uint64_t)1 << (2 * k)) - 1);  // 2k == 32 with a 32-bit word: all ones
    const KmerSampler smp = kmer_sampler(kmer_mod, k);
    const int rcsh = 2 * (k - 1);
    // a wavefront per tile (16 tiles of the group in flight), a lane per 64 positions: 64 + k - 1 roll steps yield 64
    // k-mers (a thread per 16 positions spent 16 + k - 1 on 16), and the chain tile -> offsets -> bases is walked a
    // quarter as often.  Bases stream through one 8-byte word per 8 steps, the next word in flight.
    constexpr int32_t PER = KM_TILE / LANES;
    const int32_t nroll = PER + k - 1;
    const uint64_t kones = (1ull << k) - 1ull;
    for (int32_t t = gtile[g] + (tid / LANES); t < gtile[g + 1]; t += GI_THREADS / LANES) {
        const int32_t s = tiles[t].x;
        const int64_t o = A.off[s];
        const int32_t len = (int32_t)(A.off[s + 1] - o);
        const int32_t p0 = tiles[t].y + (tid & (LANES - 1)) * PER;
        if (p0 >= len) continue;
        const uint8_t *a = A.bases + o;
        KT km = 0, rc = 0;
        int32_t valid = 0;
        // soft-mask bits of the lane's k-mers (starts p0 .. p0 + 63, up to 64 + k - 1 <= 91 bits) in two words
        uint64_t mw0 = 0, mw1 = 0;
        if (A.mask_bits) {
            const int64_t gb = o + p0;
            mw0 = load8(A.mask_bits + (gb >> 3)) >> (gb & 7);
            const uint64_t hi = load8(A.mask_bits + (gb >> 3) + 8);
            if (gb & 7) mw0 |= hi << (64 - (gb & 7));
            mw1 = hi >> (gb & 7);  // bits 64 .. 64 + 56 of the window: k - 1 <= 27 are needed
        }
        uint64_t cur = load8(a + p0);
        for (int32_t wi = 0; wi * 8 < nroll; wi++) {
            const int32_t pn = p0 + 8 * (wi + 1);
            const uint64_t nxt = (8 * (wi + 1) < nroll && pn < len) ? load8(a + pn) : 0ull;
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int32_t x = wi * 8 + u, p = p0 + x;
                if (x >= nroll || p >= len) continue;
                const uint32_t c = (uint32_t)(cur >> (8 * u)) & 0xFFu;
                if (c < 4u) {
                    km = ((km << 2) | (KT)c) & mask;
                    rc = (rc >> 2) | ((KT)(3u - c) << rcsh);
                    valid++;
                } else {
                    km = 0;
                    rc = 0;
                    valid = 0;
                }
                const KT canon = km < rc ? km : rc;
                if (x < k - 1 || valid < k) continue;
                const uint32_t rel = (uint32_t)(canon >> shift) - sl0;
                if (rel >= (uint32_t)slice) continue;  // another slice's bucket
                const int32_t d = x - (k - 1);         // offset of the k-mer's first base in the lane's window
                const uint64_t mb = d == 0 ? mw0 : ((mw0 >> d) | (mw1 << (64 - d)));
                if (!kmer_sampled((uint64_t)canon, smp) || (mb & kones) != 0ull) continue;
                const uint32_t slot = atomicAdd(&cnt[rel], 1u);
                if (FILL)
                    ent[slot] = make_ulonglong2(((grp << (2 * k)) | (uint64_t)canon) | (km != canon ? 1ull << 63 : 0ull),
                                                ((uint64_t)s << 40) | (uint64_t)(goff[s] + (p - k + 1)));
            }
            cur = nxt;
        }
    }
    __syncthreads();
    for (int32_t i = tid; i < slice; i += GI_THREADS) dir[b0 + i] = cnt[i];
}
#define GI_INST(F, T)                                                                                               \
    template __global__ void k_group_index<F, T>(DbView, const int2 *, const int32_t *, int32_t, int32_t, int32_t, int32_t, \
                                                 int32_t, uint32_t *, ulonglong2 *, const int64_t *);
GI_INST(false, uint32_t)
GI_INST(true, uint32_t)
GI_INST(false, uint64_t)
GI_INST(true, uint64_t)
#undef GI_INST

// fat directory (dh_device.h): thread per bucket
__global__ void __launch_bounds__(256)
k_fat_dir(const uint32_t *__restrict__ dir, const ulonglong2 *__restrict__ ent, int64_t nb, ulonglong2 *__restrict__ fat)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    const uint32_t s0 = dir[b - 1], e0 = dir[b];
    ulonglong2 f;
    if (e0 == s0) {
        f.x = DH_FAT_EMPTY;
        f.y = 0;
    } else if (e0 - s0 == 1u)
        f = ent[s0];
    else {
        f.x = 1ull << 62;
        f.y = (unsigned long long)s0 | ((unsigned long long)(e0 - s0) << 32);
    }
    fat[b] = f;
}

template __global__ void k_kmer_pass<false>(DbView, const int2 *, int32_t, int32_t, int32_t, int32_t,
                                            uint32_t *, ulonglong2 *, const int64_t *);
template __global__ void k_kmer_pass<true>(DbView, const int2 *, int32_t, int32_t, int32_t, int32_t,
                                           uint32_t *, ulonglong2 *, const int64_t *);

// exclusive scan of n uint32 in place: block sums, scan of sums, add-back
#define SCAN_PER_BLOCK 2048
__global__ void __launch_bounds__(256) k_scan_sums(const uint32_t *__restrict__ v, int64_t n,
                                                   uint32_t *__restrict__ sums)
{
    __shared__ uint32_t red[256];
    const int64_t base = (int64_t)blockIdx.x * SCAN_PER_BLOCK;
    uint32_t acc = 0;
    for (int i = threadIdx.x; i < SCAN_PER_BLOCK; i += 256)
        if (base + i < n) acc += v[base + i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[blockIdx.x] = red[0];
}

// total64 (optional): the sum of all elements in 64 bits -- the caller's check that the 32-bit prefix sums did not wrap
__global__ void __launch_bounds__(1024) k_scan_top(uint32_t *__restrict__ sums, int32_t nb, unsigned long long *__restrict__ total64)
{
    // single block: serial chunks per thread then a block scan of the 1024 partials
    __shared__ uint32_t part[1024];
    const int32_t per = (nb + 1023) / 1024;
    const int32_t lo = threadIdx.x * per, hi = min(nb, lo + per);
    uint32_t acc = 0;
    unsigned long long acc64 = 0;
    for (int32_t i = lo; i < hi; i++) {
        acc += sums[i];
        acc64 += sums[i];
    }
    if (total64 && acc64) atomicAdd(total64, acc64);
    part[threadIdx.x] = acc;
    __syncthreads();
    // Hillis-Steele inclusive scan
    for (int s = 1; s < 1024; s <<= 1) {
        uint32_t add = (int)threadIdx.x >= s ? part[threadIdx.x - s] : 0;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t run = threadIdx.x ? part[threadIdx.x - 1] : 0;
    for (int32_t i = lo; i < hi; i++) {
        const uint32_t x = sums[i];
        sums[i] = run;
        run += x;
    }
}

__global__ void __launch_bounds__(256) k_scan_apply(uint32_t *__restrict__ v, int64_t n,
                                                    const uint32_t *__restrict__ sums)
{
    __shared__ uint32_t part[256];
    const int64_t base = (int64_t)blockIdx.x * SCAN_PER_BLOCK;
    const int per = SCAN_PER_BLOCK / 256;
    uint32_t loc[SCAN_PER_BLOCK / 256];
    uint32_t acc = 0;
    for (int i = 0; i < per; i++) {
        const int64_t idx = base + threadIdx.x * per + i;
        loc[i] = idx < n ? v[idx] : 0;
        acc += loc[i];
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 1; s < 256; s <<= 1) {
        uint32_t add = (int)threadIdx.x >= s ? part[threadIdx.x - s] : 0;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t run = sums[blockIdx.x] + (threadIdx.x ? part[threadIdx.x - 1] : 0);
    for (int i = 0; i < per; i++) {
        const int64_t idx = base + threadIdx.x * per + i;
        if (idx < n) v[idx] = run;
        run += loc[i];
    }
}

// ------------------------------------------------------------------------------------ K4

// development / tests: 0 = reads with a bucket above SORT_BMAX hits take the bitonic network as before round 6
// (DH_SEED_NO_REFINE=1; the order is the same either way)
__device__ int g_seed_sort_refine = 1;
#define HIT_QBITS 24
#define HIT_QMASK ((1u << HIT_QBITS) - 1u)
#define SEED_THREADS 512
#define SEED_LOOKUP_THREADS 512 /* threads that roll k-mers (whole wavefronts; fewer = longer serial chains = slower) */
#define SEED_CCAP 256 /* candidate band pairs collected per (read, strand) before ranking */

__device__ __forceinline__ int64_t hitD(uint64_t h) { return (int64_t)(h >> HIT_QBITS); }
__device__ __forceinline__ int32_t hitQ(uint64_t h) { return (int32_t)(h & HIT_QMASK); }

// covered-base contribution of sorted hit i (needs hit i-1)
__device__ __forceinline__ int32_t hit_cov(const uint64_t *h, int32_t i, int32_t k)
{
    if (i > 0 && hitD(h[i - 1]) == hitD(h[i])) {
        const int32_t dq = hitQ(h[i]) - hitQ(h[i - 1]);
        return dq < k ? dq : k;
    }
    return k;
}

// LCAP > 0: hits are staged in LDS (LCAP entries); items that do not fit are marked with
// ncand = -1 and redone by the LCAP == 0 instantiation, whose hit buffer is a slab of HBM
// (gcap entries per block, items taken from item_list) -- same code, same results.
#ifdef DH_SEED_PROF
__device__ unsigned long long g_seed_prof[12];
#define SP(i) if (tid == 0) { const unsigned long long t_ = wall_clock64(); atomicAdd(&g_seed_prof[i], t_ - tp_); tp_ = t_; }
#else
#define SP(i)
#endif
// One READ (both strands), processed by the whole block; `work` = index of the read in this launch,
// `slab` = index of the block's HBM hit slab (LCAP == 0).  The k-mers are rolled once over the forward
// read together with their reverse complements; the index is keyed by canonical k-mers with the
// orientation of the A k-mer in bit 63 of the key, so ONE lookup yields the hits of both strands:
// equal orientations = the forward read matches A, opposite = its reverse complement does (at
// position blen - k - q of the reverse-complemented read).  Hits carry the strand in their top bit,
// the band filter therefore never mixes strands; candidates go to the items 2r (forward) and 2r + 1.
#define HIT_DBITS 39
// JOIN: the hits come from the per-pile-up k-mer join (dh_join.hip) -- the read's segments of the hit buffer are
// gathered instead of looking its k-mers up; everything after the hit buffer is filled is the same code.
template <int LCAP, bool JOIN, int NT, int CC>
__device__ void seed_item(const DbView &B, const IndexView &ix, const JoinView &jv,
                          const DhOpts &o, int32_t read0, int32_t work, int32_t slab,
                          DhCand *__restrict__ cand_out, int32_t *__restrict__ ncand_out,
                          int32_t *__restrict__ nhits_out, int32_t *__restrict__ status,
                          uint64_t *__restrict__ gbuf, int32_t gcap, const int32_t *__restrict__ read_list)
{
    __shared__ uint64_t lhits[LCAP > 0 ? LCAP : 1];
    __shared__ DhCand cands[2 * CC];
    __shared__ int64_t cband[2 * CC];
    __shared__ int32_t s_n, s_nc;

    const int32_t r = read_list ? read_list[work] : read0 + work;  // (the HBM variant always works from a list)
    const int32_t item = 2 * r;
    // HBM variant: the block's slab holds gcap hits, gcap 64-bit prefix sums and gcap 32-bit head positions
    uint64_t *hits = LCAP > 0 ? lhits : gbuf + (int64_t)slab * (2 * (int64_t)gcap + (gcap + 1) / 2);
    const int32_t CAP = LCAP > 0 ? LCAP : gcap;
    const int tid = threadIdx.x;
#ifdef DH_SEED_PROF
    unsigned long long tp_ = wall_clock64();
#endif
    if (tid == 0) {
        s_n = 0;
        s_nc = 0;
    }
    __syncthreads();
    const int64_t bo = B.off[r];
    const int32_t blen = (int32_t)(B.off[r + 1] - bo);
    const uint8_t *b = B.bases + bo;
    const uint64_t grp = B.group ? (uint64_t)B.group[r] : 0ull;
    const int k = o.k;
    const uint64_t mask = (1ull << (2 * k)) - 1;
    const int32_t npos = blen - k + 1;
    constexpr uint64_t ORI = 1ull << 63, PAL = 1ull << 62;

    // ---- k-mer lookups: thread t rolls over a contiguous chunk of positions.  Sampled k-mers are
    // queued in registers (SEED_QN per lane); when the queue of ANY lane of the wavefront is full
    // every lane looks up what it holds: the directory words of all queued k-mers are fetched
    // back to back, then the first (key, value) entry of every non-empty bucket -- two memory
    // round trips per flush for all lanes together.  Buckets hold one entry almost always (the
    // directory has ~8 buckets per indexed k-mer); longer ones take the generic loop.
    // The phase is bound by the latency of each lane's serial chain (measured: halving the number
    // of rolling threads makes it 40 % slower), so every thread of the block takes a chunk.
    if (JOIN) {
        // the read's segments: one per slice of its group (segtab row), first hit << 24 | count.  Their prefix sums
        // and first hits overlay the candidate arrays, which are not in use yet.
        uint64_t *segb = (uint64_t *)cands;                     // [NT] first hit of segment s
        uint32_t *sego = (uint32_t *)(cands + CC) + 1;   // [-1 .. NT) exclusive prefix sums of the counts
        __shared__ uint32_t s_jw[NT / LANES];
        const int32_t ns = jv.gns ? jv.gns[B.group[r]] : jv.ns_fixed;
        const int64_t srow = jv.gns ? jv.segrow[r] : (int64_t)(r - jv.read0) * jv.ns_fixed;
        uint32_t c = 0;
        if (tid < ns) {
            const uint64_t sg = jv.segtab[srow + tid];
            c = (uint32_t)(sg & 0xFFFFFFull);
            segb[tid] = sg >> 24;
        }
        uint32_t incl = c;
        for (int off = 1; off < LANES; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off, LANES);
            if ((tid & (LANES - 1)) >= off) incl += up;
        }
        if ((tid & (LANES - 1)) == LANES - 1) s_jw[tid / LANES] = incl;
        __syncthreads();
        uint32_t base = 0, tot = 0;
        for (int wv = 0; wv < NT / LANES; wv++) {
            if (wv < tid / LANES) base += s_jw[wv];
            tot += s_jw[wv];
        }
        sego[tid] = base + incl;  // inclusive: sego[s - 1] = hits before segment s
        if (tid == 0) {
            sego[-1] = 0;
            s_n = (int32_t)tot;
        }
        __syncthreads();
        if ((int32_t)tot <= CAP) {
            // the segment of hit e: the last s with sego[s - 1] <= e (sego[-1] = 0; empty segments repeat a value and lose to
            // the one behind them).  Four hits per thread at a time, their searches a fixed number of steps without a branch:
            // the LDS round trips of the four overlap and so do the four loads from the hit buffer (one hit per iteration
            // made the gather a chain of dependent round trips: 106 of the 170 us a block spent on a pile-up read of 166)
            constexpr int GU = 4;
            int32_t top = 1;
            while (top < ns) top <<= 1;
            for (int32_t e0 = tid; e0 < (int32_t)tot; e0 += NT * GU) {
                int32_t lo[GU];
#pragma unroll
                for (int u = 0; u < GU; u++) lo[u] = 0;
                for (int32_t step = top >> 1; step > 0; step >>= 1) {
#pragma unroll
                    for (int u = 0; u < GU; u++) {
                        const int32_t idx = lo[u] + step;
                        const uint32_t e = (uint32_t)(e0 + u * NT);
                        if (idx < ns && sego[idx - 1] <= e) lo[u] = idx;
                    }
                }
                uint64_t v[GU];
#pragma unroll
                for (int u = 0; u < GU; u++) {
                    const int32_t e = e0 + u * NT;
                    v[u] = 0;
                    if (e < (int32_t)tot) v[u] = jv.hits[segb[lo[u]] + ((uint32_t)e - sego[lo[u] - 1])];
                }
#pragma unroll
                for (int u = 0; u < GU; u++) {
                    const int32_t e = e0 + u * NT;
                    if (e < (int32_t)tot) hits[e] = v[u];
                }
            }
        }
    } else if (npos > 0 && tid < SEED_LOOKUP_THREADS) {
        constexpr int QN = 4;
        const int32_t per = (npos + SEED_LOOKUP_THREADS - 1) / SEED_LOOKUP_THREADS;
        const int32_t q0 = tid * per, q1 = min(npos, q0 + per);
        const KmerSampler smp = kmer_sampler(o.kmer_mod, k);
        uint64_t km = 0, rc = 0;
        int32_t valid = 0;
        const int32_t pend = q0 < q1 ? q1 + k - 1 : q0;
        uint64_t qk[QN];
        int32_t qq[QN];
        int32_t nq = 0;
#pragma unroll
        for (int u = 0; u < QN; u++) {
            qk[u] = 0;
            qq[u] = 0;
        }
        auto emit = [&](uint64_t v, int32_t q, int32_t strand) {
            if (!(o.strands & (1 << strand))) return;
            const int32_t aseq = (int32_t)(v >> 40);
            if (o.skip_self == 1 && aseq == r) return;
            // tandem (datander): a read against itself, below the main diagonal only (position on A > position on B)
            if (o.skip_self == 3 && (aseq != r || (int64_t)(v & ((1ull << 40) - 1)) - ix.goff[r] - q < 1)) return;
            // symmetric: each unordered pair once; which read plays B alternates with the
            // parity of a + b, so every read is B for about half of its partners
            if (o.skip_self == 2 && (aseq == r || ((aseq < r) != (((aseq + r) & 1) == 0)))) return;
            if (o.skip_self == 2 && B.pflags && !dh_pair_seeded(B.pflags, aseq, r)) return;  // neither record is wanted
            const int64_t gv = (int64_t)(v & ((1ull << 40) - 1));
            const int32_t qs = strand ? blen - k - q : q;  // position on the oriented read
            const int64_t D = gv + ix.sepv - qs;
            const int32_t slot = atomicAdd(&s_n, 1);
            if (slot < CAP) hits[slot] = ((uint64_t)strand << 63) | ((uint64_t)D << HIT_QBITS) | (uint32_t)qs;
        };
        auto flush = [&]() {
#ifdef DH_SEED_NOLOAD
            nq = 0;  // development: the rolling alone (no lookups), for the split of the lookup phase
            return;
#endif
            // the fat directory word of every queued k-mer: one 16-byte load, one memory round trip per flush
            ulonglong2 f[QN];
#pragma unroll
            for (int u = 0; u < QN; u++) {
                f[u].x = DH_FAT_EMPTY;
                f[u].y = 0;
                if (u < nq) f[u] = ix.fat[(uint32_t)((qk[u] & ~(ORI | PAL)) >> ix.shift)];
            }
#pragma unroll
            for (int u = 0; u < QN; u++) {
                if (f[u].x == DH_FAT_EMPTY) continue;
                const uint64_t key = qk[u] & ~(ORI | PAL);
                const uint64_t bori = qk[u] & ORI;
                const bool pal = (qk[u] & PAL) != 0;
                if ((f[u].x >> 62) != 1ull) {  // the bucket's only entry
                    if ((f[u].x & ~ORI) == key && o.tcap >= 1) {
                        const bool same = (f[u].x & ORI) == bori;
                        if (same || pal) emit(f[u].y, qq[u], 0);
                        if (!same || pal) emit(f[u].y, qq[u], 1);
                    }
                    continue;
                }
                const uint32_t ss_u = (uint32_t)f[u].y, ee_u = ss_u + (uint32_t)(f[u].y >> 32);
                // -t cap: a k-mer occurring more than tcap times (per orientation) is skipped.  A bucket with at most
                // tcap entries cannot hold such a k-mer, so only larger buckets are counted first -- the count pass costs
                // one dependent load per entry, which for an unsampled index (the pile-up stage's: every intact k-mer of
                // a pile-up shares a bucket with its ~coverage copies) was half of the lookup phase
                bool dof = true, dor = true;
                if (ee_u - ss_u > (uint32_t)max(o.tcap, 0)) {
                    int32_t runf = 0, runr = 0;
                    for (uint32_t t = ss_u; t < ee_u; t++) {
                        const uint64_t ex = ix.ent[t].x;
                        if ((ex & ~ORI) != key) continue;
                        const bool same = (ex & ORI) == bori;
                        runf += (same || pal) ? 1 : 0;
                        runr += (!same || pal) ? 1 : 0;
                    }
                    dof = runf > 0 && runf <= o.tcap;
                    dor = runr > 0 && runr <= o.tcap;
                    if (!dof && !dor) continue;
                }
                // ... then emit its hits (the next entry is on its way while this one is handled)
                ulonglong2 nx = ix.ent[ss_u];
                for (uint32_t t = ss_u; t < ee_u; t++) {
                    const ulonglong2 en = nx;
                    if (t + 1 < ee_u) nx = ix.ent[t + 1];
                    if ((en.x & ~ORI) != key) continue;
                    const bool same = (en.x & ORI) == bori;
                    if (dof && (same || pal)) emit(en.y, qq[u], 0);
                    if (dor && (!same || pal)) emit(en.y, qq[u], 1);
                }
            }
            nq = 0;
        };
        const int rcsh = 2 * (k - 1);
        // warm-up: the first k - 1 bases of the chunk only fill the rolling k-mers
        uint64_t w = 0, wnext = q0 < pend ? load8(b + q0) : 0ull;
        for (int32_t t = 0; t < k - 1; t++) {
            const int32_t pp = q0 + t;
            if ((t & 7) == 0) {
                w = wnext;
                if (pp + 8 < pend) wnext = load8(b + pp + 8);
            }
            const uint8_t c = (uint8_t)w;
            w >>= 8;
            if (pp < pend) {
                if (c < 4) {
                    km = ((km << 2) | c) & mask;
                    rc = (rc >> 2) | ((uint64_t)(3 - c) << rcsh);
                    valid++;
                } else {
                    km = 0;
                    rc = 0;
                    valid = 0;
                }
            }
        }
        // uniform trip count so that the wavefront flushes together
        for (int32_t t = k - 1; t < per + k - 1; t++) {
            const int32_t pp = q0 + t;
            if ((t & 7) == 0) {
                w = wnext;
                if (pp + 8 < pend) wnext = load8(b + pp + 8);
            }
            const uint8_t c = (uint8_t)w;
            w >>= 8;
            if (pp < pend) {
                if (c < 4) {
                    km = ((km << 2) | c) & mask;
                    rc = (rc >> 2) | ((uint64_t)(3 - c) << rcsh);
                    valid++;
                } else {
                    km = 0;
                    rc = 0;
                    valid = 0;
                }
                const uint64_t canon = km < rc ? km : rc;
                bool em = valid >= k && kmer_sampled(canon, smp);
                if (em && B.mask_bits && mask_touch(B.mask_bits, bo + pp - k + 1, k)) em = false;
                if (em) {
                    const uint64_t key = ((grp << (2 * k)) | canon) | (km != canon ? ORI : 0ull) | (km == rc ? PAL : 0ull);
                    const int32_t q = pp - k + 1;
#pragma unroll
                    for (int u = 0; u < QN; u++)
                        if (u == nq) {
                            qk[u] = key;
                            qq[u] = q;
                        }
                    nq++;
                }
            }
            if (__ballot(nq == QN) != 0ull) flush();
        }
        if (__ballot(nq > 0) != 0ull) flush();
    }
    __syncthreads();
    SP(0)
    int32_t n = s_n;
    if (n > CAP) {
        // capacity exceeded: never silently truncated.  LDS variant: hand the read to the HBM
        // variant (ncand = -1); HBM variant: report
        if (tid == 0) {
            if (LCAP == 0) atomicOr(status, DH_ST_HIT_OVERFLOW);
            ncand_out[item] = ncand_out[item + 1] = LCAP > 0 ? -1 : 0;
            nhits_out[item] = n;  // what the HBM slab has to hold (both strands)
            nhits_out[item + 1] = 0;
        }
        return;
    }
    if (n == 0) {
        if (tid == 0) ncand_out[item] = ncand_out[item + 1] = nhits_out[item] = nhits_out[item + 1] = 0;
        return;
    }
    // ---- sort of the hit buffer (keys are distinct: a hit is (strand, diagonal, read position))
    int32_t N = 1;
    while (N < n) N <<= 1;
    constexpr bool SMALL = NT < SEED_THREADS;  // a wavefront per read (the mapping launches' first tier): LCAP <= 8 NT
    if (SMALL) {
        // every thread takes the (at most LCAP / NT) keys tid, tid + NT, ... and counts the keys below each of them: n
        // broadcast reads for all of its keys together; keys are distinct, the ranks a permutation
        // (as many keys per thread as the read needs: 140 hits at 1/8 sampling are three)
        constexpr int E8 = LCAP / NT > 0 ? LCAP / NT : 1;
        uint64_t ky[E8];
        int32_t rk8[E8];
#pragma unroll
        for (int u = 0; u < E8; u++) {
            const int32_t i = tid + u * NT;
            ky[u] = i < n ? hits[i] : ~0ull;
            rk8[u] = 0;
        }
#define DH_RANK_KEYS(M_)                                                   \
    for (int32_t x = 0; x < n; x++) {                                      \
        const uint64_t h = hits[x];                                        \
        _Pragma("unroll") for (int u = 0; u < (M_ < E8 ? M_ : E8); u++) rk8[u] += h < ky[u] ? 1 : 0; \
    }
        if (n <= 2 * NT) {
            DH_RANK_KEYS(2)
        } else if (n <= 4 * NT) {
            DH_RANK_KEYS(4)
        } else {
            DH_RANK_KEYS(E8)
        }
#undef DH_RANK_KEYS
        __syncthreads();
#pragma unroll
        for (int u = 0; u < E8; u++)
            if (tid + u * NT < n) hits[rk8[u]] = ky[u];
        N = 1;  // the network below has nothing left to do
    } else if (LCAP > 0 && n <= NT) {
        // at most one hit per thread (the mapping launches: 140 hits per read at kmer_mod 8): every thread counts the
        // keys below its own -- n broadcast reads that do not depend on each other -- and stores its key at that rank.
        // The bitonic network below takes log^2 N dependent LDS round trips (36 for N = 256: 8.6 of the 50 us a block
        // spent per read)
        uint64_t key = 0;
        int32_t rk = 0;
        if (tid < n) {
            key = hits[tid];
#pragma unroll 4
            for (int32_t x = 0; x < n; x++) rk += hits[x] < key ? 1 : 0;
        }
        __syncthreads();
        if (tid < n)
            hits[rk] = key;
        else if (tid < N)
            hits[tid] = ~0ull;
        N = 1;  // the network below has nothing left to do
    } else if (LCAP == 0 || ((LCAP <= 8192 || JOIN) && (JOIN || LCAP >= 4096))) {  // (not the mapping launches' small variants: registers)
        // More than one hit per thread (the pile-up all-vs-all: 2 500 hits per read, where the network below was 55 of the
        // 97 us a block spent per read): the hits of a read cluster on the diagonals of its overlaps, so they are dealt
        // into 2 x 1024 diagonal buckets (strand, then equal slices of the read's diagonal range: a counting pass, a scan,
        // a scatter through registers) and every hit takes its rank among the few hits of its bucket.  A bucket that grew
        // beyond SORT_BMAX hits (a repeat) sends the read through the network instead -- the same order either way.
        // The HBM variant (the few reads with more hits than any LDS buffer holds) scatters into the slab's prefix-sum area
        // instead of registers; its network is a chain of global round trips per exchange (8 ms for 4 reads of a
        // configs[2] part).  Ranking costs (n / 512) x bucket loads from L2 per thread, the network ~0.5 ms at 16 384
        // hits: measured break-even at buckets of ~340 hits.
        constexpr int E = LCAP >= NT ? LCAP / NT : 1;
        constexpr int NB = 2048, NBH = NB / 2, SORT_BMAX = LCAP == 0 ? 384 : 256;
        constexpr uint64_t DM = (1ull << HIT_DBITS) - 1;
        static_assert(SMALL || sizeof(cands) >= NB * sizeof(uint32_t), "bucket counters overlay the candidate array");
        uint32_t *bcnt = (uint32_t *)cands;  // not in use yet (the join's segment table is done with it)
        __shared__ unsigned long long s_dmin, s_dmax;
        __shared__ uint32_t s_bw[NT / LANES];
        __shared__ uint32_t s_bmax;
        // REFINE (round 6): buckets above SORT_BMAX hits are sorted by a second counting pass over their own diagonal range
        // instead of sending the whole read through the network (below).  The mapping launches need it: a read's ~900 true
        // hits at kmer_mod 1 lie on a few hundred neighbouring diagonals while its handful of chance hits stretch the
        // diagonal range over the whole assembly, so the slices are 10^5 diagonals wide and one of them holds everything --
        // 88 % of the reads of configs[2] took the network, 25.6 of the 41 us a block spent per read.
        constexpr int HV = 8, NB2 = 1024;            // heavy buckets a read may have; slices of the second pass
        constexpr bool REFINE = LCAP > 0 && !SMALL;
        static_assert(!REFINE || sizeof(cband) >= NB2 * sizeof(uint32_t), "the second pass's counters overlay the band array");
        uint32_t *fcnt = (uint32_t *)cband;          // not in use yet
        __shared__ uint32_t s_nheavy, s_heavy[HV];
        for (int32_t i = tid; i < NB; i += NT) bcnt[i] = 0;
        if (tid == 0) {
            s_dmin = ~0ull;
            s_dmax = 0ull;
            s_bmax = 0;
            s_nheavy = 0;
        }
        unsigned long long dmin = ~0ull, dmax = 0ull;
        for (int32_t i = tid; i < n; i += NT) {
            const unsigned long long d = (hits[i] >> HIT_QBITS) & DM;
            dmin = d < dmin ? d : dmin;
            dmax = d > dmax ? d : dmax;
        }
        for (int off = LANES / 2; off > 0; off >>= 1) {
            const unsigned long long a = __shfl_xor(dmin, off, LANES), c = __shfl_xor(dmax, off, LANES);
            dmin = a < dmin ? a : dmin;
            dmax = c > dmax ? c : dmax;
        }
        __syncthreads();
        if ((tid & (LANES - 1)) == 0) {
            atomicMin(&s_dmin, dmin);
            atomicMax(&s_dmax, dmax);
        }
        __syncthreads();
        // (slices aligned to their width: the hits of a bucket then differ in their low 24 + sh bits only)
        uint64_t d0 = s_dmin;
        int sh = 0;
        while (((s_dmax - d0) >> sh) >= (uint64_t)NBH) {
            sh++;
            d0 = s_dmin & ~((1ull << sh) - 1);
        }
        auto bucket = [&](uint64_t key) {
            return (uint32_t)(key >> 63) * NBH + (uint32_t)((((key >> HIT_QBITS) & DM) - d0) >> sh);
        };
        for (int32_t i = tid; i < n; i += NT) atomicAdd(&bcnt[bucket(hits[i])], 1u);
        __syncthreads();
        // exclusive scan of the counters (4 per thread), largest bucket
        uint32_t c4[NB / NT], sum = 0, mx = 0;
#pragma unroll
        for (int u = 0; u < NB / NT; u++) {
            c4[u] = bcnt[tid * (NB / NT) + u];
            sum += c4[u];
            mx = c4[u] > mx ? c4[u] : mx;
        }
        uint32_t incl = sum;
        for (int off = 1; off < LANES; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off, LANES);
            if ((tid & (LANES - 1)) >= off) incl += up;
        }
        for (int off = LANES / 2; off > 0; off >>= 1) {
            const uint32_t a = __shfl_xor(mx, off, LANES);
            mx = a > mx ? a : mx;
        }
        if ((tid & (LANES - 1)) == LANES - 1) s_bw[tid / LANES] = incl;
        if ((tid & (LANES - 1)) == 0) atomicMax(&s_bmax, mx);
        __syncthreads();
        uint32_t base = incl - sum;
        for (int wv = 0; wv < tid / LANES; wv++) base += s_bw[wv];
#pragma unroll
        for (int u = 0; u < NB / NT; u++) {
            bcnt[tid * (NB / NT) + u] = base;
            base += c4[u];
        }
        if (REFINE) {
#pragma unroll
            for (int u = 0; u < NB / NT; u++)
                if (c4[u] > (uint32_t)SORT_BMAX) {
                    const uint32_t slot = atomicAdd(&s_nheavy, 1u);
                    if (slot < (uint32_t)HV) s_heavy[slot] = (uint32_t)(tid * (NB / NT) + u);
                }
        }
        uint64_t ke[E];
        if (LCAP > 0) {
#pragma unroll
            for (int u = 0; u < E; u++) {
                const int32_t i = tid + u * NT;
                ke[u] = i < n ? hits[i] : 0ull;
            }
        }
        __syncthreads();
        SP(5)
        const bool refine = REFINE && g_seed_sort_refine && s_bmax > (uint32_t)SORT_BMAX && s_nheavy <= (uint32_t)HV;
        if (LCAP == 0 && s_bmax <= (uint32_t)SORT_BMAX) {
            uint64_t *tmp = hits + gcap;  // the block's prefix sums live here later
            for (int32_t i = tid; i < n; i += NT) {
                const uint64_t key = hits[i];
                tmp[atomicAdd(&bcnt[bucket(key)], 1u)] = key;
            }
            __syncthreads();
            for (int32_t i = tid; i < n; i += NT) {
                const uint64_t key = tmp[i];
                const uint32_t bk = bucket(key);
                const uint32_t b0 = bk ? bcnt[bk - 1] : 0u, b1 = bcnt[bk];
                uint32_t rk = b0, x = b0;
                for (; x + 8 <= b1; x += 8) {
                    uint64_t h[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) h[j] = tmp[x + j];
#pragma unroll
                    for (int j = 0; j < 8; j++) rk += h[j] < key ? 1u : 0u;
                }
                for (; x < b1; x++) rk += tmp[x] < key ? 1u : 0u;
                hits[rk] = key;
            }
            N = 1;
        } else if (LCAP > 0 && (s_bmax <= (uint32_t)SORT_BMAX || refine)) {
            // scatter: a bucket's hits in arrival order; the counters end up at the buckets' ends
#pragma unroll
            for (int u = 0; u < E; u++) {
                const int32_t i = tid + u * NT;
                if (i < n) hits[atomicAdd(&bcnt[bucket(ke[u])], 1u)] = ke[u];
            }
            __syncthreads();
            SP(6)
            uint32_t dst[E];
            // keys below `key` among hits[x0, x1) (keys are distinct; eight loads in flight: one at a time made every compare
            // a full LDS round trip; `low`: the keys of the range agree above their low words, which then decide -- half the
            // LDS traffic of this loop, which is bound by it)
            auto count_below = [&](uint32_t x0, uint32_t x1, uint64_t key, bool low) {
                uint32_t rk = 0, x = x0;
                if (low) {
                    const uint32_t *h32 = (const uint32_t *)hits;
                    const uint32_t key32 = (uint32_t)key;
                    for (; x + 8 <= x1; x += 8) {
                        uint32_t h[8];
#pragma unroll
                        for (int j = 0; j < 8; j++) h[j] = h32[2 * (x + j)];
#pragma unroll
                        for (int j = 0; j < 8; j++) rk += h[j] < key32 ? 1u : 0u;
                    }
                }
                for (; x + 8 <= x1; x += 8) {
                    uint64_t h[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) h[j] = hits[x + j];
#pragma unroll
                    for (int j = 0; j < 8; j++) rk += h[j] < key ? 1u : 0u;
                }
                for (; x + 4 <= x1; x += 4) {
                    const uint64_t h0 = hits[x], h1 = hits[x + 1], h2 = hits[x + 2], h3 = hits[x + 3];
                    rk += (h0 < key ? 1u : 0u) + (h1 < key ? 1u : 0u) + (h2 < key ? 1u : 0u) + (h3 < key ? 1u : 0u);
                }
                for (; x < x1; x++) rk += hits[x] < key ? 1u : 0u;
                return rk;
            };
            const uint32_t nheavy = refine ? s_nheavy : 0u;
            if (REFINE) {
                // ---- second pass, one heavy bucket at a time: its hits [hb0, hb1) are dealt into NB2 slices of the bucket's
                // own diagonal range (counting pass, scan, scatter through registers) and ranked inside their slice; the
                // bucket ends up sorted in place.  A slice that is still long (hundreds of hits on one diagonal: a
                // low-complexity read) only makes its ranking loop longer.
                for (uint32_t hv = 0; hv < nheavy; hv++) {
                    const uint32_t hb = s_heavy[hv];
                    const uint32_t hb0 = hb ? bcnt[hb - 1] : 0u, hb1 = bcnt[hb];
                    if (tid == 0) {
                        s_dmin = ~0ull;
                        s_dmax = 0ull;
                    }
                    for (int32_t i = tid; i < NB2; i += NT) fcnt[i] = 0;
                    unsigned long long lmin = ~0ull, lmax = 0ull;
                    for (uint32_t i = hb0 + tid; i < hb1; i += NT) {
                        const unsigned long long d = (hits[i] >> HIT_QBITS) & DM;
                        lmin = d < lmin ? d : lmin;
                        lmax = d > lmax ? d : lmax;
                    }
                    for (int off = LANES / 2; off > 0; off >>= 1) {
                        const unsigned long long a = __shfl_xor(lmin, off, LANES), c = __shfl_xor(lmax, off, LANES);
                        lmin = a < lmin ? a : lmin;
                        lmax = c > lmax ? c : lmax;
                    }
                    __syncthreads();
                    if ((tid & (LANES - 1)) == 0) {
                        atomicMin(&s_dmin, lmin);
                        atomicMax(&s_dmax, lmax);
                    }
                    __syncthreads();
                    // (slices aligned to their width, as the buckets are: the hits of a slice then agree above their low
                    // 24 + sh2 bits, which is what lets count_below compare low words)
                    uint64_t e0 = s_dmin;
                    int sh2 = 0;
                    while (((s_dmax - e0) >> sh2) >= (uint64_t)NB2) {
                        sh2++;
                        e0 = s_dmin & ~((1ull << sh2) - 1);
                    }
                    auto slice = [&](uint64_t key) { return (uint32_t)((((key >> HIT_QBITS) & DM) - e0) >> sh2); };
                    for (uint32_t i = hb0 + tid; i < hb1; i += NT) atomicAdd(&fcnt[slice(hits[i])], 1u);
                    __syncthreads();
                    uint32_t f2[NB2 / NT > 0 ? NB2 / NT : 1], fsum = 0;
#pragma unroll
                    for (int u = 0; u < NB2 / NT; u++) {
                        f2[u] = fcnt[tid * (NB2 / NT) + u];
                        fsum += f2[u];
                    }
                    uint32_t fincl = fsum;
                    for (int off = 1; off < LANES; off <<= 1) {
                        const uint32_t up = __shfl_up(fincl, off, LANES);
                        if ((tid & (LANES - 1)) >= off) fincl += up;
                    }
                    if ((tid & (LANES - 1)) == LANES - 1) s_bw[tid / LANES] = fincl;
                    __syncthreads();
                    uint32_t fbase = hb0 + fincl - fsum;
                    for (int wv = 0; wv < tid / LANES; wv++) fbase += s_bw[wv];
#pragma unroll
                    for (int u = 0; u < NB2 / NT; u++) {
                        fcnt[tid * (NB2 / NT) + u] = fbase;
                        fbase += f2[u];
                    }
#pragma unroll
                    for (int u = 0; u < E; u++) {
                        const uint32_t i = hb0 + (uint32_t)tid + (uint32_t)u * NT;
                        ke[u] = i < hb1 ? hits[i] : 0ull;
                    }
                    __syncthreads();
#pragma unroll
                    for (int u = 0; u < E; u++) {
                        const uint32_t i = hb0 + (uint32_t)tid + (uint32_t)u * NT;
                        if (i < hb1) hits[atomicAdd(&fcnt[slice(ke[u])], 1u)] = ke[u];
                    }
                    __syncthreads();  // fcnt[f] = end of slice f (absolute positions)
#pragma unroll
                    for (int u = 0; u < E; u++) {
                        const uint32_t i = hb0 + (uint32_t)tid + (uint32_t)u * NT;
                        dst[u] = 0;
                        if (i < hb1) {
                            const uint64_t key = hits[i];
                            ke[u] = key;
                            const uint32_t f = slice(key);
                            const uint32_t f0 = f ? fcnt[f - 1] : hb0, f1 = fcnt[f];
                            dst[u] = f0 + count_below(f0, f1, key, sh2 <= 32 - HIT_QBITS);
                        }
                    }
                    __syncthreads();
#pragma unroll
                    for (int u = 0; u < E; u++)
                        if (hb0 + (uint32_t)tid + (uint32_t)u * NT < hb1) hits[dst[u]] = ke[u];
                    __syncthreads();
                }
            }
#pragma unroll
            for (int u = 0; u < E; u++) {
                const int32_t i = tid + u * NT;
                dst[u] = 0;
                if (i < n) {
                    const uint64_t key = hits[i];
                    ke[u] = key;
                    const uint32_t bk = bucket(key);
                    const uint32_t b0 = bk ? bcnt[bk - 1] : 0u, b1 = bcnt[bk];
                    bool heavy = false;
                    if (REFINE)
                        for (uint32_t hv = 0; hv < nheavy; hv++) heavy = heavy || s_heavy[hv] == bk;
                    if (heavy)  // sorted by the second pass
                        dst[u] = (uint32_t)i;
                    else
                        dst[u] = b0 + count_below(b0, b1, key, sh <= 32 - HIT_QBITS);
                }
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < E; u++)
                if (tid + u * NT < n) hits[dst[u]] = ke[u];
            N = 1;
        } else {
#ifdef DH_SEED_PROF
            if (tid == 0) atomicAdd(&g_seed_prof[10], 1ull);
#endif
            for (int32_t i = n + tid; i < N; i += NT) hits[i] = ~0ull;
        }
    } else {
        for (int32_t i = n + tid; i < N; i += NT) hits[i] = ~0ull;
    }
    __syncthreads();
    // Pair p exchanges elements i = insert-zero-bit(p, j) and i | j.  Pairs are dealt to threads in
    // runs of 64, so for strides j < 128 both elements of every pair of a wavefront live in that
    // wavefront's own 128-element blocks: those rounds need no block barrier (LDS operations of
    // one wavefront execute in order), only the rounds with j >= 128 do.
    for (int32_t kk = 2; kk <= N; kk <<= 1) {
        for (int32_t j = kk >> 1; j > 0; j >>= 1) {
            const bool cross = j >= 128;
            if (cross) __syncthreads();
            // pairs in batches of SORT_U: all loads of a batch are issued before the first exchange is stored (the pairs
            // of a round are disjoint).  One pair at a time made every pair a full memory round trip -- 64 of them in a
            // row per thread and round when 50 000 hits of a repeat-rich read are sorted in the HBM slab (17 ms for the
            // 27 such reads of a configs[2] half)
            constexpr int SORT_U = (LCAP == 0 || LCAP >= 4096) ? 8 : 4;
            for (int32_t p0 = tid; p0 < (N >> 1); p0 += NT * SORT_U) {
                uint64_t xs[SORT_U], ys[SORT_U];
#pragma unroll
                for (int u = 0; u < SORT_U; u++) {
                    const int32_t p = p0 + u * NT;
                    if (p < (N >> 1)) {
                        const int32_t i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
                        xs[u] = hits[i];
                        ys[u] = hits[i | j];
                    }
                }
#pragma unroll
                for (int u = 0; u < SORT_U; u++) {
                    const int32_t p = p0 + u * NT;
                    if (p < (N >> 1)) {
                        const int32_t i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
                        const bool up = (i & kk) == 0;
                        if ((xs[u] > ys[u]) == up) {
                            hits[i] = ys[u];
                            hits[i | j] = xs[u];
                        }
                    }
                }
            }
            if (cross)
                __syncthreads();
            else
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    if (tid == 0) {  // hits per strand: the forward strand sorts first
        int32_t lo = 0, hi = n;
        while (lo < hi) {
            const int32_t mid = (lo + hi) >> 1;
            if (hits[mid] >> 63)
                hi = mid;
            else
                lo = mid + 1;
        }
        nhits_out[item] = lo;
        nhits_out[item + 1] = n - lo;
    }
    SP(1)
    // ---- band pairs.  Small variants (FASTB): one block-wide inclusive scan over
    // (band-head flag << 18 | covered-base contribution) gives every band its coverage as a
    // difference of two prefix sums and the compacted list of band heads, so the work is spread
    // over all threads instead of one serial walk per band; large variants (no LDS to spare) walk
    // the four bands from every band head.
    // The 8192-entry variant has no LDS to spare either, but its two arrays fit a per-block slab of
    // global scratch (48 KB, L2 resident since the persistent block reuses it): the parallel scan beats
    // the serial walks by far (pile-up all-vs-all: 183 -> about 30 us per read).  18 bits of coverage and
    // 14 bits of head count hold up to 8192 hits of k <= 28.
    // The HBM variant (LCAP == 0: the few reads whose hits -- tens of thousands for a repeat-rich read -- overflow the
    // LDS buffer) scans as well, with 64-bit sums (32 bits of coverage, 32 of head count) and 32-bit head positions in
    // the block's slab behind the hits: the serial walks cost a chain of dependent L2 round trips per hit of a band,
    // 5 - 21 ms for the 27 such reads of a configs[2] half (one block each).
    // (the 16384-entry variant fed from segments -- uncapped pile-ups: 166 reads, ~10 000 hits per read -- scans as well,
    // with the wide sums of the HBM variant in a slab of 24 576 words per block; the directory-fed one keeps the walks)
    constexpr bool FASTB = LCAP <= 8192 || (JOIN && LCAP == 16384);
    constexpr bool FB_LDS = LCAP > 0 && LCAP <= 4096;
    constexpr bool FB_BIG = LCAP == 0;
    constexpr bool FB_WIDE = LCAP == 0 || LCAP == 16384;
    using bsum_t = typename std::conditional<FB_WIDE, uint64_t, uint32_t>::type;
    using bhead_t = typename std::conditional<FB_WIDE, uint32_t, uint16_t>::type;
    constexpr int HSH = FB_WIDE ? 32 : 18;
    constexpr bsum_t CMASK = ((bsum_t)1 << HSH) - 1;
    __shared__ uint32_t bsum_l[FB_LDS ? LCAP : 1];   // inclusive prefix sums
    __shared__ uint16_t bhead_l[FB_LDS ? LCAP : 1];  // positions of the band heads
    bsum_t *bsum = FB_LDS ? (bsum_t *)bsum_l
                          : (FB_BIG ? (bsum_t *)(hits + gcap) : (bsum_t *)(gbuf + (int64_t)slab * gcap));
    bhead_t *bhead = FB_LDS ? (bhead_t *)bhead_l : (bhead_t *)(bsum + (LCAP > 0 ? LCAP : gcap));
    __shared__ bsum_t s_wsum[NT / LANES];
    __shared__ int32_t s_nbig;
    constexpr int NBIG = NT < SEED_THREADS ? 8 : ((LCAP > 0 && LCAP <= 4096) ? 128 : 64);  // (the 8192-entry variant has no LDS to spare; what does not fit walks serially)
    __shared__ int32_t bigc[NBIG][4];  // candidate band pairs with long hit ranges: (first, end, P, slot)
    __shared__ unsigned long long s_bestkeys[NBIG];
    const int bs = o.band_shift;
    // seed of a band pair [i, e1): first hit of the same-diagonal run (steps <= k) covering most
    // bases; then the candidate record
    auto emit_cand = [&](int32_t slot, int32_t best_first, int32_t P, int64_t band) {
        const int64_t D = hitD(hits[best_first]) & ((1ll << HIT_DBITS) - 1);  // without the strand bit
        const int32_t q = hitQ(hits[best_first]);
        const int64_t gv = D - ix.sepv + q;
        // sequences start on 4096-base pages of the virtual axis: the page names the sequence (the binary search
        // over goff this replaces was a chain of ten dependent loads per candidate)
        const int32_t lo = ix.page_seq[gv >> 12];
        cands[slot].score = P;
        cands[slot].aseq = lo;
        cands[slot].apos = (int32_t)(gv - ix.goff[lo]);
        cands[slot].bpos = q;
        cband[slot] = band;
    };
    auto serial_seed = [&](int32_t i, int32_t e1) {
        int32_t best_first = i, best_cov = -1, run_first = i;
        for (int32_t x = i; x < e1; x++) {
            bool linked = false;
            if (x > i && hitD(hits[x]) == hitD(hits[x - 1]))
                linked = (hitQ(hits[x]) - hitQ(hits[x - 1])) <= k;
            if (!linked) run_first = x;
            const int32_t cov = k + hitQ(hits[x]) - hitQ(hits[run_first]);
            if (cov > best_cov) {
                best_cov = cov;
                best_first = run_first;
            }
        }
        return best_first;
    };
    if (FASTB) {
        if (tid == 0) s_nbig = 0;
        // -- scan: thread t owns the elements [t * per, t * per + per)
        const int32_t per = (n + NT - 1) / NT;
        const int32_t x0 = tid * per, x1 = min(n, x0 + per);
        // (LCAP > 0: the first pass only sums -- the hits are in LDS, the sums of the 8192 / 16384-entry variants in a slab of
        // global memory: storing the partial sums here and loading them back below was a chain of dependent round trips per
        // element; the second pass recomputes an element's term from the hits instead)
        bsum_t acc = 0;
        for (int32_t i = x0; i < x1; i++) {
            const bool head = i == 0 || (hitD(hits[i - 1]) >> bs) != (hitD(hits[i]) >> bs);
            acc += ((bsum_t)(head ? 1u : 0u) << HSH) | (bsum_t)hit_cov(hits, i, k);
            if (LCAP == 0) bsum[i] = acc;
        }
        bsum_t incl = acc;  // inclusive scan of the per-thread totals: inside the wavefront ...
        for (int off = 1; off < LANES; off <<= 1) {
            const bsum_t up = __shfl_up(incl, off, LANES);
            if ((tid & (LANES - 1)) >= off) incl += up;
        }
        if ((tid & (LANES - 1)) == LANES - 1) s_wsum[tid / LANES] = incl;
        __syncthreads();
        bsum_t base = incl - acc;  // ... plus the wavefronts before this one
        for (int wv = 0; wv < tid / LANES; wv++) base += s_wsum[wv];
        bsum_t run = base;
        for (int32_t i = x0; i < x1; i++) {
            const bool head = i == 0 || (hitD(hits[i - 1]) >> bs) != (hitD(hits[i]) >> bs);
            bsum_t v;
            if (LCAP == 0)
                v = bsum[i] + base;
            else {
                run += ((bsum_t)(head ? 1u : 0u) << HSH) | (bsum_t)hit_cov(hits, i, k);
                v = run;
            }
            bsum[i] = v;
            if (head) bhead[(v >> HSH) - 1] = (bhead_t)i;
        }
        __syncthreads();
        SP(2)
        const int32_t nheads = (int32_t)(bsum[n - 1] >> HSH);
        auto band_cov = [&](int32_t rnk) {  // coverage of the band with head number rnk
            const int32_t st_ = bhead[rnk], en_ = rnk + 1 < nheads ? bhead[rnk + 1] : n;
            return (int32_t)((bsum[en_ - 1] & CMASK) - (st_ ? (bsum[st_ - 1] & CMASK) : (bsum_t)0));
        };
        for (int32_t rnk = tid; rnk < nheads; rnk += NT) {
            const int32_t i = bhead[rnk];
            const int64_t band = hitD(hits[i]) >> bs;
            int32_t covm1 = 0, cov1 = 0, cov2 = 0, e1;
            const int32_t cov0 = band_cov(rnk);
            if (rnk > 0 && (hitD(hits[bhead[rnk - 1]]) >> bs) == band - 1) covm1 = band_cov(rnk - 1);
            int32_t nx = rnk + 1;  // head number of the next band present
            e1 = nx < nheads ? bhead[nx] : n;
            if (nx < nheads && (hitD(hits[bhead[nx]]) >> bs) == band + 1) {
                cov1 = band_cov(nx);
                nx++;
                e1 = nx < nheads ? bhead[nx] : n;
            }
            if (nx < nheads && (hitD(hits[bhead[nx]]) >> bs) == band + 2) cov2 = band_cov(nx);
            const int32_t P = cov0 + cov1, Pm1 = covm1 + cov0, Pp1 = cov1 + cov2;
            if (P < o.hmin || P < Pm1 || P <= Pp1) continue;
            const int32_t slot = atomicAdd(&s_nc, 1);
            if (slot >= 2 * CC) continue;
            if (e1 - i > 16) {
                // long range: the whole block picks the seed below
                const int32_t bslot = atomicAdd(&s_nbig, 1);
                if (bslot < NBIG) {
                    bigc[bslot][0] = i;
                    bigc[bslot][1] = e1;
                    bigc[bslot][2] = P;
                    bigc[bslot][3] = slot;
                    continue;
                }
            }
            emit_cand(slot, serial_seed(i, e1), P, band);
        }
        __syncthreads();
        SP(8)
        // long ranges: 16 lanes per candidate, 16 consecutive hits at a time.  The first hit of the run a hit belongs to
        // (a run = hits of one diagonal at most k apart) is the running maximum of the run heads' positions -- a scan
        // over the 16 lanes plus the carry of the lanes before --, not a walk back from every run end: the walks were a
        // chain of dependent LDS round trips as long as the longest run of the wavefront (13 of the 97 us per read)
        constexpr int GW = 16;
        const int32_t nbig = min(s_nbig, NBIG);
        const int gl = tid & (GW - 1);
        for (int32_t bc = tid / GW; bc < nbig; bc += NT / GW) {
            const int32_t i = bigc[bc][0], e1 = bigc[bc][1];
            unsigned long long best = 0ull;
            int32_t carry = i;
            for (int32_t base = i; base < e1; base += GW) {
                const int32_t x = base + gl;
                const bool valid = x < e1;
                const uint64_t h = valid ? hits[x] : 0ull;
                const uint64_t hp = valid && x > i ? hits[x - 1] : 0ull;
                const uint64_t hn = x + 1 < e1 ? hits[x + 1] : 0ull;
                const bool linked = valid && x > i && hitD(h) == hitD(hp) && (hitQ(h) - hitQ(hp)) <= k;
                int32_t f = valid && !linked ? x : -1;
                for (int off = 1; off < GW; off <<= 1) {
                    const int32_t up = __shfl_up(f, off, GW);
                    if (gl >= off) f = up > f ? up : f;
                }
                f = carry > f ? carry : f;
                carry = __shfl(f, GW - 1, GW);
                // a run ends where the next hit is not linked; its coverage is the largest of the run
                const bool last = valid && (x + 1 >= e1 || hitD(hn) != hitD(h) || (hitQ(hn) - hitQ(h)) > k);
                if (last) {
                    const uint32_t cov = (uint32_t)(k + hitQ(h) - hitQ(hits[f]));
                    // largest coverage, then the earliest run
                    const unsigned long long key = ((unsigned long long)cov << 32) | (uint32_t)(0x7FFFFFFF - f);
                    best = key > best ? key : best;
                }
            }
            for (int off = GW / 2; off > 0; off >>= 1) {
                const unsigned long long ot = __shfl_xor(best, off, GW);
                best = ot > best ? ot : best;
            }
            if (gl == 0) s_bestkeys[bc] = best;
        }
        __syncthreads();
        SP(9)
        for (int32_t bc = tid; bc < nbig; bc += NT)
            emit_cand(bigc[bc][3], 0x7FFFFFFF - (int32_t)(uint32_t)s_bestkeys[bc], bigc[bc][2], hitD(hits[bigc[bc][0]]) >> bs);
    } else {
        for (int32_t i = tid; i < n; i += NT) {
            const int64_t band = hitD(hits[i]) >> bs;
            if (i > 0 && (hitD(hits[i - 1]) >> bs) == band) continue;  // not a band head
            int32_t covm1 = 0, cov0 = 0, cov1 = 0, cov2 = 0, e1;
            for (int32_t j = i - 1; j >= 0 && (hitD(hits[j]) >> bs) == band - 1; j--)
                covm1 += hit_cov(hits, j, k);
            int32_t j = i;
            for (; j < n && (hitD(hits[j]) >> bs) == band; j++) cov0 += hit_cov(hits, j, k);
            for (; j < n && (hitD(hits[j]) >> bs) == band + 1; j++) cov1 += hit_cov(hits, j, k);
            e1 = j;
            for (; j < n && (hitD(hits[j]) >> bs) == band + 2; j++) cov2 += hit_cov(hits, j, k);
            const int32_t P = cov0 + cov1, Pm1 = covm1 + cov0, Pp1 = cov1 + cov2;
            if (P < o.hmin || P < Pm1 || P <= Pp1) continue;
            const int32_t slot = atomicAdd(&s_nc, 1);
            if (slot < 2 * CC) emit_cand(slot, serial_seed(i, e1), P, band);
        }
    }
    __syncthreads();
    SP(3)
    int32_t nc = s_nc;
    if (nc > 2 * CC) {
        // more candidate band pairs than one read can sensibly have (a repeat the -t cap did not
        // catch): the read yields no alignments and is reported (ncand = -2), the launch goes on
        // (the wavefront-per-read tier holds fewer: the read goes to the next tier, where the rule above decides)
        if (tid == 0) {
            ncand_out[item] = ncand_out[item + 1] = NT < SEED_THREADS ? -1 : -2;
            if (NT < SEED_THREADS) {
                nhits_out[item] = n;
                nhits_out[item + 1] = 0;
            }
        }
        return;
    }
    // ---- rank per strand by (score desc, band asc); bands are distinct so ranks are a permutation
    // (the strand is the top bit of the band).  Symmetric all-vs-all: the kept candidates
    // (rank < max_cand) are then grouped by A read, rank order inside a group -- groups are the only
    // candidates that depend on each other (coverage skip), which makes each of them a separate work
    // unit of the wave kernel (k_units).
    // (the ranks overlay the hit buffer, which nobody reads any more: the 2 KB they took kept the 8192-entry variant at 81.5 KB
    // of LDS -- one block per CU instead of two)
    __shared__ int32_t crank_s[LCAP > 0 ? 1 : 2 * CC];
    int32_t *crank = LCAP > 0 ? (int32_t *)lhits : crank_s;
    __shared__ int32_t s_ncs[2];
    constexpr int BSTR = HIT_DBITS;  // strand bit of a band = bit HIT_DBITS - band_shift
    auto strand_of = [&](int32_t c) { return (int32_t)((cband[c] >> (BSTR - bs)) & 1); };
    if (tid < 2) s_ncs[tid] = 0;
    __syncthreads();
    for (int32_t c = tid; c < nc; c += NT) {
        const int32_t st = strand_of(c);
        const int32_t sc = cands[c].score;
        const int64_t bc = cband[c];
        int32_t rank = 0;
        // (no branches, loads of four candidates in flight: the loop is a chain of LDS round trips otherwise)
#pragma unroll 4
        for (int32_t x = 0; x < nc; x++) {
            const int64_t bx = cband[x];
            const int32_t sx = cands[x].score;
            rank += ((int32_t)((bx >> (BSTR - bs)) & 1) == st && (sx > sc || (sx == sc && bx < bc))) ? 1 : 0;
        }
        crank[c] = rank;
        atomicAdd(&s_ncs[st], 1);
    }
    __syncthreads();
    for (int32_t c = tid; c < nc; c += NT) {
        const int32_t rank = crank[c], st = strand_of(c);
        if (rank >= o.max_cand) continue;
        int32_t pos = rank;
        if (o.skip_self == 2) {
            pos = 0;
            const int32_t ac = cands[c].aseq;
#pragma unroll 4
            for (int32_t x = 0; x < nc; x++) {
                const int32_t ax = cands[x].aseq, rx = crank[x];
                pos += (strand_of(x) == st && rx < o.max_cand && (ax < ac || (ax == ac && rx < rank))) ? 1 : 0;
            }
        }
        cand_out[(int64_t)(item + st) * o.max_cand + pos] = cands[c];
    }
    if (tid < 2) ncand_out[item + tid] = s_ncs[tid] < o.max_cand ? s_ncs[tid] : o.max_cand;
    SP(4)
#ifdef DH_SEED_PROF
    if (tid == 0) atomicAdd(&g_seed_prof[7], 1ull);
#endif
}
// Persistent blocks: the grid is sized to the resident capacity of the chip and every block pulls
// items from an atomic queue (no per-item block launch, dynamic balance over ragged read lengths).
template <int LCAP, bool JOIN, int NT = SEED_THREADS, int CC = SEED_CCAP>
__global__ void __launch_bounds__(NT, NT < SEED_THREADS ? 4 : ((LCAP > 0 && LCAP <= 2048) ? 6 : (LCAP == 16384 ? 2 : 4)))
k_seed(DbView B, IndexView ix, JoinView jv, DhOpts o, int32_t read0,
       int32_t nreads, DhCand *__restrict__ cand_out, int32_t *__restrict__ ncand_out,
       int32_t *__restrict__ nhits_out, int32_t *__restrict__ status, uint64_t *__restrict__ gbuf,
       int32_t gcap, const int32_t *__restrict__ read_list, uint32_t *__restrict__ queue)
{
    __shared__ int32_t s_work;
    // (the queue is ONE address: half a million reads of a mapping chunk were half a million returning atomics on it, ~11 ns
    // each whatever the kernel did in between -- 5.7 of the wavefront-per-read tier's 5.7 ms, SQ_WAIT_ANY 88 %.  The small
    // tiers of the segment-fed back end take eight reads per atomic.)
    constexpr int32_t BATCH = (JOIN && LCAP > 0 && LCAP <= 2048) ? 8 : 1;
    for (;;) {
        __syncthreads();  // the previous read is finished by every thread (shared state is reused)
        if (threadIdx.x == 0) s_work = (int32_t)atomicAdd(queue, (uint32_t)BATCH);
        __syncthreads();
        const int32_t work0 = s_work;
        if (work0 >= nreads) break;
#pragma unroll 1
        for (int32_t wi = 0; wi < BATCH; wi++) {
            const int32_t work = work0 + wi;
            if (work >= nreads) break;
            if (wi) __syncthreads();
            seed_item<LCAP, JOIN, NT, CC>(B, ix, jv, o, read0, work, (int32_t)blockIdx.x, cand_out, ncand_out, nhits_out,
                                          status, gbuf, gcap, read_list);
        }
    }
}
#define SEED_INST(C, J)                                                                           \
    template __global__ void k_seed<C, J>(DbView, IndexView, JoinView, DhOpts, int32_t, int32_t,  \
                                          DhCand *, int32_t *, int32_t *, int32_t *, uint64_t *, int32_t, \
                                          const int32_t *, uint32_t *);
SEED_INST(1024, false)
SEED_INST(2048, false)
SEED_INST(4096, false)
SEED_INST(8192, false)
SEED_INST(16384, false)
SEED_INST(0, false)
SEED_INST(2048, true)
template __global__ void k_seed<512, true, 64, 32>(DbView, IndexView, JoinView, DhOpts, int32_t, int32_t, DhCand *, int32_t *, int32_t *,
                                                   int32_t *, uint64_t *, int32_t, const int32_t *, uint32_t *);
SEED_INST(4096, true)
SEED_INST(8192, true)
SEED_INST(16384, true)
SEED_INST(0, true)

// ------------------------------------------------------------------------------------ K4b
// Work units of the symmetric wave launch: the candidates of an item are grouped by A read
// (k_seed), and only candidates of one group depend on each other, so every group is a unit
// (item, first candidate, end candidate) of its own -- the heavy items of an all-vs-all no longer
// serialise dozens of alignments in one wavefront.  Items with more than 64 candidates stay whole
// (the cap of 64 attempted alignments per item must see them in order).
__global__ void __launch_bounds__(256)
k_units(const DhCand *__restrict__ cand, const int32_t *__restrict__ ncand, int32_t item0, int32_t nitems,
        int32_t max_cand, int4 *__restrict__ units, uint32_t *__restrict__ nunits)
{
    const int32_t it = blockIdx.x * blockDim.x + threadIdx.x;
    if (it >= nitems) return;
    const int32_t item = item0 + it;
    const int32_t nc = max(ncand[item], 0);
    if (nc == 0) return;
    const DhCand *cl = cand + (int64_t)item * max_cand;
    if (nc > LANES) {
        units[atomicAdd(nunits, 1u)] = make_int4(it, 0, nc, 0);
        return;
    }
    int32_t c0 = 0;
    while (c0 < nc) {
        int32_t c1 = c0 + 1;
        while (c1 < nc && cl[c1].aseq == cl[c0].aseq) c1++;
        units[atomicAdd(nunits, 1u)] = make_int4(it, c0, c1, 0);
        c0 = c1;
    }
}

// ------------------------------------------------------------------------------------ K5

__device__ __forceinline__ int32_t nbound(int32_t x, int32_t tp_first, int32_t ts)
{
    return x >= tp_first ? (x - tp_first) / ts + 1 : 0;
}

// ballot straight from the compare (llvm.amdgcn.ballot): no bool -> int -> compare round trip
__device__ __forceinline__ unsigned long long wballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// ---- wave64 primitives (verified on gfx950 by scripts/dpp_probe.cpp)
// value of lane-1 / lane+1 (rotation over the whole wave): one DPP mov each, no LDS crossbar
__device__ __forceinline__ int32_t from_lower_lane(int32_t v)
{
    return __builtin_amdgcn_mov_dpp(v, 0x13C, 0xF, 0xF, false);  // wave_ror:1, every lane has a source
}
__device__ __forceinline__ int32_t from_upper_lane(int32_t v)
{
    return __builtin_amdgcn_mov_dpp(v, 0x134, 0xF, 0xF, false);  // wave_rol:1
}
// max over the 64 lanes, result uniform: 4 DPP steps inside each row of 16, then 4 readlanes
__device__ __forceinline__ int32_t wave_max_i32(int32_t v)
{
    v = max(v, __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
    v = max(v, __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
    v = max(v, __builtin_amdgcn_mov_dpp(v, 0x141, 0xF, 0xF, false));  // row_half_mirror
    v = max(v, __builtin_amdgcn_mov_dpp(v, 0x140, 0xF, 0xF, false));  // row_mirror
    const int32_t r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
    const int32_t r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
    return max(max(r0, r1), max(r2, r3));
}
// a wave-uniform global pointer pinned to an SGPR pair (explicit global address space so that
// the loads stay global_load with SGPR base + 32-bit VGPR offset)
typedef const __attribute__((address_space(1))) uint8_t *gptr_t;
struct __attribute__((packed)) PackedU64 {
    uint64_t v;
};
__device__ __forceinline__ gptr_t uniform_ptr(const uint8_t *p)
{
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (gptr_t)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ uint64_t load8g(gptr_t base, uint32_t off)
{
    return ((const __attribute__((address_space(1))) PackedU64 *)(base + off))->v;
}
// extend a run of matches: element i of A' is ap[i * step], 8 bases per compare.
// DB buffers carry 64 bytes of padding on both sides, so the wide loads stay inside them.
// lim = min(an, bn + k) bounds i on diagonal k.  ar/br = ap - an - 7 / bp - bn - 7 (reverse only):
// offsets are unsigned 32-bit values on wave-uniform bases.
template <int STEP>
__device__ __forceinline__ void slide(gptr_t ap, gptr_t ar, int32_t an, gptr_t bp, gptr_t br,
                                      int32_t bn, int32_t lim, int32_t &i, int32_t &j)
{
    for (;;) {
        const int32_t rem = lim - i;
        if (rem <= 0) break;
        int32_t m;
        if (STEP > 0) {
            const uint64_t x = load8g(ap, (uint32_t)i) ^ load8g(bp, (uint32_t)j);
            m = x ? ((__ffsll((long long)x) - 1) >> 3) : 8;
        } else {
            const uint64_t x = load8g(ar, (uint32_t)(an - i)) ^ load8g(br, (uint32_t)(bn - j));
            m = x ? (__clzll((long long)x) >> 3) : 8;
        }
        m = min(m, rem);
        i += m;
        j += m;
        if (m < 8) break;
    }
}

// The same on 2-bit packed sequences: 32 bases per 8-byte load.  Forward: element i of A' is
// base (4 * qa + ra + i) and pa points at byte qa; reverse: element i is base (4 * qa + ra - i),
// par points at byte qa - 7 - na4 and the window is the 32 bases ENDING at that base (na4 keeps
// the unsigned load offsets non-negative).  The window of a load starts at an arbitrary base of
// its first byte, so only 32 - max(phase) bases of a compare are valid.
template <int STEP>
__device__ __forceinline__ void slide_pk(gptr_t pa, int32_t ra, int32_t na4, gptr_t pb, int32_t rb,
                                         int32_t nb4, int32_t lim, int32_t &i, int32_t &j)
{
    for (;;) {
        const int32_t rem = lim - i;
        if (rem <= 0) break;
        int32_t m, valid;
        if (STEP > 0) {
            const int32_t ta = ra + i, tb = rb + j;
            const int32_t sa = (ta & 3) << 1, sb = (tb & 3) << 1;
            const uint64_t x = (load8g(pa, (uint32_t)ta >> 2) >> sa) ^ (load8g(pb, (uint32_t)tb >> 2) >> sb);
            valid = 32 - (max(sa, sb) >> 1);
            m = x ? ((__ffsll((long long)x) - 1) >> 1) : 32;
        } else {
            const int32_t ta = ra - i, tb = rb - j;
            const int32_t sa = (3 - (ta & 3)) << 1, sb = (3 - (tb & 3)) << 1;
            const uint64_t x = (load8g(pa, (uint32_t)((ta >> 2) + na4)) << sa) ^
                               (load8g(pb, (uint32_t)((tb >> 2) + nb4)) << sb);
            valid = 32 - (max(sa, sb) >> 1);
            m = x ? (__clzll((long long)x) >> 1) : 32;
        }
        m = min(min(m, valid), rem);
        i += m;
        j += m;
        if (m < valid) break;
    }
}

struct ExtResult {
    int32_t i, j, d, head, nb, headb, nbb;
};

// One-directional greedy extension by one wavefront; lane (k & 63) owns diagonal k.
// All lanes execute every cross-lane operation.  SYM additionally records the crossings of the
// B-offsets tpb_first + m*ts (value = i when j first reaches the boundary): the same path then
// also yields the trace of the transposed record (symmetric all-vs-all, each pair aligned once).
// PK: ap_ / bp_ are the 2-bit packed arrays and ag / bg the absolute base index of element 0
// (slide_pk); otherwise ap_ / bp_ point at element 0 of the byte arrays.
template <int STEP, bool SYM, bool PK>
__device__ ExtResult ext_wave(const uint8_t *ap_, int64_t ag, int32_t an, const uint8_t *bp_, int64_t bg,
                              int32_t bn, int32_t tp_first,
                              int32_t tpb_first, const DhOpts &o, DhNode *__restrict__ pool,
                              int32_t poolcap, int32_t &pool_n, unsigned long long &cells,
                              int32_t &err)
{
    const int lane = threadIdx.x & (LANES - 1);
    const int32_t ts = o.tspace, pen = o.pen, xdrop = o.xdrop;
    an = __builtin_amdgcn_readfirstlane(an);
    bn = __builtin_amdgcn_readfirstlane(bn);
    // byte arrays: forward base ap / bp, reverse base ar / br (slide); packed: one base per
    // direction in ap / bp plus the phases ra / rb and the offset biases na4 / nb4 (slide_pk)
    const int32_t ra = PK ? __builtin_amdgcn_readfirstlane((int32_t)(ag & 3)) : 0;
    const int32_t rb = PK ? __builtin_amdgcn_readfirstlane((int32_t)(bg & 3)) : 0;
    const int32_t na4 = (PK && STEP < 0) ? (an >> 2) + 2 : 0, nb4 = (PK && STEP < 0) ? (bn >> 2) + 2 : 0;
    const gptr_t ap = uniform_ptr(PK ? ap_ + (ag >> 2) - (STEP < 0 ? 7 + na4 : 0) : ap_);
    const gptr_t bp = uniform_ptr(PK ? bp_ + (bg >> 2) - (STEP < 0 ? 7 + nb4 : 0) : bp_);
    const gptr_t ar = uniform_ptr(PK ? ap_ : ap_ - an - 7), br = uniform_ptr(PK ? bp_ : bp_ - bn - 7);
    tp_first = __builtin_amdgcn_readfirstlane(tp_first);
    tpb_first = __builtin_amdgcn_readfirstlane(tpb_first);
    // per-lane state of diagonal k: R = furthest i (DEAD when dead), H = head of its trace chain,
    // NB = the first trace boundary above R (tp_first + #boundaries * ts, carried along so that the
    // loop needs neither a division nor a multiplication); HB / NBB the same for the B-offset
    // boundaries (SYM only)
    constexpr int32_t DEAD = -(1 << 30);
    int32_t R = DEAD, H = -1, NB = tp_first, HB = -1, NBB = tpb_first;
    int32_t L = 0;

    // d = 0: the seed diagonal, slid by lane 0
    int32_t i0 = 0, h0 = -1, nb0 = 0, hb0 = -1, nbb0 = 0;
    if (lane == 0) {
        int32_t j0 = 0;
        if (PK)
            slide_pk<STEP>(ap, ra, na4, bp, rb, nb4, min(an, bn), i0, j0);
        else
            slide<STEP>(ap, ar, an, bp, br, bn, min(an, bn), i0, j0);
        int32_t cnt = 0;
        for (int32_t nextb = tp_first; nextb <= i0; nextb += ts) {
            const int32_t idx = pool_n + cnt;
            if (idx < poolcap) {
                pool[idx].parent = h0;
                pool[idx].d = 0;
                pool[idx].j = nextb;
            }
            h0 = idx;
            nb0++;
            cnt++;
        }
        if (SYM)
            for (int32_t nextb = tpb_first; nextb <= i0; nextb += ts) {
                const int32_t idx = pool_n + cnt;
                if (idx < poolcap) {
                    pool[idx].parent = hb0;
                    pool[idx].d = 0;
                    pool[idx].j = nextb;
                }
                hb0 = idx;
                nbb0++;
                cnt++;
            }
        R = i0;
        H = h0;
        NB = tp_first + nb0 * ts;
        HB = hb0;
        NBB = tpb_first + nbb0 * ts;
    }
    // wave-uniform values are pinned to SGPRs (readfirstlane) so that the window arithmetic,
    // mask rotations and find-first-set below run on the scalar unit
    i0 = __builtin_amdgcn_readfirstlane(i0);
    h0 = __builtin_amdgcn_readfirstlane(h0);
    nb0 = __builtin_amdgcn_readfirstlane(nb0);
    hb0 = __builtin_amdgcn_readfirstlane(hb0);
    nbb0 = __builtin_amdgcn_readfirstlane(nbb0);
    pool_n = __builtin_amdgcn_readfirstlane(pool_n + nb0 + nbb0);
    int32_t best_score = 2 * i0, best_i = i0, best_k = 0, best_d = 0, best_head = h0;
    int32_t best_nb = tp_first + nb0 * ts, best_headb = hb0, best_nbb = tpb_first + nbb0 * ts;
    unsigned long long ncell = 1;

    for (int32_t d = 1; d <= o.dmax; d++) {
        const int32_t nL = L - 1;
        const int32_t kidx = (lane - nL) & (LANES - 1);
        const int32_t k = nL + kidx;
        const int32_t Rm = from_lower_lane(R), Hm = from_lower_lane(H), Nm = from_lower_lane(NB);
        const int32_t Rp = from_upper_lane(R), Hp = from_upper_lane(H), Np = from_upper_lane(NB);
        int32_t HBm = -1, NBm = tpb_first, HBp = -1, NBp = tpb_first;
        if (SYM) {
            HBm = from_lower_lane(HB);
            NBm = from_lower_lane(NBB);
            HBp = from_upper_lane(HB);
            NBp = from_upper_lane(NBB);
        }
        // substitution on k, deletion from k-1 (consumes A), insertion from k+1 (consumes B); ties
        // prefer sub, then del.  Dead diagonals carry R = DEAD (very negative), so a candidate from
        // a dead source never beats ni = -1; sources are valid points, hence j >= 0 holds for all
        // three moves and only i <= an, j <= bn (i <= lim) has to be checked.  Lanes outside the
        // window see dead sources only (width <= 62) and stay dead.
        int32_t ni = -1, hd = -1, nbp = tp_first, hb = -1, nbbp = tpb_first;
        const int32_t lim = min(an, bn + k);  // i <= an and i - k <= bn
        {
            const int32_t cs = R + 1, cd = Rm + 1, ci = Rp;
            if (cs <= lim && cs > ni) {
                ni = cs;
                hd = H;
                nbp = NB;
                hb = HB;
                nbbp = NBB;
            }
            if (cd <= lim && cd > ni) {
                ni = cd;
                hd = Hm;
                nbp = Nm;
                hb = HBm;
                nbbp = NBm;
            }
            if (ci <= lim && ci > ni) {
                ni = ci;
                hd = Hp;
                nbp = Np;
                hb = HBp;
                nbbp = NBp;
            }
        }
        bool alive = ni >= 0;
        int32_t j = ni - k;
        if (alive) {
            if (PK)
                slide_pk<STEP>(ap, ra, na4, bp, rb, nb4, lim, ni, j);
            else
                slide<STEP>(ap, ar, an, bp, br, bn, lim, ni, j);
        }
        const unsigned long long amask = wballot(alive);
        if (amask == 0ull) break;
        ncell += __popcll(amask);
        // trace nodes for the boundaries crossed in (prev_i, ni]: nbp is the first one above prev_i
        int32_t nextb = nbp;
        bool cross = alive && ni >= nextb;
        for (;;) {
            const unsigned long long m = wballot(cross);
            if (m == 0ull) break;
            if (cross) {
                const int32_t idx = pool_n + __popcll(m & ((1ull << lane) - 1ull));
                if (idx < poolcap) {
                    pool[idx].parent = hd;
                    pool[idx].d = d;
                    pool[idx].j = nextb - k;
                }
                hd = idx;
                nextb += ts;
                cross = ni >= nextb;
            }
            pool_n = __builtin_amdgcn_readfirstlane(pool_n + __popcll(m));
        }
        int32_t nextbb_out = nbbp;
        if (SYM) {
            int32_t nextbb = nbbp;
            bool crossb = alive && j >= nextbb;
            for (;;) {
                const unsigned long long m = wballot(crossb);
                if (m == 0ull) break;
                if (crossb) {
                    const int32_t idx = pool_n + __popcll(m & ((1ull << lane) - 1ull));
                    if (idx < poolcap) {
                        pool[idx].parent = hb;
                        pool[idx].d = d;
                        pool[idx].j = nextbb + k;
                    }
                    hb = idx;
                    nextbb += ts;
                    crossb = j >= nextbb;
                }
                pool_n = __builtin_amdgcn_readfirstlane(pool_n + __popcll(m));
            }
            nextbb_out = nextbb;
        }
        if (pool_n > poolcap) {
            err |= DH_ST_POOL_OVERFLOW;
            break;
        }
        R = alive ? ni : DEAD;
        H = hd;
        NB = nextb;
        HB = hb;
        NBB = SYM ? nextbb_out : NBB;
        // best of this step: highest score, then lowest diagonal (ballot of the max holders,
        // rotated so that bit x is diagonal nL + x)
        const int32_t sc = alive ? 2 * ni - k - pen * d : INT32_MIN;
        const int32_t step_best = wave_max_i32(sc);
        const int rot = nL & (LANES - 1);
        if (step_best > best_score) {
            const unsigned long long hm = wballot(alive && sc == step_best);
            const unsigned long long hr = rot ? ((hm >> rot) | (hm << (LANES - rot))) : hm;
            const int32_t step_kidx = __ffsll((long long)hr) - 1;
            const int src = __builtin_amdgcn_readfirstlane((nL + step_kidx) & (LANES - 1));
            best_score = step_best;
            best_k = nL + step_kidx;
            best_i = __builtin_amdgcn_readlane(R, src);
            best_head = __builtin_amdgcn_readlane(H, src);
            best_nb = __builtin_amdgcn_readlane(NB, src);
            if (SYM) {
                best_headb = __builtin_amdgcn_readlane(HB, src);
                best_nbb = __builtin_amdgcn_readlane(NBB, src);
            }
            best_d = d;
        }
        // trim to xdrop of the best
        if (alive && sc < best_score - xdrop) {
            alive = false;
            R = DEAD;
        }
        unsigned long long lm = wballot(alive);
        if (lm == 0ull) break;
        unsigned long long rm = rot ? ((lm >> rot) | (lm << (LANES - rot))) : lm;
        int32_t l2 = __builtin_amdgcn_readfirstlane(nL + (__ffsll((long long)rm) - 1));
        int32_t u2 = __builtin_amdgcn_readfirstlane(nL + (63 - __clzll((long long)rm)));
        while (u2 - l2 + 1 > o.width) {
            // drop the lower-scoring edge (same d: compare 2R - k), ties drop the low edge
            const int32_t val = 2 * R - k;
            const int32_t sl = __builtin_amdgcn_readlane(val, l2 & (LANES - 1));
            const int32_t su = __builtin_amdgcn_readlane(val, u2 & (LANES - 1));
            const int32_t kill = sl <= su ? l2 : u2;
            if (k == kill) {
                alive = false;
                R = DEAD;
            }
            lm = wballot(alive);
            rm = rot ? ((lm >> rot) | (lm << (LANES - rot))) : lm;
            l2 = __builtin_amdgcn_readfirstlane(nL + (__ffsll((long long)rm) - 1));
            u2 = __builtin_amdgcn_readfirstlane(nL + (63 - __clzll((long long)rm)));
        }
        L = l2;
    }
    cells += ncell;
    ExtResult res;
    res.i = best_i;
    res.j = best_i - best_k;
    res.d = best_d;
    res.head = best_head;
    res.nb = (best_nb - tp_first) / ts;
    res.headb = best_headb;
    res.nbb = SYM ? (best_nbb - tpb_first) / ts : 0;
    return res;
}

// walk a trace chain (serial, one lane); writes cd/cj[m], returns diagonal excursion
__device__ void walk_chain(const DhNode *__restrict__ pool, int32_t head, int32_t nb,
                           int32_t tp_first, int32_t ts, int32_t best_k, int32_t *cd, int32_t *cj,
                           int32_t &lo, int32_t &hi)
{
    lo = best_k < 0 ? best_k : 0;
    hi = best_k > 0 ? best_k : 0;
    int32_t h = head;
#ifdef DH_SKIP_WALK
    return;
#endif
    for (int32_t m = nb - 1; m >= 0 && h >= 0; m--) {
        const DhNode nd = pool[h];
        cd[m] = nd.d;
        cj[m] = nd.j;
        const int32_t kk = (tp_first + m * ts) - nd.j;
        lo = kk < lo ? kk : lo;
        hi = kk > hi ? kk : hi;
        h = nd.parent;
    }
}

// the pairs (delta diffs, delta other) of a trace between consecutive grid boundaries, written by
// the whole wavefront.  grid = the coordinate the trace spacing refers to (boundaries at
// grid = res mod ts), other = the opposite sequence; gs/os = seed on the two axes; rd/ro, fd/fo =
// boundary records of the reverse / forward extension (diffs, offset on `other`).  `reverse`
// writes the pairs back to front (transposed record of a complemented alignment).
__device__ int32_t emit_trace(int lane, int stride, int32_t ts, int32_t res, int32_t gs, int32_t os,
                              int32_t gbeg, int32_t gend, int32_t obeg, int32_t oend, int32_t rdv,
                              int32_t fdv, int32_t rev_first, int32_t nr, const int32_t *rd,
                              const int32_t *ro, int32_t fwd_first, int32_t nf, const int32_t *fd,
                              const int32_t *fo, bool reverse, uint16_t *__restrict__ tr)
{
    const int32_t nrv = nr - ((nr > 0 && rev_first + (nr - 1) * ts == gs - gbeg) ? 1 : 0);
    const int32_t nfv = nf - ((nf > 0 && fwd_first + (nf - 1) * ts == gend - gs) ? 1 : 0);
    int32_t gm = (gs - res) % ts;
    gm = gm < 0 ? gm + ts : gm;
    const int32_t seedb = (gm == 0 && gs > gbeg && gs < gend) ? 1 : 0;
    const int32_t npairs = nrv + seedb + nfv + 1;
    for (int32_t e = lane; e < npairs; e += stride) {
        int32_t po[2], pD[2];
#pragma unroll
        for (int w = 0; w < 2; w++) {
            const int32_t idx = e + w;
            if (idx == 0) {
                po[w] = obeg;
                pD[w] = -rdv;
            } else if (idx <= nrv) {
                const int32_t m = nrv - idx;
                po[w] = os - ro[m];
                pD[w] = -rd[m];
            } else if (idx <= nrv + seedb) {
                po[w] = os;
                pD[w] = 0;
            } else if (idx <= nrv + seedb + nfv) {
                const int32_t m = idx - 1 - nrv - seedb;
                po[w] = os + fo[m];
                pD[w] = fd[m];
            } else {
                po[w] = oend;
                pD[w] = fdv;
            }
        }
        const int32_t pos = reverse ? npairs - 1 - e : e;
        tr[2 * pos] = (uint16_t)(pD[1] - pD[0]);
        tr[2 * pos + 1] = (uint16_t)(po[1] - po[0]);
    }
    return npairs;
}

// SYM (all-vs-all inside one DB, skip_self == 2): each unordered pair has candidates in one item only; every
// accepted alignment emits the record (a, b) into the slots of item (a, strand) and the transposed
// record (b, a) into the slots of item (b, strand); slots are claimed with atomics because any
// wavefront may add records to any item (the final LAsort makes the output order unique).
// PK: the wave slides over the 2-bit packed copies apk (A), bpk / brcpk (B, B reverse-complemented)
template <bool SYM, bool PK>
__global__ void __launch_bounds__(LANES)
k_wave(DbView A, DbView B, const uint8_t *__restrict__ brc, const uint8_t *__restrict__ apk,
       const uint8_t *__restrict__ bpk, const uint8_t *__restrict__ brcpk, DhOpts o, int32_t item0,
       int32_t nitems, const DhCand *__restrict__ cand, const int32_t *__restrict__ ncand,
       WaveScratch ws, DhLa *__restrict__ out_la, uint16_t *__restrict__ out_trace,
       int32_t trmax, int32_t *__restrict__ out_nla, int32_t *__restrict__ out_ntr,
       unsigned long long *__restrict__ counters,
       int32_t *__restrict__ status)
{
    const int lane = threadIdx.x;
    DhNode *pool = ws.pool + (int64_t)blockIdx.x * ws.poolcap;
    int32_t *cdj = ws.cdj + (int64_t)blockIdx.x * 8 * ws.nbmax;
    int32_t *fd = cdj, *fj = cdj + ws.nbmax, *rd = cdj + 2 * ws.nbmax, *rj = cdj + 3 * ws.nbmax;
    int32_t *fdb = cdj + 4 * ws.nbmax, *fib = cdj + 5 * ws.nbmax, *rdb = cdj + 6 * ws.nbmax,
            *rib = cdj + 7 * ws.nbmax;
    const int32_t ts = o.tspace;
    unsigned long long cells = 0, naln = 0;
    int32_t err = 0;

    for (;;) {
        int32_t it = 0;
        if (lane == 0) it = (int32_t)atomicAdd(ws.queue, 1u);
        it = __builtin_amdgcn_readfirstlane(it);
        if (it >= (ws.units ? (int32_t)*ws.nunits : nitems)) break;
        // work unit: a whole item, or (symmetric mode) one group of candidates of an item
        int32_t c0 = 0, c1 = INT32_MAX, ui = it;
        if (ws.units) {
            const int4 u = ws.units[it];
            ui = u.x;
            c0 = u.y;
            c1 = u.z;
        }
        const int32_t item = item0 + ui;
        const int32_t r = item >> 1, strand = item & 1;
        const int32_t nc = min(max(ncand[item], 0), c1);
        const int64_t bo = B.off[r];
        const int32_t blen = (int32_t)(B.off[r + 1] - bo);
        const uint8_t *b = (strand ? brc : B.bases) + bo;
        // regions already aligned for this (read, strand): kept in registers of lanes 0..nd-1
        int32_t g_aseq = -1, g_ab = 0, g_ae = 0, g_bb = 0, g_be = 0, g_lo = 0, g_hi = 0;
        int32_t nd = 0, nacc = 0, ntr = 0;
        for (int32_t c = c0; c < nc && (SYM || nacc < o.max_la) && nd < LANES; c++) {
            const DhCand cd = cand[(int64_t)item * o.max_cand + c];
            const int32_t sd = cd.apos - cd.bpos;
            const bool cov = lane < nd && g_aseq == cd.aseq && cd.apos >= g_ab && cd.apos < g_ae &&
                             cd.bpos >= g_bb && cd.bpos < g_be && sd >= g_lo - 64 && sd <= g_hi + 64;
            if (wballot(cov) != 0ull) continue;
            const int64_t ao = A.off[cd.aseq];
            const int32_t alen = (int32_t)(A.off[cd.aseq + 1] - ao);
            const uint8_t *a = A.bases + ao;
            const int32_t as = cd.apos, bs = cd.bpos;
            const int32_t fwd_first = ts - (as % ts);
            const int32_t rev_first = (as % ts) ? (as % ts) : ts;
            // B grid of the transposed record: forward strand of the read behind B
            const int32_t resb = strand ? blen % ts : 0;
            int32_t bm = (bs - resb) % ts;
            bm = bm < 0 ? bm + ts : bm;
            const int32_t fwdb_first = ts - bm, revb_first = bm ? bm : ts;
            int32_t pool_n = 0;
            const uint8_t *bsrc = PK ? (strand ? brcpk : bpk) : b;
            const ExtResult fw = ext_wave<1, SYM, PK>(PK ? apk : a + as, ao + as, alen - as,
                                                      PK ? bsrc : b + bs, bo + bs, blen - bs, fwd_first,
                                                      fwdb_first, o, pool, ws.poolcap, pool_n, cells, err);
            const ExtResult rv = ext_wave<-1, SYM, PK>(PK ? apk : a + as - 1, ao + as - 1, as,
                                                       PK ? bsrc : b + bs - 1, bo + bs - 1, bs, rev_first,
                                                       revb_first, o, pool, ws.poolcap, pool_n, cells, err);
            naln++;
            if (err || fw.nb > ws.nbmax || rv.nb > ws.nbmax || fw.nbb > ws.nbmax || rv.nbb > ws.nbmax) {
                err |= DH_ST_POOL_OVERFLOW;
                break;
            }
            // chains: lanes 0..3 walk the forward / reverse chains of the two boundary families
            int32_t flo = 0, fhi = 0, rlo = 0, rhi = 0;
            if (lane == 0)
                walk_chain(pool, fw.head, fw.nb, fwd_first, ts, fw.i - fw.j, fd, fj, flo, fhi);
            if (lane == 1)
                walk_chain(pool, rv.head, rv.nb, rev_first, ts, rv.i - rv.j, rd, rj, rlo, rhi);
            if (SYM) {
                int32_t x0, x1;
                if (lane == 2) walk_chain(pool, fw.headb, fw.nbb, fwdb_first, ts, 0, fdb, fib, x0, x1);
                if (lane == 3) walk_chain(pool, rv.headb, rv.nbb, revb_first, ts, 0, rdb, rib, x0, x1);
            }
            __threadfence_block();
            flo = __shfl(flo, 0, LANES);
            fhi = __shfl(fhi, 0, LANES);
            rlo = __shfl(rlo, 1, LANES);
            rhi = __shfl(rhi, 1, LANES);
            const int32_t abpos = as - rv.i, bbpos = bs - rv.j, aepos = as + fw.i, bepos = bs + fw.j;
            const int32_t diffs = fw.d + rv.d;
            int32_t lo = sd + flo, hi = sd + fhi;
            lo = (sd - rhi) < lo ? (sd - rhi) : lo;
            hi = (sd - rlo) > hi ? (sd - rlo) : hi;
            if (lane == nd) {
                g_aseq = cd.aseq;
                g_ab = abpos;
                g_ae = aepos;
                g_bb = bbpos;
                g_be = bepos;
                g_lo = lo;
                g_hi = hi;
            }
            nd++;
            const int64_t al = aepos - abpos, bl = bepos - bbpos;
            const bool accept = al >= o.min_len &&
                                (int64_t)2 * diffs * 1000000ll <= (int64_t)o.max_err_ppm * (al + bl);
            if (!accept) continue;
            // ---- the record (a, b): trace on the grid of A.  SYM: it goes to the slots of item
            // (a, strand) and the transposed record to those of (b, strand), so that the output
            // is grouped by A read.
            const int32_t item_a = SYM ? 2 * cd.aseq + strand : item;
            int32_t s1 = nacc;
            if (SYM) {
                if (lane == 0) s1 = atomicAdd(&out_nla[item_a], 1);
                s1 = __shfl(s1, 0, LANES);
                if (s1 >= o.max_la) {  // more overlaps than slots: drop the pair, report both items
                    if (lane == 0) {
                        atomicSub(&out_nla[item_a], 1);
                        ws.item_ovf[item_a] = 1;
                        ws.item_ovf[item] = 1;
                    }
                    continue;
                }
            }
            const int64_t slot = (int64_t)item_a * o.max_la + s1;
            const int32_t npairs = emit_trace(lane, LANES, ts, 0, as, bs, abpos, aepos, bbpos, bepos, rv.d, fw.d,
                                              rev_first, rv.nb, rd, rj, fwd_first, fw.nb, fd, fj, false,
                                              out_trace + slot * trmax);
            if (lane == 0) {
                DhLa la;
                la.tlen = 2 * npairs;
                la.diffs = diffs;
                la.abpos = abpos;
                la.bbpos = bbpos;
                la.aepos = aepos;
                la.bepos = bepos;
                la.flags = strand ? 1u : 0u;
                la.aread = cd.aseq;
                la.bread = r;
                la.pad = 0;
                la.toff = 0;
                out_la[slot] = la;
                if (SYM) atomicAdd(&out_ntr[item_a], 2 * npairs);
            }
            nacc++;
            ntr += 2 * npairs;
            if (SYM) {
                // ---- the transposed record (b, a): same path, trace on the grid of B
                const int32_t item2 = item;
                int32_t s2 = 0;
                if (lane == 0) s2 = atomicAdd(&out_nla[item2], 1);
                s2 = __shfl(s2, 0, LANES);
                if (s2 >= o.max_la) {
                    if (lane == 0) {
                        atomicSub(&out_nla[item2], 1);
                        ws.item_ovf[item2] = 1;
                        ws.item_ovf[item_a] = 1;
                    }
                    continue;
                }
                const int64_t slot2 = (int64_t)item2 * o.max_la + s2;
                const int32_t np2 = emit_trace(lane, LANES, ts, resb, bs, as, bbpos, bepos, abpos, aepos, rv.d, fw.d,
                                               revb_first, rv.nbb, rdb, rib, fwdb_first, fw.nbb, fdb, fib,
                                               strand != 0, out_trace + slot2 * trmax);
                if (lane == 0) {
                    DhLa la;
                    la.tlen = 2 * np2;
                    la.diffs = diffs;
                    la.abpos = strand ? blen - bepos : bbpos;
                    la.aepos = strand ? blen - bbpos : bepos;
                    la.bbpos = strand ? alen - aepos : abpos;
                    la.bepos = strand ? alen - abpos : aepos;
                    la.flags = strand ? 1u : 0u;
                    la.aread = r;
                    la.bread = cd.aseq;
                    la.pad = 0;
                    la.toff = 0;
                    out_la[slot2] = la;
                    atomicAdd(&out_ntr[item2], 2 * np2);
                }
            }
        }
        if (!SYM && lane == 0) {
            out_nla[item] = nacc;
            out_ntr[item] = ntr;
        }
        if (err) break;
    }
    if (lane == 0) {
        atomicAdd(&counters[0], cells);
        atomicAdd(&counters[1], naln);
        if (err) atomicOr(status, err);
    }
}

// ------------------------------------------------------------------------------------ K5b
//
// k_wave2: two alignments per wavefront.  On average only ~19 of the 64 diagonals of a wavefront
// are alive and the kernel is bound by VALU issue, so with a wave width of at most 30 diagonals
// (DhOpts.width <= 30) each 32-lane half runs its own alignment: lane (k & 31) of a half owns
// diagonal k.  The halves are independent state machines sharing one instruction stream -- a half
// that finishes an extension runs its bookkeeping (next candidate, chains, trace, records, next
// item) while the other half keeps stepping -- and everything that is wave-uniform in k_wave is
// half-uniform here (kept per lane, broadcast with readlane pairs / ds_bpermute, ballots split
// into their 32-bit halves).  Reverse extensions are forward extensions over the
// reverse-complemented copies, so both halves always run the same slide code.
// The arithmetic is that of ext_wave / k_wave, bit for bit.

enum { W2_FETCH = 0, W2_CAND = 1, W2_EXT = 2, W2_EXT_END = 3, W2_DONE = 4, W2_POST_CHAIN = 5, W2_POST_REC1 = 6,
       W2_POST_REC2 = 7 };

// G lanes per alignment (32: two per wavefront, 16: four); hb = first lane of my group
template <int G>
__device__ __forceinline__ uint32_t hballot(bool p, int hb)
{
    const uint64_t m = wballot(p);
    if (G == 32) return hb ? (uint32_t)(m >> 32) : (uint32_t)m;
    return (uint32_t)(m >> hb) & 0xFFFFu;
}
// value of lane `l` (constant) of my group
template <int G, int L>
__device__ __forceinline__ int32_t hlane(int32_t v, int hb)
{
    if (G == 32) {
        const int32_t a = __builtin_amdgcn_readlane(v, L), b = __builtin_amdgcn_readlane(v, 32 + L);
        return hb ? b : a;
    }
    // four groups: one trip through the LDS crossbar (the group's lanes are all active wherever this
    // is used) instead of four readlanes and three selects
    return __builtin_amdgcn_ds_bpermute((hb | L) << 2, v);
}
// value of lane l (group-uniform, 0..G-1) of my group; every lane of the group must be active
__device__ __forceinline__ int32_t hread(int32_t v, int32_t l, int hb)
{
    return __builtin_amdgcn_ds_bpermute((hb | l) << 2, v);
}
// the same for the serial edge trimming.  Two groups: two readlanes per group on scalar indices
// (no LDS crossbar round trip on the critical path); four groups: the crossbar after all (eight
// readlanes plus selects cost more issue slots than the round trip costs latency)
template <int G>
__device__ __forceinline__ int32_t hread_fast(int32_t v, int32_t l, int hb)
{
    if (G == 32) {
        const int32_t l0 = __builtin_amdgcn_readlane(l, 0) & 31, l1 = __builtin_amdgcn_readlane(l, 32) & 31;
        const int32_t a = __builtin_amdgcn_readlane(v, l0), b = __builtin_amdgcn_readlane(v, 32 + l1);
        return hb ? b : a;
    }
    return __builtin_amdgcn_ds_bpermute((hb | (l & 15)) << 2, v);
}
// max over the G lanes of my group
template <int G>
__device__ __forceinline__ int32_t hmax_i32(int32_t v, int hb)
{
    v = max(v, __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true));
    v = max(v, __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true));
    v = max(v, __builtin_amdgcn_mov_dpp(v, 0x141, 0xF, 0xF, true));
    v = max(v, __builtin_amdgcn_mov_dpp(v, 0x140, 0xF, 0xF, true));
    if (G == 16) return v;  // a DPP row is a group: every lane holds its row's maximum
    const int32_t r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
    const int32_t r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
    return hb ? max(r2, r3) : max(r0, r1);
}
// rotate the G-bit group mask right by r (0 <= r < G)
template <int G>
__device__ __forceinline__ uint32_t hrotr(uint32_t x, uint32_t r)
{
    if (G == 32) return __builtin_rotateright32(x, r);
    return ((x | (x << 16)) >> r) & 0xFFFFu;
}
// forward slide with per-lane base pointers (bytes: p + i; packed: base index 4 * q + r + i,
// p points at byte q)
template <bool PK>
__device__ __forceinline__ void slide2(const uint8_t *pa, int32_t ra, const uint8_t *pb, int32_t rb,
                                       int32_t lim, int32_t &i, int32_t &j)
{
    for (;;) {
        const int32_t rem = lim - i;
        if (rem <= 0) break;
        int32_t m, valid;
        if (PK) {
            const int32_t ta = ra + i, tb = rb + j;
            const int32_t sa = (ta & 3) << 1, sb = (tb & 3) << 1;
            const uint64_t x = (load8(pa + ((uint32_t)ta >> 2)) >> sa) ^ (load8(pb + ((uint32_t)tb >> 2)) >> sb);
            valid = 32 - (max(sa, sb) >> 1);
            m = x ? ((__ffsll((long long)x) - 1) >> 1) : 32;
        } else {
            const uint64_t x = load8(pa + (uint32_t)i) ^ load8(pb + (uint32_t)j);
            valid = 8;
            m = x ? ((__ffsll((long long)x) - 1) >> 3) : 8;
        }
        m = min(min(m, valid), rem);
        i += m;
        j += m;
        if (m < valid) break;
    }
}

struct W2Cold {
    int32_t item, r, strand, nc, c, blen, nd, nacc, ntr;
    int32_t c_aseq, as, bs, alen, sd;
    int32_t fwd_first, rev_first, fwdb_first, revb_first, resb;
    int64_t bo, ao;
    int32_t fw_i, fw_j, fw_d, fw_head, fw_nb, fw_headb, fw_nbb;
    int32_t rv_i, rv_j, rv_d, rv_head, rv_nb, rv_headb, rv_nbb;
    int32_t abpos, bbpos, aepos, bepos, diffs;
    unsigned long long cells, naln;
};

template <bool SYM, bool PK, int G>
__global__ void __launch_bounds__(LANES, G == 16 ? 5 : 6)  // G = 32: 80 VGPRs, measured best of 4 / 5 / 6 / 8 waves per SIMD
k_wave2(DbView A, DbView B, const uint8_t *__restrict__ arc, const uint8_t *__restrict__ brc,
        const uint8_t *__restrict__ apk, const uint8_t *__restrict__ arcpk,
        const uint8_t *__restrict__ bpk, const uint8_t *__restrict__ brcpk, DhOpts o, int32_t item0,
        int32_t nitems, const DhCand *__restrict__ cand, const int32_t *__restrict__ ncand,
        WaveScratch ws, DhLa *__restrict__ out_la, uint16_t *__restrict__ out_trace,
        int32_t trmax, int32_t *__restrict__ out_nla, int32_t *__restrict__ out_ntr,
        unsigned long long *__restrict__ counters, int32_t *__restrict__ status)
{
    constexpr int NG = LANES / G;  // alignments per wavefront
    const int lane = threadIdx.x, hl = lane & (G - 1), hb = lane & (LANES - G), grp = lane / G;
    const int64_t slot = (int64_t)blockIdx.x * NG + grp;
    DhNode *pool = ws.pool + slot * ws.poolcap;
    // every lane pushes its trace nodes into its own stretch of the slot's pool (node = lbase + pn):
    // no ballot / prefix count per boundary crossing, and a level on which nothing crosses costs
    // one compare per family
    const int32_t ts = o.tspace, pen = o.pen, xdrop = o.xdrop, lanecap = ws.poolcap / G, lbase = hl * lanecap;
    const int32_t addr_lo = (hb | ((hl - 1) & (G - 1))) << 2, addr_hi = (hb | ((hl + 1) & (G - 1))) << 2;
    constexpr int32_t DEAD = -(1 << 30);

    int32_t st = W2_FETCH, err = 0;
    // ---- cold state of the half (item, candidate, results, counters): half-uniform values that
    // only the bookkeeping touches live in LDS (every lane of the half writes the same value), so
    // that the stepping loop keeps its registers -- two alignments per wavefront at 8 waves/SIMD
    __shared__ W2Cold cold_[NG];
    __shared__ int32_t greg_[NG][7 * NG][G];  // regions already aligned: region x in lane x % G, set x / G
    W2Cold &cs = cold_[grp];
    int32_t(*gr)[G] = greg_[grp];
    cs.cells = 0;
    cs.naln = 0;
    // ---- the running extension (hot)
    int32_t dir = 0, ra = 0, rb = 0, an = 0, bn = 0, tp_first = 0, tpb_first = 0;
    const uint8_t *pa = nullptr, *pb = nullptr;
    int32_t R = DEAD, H = -1, NB = 0, HB = -1, NBB = 0;  // per lane
    int32_t L = 0, d = 0, pn = 0;  // pn: nodes of this lane (per candidate, both extensions)
    int32_t best_score = 0, best_i = 0, best_k = 0, best_d = 0, best_head = -1, best_nb = 0, best_headb = -1,
            best_nbb = 0;
    uint32_t ncell = 0;

    // start the extension `dir` (0 forward, 1 reverse) of the current candidate
    auto ext_begin = [&](int32_t nd_) {
        dir = nd_;
        // reverse = forward over the reverse complements: base (len - pos) of the rc copy
        const int32_t as = cs.as, bs = cs.bs, alen = cs.alen, blen = cs.blen;
        const int64_t ga = cs.ao + (dir ? alen - as : as), gb = cs.bo + (dir ? blen - bs : bs);
        an = dir ? as : alen - as;
        bn = dir ? bs : blen - bs;
        const bool brc_side = (cs.strand != 0) != (dir != 0);
        if (PK) {
            pa = (dir ? arcpk : apk) + (ga >> 2);
            pb = (brc_side ? brcpk : bpk) + (gb >> 2);
            ra = (int32_t)(ga & 3);
            rb = (int32_t)(gb & 3);
        } else {
            pa = (dir ? arc : A.bases) + ga;
            pb = (brc_side ? brc : B.bases) + gb;
            ra = rb = 0;
        }
        tp_first = dir ? cs.rev_first : cs.fwd_first;
        tpb_first = dir ? cs.revb_first : cs.fwdb_first;
        R = DEAD;
        H = -1;
        NB = tp_first;
        HB = -1;
        NBB = tpb_first;
        L = 0;
        // d = 0: the seed diagonal, slid by lane 0 of the half
        int32_t i0 = 0, h0 = -1, nb0 = 0, hb0 = -1, nbb0 = 0;
        if (hl == 0) {
            int32_t j0 = 0;
            slide2<PK>(pa, ra, pb, rb, min(an, bn), i0, j0);
            int32_t cnt = 0;
            for (int32_t nextb = tp_first; nextb <= i0; nextb += ts) {
                const int32_t idx = lbase + pn + cnt;
                if (pn + cnt < lanecap) {
                    pool[idx].parent = h0;
                    pool[idx].d = 0;
                    pool[idx].j = nextb;
                }
                h0 = idx;
                nb0++;
                cnt++;
            }
            if (SYM)
                for (int32_t nextb = tpb_first; nextb <= i0; nextb += ts) {
                    const int32_t idx = lbase + pn + cnt;
                    if (pn + cnt < lanecap) {
                        pool[idx].parent = hb0;
                        pool[idx].d = 0;
                        pool[idx].j = nextb;
                    }
                    hb0 = idx;
                    nbb0++;
                    cnt++;
                }
            R = i0;
            H = h0;
            NB = tp_first + nb0 * ts;
            HB = hb0;
            NBB = tpb_first + nbb0 * ts;
            pn += cnt;
        }
        i0 = hlane<G, 0>(i0, hb);
        h0 = hlane<G, 0>(h0, hb);
        nb0 = hlane<G, 0>(nb0, hb);
        hb0 = hlane<G, 0>(hb0, hb);
        nbb0 = hlane<G, 0>(nbb0, hb);
        best_score = 2 * i0;
        best_i = i0;
        best_k = 0;
        best_d = 0;
        best_head = h0;
        best_nb = tp_first + nb0 * ts;
        best_headb = hb0;
        best_nbb = tpb_first + nbb0 * ts;
        ncell = hl == 0 ? 1u : 0u;
        d = 1;
        st = d <= o.dmax ? W2_EXT : W2_EXT_END;
    };

#ifdef DH_WAVE_GUARD
    uint32_t guard_ = 0;
#endif
    for (;;) {
#ifdef DH_WAVE_GUARD
        if (++guard_ > (1u << 22)) {  // debug builds: a stuck state machine reports instead of hanging
            if (hl == 0) printf("k_wave2 guard: block %d grp %d st %d d %d L %d item %d c %d nc %d nd %d\n", (int)blockIdx.x, grp,
                                st, d, L, cs.item, cs.c, cs.nc, cs.nd);
            err |= 8;
            break;
        }
#endif
        // the stepping loop proper: left only when a half needs bookkeeping (or both are done)
        if (wballot(st != W2_EXT && st != W2_DONE) == 0ull && wballot(st == W2_EXT) != 0ull) do {
          {
            // ======================================================== one difference level
            // (executed by every lane: a half that is done carries dead diagonals only, so the
            // step is a no-op for it and the loop body needs no divergent region)
            R = st == W2_EXT ? R : DEAD;
            const int32_t nL = L - 1;
            const int32_t kidx = (hl - nL) & (G - 1);
            const int32_t k = nL + kidx;
            const int32_t Rm = __builtin_amdgcn_ds_bpermute(addr_lo, R), Hm = __builtin_amdgcn_ds_bpermute(addr_lo, H),
                          Nm = __builtin_amdgcn_ds_bpermute(addr_lo, NB);
            const int32_t Rp = __builtin_amdgcn_ds_bpermute(addr_hi, R), Hp = __builtin_amdgcn_ds_bpermute(addr_hi, H),
                          Np = __builtin_amdgcn_ds_bpermute(addr_hi, NB);
            int32_t HBm = -1, NBm = tpb_first, HBp = -1, NBp = tpb_first;
            if (SYM) {
                HBm = __builtin_amdgcn_ds_bpermute(addr_lo, HB);
                NBm = __builtin_amdgcn_ds_bpermute(addr_lo, NBB);
                HBp = __builtin_amdgcn_ds_bpermute(addr_hi, HB);
                NBp = __builtin_amdgcn_ds_bpermute(addr_hi, NBB);
            }
            int32_t ni = -1, hd = -1, nbp = tp_first, hbn = -1, nbbp = tpb_first;
            const int32_t lim = min(an, bn + k);
            {
                const int32_t cs = R + 1, cdl = Rm + 1, ci = Rp;
                if (cs <= lim && cs > ni) {
                    ni = cs;
                    hd = H;
                    nbp = NB;
                    hbn = HB;
                    nbbp = NBB;
                }
                if (cdl <= lim && cdl > ni) {
                    ni = cdl;
                    hd = Hm;
                    nbp = Nm;
                    hbn = HBm;
                    nbbp = NBm;
                }
                if (ci <= lim && ci > ni) {
                    ni = ci;
                    hd = Hp;
                    nbp = Np;
                    hbn = HBp;
                    nbbp = NBp;
                }
            }
            bool alive = ni >= 0;
            int32_t j = ni - k;
            if (alive) slide2<PK>(pa, ra, pb, rb, lim, ni, j);
            // live diagonals of this level are counted per lane and summed when the extension ends;
            // a level without any falls through: nothing crosses, nothing beats the best, and the
            // window test below ends the extension
            ncell += alive ? 1u : 0u;
            bool ended = false;
            {
                // trace nodes for the boundaries crossed in (prev_i, ni]
                int32_t nextb = nbp;
                if (alive)
                    while (ni >= nextb) {
                        const int32_t idx = lbase + pn;
                        if (pn < lanecap) {
                            pool[idx].parent = hd;
                            pool[idx].d = d;
                            pool[idx].j = nextb - k;
                        }
                        hd = idx;
                        pn++;
                        nextb += ts;
                    }
                int32_t nextbb = nbbp;
                if (SYM && alive)
                    while (j >= nextbb) {
                        const int32_t idx = lbase + pn;
                        if (pn < lanecap) {
                            pool[idx].parent = hbn;
                            pool[idx].d = d;
                            pool[idx].j = nextbb + k;
                        }
                        hbn = idx;
                        pn++;
                        nextbb += ts;
                    }
                {
                    R = alive ? ni : DEAD;
                    H = hd;
                    NB = nextb;
                    HB = hbn;
                    NBB = nextbb;
                    const int32_t sc = alive ? 2 * ni - k - pen * d : INT32_MIN;
                    const int32_t step_best = hmax_i32<G>(sc, hb);
                    const uint32_t rot = (uint32_t)nL & (uint32_t)(G - 1);
                    if (step_best > best_score) {
                        const uint32_t hm = hballot<G>(alive && sc == step_best, hb);
                        const uint32_t hr = hrotr<G>(hm, rot);
                        const int32_t step_kidx = __ffs((int)hr) - 1;
                        const int32_t src = (nL + step_kidx) & (G - 1);
                        best_score = step_best;
                        best_k = nL + step_kidx;
                        best_i = hread(R, src, hb);
                        best_head = hread(H, src, hb);
                        best_nb = hread(NB, src, hb);
                        if (SYM) {
                            best_headb = hread(HB, src, hb);
                            best_nbb = hread(NBB, src, hb);
                        }
                        best_d = d;
                    }
                    if (alive && sc < best_score - xdrop) {
                        alive = false;
                        R = DEAD;
                    }
                    uint32_t lm = hballot<G>(alive, hb);
                    if (lm == 0u) {
                        ended = true;
                    } else {
                        uint32_t rm = hrotr<G>(lm, rot);
                        int32_t l2 = nL + (__ffs((int)rm) - 1);
                        int32_t u2 = nL + (31 - __clz((int)rm));
                        if (u2 - l2 + 1 > o.width) {
                            // Narrow windows trim on most levels.  A level adds at most one diagonal on
                            // each side, so at most two edges go: fetch the scores of the two lowest and
                            // the two highest live diagonals in one crossbar round trip and replay the
                            // rule (drop the lower-scoring edge, ties the low edge) on them.
                            const uint32_t rml = rm & (rm - 1u);
                            const int32_t pl0 = __ffs((int)rm) - 1, pu0 = 31 - __clz((int)rm);
                            const int32_t pl1 = __ffs((int)rml) - 1, pu1 = 31 - __clz((int)(rm & ~(1u << pu0)));
                            const int32_t val = 2 * R - k;
                            const int32_t sl0 = hread(val, (nL + pl0) & (G - 1), hb), sl1 = hread(val, (nL + pl1) & (G - 1), hb);
                            const int32_t su0 = hread(val, (nL + pu0) & (G - 1), hb), su1 = hread(val, (nL + pu1) & (G - 1), hb);
                            const bool low1 = sl0 <= su0;
                            const int32_t kill1 = low1 ? pl0 : pu0;
                            const int32_t nl = low1 ? pl1 : pl0, nu = low1 ? pu0 : pu1;
                            const bool low2 = (low1 ? sl1 : sl0) <= (low1 ? su0 : su1);
                            const int32_t kill2 = nu - nl + 1 > o.width ? (low2 ? nl : nu) : -1;
                            if (kidx == kill1 || kidx == kill2) {
                                alive = false;
                                R = DEAD;
                            }
                            lm = hballot<G>(alive, hb);
                            rm = hrotr<G>(lm, rot);
                            l2 = nL + (__ffs((int)rm) - 1);
                            u2 = nL + (31 - __clz((int)rm));
                        }
                        while (u2 - l2 + 1 > o.width) {
                            const int32_t val = 2 * R - k;
                            const int32_t sl = hread_fast<G>(val, l2, hb);
                            const int32_t su = hread_fast<G>(val, u2, hb);
                            const int32_t kill = sl <= su ? l2 : u2;
                            if (k == kill) {
                                alive = false;
                                R = DEAD;
                            }
                            lm = hballot<G>(alive, hb);
                            rm = hrotr<G>(lm, rot);
                            l2 = nL + (__ffs((int)rm) - 1);
                            u2 = nL + (31 - __clz((int)rm));
                        }
                        L = l2;
                    }
                }
            }
            d++;
            st = ((ended || d > o.dmax) && st == W2_EXT) ? W2_EXT_END : st;
          }
        } while (wballot(st == W2_EXT_END) == 0ull);  // (no half can run out of work inside the loop)
        if (st != W2_EXT && st != W2_DONE) {
            // ======================================================== bookkeeping of this half
            while (st != W2_EXT && st != W2_DONE) {
                if (st == W2_FETCH) {
                    int32_t it = 0;
                    if (hl == 0) it = (int32_t)atomicAdd(ws.queue, 1u);
                    it = hlane<G, 0>(it, hb);
                    if (it >= (ws.units ? (int32_t)*ws.nunits : nitems)) {
                        st = W2_DONE;
                        break;
                    }
                    // work unit: a whole item, or (symmetric mode) one group of candidates of an item
                    int32_t c0 = 0, c1 = INT32_MAX, ui = it;
                    if (ws.units) {
                        const int4 u = ws.units[it];
                        ui = u.x;
                        c0 = u.y;
                        c1 = u.z;
                    }
                    const int32_t item = item0 + ui;
                    cs.item = item;
                    cs.r = item >> 1;
                    cs.strand = item & 1;
                    cs.nc = min(max(ncand[item], 0), c1);
                    const int64_t bo = B.off[item >> 1];
                    cs.bo = bo;
                    cs.blen = (int32_t)(B.off[(item >> 1) + 1] - bo);
#pragma unroll
                    for (int sx = 0; sx < NG; sx++) gr[7 * sx][hl] = -1;
                    cs.nd = cs.nacc = cs.ntr = 0;
                    cs.c = c0;
                    st = W2_CAND;
                } else if (st == W2_CAND) {
                    bool started = false;
                    const int32_t item = cs.item, nc = cs.nc, nd = cs.nd;
                    int32_t c = cs.c;
                    while (c < nc && (SYM || cs.nacc < o.max_la) && nd < LANES) {
                        const DhCand cd = cand[(int64_t)item * o.max_cand + c];
                        const int32_t sdc = cd.apos - cd.bpos;
                        bool covd = false;
#pragma unroll 1
                        for (int sx = 0; sx < NG; sx++)
                            covd = covd || (hl + sx * G < nd && gr[7 * sx][hl] == cd.aseq && cd.apos >= gr[7 * sx + 1][hl] &&
                                            cd.apos < gr[7 * sx + 2][hl] && cd.bpos >= gr[7 * sx + 3][hl] &&
                                            cd.bpos < gr[7 * sx + 4][hl] && sdc >= gr[7 * sx + 5][hl] - 64 &&
                                            sdc <= gr[7 * sx + 6][hl] + 64);
                        if (hballot<G>(covd, hb) != 0u) {
                            c++;
                            continue;
                        }
                        const int32_t as = cd.apos, bs = cd.bpos;
                        cs.c_aseq = cd.aseq;
                        cs.as = as;
                        cs.bs = bs;
                        cs.sd = sdc;
                        const int64_t ao = A.off[cd.aseq];
                        cs.ao = ao;
                        cs.alen = (int32_t)(A.off[cd.aseq + 1] - ao);
                        cs.fwd_first = ts - (as % ts);
                        cs.rev_first = (as % ts) ? (as % ts) : ts;
                        const int32_t resb = cs.strand ? cs.blen % ts : 0;
                        cs.resb = resb;
                        int32_t bm = (bs - resb) % ts;
                        bm = bm < 0 ? bm + ts : bm;
                        cs.fwdb_first = ts - bm;
                        cs.revb_first = bm ? bm : ts;
                        pn = 0;
                        ext_begin(0);
                        started = true;
                        break;
                    }
                    cs.c = c;
                    if (!started) {
                        if (!SYM && hl == 0) {
                            out_nla[item] = cs.nacc;
                            out_ntr[item] = cs.ntr;
                        }
                        st = W2_FETCH;
                    }
                } else if (st == W2_EXT_END) {
                    {
                        // sum of the per-lane counts over the half
                        uint32_t tot = ncell;
                        for (int off = G / 2; off > 0; off >>= 1) tot += (uint32_t)__shfl_xor((int)tot, off, LANES);
                        cs.cells += tot;
                        // a lane that ran out of node slots wrote nothing past its stretch; its chains are
                        // broken, so the alignment is reported instead of used
                        if (hballot<G>(pn > lanecap, hb) != 0u) err |= DH_ST_POOL_OVERFLOW;
                    }
                    const int32_t r_nb = (best_nb - tp_first) / ts, r_nbb = SYM ? (best_nbb - tpb_first) / ts : 0;
                    if (dir == 0 && !err) {
                        cs.fw_i = best_i;
                        cs.fw_j = best_i - best_k;
                        cs.fw_d = best_d;
                        cs.fw_head = best_head;
                        cs.fw_nb = r_nb;
                        cs.fw_headb = best_headb;
                        cs.fw_nbb = r_nbb;
                        ext_begin(1);
                        continue;
                    }
                    cs.rv_i = best_i;
                    cs.rv_j = best_i - best_k;
                    cs.rv_d = best_d;
                    cs.rv_head = best_head;
                    cs.rv_nb = r_nb;
                    cs.rv_headb = best_headb;
                    cs.rv_nbb = r_nbb;
                    cs.naln += 1;
                    if (err || cs.fw_nb > ws.nbmax || r_nb > ws.nbmax || cs.fw_nbb > ws.nbmax || r_nbb > ws.nbmax) {
                        err |= DH_ST_POOL_OVERFLOW;
                        st = W2_DONE;
                        break;
                    }
                    st = W2_POST_CHAIN;
                } else if (st == W2_POST_CHAIN) {
                    // chains: lane 0 forward, lane 1 reverse, lanes 2 / 3 the B-boundary families (SYM)
                    int32_t *cdj = ws.cdj + slot * 8 * ws.nbmax;
                    int32_t clo = 0, chi = 0;
                    if (hl < (SYM ? 4 : 2)) {
                        const bool isf = (hl & 1) == 0, isb = hl >= 2;
                        const int32_t head = isb ? (isf ? cs.fw_headb : cs.rv_headb) : (isf ? cs.fw_head : cs.rv_head);
                        const int32_t nb = isb ? (isf ? cs.fw_nbb : cs.rv_nbb) : (isf ? cs.fw_nb : cs.rv_nb);
                        const int32_t first = isb ? (isf ? cs.fwdb_first : cs.revb_first)
                                                  : (isf ? cs.fwd_first : cs.rev_first);
                        const int32_t bk = isb ? 0 : (isf ? cs.fw_i - cs.fw_j : cs.rv_i - cs.rv_j);
                        // layout of cdj: fd fj rd rj fdb fib rdb rib (nbmax each)
                        int32_t *cd = cdj + (int64_t)((isb ? 4 : 0) + (isf ? 0 : 2)) * ws.nbmax;
                        walk_chain(pool, head, nb, first, ts, bk, cd, cd + ws.nbmax, clo, chi);
                    }
                    __threadfence_block();
                    const int32_t flo = hlane<G, 0>(clo, hb), fhi = hlane<G, 0>(chi, hb);
                    const int32_t rlo = hlane<G, 1>(clo, hb), rhi = hlane<G, 1>(chi, hb);
                    const int32_t as = cs.as, bs = cs.bs, sd = cs.sd, nd = cs.nd;
                    const int32_t abpos = as - cs.rv_i, bbpos = bs - cs.rv_j, aepos = as + cs.fw_i, bepos = bs + cs.fw_j;
                    const int32_t diffs = cs.fw_d + cs.rv_d;
                    int32_t lo = sd + flo, hi = sd + fhi;
                    lo = (sd - rhi) < lo ? (sd - rhi) : lo;
                    hi = (sd - rlo) > hi ? (sd - rlo) : hi;
                    if (hl == (nd & (G - 1))) {
                        const int g0 = 7 * (nd / G);
                        gr[g0 + 0][hl] = cs.c_aseq;
                        gr[g0 + 1][hl] = abpos;
                        gr[g0 + 2][hl] = aepos;
                        gr[g0 + 3][hl] = bbpos;
                        gr[g0 + 4][hl] = bepos;
                        gr[g0 + 5][hl] = lo;
                        gr[g0 + 6][hl] = hi;
                    }
                    cs.nd = nd + 1;
                    cs.c = cs.c + 1;
                    cs.abpos = abpos;
                    cs.bbpos = bbpos;
                    cs.aepos = aepos;
                    cs.bepos = bepos;
                    cs.diffs = diffs;
                    const int64_t al = aepos - abpos, bl = bepos - bbpos;
                    const bool accept = al >= o.min_len &&
                                        (int64_t)2 * diffs * 1000000ll <= (int64_t)o.max_err_ppm * (al + bl);
                    st = accept ? W2_POST_REC1 : W2_CAND;
                } else if (st == W2_POST_REC1) {
                    // ---- the record (a, b): trace on the grid of A
                    int32_t *cdj = ws.cdj + slot * 8 * ws.nbmax;
                    const int32_t item = cs.item, strand = cs.strand, c_aseq = cs.c_aseq;
                    const int32_t item_a = SYM ? 2 * c_aseq + strand : item;
                    int32_t s1 = cs.nacc;
                    if (SYM) {
                        if (hl == 0) s1 = atomicAdd(&out_nla[item_a], 1);
                        s1 = hlane<G, 0>(s1, hb);
                        if (s1 >= o.max_la) {
                            // more overlaps than slots: the pair is dropped, both items are reported
                            // (their pile-up is skipped by the caller), everything else goes on
                            if (hl == 0) {
                                atomicSub(&out_nla[item_a], 1);
                                ws.item_ovf[item_a] = 1;
                                ws.item_ovf[item] = 1;
                            }
                            st = W2_CAND;
                            continue;
                        }
                    }
                    const int64_t oslot = (int64_t)item_a * o.max_la + s1;
                    const int32_t npairs = emit_trace(hl, G, ts, 0, cs.as, cs.bs, cs.abpos, cs.aepos, cs.bbpos,
                                                      cs.bepos, cs.rv_d, cs.fw_d, cs.rev_first, cs.rv_nb,
                                                      cdj + 2 * (int64_t)ws.nbmax, cdj + 3 * (int64_t)ws.nbmax,
                                                      cs.fwd_first, cs.fw_nb, cdj, cdj + ws.nbmax, false,
                                                      out_trace + oslot * trmax);
                    if (hl == 0) {
                        DhLa la;
                        la.tlen = 2 * npairs;
                        la.diffs = cs.diffs;
                        la.abpos = cs.abpos;
                        la.bbpos = cs.bbpos;
                        la.aepos = cs.aepos;
                        la.bepos = cs.bepos;
                        la.flags = strand ? 1u : 0u;
                        la.aread = c_aseq;
                        la.bread = cs.r;
                        la.pad = 0;
                        la.toff = 0;
                        out_la[oslot] = la;
                        if (SYM) atomicAdd(&out_ntr[item_a], 2 * npairs);
                    }
                    cs.nacc = cs.nacc + 1;
                    cs.ntr = cs.ntr + 2 * npairs;
                    st = SYM ? W2_POST_REC2 : W2_CAND;
                } else {  // W2_POST_REC2: the transposed record (b, a), trace on the grid of B
                    int32_t *cdj = ws.cdj + slot * 8 * ws.nbmax;
                    const int32_t item = cs.item, strand = cs.strand;
                    int32_t s2 = 0;
                    if (hl == 0) s2 = atomicAdd(&out_nla[item], 1);
                    s2 = hlane<G, 0>(s2, hb);
                    if (s2 >= o.max_la) {
                        if (hl == 0) {
                            atomicSub(&out_nla[item], 1);
                            ws.item_ovf[item] = 1;
                            ws.item_ovf[2 * cs.c_aseq + strand] = 1;
                        }
                        st = W2_CAND;
                        continue;
                    }
                    const int64_t slot2 = (int64_t)item * o.max_la + s2;
                    const int32_t np2 = emit_trace(hl, G, ts, cs.resb, cs.bs, cs.as, cs.bbpos, cs.bepos, cs.abpos,
                                                   cs.aepos, cs.rv_d, cs.fw_d, cs.revb_first, cs.rv_nbb,
                                                   cdj + 6 * (int64_t)ws.nbmax, cdj + 7 * (int64_t)ws.nbmax,
                                                   cs.fwdb_first, cs.fw_nbb, cdj + 4 * (int64_t)ws.nbmax,
                                                   cdj + 5 * (int64_t)ws.nbmax, strand != 0, out_trace + slot2 * trmax);
                    if (hl == 0) {
                        const int32_t blen = cs.blen, alen = cs.alen;
                        DhLa la;
                        la.tlen = 2 * np2;
                        la.diffs = cs.diffs;
                        la.abpos = strand ? blen - cs.bepos : cs.bbpos;
                        la.aepos = strand ? blen - cs.bbpos : cs.bepos;
                        la.bbpos = strand ? alen - cs.aepos : cs.abpos;
                        la.bepos = strand ? alen - cs.abpos : cs.aepos;
                        la.flags = strand ? 1u : 0u;
                        la.aread = cs.r;
                        la.bread = cs.c_aseq;
                        la.pad = 0;
                        la.toff = 0;
                        out_la[slot2] = la;
                        atomicAdd(&out_ntr[item], 2 * np2);
                    }
                    st = W2_CAND;
                }
            }
        }
        if (wballot(st != W2_DONE) == 0ull) break;
    }
    if (hl == 0) {
        atomicAdd(&counters[0], cs.cells);
        atomicAdd(&counters[1], cs.naln);
        if (err) atomicOr(status, err);
    }
}

// compaction of the per-item output slots: la_off / tr_off are the exclusive scans of the
// per-item LA counts and trace lengths; one wavefront per item copies its records and traces.
// ordered != 0 (symmetric mode: slots are claimed in racy order): the records of an item with at
// most 64 of them are written ordered by (bread, strand, abpos, bbpos, aepos, bepos), which makes
// the output deterministic and hands the host (A, B) pairs that are already adjacent.
__global__ void __launch_bounds__(LANES)
k_compact(const DhLa *__restrict__ la_slots, const uint16_t *__restrict__ tr_slots, int32_t trmax,
          int32_t max_la, int32_t ordered, int32_t nitems, const uint32_t *__restrict__ la_off,
          const uint32_t *__restrict__ tr_off, int64_t tr_base, DhLa *__restrict__ la_out,
          uint16_t *__restrict__ tr_out)
{
    const int32_t it = blockIdx.x;
    if (it >= nitems) return;
    const int lane = threadIdx.x;
    const uint32_t l0 = la_off[it], n = la_off[it + 1] - l0;
    uint32_t t = tr_off[it];
    if (ordered && n <= 256u) {
        // keys of the item's records in LDS; lane l ranks the records l, l + 64, ...
        __shared__ uint64_t sk1[256], sk2[256];
        __shared__ int32_t stl[256], sso[256];
        for (uint32_t x = (uint32_t)lane; x < n; x += LANES) {
            const DhLa la = la_slots[(int64_t)it * max_la + x];
            sk1[x] = ((uint64_t)(uint32_t)la.bread << 32) | ((uint64_t)(la.flags & 1u) << 31) | (uint32_t)la.abpos;
            sk2[x] = ((uint64_t)(uint32_t)la.bbpos << 32) | (uint32_t)la.aepos;
            stl[x] = la.tlen;
            sso[x] = (int32_t)la.toff;  // where the pairs start inside the slot (k_tile; 0 for the wave kernels)
        }
        __syncthreads();
        // rank among the records of the item, trace offset = prefix sum of the slot order
        for (uint32_t x = (uint32_t)lane; x < n; x += LANES) {
            const uint64_t k1 = sk1[x], k2 = sk2[x];
            int32_t rank = 0, toff = 0;
            for (uint32_t y = 0; y < n; y++) {
                const uint64_t y1 = sk1[y], y2 = sk2[y];
                const bool less = y1 < k1 || (y1 == k1 && (y2 < k2 || (y2 == k2 && y < x)));
                rank += less ? 1 : 0;
                toff += y < x ? stl[y] : 0;
            }
            DhLa la = la_slots[(int64_t)it * max_la + x];
            la.toff = tr_base + t + toff;
            la_out[l0 + rank] = la;
        }
        for (uint32_t x = 0; x < n; x++) {
            const int32_t xl = stl[x];
            const uint16_t *src = tr_slots + ((int64_t)it * max_la + x) * trmax + sso[x];
            for (int32_t e = lane; e < xl; e += LANES) tr_out[t + e] = src[e];
            t += xl;
        }
        return;
    }
    for (uint32_t x = 0; x < n; x++) {
        const int64_t slot = (int64_t)it * max_la + x;
        DhLa la = la_slots[slot];
        const uint16_t *src = tr_slots + slot * trmax + la.toff;
        for (int32_t e = lane; e < la.tlen; e += LANES) tr_out[t + e] = src[e];
        if (lane == 0) {
            la.toff = tr_base + t;
            la_out[l0 + x] = la;
        }
        t += la.tlen;
    }
}

// ------------------------------------------------------------------------------------ launchers

// resident blocks of a seed variant on the whole chip (persistent grid size)
template <int C, bool J = false, int NT = SEED_THREADS, int CC = SEED_CCAP>
static int seed_grid(int32_t nitems, int32_t ncu)
{
    static int per_cu = 0;
    if (per_cu == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_seed<C, J, NT, CC>, NT, 0) != hipSuccess || nb < 1)
            nb = 1;
        per_cu = nb;
    }
    int use = per_cu;
    if (const char *e = getenv("DH_SEED_BLOCKS_PER_CU")) use = std::max(1, std::min(per_cu, atoi(e)));  // development
    const int64_t g = (int64_t)use * ncu;
    return (int)(g < nitems ? g : nitems);
}

// ------------------------------------------------------------------------------------ DUST
// Low-complexity mask (the role of DBdust, symmetric DUST with -w64 -t2.0 -m10): a window of L bases
// (L = 16, 32, 64) is low-complexity when the triplets inside it repeat too often,
//     S = sum over triplet codes of c (c - 1) / 2  >  2 (l - 1),   l = L - 2 triplets,
// i.e. a DUST score above 2.0; the mask is the union of all such windows (windows holding a
// non-ACGT base are skipped).  The score depends on the multiset of triplets only, so a sequence and
// its reverse complement get mirrored masks.  One launch per window length; a thread slides its
// window over a chunk of `chunk` starts (a tile = 256 chunks) with byte counters in LDS ([code][thread]).
template <int L>
__global__ void __launch_bounds__(256)
k_dust(const uint8_t *__restrict__ bases, const int64_t *__restrict__ off, const int2 *__restrict__ tiles,
       int32_t ntiles, int32_t chunk, uint32_t *__restrict__ bits)
{
    __shared__ uint8_t cnt[64][256];
    const int32_t t = blockIdx.x;
    if (t >= ntiles) return;
    const int tid = threadIdx.x;
    const int32_t s = tiles[t].x;
    const int64_t o = off[s];
    const int32_t len = (int32_t)(off[s + 1] - o);
    const int32_t a0 = tiles[t].y + tid * chunk;      // first window start of this thread
    const int32_t a1 = min(a0 + chunk, len - L + 1);  // end of its window starts
    if (a0 >= a1) return;
    for (int c = 0; c < 64; c++) cnt[c][tid] = 0;
    const uint8_t *b = bases + o;
    // the triplets that enter and leave the window are two sequential streams: each keeps 8 bases in a register (one
    // unaligned 8-byte load per 6 triplets; the DB is padded) instead of three byte loads per triplet
    struct Stream {
        uint64_t w;
        int32_t p;
    };
    Stream sin{0, INT32_MIN / 2}, sout{0, INT32_MIN / 2};
    auto trip_of = [&](Stream &st, int32_t i) -> int32_t {  // code of the triplet at i, -1 when it holds a non-base
        if (i < st.p || i + 2 >= st.p + 8) {
            __builtin_memcpy(&st.w, b + i, 8);
            st.p = i;
        }
        const uint32_t v = (uint32_t)(st.w >> (8 * (i - st.p)));
        const uint32_t x = v & 0xFFu, y = (v >> 8) & 0xFFu, z = (v >> 16) & 0xFFu;
        return (x | y | z) > 3u ? -1 : (int32_t)(x << 4 | y << 2 | z);
    };
    auto trip = [&](int32_t i) -> int32_t { return trip_of(sin, i); };
    int32_t S = 0, bad = 0;
    for (int32_t i = a0; i < a0 + L - 2; i++) {
        const int32_t c = trip(i);
        if (c < 0)
            bad++;
        else
            S += cnt[c][tid]++;
    }
    for (int32_t a = a0; a < a1; a++) {
        if (bad == 0 && S > 2 * (L - 3)) {
            const int64_t g0 = o + a, g1 = g0 + L;  // mask [g0, g1)
            for (int64_t wd = g0 >> 5; wd <= (g1 - 1) >> 5; wd++) {
                const int64_t lo = max(g0, wd << 5), hi = min(g1, (wd + 1) << 5);
                const uint32_t m = (hi - lo == 32) ? 0xFFFFFFFFu : (((1u << (hi - lo)) - 1u) << (lo & 31));
                if ((bits[wd] & m) != m) atomicOr(&bits[wd], m);
            }
        }
        if (a + 1 < a1) {  // slide: triplet a leaves, triplet a + L - 2 enters
            const int32_t c0 = trip_of(sout, a), c1 = trip(a + L - 2);
            if (c0 < 0)
                bad--;
            else
                S -= --cnt[c0][tid];
            if (c1 < 0)
                bad++;
            else
                S += cnt[c1][tid]++;
        }
    }
}

// mask bits of slices: destination sequence i = source sequence sidx[i] from sbeg[i] on
__global__ void __launch_bounds__(256)
k_mask_slices(const uint32_t *__restrict__ src_bits, const int64_t *__restrict__ src_off,
              const int32_t *__restrict__ sidx, const int32_t *__restrict__ sbeg,
              const int64_t *__restrict__ dst_off, int32_t n, uint32_t *__restrict__ dst_bits)
{
    const int32_t i = blockIdx.y;
    if (i >= n) return;
    const int64_t d0 = dst_off[i], len = dst_off[i + 1] - d0, s0 = src_off[sidx[i]] + sbeg[i];
    for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < len; x += (int64_t)gridDim.x * blockDim.x) {
        const int64_t g = s0 + x;
        if (src_bits[g >> 5] >> (g & 31) & 1u) atomicOr(&dst_bits[(d0 + x) >> 5], 1u << ((d0 + x) & 31));
    }
}

// ---- alignment-coverage mask (maskRepetitiveRegions.d:238-430 BadAlignmentCoverageAssessor): the
// alignment intervals become +1 / -1 events in a difference array laid out like the DB plus one slot
// per sequence (an interval may end at the sequence's length); the event of position p sits at slot
// p + 1, so the exclusive scan leaves the coverage of base p at slot p + 2; coverage drops to zero at every sequence end, so
// one scan over the whole array serves all sequences.
__global__ void __launch_bounds__(256)
k_cov_events(const DhLa *__restrict__ las, int64_t n, const int64_t *__restrict__ off,
             const int64_t *__restrict__ roff, int32_t improper_only, int32_t allowance,
             uint32_t *__restrict__ diff)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const DhLa l = las[i];
    const int64_t o = off[l.aread];
    if (improper_only) {  // AlignmentChain.isProper, base.d:537-557
        const int32_t alen = (int32_t)(off[l.aread + 1] - o), blen = (int32_t)(roff[l.bread + 1] - roff[l.bread]);
        const bool proper = (l.abpos <= allowance || l.bbpos <= allowance) &&
                            (l.aepos + allowance >= alen || l.bepos + allowance >= blen);
        if (proper) return;
    }
    if (l.aepos <= l.abpos) return;
    const int64_t slot = o - off[0] + l.aread + 1;
    atomicAdd(&diff[slot + l.abpos], 1u);
    atomicAdd(&diff[slot + l.aepos], 0xFFFFFFFFu);
}

// bases whose coverage is outside [lower, upper] get their mask bit (runs of them are the intervals
// the assessor's event machine emits: it masks from an event entering a bad zone to the next event
// entering the ok zone, sequence ends closing a run)
__global__ void __launch_bounds__(256)
k_cov_mask_at(const uint32_t *__restrict__ cov, const int64_t *__restrict__ off, int32_t s0, int32_t nseq, int32_t lower,
              int32_t upper, uint32_t *__restrict__ bits)
{
    if ((int32_t)blockIdx.y >= nseq) return;
    const int32_t s = s0 + blockIdx.y;
    const int64_t o = off[s], e = off[s + 1], slot = o - off[0] + s + 1;
    // one thread per 32-bit word of the bitmap that the sequence touches
    const int64_t w0 = o >> 5, w1 = (e + 31) >> 5;
    for (int64_t w = w0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < w1; w += (int64_t)gridDim.x * blockDim.x) {
        const int64_t g0 = max(o, w << 5), g1 = min(e, (w << 5) + 32);
        uint32_t m = 0;
        for (int64_t g = g0; g < g1; g++) {
            const int32_t c = (int32_t)cov[slot + (g - o) + 1];  // exclusive scan: events at positions <= g - o
            if (c < lower || c > upper) m |= 1u << (g & 31);
        }
        if (m) atomicOr(&bits[w], m);
    }
}

// memset for large buffers: the runtime's fill kernel runs a fixed grid of 256 workgroups (one wavefront per SIMD on a
// quarter of the SIMDs) -- 2 GB took 5.6 ms = 0.36 TB/s in the chunk set-up of the mapping.  16-byte stores, a grid that
// fills the chip.
__global__ void __launch_bounds__(256) k_fill16(uint4 *__restrict__ p, int64_t n16, uint32_t v)
{
    const uint4 w = make_uint4(v, v, v, v);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) p[i] = w;
}

extern "C" {

// head and tail up to the next 16-byte boundary go through the runtime, the body through k_fill16
hipError_t dhk_memset(hipStream_t st, void *ptr, int value, size_t nbytes)
{
    if (nbytes < (1u << 20)) return hipMemsetAsync(ptr, value, nbytes, st);
    uint8_t *p = (uint8_t *)ptr;
    const size_t head = (size_t)((16 - ((uintptr_t)p & 15)) & 15);
    if (head) {
        const hipError_t e = hipMemsetAsync(p, value, head, st);
        if (e != hipSuccess) return e;
    }
    const size_t body = (nbytes - head) & ~(size_t)15, tail = nbytes - head - body;
    const uint32_t v = 0x01010101u * (uint32_t)(value & 255);
    const int64_t n16 = (int64_t)(body / 16);
    const int grid = (int)std::min<int64_t>((n16 + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(k_fill16, dim3(grid), dim3(256), 0, st, (uint4 *)(p + head), n16, v);
    if (tail) return hipMemsetAsync(p + head + body, value, tail, st);
    return hipGetLastError();
}

void dhk_revcomp(hipStream_t st, const uint8_t *src, uint8_t *dst, const int64_t *off, int32_t n,
                 int32_t max_len)
{
    if (n <= 0) return;
    int gx = (max_len + 2047) / 2048;  // 256 threads x 8 bases per block and step
    if (gx > 64) gx = 64;
    if (gx < 1) gx = 1;
    // grid.y is limited to 65535: loop in slabs
    for (int32_t s0 = 0; s0 < n; s0 += 65535) {
        const int32_t cnt = n - s0 < 65535 ? n - s0 : 65535;
        // shifted views: off + s0 keeps absolute offsets into src/dst
        hipLaunchKernelGGL(k_revcomp, dim3(gx, cnt), dim3(256), 0, st, src, dst, off + s0, cnt);
    }
}

void dhk_kmer_pass(hipStream_t st, int fill, DbView A, const int2 *tiles, int32_t ntiles, int32_t k,
                   int32_t kmer_mod, int32_t shift, uint32_t *dir, ulonglong2 *ent, const int64_t *goff)
{
    if (ntiles <= 0) return;
    if (fill)
        hipLaunchKernelGGL(k_kmer_pass<true>, dim3(ntiles), dim3(256), 0, st, A, tiles, ntiles, k,
                           kmer_mod, shift, dir, ent, goff);
    else
        hipLaunchKernelGGL(k_kmer_pass<false>, dim3(ntiles), dim3(256), 0, st, A, tiles, ntiles, k,
                           kmer_mod, shift, dir, ent, goff);
}

void dhk_group_index(hipStream_t st, int fill, DbView A, const int2 *tiles, const int32_t *gtile, int32_t ngroups,
                     int32_t slices_per_group, int32_t slice, int32_t k, int32_t kmer_mod, int32_t shift, uint32_t *dir,
                     ulonglong2 *ent, const int64_t *goff)
{
    if (ngroups <= 0) return;
    const dim3 grid((uint32_t)ngroups * (uint32_t)slices_per_group);
#define GI_LAUNCH(F, T)                                                                                              \
    hipLaunchKernelGGL((k_group_index<F, T>), grid, dim3(GI_THREADS), 0, st, A, tiles, gtile, slices_per_group, slice, k, \
                       kmer_mod, shift, dir, ent, goff)
    if (k <= 16 && shift < 32) {  // (a 32-bit word shifted by 32 would be undefined)
        if (fill)
            GI_LAUNCH(true, uint32_t);
        else
            GI_LAUNCH(false, uint32_t);
    } else {
        if (fill)
            GI_LAUNCH(true, uint64_t);
        else
            GI_LAUNCH(false, uint64_t);
    }
#undef GI_LAUNCH
}

// per-chunk summary of the seed filter's per-item results, so that the host fetches the per-item arrays only when it
// has to: out[0] = sum of hits, out[1] = sum of candidates, out[2] = items handed to the HBM variant (-1),
// out[3] = items the filter gave up on (-2)
__global__ void __launch_bounds__(256)
k_seed_summary(const int32_t *__restrict__ ncand, const int32_t *__restrict__ nhits, int32_t n, unsigned long long *__restrict__ out)
{
    unsigned long long h = 0, c = 0, big = 0, gave = 0;
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int32_t nc = ncand[i];
        h += (unsigned long long)max(nhits[i], 0);
        c += (unsigned long long)max(nc, 0);
        big += nc == -1 ? 1ull : 0ull;
        gave += nc == -2 ? 1ull : 0ull;
    }
    for (int off = 32; off > 0; off >>= 1) {
        h += __shfl_xor(h, off, 64);
        c += __shfl_xor(c, off, 64);
        big += __shfl_xor(big, off, 64);
        gave += __shfl_xor(gave, off, 64);
    }
    // one atomic per block and counter: per wavefront they were 16 000 returning-order atomics on four addresses for a
    // mapping chunk (2.7 ms for a kernel that reads 16 MB)
    __shared__ unsigned long long s_part[4][4];
    if ((threadIdx.x & 63) == 0) {
        s_part[threadIdx.x >> 6][0] = h;
        s_part[threadIdx.x >> 6][1] = c;
        s_part[threadIdx.x >> 6][2] = big;
        s_part[threadIdx.x >> 6][3] = gave;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const unsigned long long v = s_part[0][threadIdx.x] + s_part[1][threadIdx.x] + s_part[2][threadIdx.x] + s_part[3][threadIdx.x];
        if (v) atomicAdd(&out[threadIdx.x], v);
    }
}

void dhk_seed_summary(hipStream_t st, const int32_t *ncand, const int32_t *nhits, int32_t n, unsigned long long *out)
{
    (void)hipMemsetAsync(out, 0, 4 * sizeof(unsigned long long), st);
    if (n <= 0) return;
    hipLaunchKernelGGL(k_seed_summary, dim3(std::min((n + 255) / 256, 512)), dim3(256), 0, st, ncand, nhits, n, out);
}

void dhk_fat_dir(hipStream_t st, const uint32_t *dir, const ulonglong2 *ent, int64_t nb, ulonglong2 *fat)
{
    if (nb <= 0) return;
    hipLaunchKernelGGL(k_fat_dir, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st, dir, ent, nb, fat);
}

// exclusive scan in place; sums must hold ceil(n / 2048) uint32
void dhk_scan(hipStream_t st, uint32_t *v, int64_t n, uint32_t *sums)
{
    const int32_t nb = (int32_t)((n + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK);
    hipLaunchKernelGGL(k_scan_sums, dim3(nb), dim3(256), 0, st, v, n, sums);
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, st, sums, nb, (unsigned long long *)nullptr);
    hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(256), 0, st, v, n, sums);
}

// the same, adding the 64-bit total of the elements to *total64 (zeroed by the caller): a block's sum of SCAN_PER_BLOCK
// counters fits 32 bits as long as the counters themselves did not wrap, so the total tells whether the prefix sums did
void dhk_scan_total(hipStream_t st, uint32_t *v, int64_t n, uint32_t *sums, unsigned long long *total64)
{
    const int32_t nb = (int32_t)((n + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK);
    hipLaunchKernelGGL(k_scan_sums, dim3(nb), dim3(256), 0, st, v, n, sums);
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, st, sums, nb, total64);
    hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(256), 0, st, v, n, sums);
}

// queue: one zeroed uint32 (work counter of the persistent blocks)
// item0 / nitems: even (both strands of the reads [item0 / 2, (item0 + nitems) / 2))
// development / tests: DH_SEED_NO_REFINE=1 switches the second counting pass of the seed sort off (read per launch)
static void seed_sort_switch()
{
    static int cur = 1;
    const int want = getenv("DH_SEED_NO_REFINE") ? 0 : 1;
    if (want != cur) {
        (void)hipDeviceSynchronize();
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_seed_sort_refine), &want, sizeof(int));
        cur = want;
    }
}

void dhk_seed(hipStream_t st, int cap, DbView B, IndexView ix, DhOpts o,
              int32_t item0, int32_t nitems, DhCand *cand, int32_t *ncand, int32_t *nhits,
              int32_t *status, uint32_t *queue, int32_t ncu, uint64_t *fscr)
{
    if (nitems <= 0) return;
    seed_sort_switch();
    const int32_t read0 = item0 / 2, nreads = nitems / 2;
    const JoinView jv = {};
#define SEED_LAUNCH(C)                                                                            \
    hipLaunchKernelGGL((k_seed<C, false>), dim3(seed_grid<C>(nreads, ncu)), dim3(SEED_THREADS), 0, st, B, ix, jv, o, \
                       read0, nreads, cand, ncand, nhits, status, C == 8192 ? fscr : (uint64_t *)nullptr,    \
                       C == 8192 ? DH_SEED_FSCR_WORDS : 0, (const int32_t *)nullptr, queue)
    if (cap <= 1024)
        SEED_LAUNCH(1024);
    else if (cap <= 2048)
        SEED_LAUNCH(2048);
    else if (cap <= 4096)
        SEED_LAUNCH(4096);
    else if (cap <= 8192)
        SEED_LAUNCH(8192);
    else
        SEED_LAUNCH(16384);
#undef SEED_LAUNCH
}

#ifdef DH_SEED_PROF
void dhk_seed_prof_dump()
{
    unsigned long long h[12];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_seed_prof), sizeof(h));
    fprintf(stderr, "[seed prof] blocks %llu: lookup %.1f sort %.1f bcov %.1f bands %.1f rank %.1f us/block (of the sort: buckets %.1f scatter %.1f; of the bands: heads %.1f long ranges %.1f; %llu reads through the network)\n", h[7], h[0] / 100.0 / h[7], h[1] / 100.0 / h[7], h[2] / 100.0 / h[7], h[3] / 100.0 / h[7], h[4] / 100.0 / h[7], h[5] / 100.0 / h[7], h[6] / 100.0 / h[7], h[8] / 100.0 / h[7], h[9] / 100.0 / h[7], h[10]);
    unsigned long long z[12] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_seed_prof), z, sizeof(z));
}
#endif
// the items listed in item_list (absolute ids) with their hits staged in HBM: block x owns the
// slab gbuf[x * gcap ..]; nslabs bounds the grid
void dhk_seed_big(hipStream_t st, DbView B, IndexView ix, DhOpts o,
                  const int32_t *read_list, int32_t nreads, uint64_t *gbuf, int32_t gcap, DhCand *cand,
                  int32_t *ncand, int32_t *nhits, int32_t *status, uint32_t *queue, int32_t ncu)
{
    if (nreads <= 0) return;
    const JoinView jv = {};
    hipLaunchKernelGGL((k_seed<0, false>), dim3(seed_grid<0>(nreads, ncu)), dim3(SEED_THREADS), 0, st, B, ix, jv, o, 0,
                       nreads, cand, ncand, nhits, status, gbuf, gcap, read_list, queue);
}

// the same back end fed from the hit segments of the per-pile-up k-mer join (dh_join.hip)
void dhk_seed_join(hipStream_t st, int cap, DbView B, IndexView ix, DhOpts o, JoinView jv, int32_t item0, int32_t nitems,
                   DhCand *cand, int32_t *ncand, int32_t *nhits, int32_t *status, uint32_t *queue, int32_t ncu,
                   uint64_t *fscr, const int32_t *read_list, int32_t nlist)
{
    if (nitems <= 0 || (read_list && nlist <= 0)) return;
    seed_sort_switch();
    // read_list (device, nlist absolute read ids): only those reads -- the second tier of the join path, the reads whose
    // hits overflowed the first tier's LDS buffer
    const int32_t read0 = item0 / 2, nreads = read_list ? nlist : nitems / 2;
#define SEED_LAUNCH_J(C)                                                                          \
    hipLaunchKernelGGL((k_seed<C, true>), dim3(seed_grid<C, true>(nreads, ncu)), dim3(SEED_THREADS), 0, st, B, ix, jv, o, \
                       read0, nreads, cand, ncand, nhits, status, C >= 8192 ? fscr : (uint64_t *)nullptr,    \
                       C == 8192 ? DH_SEED_FSCR_WORDS : (C == 16384 ? DH_SEED_FSCR_WORDS16 : 0), read_list, queue)
    if (cap <= 512)  // the first tier of a mapping: a wavefront per read (140 hits at 1/8 sampling), 32 candidate band pairs
        hipLaunchKernelGGL((k_seed<512, true, 64, 32>), dim3(seed_grid<512, true, 64, 32>(nreads, ncu)), dim3(64), 0, st, B, ix, jv, o,
                           read0, nreads, cand, ncand, nhits, status, (uint64_t *)nullptr, 0, read_list, queue);
    else if (cap <= 2048)
        SEED_LAUNCH_J(2048);
    else if (cap <= 4096)
        SEED_LAUNCH_J(4096);
    else if (cap <= 8192)
        SEED_LAUNCH_J(8192);
    else
        SEED_LAUNCH_J(16384);
#undef SEED_LAUNCH_J
}

void dhk_seed_big_join(hipStream_t st, DbView B, IndexView ix, DhOpts o, JoinView jv, const int32_t *read_list,
                       int32_t nreads, uint64_t *gbuf, int32_t gcap, DhCand *cand, int32_t *ncand, int32_t *nhits,
                       int32_t *status, uint32_t *queue, int32_t ncu)
{
    if (nreads <= 0) return;
    hipLaunchKernelGGL((k_seed<0, true>), dim3(seed_grid<0, true>(nreads, ncu)), dim3(SEED_THREADS), 0, st, B, ix, jv, o, 0,
                       nreads, cand, ncand, nhits, status, gbuf, gcap, read_list, queue);
}

// apk / bpk / brcpk: 2-bit packed copies (all three or none)
void dhk_wave(hipStream_t st, int32_t nslots, DbView A, DbView B, const uint8_t *brc, const uint8_t *apk,
              const uint8_t *bpk, const uint8_t *brcpk, DhOpts o,
              int32_t item0, int32_t nitems, const DhCand *cand, const int32_t *ncand,
              WaveScratch ws, DhLa *out_la, uint16_t *out_trace, int32_t trmax, int32_t *out_nla,
              int32_t *out_ntr, unsigned long long *counters, int32_t *status)
{
    if (nitems <= 0) return;
    const bool pk = apk && bpk && brcpk;
#define WAVE_LAUNCH(S, P)                                                                          \
    hipLaunchKernelGGL((k_wave<S, P>), dim3(nslots), dim3(LANES), 0, st, A, B, brc, apk, bpk, brcpk, o, item0, \
                       nitems, cand, ncand, ws, out_la, out_trace, trmax, out_nla, out_ntr, counters, status)
    if (o.skip_self == 2) {
        if (pk)
            WAVE_LAUNCH(true, true);
        else
            WAVE_LAUNCH(true, false);
    } else {
        if (pk)
            WAVE_LAUNCH(false, true);
        else
            WAVE_LAUNCH(false, false);
    }
#undef WAVE_LAUNCH
}

// two (o.width <= 30) or four (o.width <= 14) alignments per wavefront: nslots blocks with 2 or 4
// scratch slots each; needs the reverse complement of A as well (arc, arcpk); apk / arcpk / bpk /
// brcpk all four or none
void dhk_wave2(hipStream_t st, int32_t nslots, DbView A, DbView B, const uint8_t *arc, const uint8_t *brc,
               const uint8_t *apk, const uint8_t *arcpk, const uint8_t *bpk, const uint8_t *brcpk, DhOpts o,
               int32_t item0, int32_t nitems, const DhCand *cand, const int32_t *ncand, WaveScratch ws,
               DhLa *out_la, uint16_t *out_trace, int32_t trmax, int32_t *out_nla, int32_t *out_ntr,
               unsigned long long *counters, int32_t *status)
{
    if (nitems <= 0) return;
    const bool pk = apk && arcpk && bpk && brcpk;
#define WAVE2_LAUNCH(S, P)                                                                         \
    do {                                                                                           \
        if (o.width <= 14 && !getenv("DH_WAVE_G32"))                                               \
            hipLaunchKernelGGL((k_wave2<S, P, 16>), dim3(nslots), dim3(LANES), 0, st, A, B, arc, brc, apk, arcpk, bpk, \
                               brcpk, o, item0, nitems, cand, ncand, ws, out_la, out_trace, trmax, out_nla, out_ntr,  \
                               counters, status);                                                  \
        else                                                                                       \
            hipLaunchKernelGGL((k_wave2<S, P, 32>), dim3(nslots), dim3(LANES), 0, st, A, B, arc, brc, apk, arcpk, bpk, \
                               brcpk, o, item0, nitems, cand, ncand, ws, out_la, out_trace, trmax, out_nla, out_ntr,  \
                               counters, status);                                                  \
    } while (0)
    if (o.skip_self == 2) {
        if (pk)
            WAVE2_LAUNCH(true, true);
        else
            WAVE2_LAUNCH(true, false);
    } else {
        if (pk)
            WAVE2_LAUNCH(false, true);
        else
            WAVE2_LAUNCH(false, false);
    }
#undef WAVE2_LAUNCH
}

void dhk_units(hipStream_t st, const DhCand *cand, const int32_t *ncand, int32_t item0, int32_t nitems,
               int32_t max_cand, void *units, uint32_t *nunits)
{
    if (nitems <= 0) return;
    hipLaunchKernelGGL(k_units, dim3((nitems + 255) / 256), dim3(256), 0, st, cand, ncand, item0, nitems, max_cand,
                       (int4 *)units, nunits);
}

void dhk_pack2(hipStream_t st, const uint8_t *src, int64_t total, uint8_t *dst, int32_t *flag)
{
    const int64_t nw = (total + 31) >> 5;
    if (nw <= 0) return;
    hipLaunchKernelGGL(k_pack2<false>, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, st, src, total, (uint64_t *)dst,
                       flag);
}
// plane-packed forward / reverse-complement copies of a chunk for k_tile, straight from the bytes
void dhk_pack2_planes(hipStream_t st, const uint8_t *src, int64_t total, uint8_t *dst, int32_t *flag)
{
    const int64_t nw = (total + 31) >> 5;
    if (nw <= 0) return;
    hipLaunchKernelGGL(k_pack2<true>, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, st, src, total, (uint64_t *)dst,
                       flag);
}

// zeroes the (at most two) destination words every sequence shares with its neighbours: what k_pack2_rc ORs into.
// Interior words are stored whole, so the rest of the buffer needs no memset (2 GB per chunk of the mapping).
__global__ void __launch_bounds__(256)
k_pack2_rc_bounds(const int64_t *__restrict__ off, int32_t n, int64_t a0, uint32_t *__restrict__ dst)
{
    const int32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const int64_t o = off[s], len = off[s + 1] - o;
    if (len <= 0) return;
    const int64_t w0 = (o - a0) >> 4, w1 = (o + len - 1 - a0) >> 4;
    const int64_t g0 = a0 + (w0 << 4), g1 = a0 + (w1 << 4);
    if (!(g0 >= o && g0 + 16 <= o + len)) dst[w0] = 0;
    if (!(g1 >= o && g1 + 16 <= o + len)) dst[w1] = 0;
}

void dhk_pack2_rc_bounds(hipStream_t st, const int64_t *off, int32_t n, int64_t a0, uint8_t *dst)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_pack2_rc_bounds, dim3((n + 255) / 256), dim3(256), 0, st, off, n, a0, (uint32_t *)dst);
}

void dhk_pack2_rc(hipStream_t st, const uint8_t *src, const int64_t *off, int32_t n, int32_t max_len, int64_t a0,
                  uint8_t *dst)
{
    if (n <= 0) return;
    int gx = (max_len / 16 + 255) / 256;
    gx = gx < 1 ? 1 : (gx > 64 ? 64 : gx);
    for (int32_t s0 = 0; s0 < n; s0 += 65535) {
        const int32_t cnt = n - s0 < 65535 ? n - s0 : 65535;
        hipLaunchKernelGGL(k_pack2_rc, dim3(gx, cnt), dim3(256), 0, st, src, off + s0, cnt, a0, (uint32_t *)dst);
    }
}

void dhk_pack2_rc_planes(hipStream_t st, const uint8_t *src, const int64_t *off, int32_t n, int32_t max_len, int64_t a0,
                         uint8_t *dst)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_pack2_rc_bounds32, dim3((n + 255) / 256), dim3(256), 0, st, off, n, a0, (uint64_t *)dst);
    int gx = (max_len / 32 + 255) / 256;
    gx = gx < 1 ? 1 : (gx > 64 ? 64 : gx);
    for (int32_t s0 = 0; s0 < n; s0 += 65535) {
        const int32_t cnt = n - s0 < 65535 ? n - s0 : 65535;
        hipLaunchKernelGGL(k_pack2_rc_planes, dim3(gx, cnt), dim3(256), 0, st, src, off + s0, cnt, a0, (unsigned long long *)dst);
    }
}

// the same result from the plane-packed forward copy `fwd` of the chunk (dhk_pack2_planes ran before on this stream)
void dhk_planes_rc(hipStream_t st, const uint8_t *fwd, const int64_t *off, int32_t n, int32_t max_len, int64_t a0, uint8_t *dst)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_pack2_rc_bounds32, dim3((n + 255) / 256), dim3(256), 0, st, off, n, a0, (uint64_t *)dst);
    int gx = (max_len / 32 + 255) / 256;
    gx = gx < 1 ? 1 : (gx > 64 ? 64 : gx);
    for (int32_t s0 = 0; s0 < n; s0 += 65535) {
        const int32_t cnt = n - s0 < 65535 ? n - s0 : 65535;
        hipLaunchKernelGGL(k_planes_rc, dim3(gx, cnt), dim3(256), 0, st, (const unsigned long long *)fwd, off + s0, cnt, a0,
                           (unsigned long long *)dst);
    }
}

void dhk_compact(hipStream_t st, const DhLa *la_slots, const uint16_t *tr_slots, int32_t trmax,
                 int32_t max_la, int32_t ordered, int32_t nitems, const uint32_t *la_off,
                 const uint32_t *tr_off, int64_t tr_base, DhLa *la_out, uint16_t *tr_out)
{
    if (nitems <= 0) return;
    hipLaunchKernelGGL(k_compact, dim3(nitems), dim3(LANES), 0, st, la_slots, tr_slots, trmax, max_la, ordered,
                       nitems, la_off, tr_off, tr_base, la_out, tr_out);
}

__global__ void __launch_bounds__(256) k_or_words(uint32_t *__restrict__ dst, const uint32_t *__restrict__ a,
                                                    const uint32_t *__restrict__ b, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = a[i] | b[i];
}
void dhk_or_words(hipStream_t st, uint32_t *dst, const uint32_t *a, const uint32_t *b, int64_t n)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_or_words, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dst, a, b, n);
}

void dhk_dust(hipStream_t st, const uint8_t *bases, const int64_t *off, const int2 *tiles, int32_t ntiles,
              int32_t chunk, uint32_t *bits)
{
    if (ntiles <= 0) return;
    hipLaunchKernelGGL(k_dust<16>, dim3(ntiles), dim3(256), 0, st, bases, off, tiles, ntiles, chunk, bits);
    hipLaunchKernelGGL(k_dust<32>, dim3(ntiles), dim3(256), 0, st, bases, off, tiles, ntiles, chunk, bits);
    hipLaunchKernelGGL(k_dust<64>, dim3(ntiles), dim3(256), 0, st, bases, off, tiles, ntiles, chunk, bits);
}

void dhk_cov_events(hipStream_t st, const DhLa *las, int64_t n, const int64_t *off, const int64_t *roff,
                    int32_t improper_only, int32_t allowance, uint32_t *diff)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_cov_events, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, las, n, off, roff, improper_only,
                       allowance, diff);
}

void dhk_cov_mask(hipStream_t st, const uint32_t *cov, const int64_t *off, int32_t nseq, int32_t max_len, int32_t lower,
                  int32_t upper, uint32_t *bits)
{
    if (nseq <= 0) return;
    int gx = (max_len / 32 + 255) / 256;
    gx = gx < 1 ? 1 : (gx > 64 ? 64 : gx);
    for (int32_t s0 = 0; s0 < nseq; s0 += 65535) {
        const int32_t cnt = nseq - s0 < 65535 ? nseq - s0 : 65535;
        // shifted views keep absolute offsets; the slot formula needs off[0] of the whole DB
        hipLaunchKernelGGL(k_cov_mask_at, dim3(gx, cnt), dim3(256), 0, st, cov, off, s0, cnt, lower, upper, bits);
    }
}

void dhk_mask_slices(hipStream_t st, const uint32_t *src_bits, const int64_t *src_off, const int32_t *sidx,
                     const int32_t *sbeg, const int64_t *dst_off, int32_t n, int32_t max_len, uint32_t *dst_bits)
{
    if (n <= 0) return;
    int gx = (max_len + 255) / 256;
    gx = gx < 1 ? 1 : (gx > 64 ? 64 : gx);
    for (int32_t s0 = 0; s0 < n; s0 += 65535) {
        const int32_t cnt = n - s0 < 65535 ? n - s0 : 65535;
        hipLaunchKernelGGL(k_mask_slices, dim3(gx, cnt), dim3(256), 0, st, src_bits, src_off, sidx + s0, sbeg + s0,
                           dst_off + s0, cnt, dst_bits);
    }
}

}  // extern "C"
