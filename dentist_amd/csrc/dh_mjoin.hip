// dh_mjoin.hip -- gfx950 kernels of the radix-partitioned k-mer join that seeds a mapping pass (design: dh_mjoin.h).
// Replaces, for `damapper ref reads.block` (source/dentist/dazzler.d:6158-6170), the directory lookups of the seed filter
// (k_seed, dh_kernels.hip): the hits a read gets are the same multiset.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

#include "dh_kmer.h"
#include "dh_mjoin.h"

#define LANES 64

#ifdef DH_MJ_PROF
__device__ unsigned long long g_mj_prof[16];
#define MP(i) if (tid == 0) { const unsigned long long t_ = wall_clock64(); atomicAdd(&g_mj_prof[i], t_ - tp_); tp_ = t_; }
#define MP_BEGIN unsigned long long tp_ = wall_clock64();
extern "C" void dhk_mj_prof_dump()
{
    unsigned long long h[16];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_mj_prof), sizeof(h));
    if (h[15])
        fprintf(stderr, "[mj prof] k_mj_part, %llu tiles: setup %.1f roll %.1f entries %.1f scan %.1f scatter %.1f write %.1f us per tile and block\n", h[15],
                h[0] / 100.0 / h[15], h[1] / 100.0 / h[15], h[2] / 100.0 / h[15], h[3] / 100.0 / h[15], h[4] / 100.0 / h[15], h[5] / 100.0 / h[15]);
    if (h[11])
        fprintf(stderr, "[mj prof] k_mj_filter2, %llu wavefronts: lifetime %.1f us, of it resolve %.1f, bitmap slice + item wait %.1f\n", h[11],
                h[8] / 100.0 / h[11], h[9] / 100.0 / h[11], h[10] / 100.0 / h[11]);
    unsigned long long z[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_mj_prof), z, sizeof(z));
}
#else
#define MP(i)
#define MP_BEGIN
#endif

namespace {

__device__ __forceinline__ uint32_t mj_xcc_id()
{
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}
// eight base codes (one per byte, first base in the low byte) -> 16 bits, base j at bits [2j, 2j + 1]
__device__ __forceinline__ uint64_t mj_pack8(uint64_t x)
{
    x &= 0x0303030303030303ull;
    x = (x | (x >> 6)) & 0x000F000F000F000Full;
    x = (x | (x >> 12)) & 0x000000FF000000FFull;
    return (x | (x >> 24)) & 0xFFFFull;
}
// order of the 32 two-bit groups reversed
__device__ __forceinline__ uint64_t mj_revpairs(uint64_t x)
{
    const uint64_t r = __brevll(x);
    return ((r >> 1) & 0x5555555555555555ull) | ((r & 0x5555555555555555ull) << 1);
}
__device__ __forceinline__ uint32_t mj_lanes_below(unsigned long long m)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
// an entry of the stream (read once: the non-temporal hint keeps it from pushing the partition's directory slice out of
// the XCD's L2, which the resolve sweep of k_mj_filter lives on)
__device__ __forceinline__ uint64_t mj_load8(const uint64_t *p) { return __builtin_nontemporal_load(p); }

// exclusive prefix sum over the block's threads (one value each); *total = sum.  s_w: one word per wavefront.
template <int THREADS>
__device__ __forceinline__ uint32_t mj_block_scan(uint32_t v, int tid, uint32_t *s_w, uint32_t *total)
{
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < LANES; off <<= 1) {
        const uint32_t up = __shfl_up(incl, off, LANES);
        if ((tid & (LANES - 1)) >= off) incl += up;
    }
    __syncthreads();  // s_w may still be read from a previous scan
    if ((tid & (LANES - 1)) == LANES - 1) s_w[tid / LANES] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < THREADS / LANES; w++) {
        const uint32_t x = s_w[w];
        if (w < tid / LANES) base += x;
        tot += x;
    }
    *total = tot;
    return base + incl - v;
}

constexpr uint64_t MJ_ORI_BIT = 1ull << MJ_POSBITS, MJ_PAL_BIT = 1ull << (MJ_POSBITS + 1);
constexpr int MJ_REMSH = MJ_POSBITS + 2, MJ_PSH = 53;

}  // namespace

// ------------------------------------------------------------------------------------ presence bitmap of the index
// two bits per key inside the slice of its partition: the bucket of a directory of 2^nbbits buckets, and a hash of the key's
// remainder (a one-probe filter of 11 bits per key lets 9 % of the absent k-mers through, the pair 3 %)
// (one 32-bit multiply: the 64-bit product this began with was six multiplies in front of every second probe)
__device__ __forceinline__ uint32_t mj_bit2(uint64_t rem, int sbits)
{
    const uint32_t x = (uint32_t)rem ^ (uint32_t)(rem >> 19) ^ ((uint32_t)(rem >> 32) << 13);
    return (x * 0x9E3779B1u) >> (32 - sbits);
}
__global__ void __launch_bounds__(256) k_mj_bitmap(const ulonglong2 *__restrict__ ent, int64_t n, int32_t k, int32_t nbbits, uint32_t *__restrict__ bm)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int remsh = 2 * k - MJ_PBITS, sbits = nbbits - MJ_PBITS;
    const uint64_t key = ent[i].x & ~(1ull << 63);
    const uint64_t p = key >> remsh, rem = key & ((1ull << remsh) - 1);
    const uint64_t b1 = (p << sbits) | (rem >> (remsh - sbits)), b2 = (p << sbits) | mj_bit2(rem, sbits);
    atomicOr(&bm[b1 >> 5], 1u << (b1 & 31));
    atomicOr(&bm[b2 >> 5], 1u << (b2 & 31));
}

// The same bits without a global atomic (round 6): the entries of the index lie in bucket order, so the keys of a partition
// are one contiguous range of `ent` -- a block finds it by binary search, sets the partition's slice of the bitmap in LDS and
// writes it out.  (k_mj_bitmap: 2 x 10^8 scattered atomics at the chip's ~30 G/s = 6.7 ms of every index build.)
__global__ void __launch_bounds__(1024) k_mj_bitmap_part(const ulonglong2 *__restrict__ ent, int64_t n, int32_t k, int32_t nbbits, uint32_t *__restrict__ bm)
{
    extern __shared__ uint32_t s_slice[];
    const int remsh = 2 * k - MJ_PBITS, sbits = nbbits - MJ_PBITS;
    const int32_t words = 1 << (sbits - 5);
    const uint64_t p = blockIdx.x;
    const int tid = threadIdx.x;
    for (int32_t i = tid; i < words; i += 1024) s_slice[i] = 0u;
    // first entry whose partition is >= q (every thread the same search: 28 broadcast loads)
    auto lower = [&](uint64_t q) {
        int64_t lo = 0, hi = n;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (((ent[mid].x & ~(1ull << 63)) >> remsh) < q)
                lo = mid + 1;
            else
                hi = mid;
        }
        return lo;
    };
    const int64_t e0 = lower(p), e1 = lower(p + 1);
    __syncthreads();
    for (int64_t i = e0 + tid; i < e1; i += 1024) {
        const uint64_t rem = ent[i].x & ((1ull << remsh) - 1);  // (the group bits of a grouped index never get here: A is ungrouped)
        const uint32_t b1 = (uint32_t)(rem >> (remsh - sbits)), b2 = mj_bit2(rem, sbits);
        atomicOr(&s_slice[b1 >> 5], 1u << (b1 & 31));
        atomicOr(&s_slice[b2 >> 5], 1u << (b2 & 31));
    }
    __syncthreads();
    for (int32_t i = tid; i < words; i += 1024) bm[(int64_t)p * words + i] = s_slice[i];
}

// ------------------------------------------------------------------------------------ partition
// first read that starts behind the first base of tile t (a binary search per tile, all tiles at once: inside k_mj_part
// it was a chain of 19 dependent loads in front of every tile)
__global__ void __launch_bounds__(256) k_mj_tile_reads(DbView B, MjView m)
{
    const int32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m.ntiles) return;
    const int64_t tb0 = m.c0 + (int64_t)t * m.tb;
    int32_t lo = m.r0, hi = m.r1;  // (off[r1] = c1 is the last "start": nothing crosses it)
    while (lo < hi) {
        const int32_t mid = (lo + hi) >> 1;
        if (B.off[mid] > tb0)
            hi = mid;
        else
            lo = mid + 1;
    }
    m.tile_r[t] = lo;
}

// the sampler with its kind known at compile time (0: every k-mer, 1: power-of-two modulus, 2: any modulus): the plain
// step of k_mj_part has no uniform branch left
template <int SAMP>
__device__ __forceinline__ bool mj_sampled(uint64_t km, const KmerSampler &s)
{
    if (SAMP == 0) return true;
    const uint32_t lo = (uint32_t)km, hi = (uint32_t)(km >> 32);
    uint32_t h = __umulhi(lo, 0x7F4A7C15u) + lo * 0x9E3779B9u;
    if (!s.small_k) h += hi * 0x7F4A7C15u;
    if (SAMP == 1) return (h & ((1u << s.rot) - 1u)) == 0u;
    const uint32_t t = h * s.inv;
    return __builtin_rotateright32(t, s.rot) <= s.thresh;
}

template <int SAMP, bool MASK>
__global__ void __launch_bounds__(MJ_THREADS, 4)
k_mj_part(DbView B, MjView m)
{
    __shared__ uint64_t buf[MJ_CAP];
    __shared__ uint32_t cnt[MJ_P];
    __shared__ int32_t rs[MJ_RS];
    __shared__ uint32_t s_w[MJ_THREADS / LANES];
    __shared__ uint32_t s_nrs;
    const int tid = threadIdx.x, lane = tid & (LANES - 1), wave = tid / LANES;
    constexpr int NW = MJ_THREADS / LANES, WCAP = MJ_CAP / NW;  // a wavefront stages its entries in its own eighth of buf
    const int k = m.k, mm = k - 1;
    const uint64_t mask = (1ull << (2 * k)) - 1;
    const int rcsh = 2 * (k - 1), remsh = 2 * k - MJ_PBITS;
    const uint64_t remmask = (1ull << remsh) - 1;
    const KmerSampler smp = kmer_sampler(m.kmer_mod, k);
    const int32_t per = m.tb / MJ_THREADS;
    for (int32_t t = blockIdx.x; t < m.ntiles; t += gridDim.x) {
        MP_BEGIN
        const int64_t tb0 = m.c0 + (int64_t)t * m.tb;
        const int32_t tlen = (int32_t)std::min<int64_t>(m.tb, m.c1 - tb0);  // k-mer starts of this tile: [0, tlen)
        for (int i = tid; i < MJ_P; i += MJ_THREADS) cnt[i] = 0;
        if (tid == 0) s_nrs = 0;
        const int32_t rfirst = m.tile_r[t];
        __syncthreads();
        // read starts inside (tb0, tb0 + tlen + k - 1): a k-mer may begin at one, never contain one
        for (int32_t i = tid;; i += MJ_THREADS) {
            const int32_t r = rfirst + i;
            if (r > m.r1) break;
            const int64_t o = B.off[r] - tb0;
            if (o >= (int64_t)tlen + k - 1) break;
            if (i < MJ_RS) rs[i] = (int32_t)o;
            atomicMax(&s_nrs, (uint32_t)i + 1u);
        }
        __syncthreads();
        int32_t nrs = (int32_t)s_nrs;
        if (nrs > MJ_RS) {
            if (tid == 0) atomicOr(m.status, DH_ST_MJ_OVERFLOW);
            nrs = MJ_RS;
        }
        MP(0)
        const uint8_t *b = B.bases + tb0;
        const int32_t xr0 = tid * per;
        // ---- the rolling pair after the first k - 1 bases of the lane's stretch, from three packed words
        uint64_t km, rc;
        int32_t valid = mm, jn = 0;
        {
            // (lanes behind the end of the chunk's last tile load nothing: the DB's padding ends 64 bytes behind its bases)
            const bool live = xr0 < tlen;
            const uint64_t w = !live ? 0ull
                                     : mj_pack8(load8(b + xr0)) | (mj_pack8(load8(b + xr0 + 8)) << 16) | (mj_pack8(load8(b + xr0 + 16)) << 32);
            const uint64_t mskm = (1ull << (2 * mm)) - 1;
            km = mj_revpairs(w & mskm) >> (64 - 2 * mm);
            rc = ((~w) & mskm) << 2;
            // first read start behind the lane's first position; a start inside the first k - 1 bases cuts `valid`
            int32_t lo = 0, hi = nrs;
            while (lo < hi) {
                const int32_t mid = (lo + hi) >> 1;
                if (rs[mid] > xr0)
                    hi = mid;
                else
                    lo = mid + 1;
            }
            jn = lo;
            while (jn < nrs && rs[jn] <= xr0 + mm - 1) {
                valid = xr0 + mm - rs[jn];
                jn++;
            }
        }
        int32_t nxt = jn < nrs ? rs[jn] : 0x7FFFFFFF;
        uint32_t wcount = 0;  // entries of this wavefront so far (uniform)
        // The loop does the least per base: roll, sample, and park the sampled k-mer (k-mer << 17 | position) in the
        // wavefront's eighth of buf -- one lane in kmer_mod is active there; the entry is made of it afterwards with every
        // lane busy.  Eight bases per loaded word, the next word on its way.
        uint64_t wnext = xr0 < tlen ? load8(b + xr0 + mm) : 0ull;
        const int32_t tmax = tlen - xr0;  // k-mer starts of this lane: tt < tmax
        // A group of eight bases is CAREFUL when, somewhere in the wavefront, a read begins inside it, a k-mer is not yet whole
        // behind an earlier read start, the tile is the chunk's last (short) one, or the wavefront's part of buf could run
        // full: then every step tests what the plain step takes for granted.  The plain step is roll, canonical choice,
        // sampler, ballot, store -- nothing else.
        const bool short_tile = tlen < m.tb;
#define MJ_STEP(CAREFUL_)                                                                                                        \
    {                                                                                                                             \
        const int32_t tt = tt0 + u, pp = xr0 + mm + tt; /* pp: the base that completes the k-mer starting at xr0 + tt */         \
        const uint32_t c = (uint32_t)(w >> (8 * u)) & 3u;                                                                         \
        if (CAREFUL_ && pp == nxt) {                                                                                              \
            valid = 0;                                                                                                            \
            do jn++;                                                                                                              \
            while (jn < nrs && rs[jn] == pp);                                                                                     \
            nxt = jn < nrs ? rs[jn] : 0x7FFFFFFF;                                                                                 \
        }                                                                                                                         \
        km = ((km << 2) | c) & mask;                                                                                              \
        rc = (rc >> 2) | ((uint64_t)(3u - c) << rcsh);                                                                            \
        if (CAREFUL_) valid++;                                                                                                    \
        const uint64_t canon = km < rc ? km : rc;                                                                                 \
        /* (no short circuits: every `&&` of lane-varying terms would be a branch on the execution mask) */                       \
        bool em = mj_sampled<SAMP>(canon, smp);                                                                                   \
        if (CAREFUL_) em = em & (tt < tmax) & (valid >= k);                                                                       \
        if (MASK) em = em && !mask_touch(B.mask_bits, tb0 + xr0 + tt, k);                                                         \
        const unsigned long long bal = __ballot(em);                                                                              \
        const uint32_t slot = wcount + mj_lanes_below(bal);                                                                       \
        /* SAMP == 0 (every k-mer is an entry: the reference's behaviour) parks the FINISHED entry -- canonical k-mer split into */ \
        /* partition and remainder, orientation, palindrome, position: the canonical choice is at hand here, the second phase  */ \
        /* then only counts; with sampling one lane in kmer_mod stores, so the k-mer is parked and finished by all lanes later  */ \
        if (CAREFUL_ ? (em & (slot < (uint32_t)WCAP)) : em)                                                                      \
            buf[wave * WCAP + slot] = SAMP == 0 ? ((1ull << 63) | ((canon >> remsh) << MJ_PSH) | ((canon & remmask) << MJ_REMSH) | \
                                                   (km == rc ? MJ_PAL_BIT : 0ull) | (rc < km ? MJ_ORI_BIT : 0ull) |                 \
                                                   (uint64_t)(xr0 + tt))                                                            \
                                                : ((km << MJ_POSBITS) | (uint64_t)(xr0 + tt));                                      \
        wcount += (uint32_t)__popcll(bal);                                                                                        \
    }
        for (int32_t tt0 = 0; tt0 < per; tt0 += 8) {
            const uint64_t w = wnext;
            wnext = tt0 + 8 < tmax ? load8(b + xr0 + mm + tt0 + 8) : 0ull;
            const bool careful = short_tile || wcount + 8u * LANES > (uint32_t)WCAP ||
                                 __ballot((nxt < xr0 + mm + tt0 + 8) | (valid < k)) != 0ull;
            if (careful) {
#pragma unroll
                for (int u = 0; u < 8; u++) MJ_STEP(true)
            } else {
#pragma unroll
                for (int u = 0; u < 8; u++) MJ_STEP(false)
                // (valid stays >= k: it only matters as a threshold until the next read start)
            }
        }
#undef MJ_STEP
        if (wcount > WCAP) {
            if (lane == 0) atomicOr(m.status, DH_ST_MJ_OVERFLOW);
            wcount = WCAP;
        }
        MP(1)
        // ---- entries from the parked k-mers (every lane busy), counted per partition
        uint64_t e16[WCAP / LANES];
#pragma unroll
        for (int i = 0; i < WCAP / LANES; i++) {
            const uint32_t idx = lane + i * LANES;
            e16[i] = 0;
            if (SAMP == 0) {
                if (idx < wcount) {
                    e16[i] = buf[wave * WCAP + idx];
                    atomicAdd(&cnt[(e16[i] >> MJ_PSH) & (MJ_P - 1)], 1u);
                }
            } else if (idx < wcount) {
                const uint64_t x = buf[wave * WCAP + idx];
                const uint64_t kmx = x >> MJ_POSBITS;
                const uint64_t rcx = (~(mj_revpairs(kmx) >> (64 - 2 * k))) & mask;
                const uint64_t canon = kmx < rcx ? kmx : rcx;
                const uint64_t p = canon >> remsh;
                e16[i] = (1ull << 63) | (p << MJ_PSH) | ((canon & remmask) << MJ_REMSH) | (kmx == rcx ? MJ_PAL_BIT : 0ull) |
                         (kmx != canon ? MJ_ORI_BIT : 0ull) | (x & ((1ull << MJ_POSBITS) - 1));
                atomicAdd(&cnt[p], 1u);
            }
        }
        // ---- counting sort by partition: offsets (two counters per thread), then the entries from the registers
        __syncthreads();
        MP(2)
        const uint32_t c0 = cnt[2 * tid], c1 = cnt[2 * tid + 1];
        uint32_t n;
        const uint32_t ex = mj_block_scan<MJ_THREADS>(c0 + c1, tid, s_w, &n);
        ((uint32_t *)m.segoff)[(int64_t)t * (MJ_P / 2) + tid] = ex | ((ex + c0) << 16);
        cnt[2 * tid] = ex;
        cnt[2 * tid + 1] = ex + c0;
        __syncthreads();
        MP(3)
#pragma unroll
        for (int i = 0; i < WCAP / LANES; i++) {
            const uint32_t idx = lane + i * LANES;
            if (idx < wcount) buf[atomicAdd(&cnt[(e16[i] >> MJ_PSH) & (MJ_P - 1)], 1u)] = e16[i];
        }
        __syncthreads();
        MP(4)
        uint4 *dst = (uint4 *)(m.ent + (int64_t)t * MJ_CAP);
        for (uint32_t i = tid; i < (n + 1) / 2; i += MJ_THREADS) dst[i] = ((const uint4 *)buf)[i];
        if (tid == 0) m.tile_n[t] = n;
        __syncthreads();
        MP(5)
#ifdef DH_MJ_PROF
        if (tid == 0) atomicAdd(&g_mj_prof[15], 1ull);
#endif
    }
}

// ------------------------------------------------------------------------------------ segment offsets by partition
// block: 32 tiles x all partitions.  seg[p][t] = start << 16 | count
__global__ void __launch_bounds__(256)
k_mj_transpose(MjView m)
{
    __shared__ uint16_t so[32][MJ_P + 2];
    __shared__ uint32_t tn[32];
    const int tid = threadIdx.x;
    const int32_t t0 = blockIdx.x * 32;
    for (int row = 0; row < 32; row++) {
        const int32_t t = t0 + row;
        for (int i = tid; i < MJ_P / 2; i += 256) {
            const uint32_t v = t < m.ntiles ? ((const uint32_t *)m.segoff)[(int64_t)t * (MJ_P / 2) + i] : 0u;
            so[row][2 * i] = (uint16_t)v;
            so[row][2 * i + 1] = (uint16_t)(v >> 16);
        }
    }
    if (tid < 32) tn[tid] = t0 + tid < m.ntiles ? m.tile_n[t0 + tid] : 0u;
    __syncthreads();
    for (int p = tid; p < MJ_P; p += 256) {
        uint32_t out[32];
#pragma unroll
        for (int row = 0; row < 32; row++) {
            const uint32_t a = so[row][p], e = p + 1 < MJ_P ? so[row][p + 1] : tn[row];
            out[row] = (a << 16) | (e - a);
        }
        uint4 *dst = (uint4 *)(m.seg + (int64_t)p * m.ntiles_pad + t0);
#pragma unroll
        for (int q = 0; q < 8; q++) dst[q] = make_uint4(out[4 * q], out[4 * q + 1], out[4 * q + 2], out[4 * q + 3]);
    }
}

// the lookup of one k-mer with the rules of seed_item's flush() (dh_kernels.hip): the bucket's only entry from the fat
// directory word, or the walk of a bucket with the -t cap per orientation class; strands as o.strands allows
namespace {
template <typename F>
__device__ __forceinline__ void mj_lookup(const IndexView &ix, const DhOpts &o, uint64_t key, bool ori, bool pal, F &&emit)
{
    constexpr uint64_t ORI = 1ull << 63;
    const ulonglong2 f = ix.fat[(uint32_t)(key >> ix.shift)];
    if (f.x == DH_FAT_EMPTY) return;
    const uint64_t bori = ori ? ORI : 0ull;
    if ((f.x >> 62) != 1ull) {
        if ((f.x & ~ORI) == key && o.tcap >= 1) {
            const bool same = (f.x & ORI) == bori;
            if ((same || pal) && (o.strands & 1)) emit(f.y, 0);
            if ((!same || pal) && (o.strands & 2)) emit(f.y, 1);
        }
        return;
    }
    const uint32_t ss = (uint32_t)f.y, ee = ss + (uint32_t)(f.y >> 32);
    bool dof = true, dor = true;
    if (ee - ss > (uint32_t)max(o.tcap, 0)) {
        int32_t runf = 0, runr = 0;
        for (uint32_t t = ss; t < ee; t++) {
            const uint64_t ex = ix.ent[t].x;
            if ((ex & ~ORI) != key) continue;
            const bool same = (ex & ORI) == bori;
            runf += (same || pal) ? 1 : 0;
            runr += (!same || pal) ? 1 : 0;
        }
        dof = runf > 0 && runf <= o.tcap;
        dor = runr > 0 && runr <= o.tcap;
        if (!dof && !dor) return;
    }
    dof = dof && (o.strands & 1);
    dor = dor && (o.strands & 2);
    for (uint32_t t = ss; t < ee; t++) {
        const ulonglong2 en = ix.ent[t];
        if ((en.x & ~ORI) != key) continue;
        const bool same = (en.x & ORI) == bori;
        if (dof && (same || pal)) emit(en.y, 0);
        if (dor && (!same || pal)) emit(en.y, 1);
    }
}
}  // namespace

// ------------------------------------------------------------------------------------ filter
// A block keeps the bitmap slice of ONE partition in LDS (blocks pull (partition, slice of the tile groups) items, the
// blocks of an XCD from the same queue).  A wavefront takes one TILE GROUP of the partition at a time: 16 segments, 16
// lanes each (a segment holds 7 - 8 entries on average, so one coalesced load per lane covers it; longer segments take
// further rounds), the entries of the groups behind it already in flight.  The entries that pass the filter -- the
// k-mers of A among them, and 3 % of the others -- go, compacted, to the wavefront's page of the survivor pool, one range
// per (group, partition): sseg[group][partition] = first << 24 | count.  No dependent global load: the kernel streams.
__global__ void __launch_bounds__(MJ_PROBE_THREADS)
k_mj_filter(IndexView ix, DhOpts o, MjView m)
{
    __shared__ uint32_t bm[1 << (MJ_MAXBITS - MJ_PBITS - 5)];
    __shared__ int32_t s_item;
    const int tid = threadIdx.x, lane = tid & (LANES - 1), wave = tid / LANES;
    constexpr int NW = MJ_PROBE_THREADS / LANES;
    constexpr int SPI = LANES / MJ_GROUP;          // segments per load instruction (4)
    constexpr int NIT = MJ_GROUP / SPI;            // load instructions per group and round (4)
    constexpr uint32_t SAFE = MJ_PAGE / 4;         // survivors one group may add to the wavefront's page
    constexpr int DEPTH = 3;                       // groups in flight behind the one at hand
    const int k = m.k, remsh = 2 * k - MJ_PBITS, sbits = m.nbbits - MJ_PBITS, bshift = remsh - sbits;
    const uint64_t remmask = (1ull << remsh) - 1;
    const int32_t slice_words = 1 << (sbits - 5);
    const int32_t ng = m.ntiles_pad / MJ_GROUP;
    const int32_t nitems_q = (MJ_P / 8) * MJ_SLICES;
    const uint32_t xcc = mj_xcc_id();
    const int sub = lane / MJ_GROUP, j = lane & (MJ_GROUP - 1);
    uint64_t page_base = 0;
    uint32_t fill = MJ_PAGE + 1;  // the wavefront has no page (yet, or the pool ran out)
    uint64_t sweep_from = 0;  // survivors of this wavefront from here on are not resolved yet
    bool pool_out = false;
    int32_t curp = -1;
    // RESOLVE: the survivors a wavefront has written for the partition at hand are looked up before it leaves the partition
    // (and before it leaves a page) -- its XCD's L2 then holds the partition's 2 MB slice of the directory, because the
    // XCD's blocks work on the same partition: 255 G lookups/s there against 54 G/s at random lines of HBM
    // (scripts/join_probe.cpp).  A survivor becomes, in place: 0 = no hit; its only hit, strand << 63 | virtual A position <<
    // 23 | position in the group + 1; or itself with bit 62 set = several hits (a palindrome, a repeat): k_mj_hits looks
    // those up again.  Four lookups per lane in flight; nothing here is on the streaming loop's critical path.
    auto resolve = [&](uint64_t from, uint64_t to, int32_t p) {
        constexpr uint64_t ORI = 1ull << 63;
        constexpr int RU = 4;
        const uint64_t keytop = (uint64_t)p << remsh;
        for (uint64_t i0 = from; i0 < to; i0 += RU * LANES) {
            uint64_t sv[RU];
            ulonglong2 f[RU];
#pragma unroll
            for (int u = 0; u < RU; u++) {
                const uint64_t i = i0 + (uint64_t)u * LANES + lane;
                sv[u] = i < to ? m.hits[i] : 0ull;
            }
#pragma unroll
            for (int u = 0; u < RU; u++) {  // the directory words of all of them, together
                f[u].x = DH_FAT_EMPTY;
                f[u].y = 0;
                if (sv[u]) f[u] = ix.fat[(uint32_t)((keytop | ((sv[u] >> MJ_REMSH) & remmask)) >> ix.shift)];
            }
#pragma unroll
            for (int u = 0; u < RU; u++) {
                const uint64_t i = i0 + (uint64_t)u * LANES + lane;
                if (i >= to) continue;
                const uint64_t key = keytop | ((sv[u] >> MJ_REMSH) & remmask);
                const uint64_t bori = (sv[u] & MJ_ORI_BIT) ? ORI : 0ull;
                const bool pal = (sv[u] & MJ_PAL_BIT) != 0;
                const uint64_t posg1 = ((sv[u] >> MJ_PSH) & (uint64_t)(MJ_GROUP - 1)) * (uint64_t)m.tb + (sv[u] & ((1ull << MJ_POSBITS) - 1)) + 1;
                uint64_t out = 0;
                if (f[u].x != DH_FAT_EMPTY) {
                    if ((f[u].x >> 62) != 1ull) {  // the bucket's only entry
                        if ((f[u].x & ~ORI) == key && o.tcap >= 1) {
                            const bool same = (f[u].x & ORI) == bori;
                            const bool h0 = (same || pal) && (o.strands & 1), h1 = (!same || pal) && (o.strands & 2);
                            if (h0 && h1)
                                out = sv[u] | (1ull << 62);
                            else if (h0 || h1)
                                out = ((uint64_t)(h1 ? 1 : 0) << 63) | ((f[u].y & ((1ull << 39) - 1)) << 23) | posg1;
                        }
                    } else {  // several entries: counted with the full rules; one hit is taken from the walk
                        uint32_t c = 0;
                        uint64_t hh = 0;
                        mj_lookup(ix, o, key, bori != 0, pal, [&](uint64_t v, int strand) {
                            hh = ((uint64_t)strand << 63) | ((v & ((1ull << 39) - 1)) << 23) | posg1;
                            c++;
                        });
                        out = c == 0 ? 0ull : (c == 1 ? hh : (sv[u] | (1ull << 62)));
                    }
                }
                m.hits[i] = out;
            }
        }
    };
    for (int qq = 0; qq < 8; qq++) {
        const uint32_t x = (xcc + qq) & 7u;  // own XCD's queue first, then whatever is left of the others'
        for (;;) {
            __syncthreads();
            if (tid == 0) s_item = (int32_t)atomicAdd(&m.ctr[x], 1u);
            __syncthreads();
            const int32_t item = s_item;
            if (item >= nitems_q) break;
            const int32_t p = (int32_t)x + 8 * (item / MJ_SLICES), sl = item % MJ_SLICES;
            if (p != curp) {
                const uint4 *src = (const uint4 *)(m.bitmap + (int64_t)p * slice_words);
                for (int i = tid; i < slice_words / 4; i += MJ_PROBE_THREADS) ((uint4 *)bm)[i] = src[i];
                if (slice_words < 4 && tid < slice_words) bm[tid] = m.bitmap[(int64_t)p * slice_words + tid];
                curp = p;
                __syncthreads();
            }
            const int32_t g0 = (int32_t)((int64_t)ng * sl / MJ_SLICES), g1 = (int32_t)((int64_t)ng * (sl + 1) / MJ_SLICES);
            const uint32_t *segp = m.seg + (int64_t)p * m.ntiles_pad;
            // descriptors of group g (one per lane of a quarter) and, from them, the lane's entries of round 0
            auto load_desc = [&](int32_t g) { return g < g1 ? segp[(int64_t)g * MJ_GROUP + j] : 0u; };
            auto load_ent = [&](int32_t g, uint32_t sdv, uint64_t *e) {
                bool more = false;
#pragma unroll
                for (int it = 0; it < NIT; it++) {
                    const int sidx = it * SPI + sub;  // segment of the group this lane serves in load `it`
                    const uint32_t sd = __shfl(sdv, sidx, LANES);
                    const int32_t t = g * MJ_GROUP + sidx;
                    const uint32_t cn = t < m.ntiles ? (sd & 0xFFFFu) : 0u;
                    more = more || cn > (uint32_t)MJ_GROUP;
                    e[it] = (uint32_t)j < cn ? mj_load8(m.ent + (int64_t)t * MJ_CAP + (sd >> 16) + j) : 0ull;  // (entries are never 0)
                }
                return more;
            };
            // pipeline: descriptors DEPTH + 1 groups ahead, entries DEPTH groups ahead
            uint32_t sdv[DEPTH + 2];
            uint64_t e[DEPTH + 1][NIT];
            bool more[DEPTH + 1];
            const int32_t gw = g0 + wave;
#pragma unroll
            for (int d = 0; d < DEPTH + 2; d++) sdv[d] = load_desc(gw + d * NW);
#pragma unroll
            for (int d = 0; d < DEPTH + 1; d++) {
#pragma unroll
                for (int it = 0; it < NIT; it++) e[d][it] = 0;
                more[d] = false;
                if (d < DEPTH && gw + d * NW < g1) more[d] = load_ent(gw + d * NW, sdv[d], e[d]);
            }
            for (int32_t g = gw; g < g1; g += NW) {
                if (g + DEPTH * NW < g1) more[DEPTH] = load_ent(g + DEPTH * NW, sdv[DEPTH], e[DEPTH]);
                const uint32_t sd_new = load_desc(g + (DEPTH + 2) * NW);
                if (fill + SAFE > MJ_PAGE && !pool_out) {  // room for the group's survivors: they form one range
                    if (fill <= MJ_PAGE && page_base + fill > sweep_from) resolve(sweep_from, page_base + fill, p);
                    uint32_t pg = 0;
                    if (lane == 0) pg = atomicAdd(&m.ctr[8], 1u);
                    pg = __shfl(pg, 0, LANES);
                    if (pg >= (uint32_t)m.npages) {
                        // (the survivors that find no page are counted: the host sizes the pool by them and runs the chunk again)
                        if (lane == 0) atomicOr(m.status, DH_ST_MJ_POOL);
                        fill = MJ_PAGE + 1;
                        pool_out = true;
                    } else {
                        page_base = (uint64_t)pg * MJ_PAGE;
                        fill = 0;
                    }
                    sweep_from = page_base + fill;
                }
                const bool have_page = fill + SAFE <= MJ_PAGE;
                const uint64_t first = page_base + fill;
                uint32_t gtot = 0;
                const bool any_more = __ballot(more[0]) != 0ull;
                for (uint32_t rnd = 0;; rnd++) {
                    bool again = false;
                    if (rnd > 0) {  // (rare: a segment with more than 16 entries)
#pragma unroll
                        for (int it = 0; it < NIT; it++) {
                            const int sidx = it * SPI + sub;
                            const uint32_t sd = __shfl(sdv[0], sidx, LANES);
                            const int32_t t = g * MJ_GROUP + sidx;
                            const uint32_t cn = t < m.ntiles ? (sd & 0xFFFFu) : 0u, jj = rnd * MJ_GROUP + j;
                            again = again || cn > (rnd + 1) * MJ_GROUP;
                            e[0][it] = jj < cn ? mj_load8(m.ent + (int64_t)t * MJ_CAP + (sd >> 16) + jj) : 0ull;
                        }
                    } else
                        again = more[0];
                    // ---- the two bits of every entry; survivors compacted behind the group's earlier ones
                    uint32_t before = 0;
#pragma unroll
                    for (int it = 0; it < NIT; it++) {
                        const uint64_t ev = e[0][it];
                        const uint64_t rem = (ev >> MJ_REMSH) & remmask;
                        const uint32_t b1 = (uint32_t)(rem >> bshift);
                        bool pass = ev != 0ull && ((bm[b1 >> 5] >> (b1 & 31)) & 1u);
                        if (pass) {
                            const uint32_t b2 = mj_bit2(rem, sbits);
                            pass = (bm[b2 >> 5] >> (b2 & 31)) & 1u;
                        }
                        const unsigned long long bal = __ballot(pass);
                        const uint32_t at = gtot + before + mj_lanes_below(bal);
                        if (pass && have_page && at < SAFE)
                            m.hits[first + at] = (ev & ~(0x3FFull << MJ_PSH)) | ((uint64_t)(it * SPI + sub) << MJ_PSH);
                        before += (uint32_t)__popcll(bal);
                    }
                    gtot += before;
                    if (!any_more || __ballot(again) == 0ull) break;
                }
                if (gtot > SAFE) {  // more survivors than a group may hold: a degenerate tile (one k-mer all over)
                    if (lane == 0) atomicOr(m.status, DH_ST_MJ_OVERFLOW);
                    gtot = 0;
                }
                if (lane == 0) m.hseg[(int64_t)g * MJ_P + p] = ((unsigned long long)first << 24) | (have_page ? gtot : 0u);
                if (!have_page && gtot && lane == 0) atomicAdd((unsigned long long *)(m.ctr + 12), (unsigned long long)gtot);
                if (have_page) fill += gtot;
                // rotate the pipeline
#pragma unroll
                for (int d = 0; d < DEPTH; d++) {
#pragma unroll
                    for (int it = 0; it < NIT; it++) e[d][it] = e[d + 1][it];
                    more[d] = more[d + 1];
                }
#pragma unroll
                for (int d = 0; d < DEPTH + 1; d++) sdv[d] = sdv[d + 1];
                sdv[DEPTH + 1] = sd_new;
            }
            if (fill <= MJ_PAGE && page_base + fill > sweep_from) {
                resolve(sweep_from, page_base + fill, p);
                sweep_from = page_base + fill;
            }
        }
    }
}

// ------------------------------------------------------------------------------------ filter, flat lane mapping (round 6)
// The filter above deals 16 lanes to every segment (tile, partition) -- 7-8 entries on average: half of the lanes idle in
// every load, bit test, ballot and store, ~400 wave-instructions per 128 entries, the kernel VALU-bound at 62 ms per launch
// unsampled (58 % VALU-busy SIMDs at 63 % lane use).  Here a wavefront takes 64 TILES of its partition at a time -- one
// descriptor per lane, one coalesced load -- and maps the ~512 entries of their 64 segments FLAT onto its lanes: an
// exclusive prefix sum of the counts (six DPP steps), a mark per segment start in a byte array of LDS, and for every round
// of 64 consecutive entries the segment of a lane's entry is the running maximum of the marks (six more DPP steps and the
// carry of the round before).  Every lane of a round tests an entry; ~55 wave-instructions per 64 entries.  The loads of
// the first MJ_F2_R rounds are issued together before the first entry is tested.  Survivors, ranges, pages and the resolve
// sweep are those of k_mj_filter: sseg[group][partition] still describes one range per group of 16 tiles.
#define MJ_F2_R 10          /* rounds whose loads are in flight together (640 entries; 512 expected) */
#define MJ_F2_MARKS 1024    /* flat indices one pass over the marks covers */
#define MJ_F2_TMAX 16383    /* entries of one partition in 64 tiles (14 bits of prefix; 512 expected -- repeat-rich reads put many
                               copies of a k-mer into one partition); more = a degenerate chunk */
namespace {
template <int CTRL, int ROWMASK>
__device__ __forceinline__ uint32_t mj_dpp0(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xF, false);
}
// inclusive scans over the 64 lanes: row_shr 1, 2, 4, 8 inside the rows of 16, then row_bcast:15 / row_bcast:31
__device__ __forceinline__ uint32_t mj_scan_add(uint32_t v)
{
    v += mj_dpp0<0x111, 0xF>(v);
    v += mj_dpp0<0x112, 0xF>(v);
    v += mj_dpp0<0x114, 0xF>(v);
    v += mj_dpp0<0x118, 0xF>(v);
    v += mj_dpp0<0x142, 0xA>(v);
    v += mj_dpp0<0x143, 0xC>(v);
    return v;
}
__device__ __forceinline__ uint32_t mj_scan_max(uint32_t v)  // (values >= 0; 0 = nothing)
{
    v = max(v, mj_dpp0<0x111, 0xF>(v));
    v = max(v, mj_dpp0<0x112, 0xF>(v));
    v = max(v, mj_dpp0<0x114, 0xF>(v));
    v = max(v, mj_dpp0<0x118, 0xF>(v));
    v = max(v, mj_dpp0<0x142, 0xA>(v));
    v = max(v, mj_dpp0<0x143, 0xC>(v));
    return v;
}
}  // namespace

__global__ void __launch_bounds__(MJ_PROBE_THREADS)
k_mj_filter2(IndexView ix, DhOpts o, MjView m)
{
    __shared__ uint32_t bm[1 << (MJ_MAXBITS - MJ_PBITS - 5)];
    __shared__ __attribute__((aligned(16))) uint8_t s_marks[MJ_PROBE_THREADS / LANES][MJ_F2_MARKS];
    __shared__ int32_t s_item;
    const int tid = threadIdx.x, lane = tid & (LANES - 1), wave = tid / LANES;
    constexpr int NW = MJ_PROBE_THREADS / LANES;
    constexpr int SG = LANES;                      // tiles a wavefront takes at a time
    constexpr int R = MJ_F2_R;
    const int k = m.k, remsh = 2 * k - MJ_PBITS, sbits = m.nbbits - MJ_PBITS, bshift = remsh - sbits;
    const uint64_t remmask = (1ull << remsh) - 1;
    const int32_t slice_words = 1 << (sbits - 5);
    const int32_t ng = m.ntiles_pad / MJ_GROUP;
    const int32_t nitems_q = (MJ_P / 8) * MJ_SLICES;
    const uint32_t xcc = mj_xcc_id();
    uint8_t *marks = s_marks[wave];
    uint64_t page_base = 0;
    uint32_t fill = MJ_PAGE + 1;  // the wavefront has no page (yet, or the pool ran out)
    uint64_t sweep_from = 0;      // survivors of this wavefront from here on are not resolved yet
    bool pool_out = false;
    int32_t curp = -1;
#ifdef DH_MJ_PROF
    unsigned long long t_res_ = 0, t_bm_ = 0;
    const unsigned long long t_all_ = wall_clock64();
#endif
    // RESOLVE: as in k_mj_filter
    auto resolve = [&](uint64_t from, uint64_t to, int32_t p) {
        constexpr uint64_t ORI = 1ull << 63;
        constexpr int RU = 4;
#ifdef DH_MJ_PROF
        const unsigned long long tr0_ = wall_clock64();
#endif
        const uint64_t keytop = (uint64_t)p << remsh;
        for (uint64_t i0 = from; i0 < to; i0 += RU * LANES) {
            uint64_t sv[RU];
            ulonglong2 f[RU];
#pragma unroll
            for (int u = 0; u < RU; u++) {
                const uint64_t i = i0 + (uint64_t)u * LANES + lane;
                sv[u] = i < to ? m.hits[i] : 0ull;
            }
#pragma unroll
            for (int u = 0; u < RU; u++) {  // the directory words of all of them, together
                f[u].x = DH_FAT_EMPTY;
                f[u].y = 0;
                if (sv[u]) f[u] = ix.fat[(uint32_t)((keytop | ((sv[u] >> MJ_REMSH) & remmask)) >> ix.shift)];
            }
            // the buckets that hold several entries are walked for the RU survivors of a lane TOGETHER, two entries of each per
            // trip (round 6): one survivor after the other, entry by entry (mj_lookup) the sweep was a chain of ~16 dependent
            // loads from the index per 256 survivors -- half of A's k-mers share their bucket -- and a third of the kernel
            uint64_t key[RU], posg1[RU], hh[RU], out[RU];
            uint32_t ss[RU], nb[RU], c[RU];
            bool slow[RU];
#pragma unroll
            for (int u = 0; u < RU; u++) {
                const uint64_t i = i0 + (uint64_t)u * LANES + lane;
                key[u] = keytop | ((sv[u] >> MJ_REMSH) & remmask);
                posg1[u] = ((sv[u] >> MJ_PSH) & (uint64_t)(MJ_GROUP - 1)) * (uint64_t)m.tb + (sv[u] & ((1ull << MJ_POSBITS) - 1)) + 1;
                out[u] = 0;
                hh[u] = 0;
                ss[u] = nb[u] = c[u] = 0;
                slow[u] = false;
                if (i >= to || f[u].x == DH_FAT_EMPTY) continue;
                const uint64_t bori = (sv[u] & MJ_ORI_BIT) ? ORI : 0ull;
                const bool pal = (sv[u] & MJ_PAL_BIT) != 0;
                if ((f[u].x >> 62) != 1ull) {  // the bucket's only entry
                    if ((f[u].x & ~ORI) == key[u] && o.tcap >= 1) {
                        const bool same = (f[u].x & ORI) == bori;
                        const bool h0 = (same || pal) && (o.strands & 1), h1 = (!same || pal) && (o.strands & 2);
                        if (h0 && h1)
                            out[u] = sv[u] | (1ull << 62);
                        else if (h0 || h1)
                            out[u] = ((uint64_t)(h1 ? 1 : 0) << 63) | ((f[u].y & ((1ull << 39) - 1)) << 23) | posg1[u];
                    }
                } else {  // several entries: counted with the full rules; one hit is taken from the walk
                    ss[u] = (uint32_t)f[u].y;
                    nb[u] = (uint32_t)(f[u].y >> 32);
                    if (nb[u] > (uint32_t)max(o.tcap, 0)) {  // (the -t cap may apply: the general walk)
                        slow[u] = true;
                        nb[u] = 0;
                    }
                }
            }
            for (uint32_t j = 0;; j += 2) {
                bool more = false;
#pragma unroll
                for (int u = 0; u < RU; u++) more = more || j < nb[u];
                if (__ballot(more) == 0ull) break;
                ulonglong2 en[RU][2];
#pragma unroll
                for (int u = 0; u < RU; u++)
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        en[u][q].x = en[u][q].y = 0ull;
                        if (j + q < nb[u]) en[u][q] = ix.ent[ss[u] + j + q];
                    }
#pragma unroll
                for (int u = 0; u < RU; u++) {
                    const uint64_t bori = (sv[u] & MJ_ORI_BIT) ? ORI : 0ull;
                    const bool pal = (sv[u] & MJ_PAL_BIT) != 0;
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        if (j + q >= nb[u] || (en[u][q].x & ~ORI) != key[u]) continue;
                        const bool same = (en[u][q].x & ORI) == bori;
                        const uint64_t hv = ((en[u][q].y & ((1ull << 39) - 1)) << 23) | posg1[u];
                        if ((same || pal) && (o.strands & 1)) {
                            hh[u] = hv;
                            c[u]++;
                        }
                        if ((!same || pal) && (o.strands & 2)) {
                            hh[u] = (1ull << 63) | hv;
                            c[u]++;
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < RU; u++) {
                const uint64_t i = i0 + (uint64_t)u * LANES + lane;
                if (i >= to) continue;
                if (slow[u]) {
                    const uint64_t bori = (sv[u] & MJ_ORI_BIT) ? ORI : 0ull;
                    const bool pal = (sv[u] & MJ_PAL_BIT) != 0;
                    uint32_t cc = 0;
                    uint64_t h1 = 0;
                    const uint64_t pg1 = posg1[u];
                    mj_lookup(ix, o, key[u], bori != 0, pal, [&](uint64_t v, int strand) {
                        h1 = ((uint64_t)strand << 63) | ((v & ((1ull << 39) - 1)) << 23) | pg1;
                        cc++;
                    });
                    out[u] = cc == 0 ? 0ull : (cc == 1 ? h1 : (sv[u] | (1ull << 62)));
                } else if (nb[u])
                    out[u] = c[u] == 0 ? 0ull : (c[u] == 1 ? hh[u] : (sv[u] | (1ull << 62)));
                m.hits[i] = out[u];
            }
        }
#ifdef DH_MJ_PROF
        t_res_ += wall_clock64() - tr0_;
#endif
    };
    for (int qq = 0; qq < 8; qq++) {
        const uint32_t x = (xcc + qq) & 7u;  // own XCD's queue first, then whatever is left of the others'
        for (;;) {
            __syncthreads();
            if (tid == 0) s_item = (int32_t)atomicAdd(&m.ctr[x], 1u);
            __syncthreads();
            const int32_t item = s_item;
            if (item >= nitems_q) break;
            const int32_t p = (int32_t)x + 8 * (item / MJ_SLICES), sl = item % MJ_SLICES;
#ifdef DH_MJ_PROF
            const unsigned long long tb0_ = wall_clock64();
#endif
            if (p != curp) {
                const uint4 *src = (const uint4 *)(m.bitmap + (int64_t)p * slice_words);
                for (int i = tid; i < slice_words / 4; i += MJ_PROBE_THREADS) ((uint4 *)bm)[i] = src[i];
                if (slice_words < 4 && tid < slice_words) bm[tid] = m.bitmap[(int64_t)p * slice_words + tid];
                curp = p;
                __syncthreads();
            }
#ifdef DH_MJ_PROF
            t_bm_ += wall_clock64() - tb0_;
#endif
            const int32_t g0 = (int32_t)((int64_t)ng * sl / MJ_SLICES), g1 = (int32_t)((int64_t)ng * (sl + 1) / MJ_SLICES);
            const int32_t t_end = g1 * MJ_GROUP;  // tiles [g0 * MJ_GROUP, t_end) of partition p
            const uint32_t *segp = m.seg + (int64_t)p * m.ntiles_pad;
            auto load_desc = [&](int32_t tb0) { return tb0 + lane < t_end ? segp[tb0 + lane] : 0u; };
            int32_t tb0 = g0 * MJ_GROUP + wave * SG;
            uint32_t sd_next = load_desc(tb0);
            for (; tb0 < t_end; tb0 += NW * SG) {
                const uint32_t sd = sd_next;
                sd_next = load_desc(tb0 + NW * SG);
                const uint32_t cn = tb0 + lane < m.ntiles ? (sd & 0xFFFFu) : 0u, st0 = sd >> 16;
                const uint32_t incl = mj_scan_add(cn);
                const uint32_t excl = incl - cn;
                uint32_t T = (uint32_t)__builtin_amdgcn_readlane((int)incl, LANES - 1);
                if (T > (uint32_t)MJ_F2_TMAX) {  // (a degenerate chunk: one k-mer all over; the directory path takes it)
                    if (lane == 0) atomicOr(m.status, DH_ST_MJ_OVERFLOW);
                    T = 0;
                }
                const uint32_t pk = excl | (st0 << 14);  // (start < 2^13)
                // ---- room for the survivors of these tiles (at most their T entries; T <= MJ_F2_TMAX < MJ_PAGE): they form one
                // range per group of 16.  A page is left only when the entries at hand might not fit: pages fill to the brim.
                static_assert(MJ_F2_TMAX < MJ_PAGE, "the entries of 64 tiles fit a page");
                if (fill + T > MJ_PAGE && !pool_out) {
                    if (fill <= MJ_PAGE && page_base + fill > sweep_from) resolve(sweep_from, page_base + fill, p);
                    uint32_t pg = 0;
                    if (lane == 0) pg = atomicAdd(&m.ctr[8], 1u);
                    pg = __shfl(pg, 0, LANES);
                    if (pg >= (uint32_t)m.npages) {
                        // (the survivors that find no page are counted: the host sizes the pool by them and runs the chunk again)
                        if (lane == 0) atomicOr(m.status, DH_ST_MJ_POOL);
                        fill = MJ_PAGE + 1;
                        pool_out = true;
                    } else {
                        page_base = (uint64_t)pg * MJ_PAGE;
                        fill = 0;
                    }
                    sweep_from = page_base + fill;
                }
                const bool have_page = fill + T <= MJ_PAGE;
                const uint64_t first = page_base + fill;
                const uint64_t *ebase = m.ent + (int64_t)tb0 * MJ_CAP;
                uint32_t gtot = 0, gq0 = 0, gq1 = 0, gq2 = 0;  // survivors so far; of the first three groups of 16 tiles
                uint32_t carry = 0;
                // one round: the entries wb + 64 r + lane.  issue(): segment of the lane's entry and its load
                auto issue = [&](uint32_t wb, uint32_t r, uint32_t *s_out) {
                    const uint32_t i = wb + r * LANES + lane;
                    const uint32_t mk = i < T ? (uint32_t)marks[r * LANES + lane] : 0u;
                    uint32_t s1 = max(mj_scan_max(mk), carry);
                    carry = (uint32_t)__builtin_amdgcn_readlane((int)s1, LANES - 1);
                    const uint32_t s = s1 ? s1 - 1u : 0u;
                    const uint32_t pks = (uint32_t)__shfl((int)pk, (int)s, LANES);
                    *s_out = s;
                    const uint32_t off = s * (uint32_t)MJ_CAP + (pks >> 14) + (i - (pks & 0x3FFFu));
                    return i < T ? mj_load8(ebase + off) : 0ull;  // (entries are never 0)
                };
                auto process = [&](uint64_t ev, uint32_t s) {
                    const uint64_t rem = (ev >> MJ_REMSH) & remmask;
                    const uint32_t b1 = (uint32_t)(rem >> bshift);
                    bool pass = ev != 0ull && ((bm[b1 >> 5] >> (b1 & 31)) & 1u);
                    if (pass) {
                        const uint32_t b2 = mj_bit2(rem, sbits);
                        pass = (bm[b2 >> 5] >> (b2 & 31)) & 1u;
                    }
                    const unsigned long long bal = __ballot(pass);
                    if (bal == 0ull) return;
                    const uint32_t at = gtot + mj_lanes_below(bal);
                    if (pass && have_page)  // (at < T)
                        m.hits[first + at] = (ev & ~(0x3FFull << MJ_PSH)) | ((uint64_t)(s & (MJ_GROUP - 1)) << MJ_PSH);
                    gtot += (uint32_t)__popcll(bal);
                    gq0 += (uint32_t)__popcll(__ballot(pass && s < 16u));
                    gq1 += (uint32_t)__popcll(__ballot(pass && s < 32u));
                    gq2 += (uint32_t)__popcll(__ballot(pass && s < 48u));
                };
                for (uint32_t wb = 0; wb < T; wb += MJ_F2_MARKS) {
                    // marks of this window: segment + 1 at the flat index its first entry has
                    ((uint4 *)marks)[lane] = make_uint4(0u, 0u, 0u, 0u);
                    if (cn && excl >= wb && excl < wb + (uint32_t)MJ_F2_MARKS) marks[excl - wb] = (uint8_t)(lane + 1);
                    if (wb) {  // the segment that holds entry wb - 1: the last one that begins before the window
                        const uint32_t before = mj_scan_max(cn && excl < wb ? (uint32_t)lane + 1u : 0u);
                        carry = (uint32_t)__builtin_amdgcn_readlane((int)before, LANES - 1);
                    }
                    const uint32_t nr = (min(T - wb, (uint32_t)MJ_F2_MARKS) + LANES - 1) / LANES;
                    uint64_t e[R];
                    uint32_t sg[R];
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        e[r] = 0ull;
                        sg[r] = 0;
                        if ((uint32_t)r < nr) e[r] = issue(wb, (uint32_t)r, &sg[r]);
                    }
#pragma unroll
                    for (int r = 0; r < R; r++)
                        if ((uint32_t)r < nr) process(e[r], sg[r]);
                    for (uint32_t r = R; r < nr; r++) {  // (rare: more than 640 entries in the window)
                        uint32_t s;
                        const uint64_t ev = issue(wb, r, &s);
                        process(ev, s);
                    }
                }
                if (lane < SG / MJ_GROUP) {
                    const int32_t g = tb0 / MJ_GROUP + lane;
                    const uint32_t lo = lane == 0 ? 0u : (lane == 1 ? gq0 : (lane == 2 ? gq1 : gq2));
                    const uint32_t hi = lane == 0 ? gq0 : (lane == 1 ? gq1 : (lane == 2 ? gq2 : gtot));
                    if (g < g1) m.hseg[(int64_t)g * MJ_P + p] = ((unsigned long long)(first + lo) << 24) | (have_page ? hi - lo : 0u);
                }
                if (!have_page && gtot && lane == 0) atomicAdd((unsigned long long *)(m.ctr + 12), (unsigned long long)gtot);
                if (have_page) fill += gtot;
            }
            if (fill <= MJ_PAGE && page_base + fill > sweep_from) {
                resolve(sweep_from, page_base + fill, p);
                sweep_from = page_base + fill;
            }
        }
    }
#ifdef DH_MJ_PROF
    if (lane == 0) {
        atomicAdd(&g_mj_prof[8], wall_clock64() - t_all_);
        atomicAdd(&g_mj_prof[9], t_res_);
        atomicAdd(&g_mj_prof[10], t_bm_);
        atomicAdd(&g_mj_prof[11], 1ull);
    }
#endif
}

// ------------------------------------------------------------------------------------ hits, by read
// A block per tile group: the survivors of the group's 1024 (partition) lists are looked up in the fat directory with
// exactly the rules of seed_item's flush() (dh_kernels.hip: the bucket's only entry from the directory word, or the walk
// of a bucket with the -t cap per orientation class), the hits are counted per read, the group's range of rhits is
// reserved, and a second pass over the survivors that had hits writes them, finished -- strand << 63 | diagonal << 24 |
// position on the oriented read: the hit encoding of k_seed --, grouped by read.  No buffer bounds the hits of a group.
#define MJ_HB_WORDS 2048 /* survivors of one tile group the second pass can mark: 65 536 */
__global__ void __launch_bounds__(MJ_THREADS)
k_mj_hits(DbView B, IndexView ix, DhOpts o, MjView m)
{
    __shared__ uint32_t hoff[MJ_P + 1];
    __shared__ uint64_t hfirst[MJ_P];
    __shared__ int32_t rsl[MJ_RG_READS + 2];
    __shared__ uint32_t rcnt[MJ_RG_READS], rcur[MJ_RG_READS];
    __shared__ uint32_t hb[MJ_HB_WORDS], mb[MJ_HB_WORDS];  // survivors with hits / with more than one
    __shared__ uint32_t s_w[MJ_THREADS / LANES];
    __shared__ int32_t s_ra, s_nrd;
    __shared__ unsigned long long s_base;
    const int tid = threadIdx.x;
    const int32_t g = blockIdx.x;
    const int64_t tbg = (int64_t)m.tb * MJ_GROUP;
    const int64_t gb0 = m.c0 + (int64_t)g * tbg;
    if (gb0 >= m.c1) return;
    const int64_t gb1 = std::min<int64_t>(m.c1, gb0 + tbg);
    const int k = m.k, remsh = 2 * k - MJ_PBITS;
    const uint64_t remmask = (1ull << remsh) - 1;
    // ---- the group's survivor lists, one per partition
    const unsigned long long hs0 = m.hseg[(int64_t)g * MJ_P + 2 * tid], hs1 = m.hseg[(int64_t)g * MJ_P + 2 * tid + 1];
    const uint32_t c0 = (uint32_t)(hs0 & 0xFFFFFFull), c1 = (uint32_t)(hs1 & 0xFFFFFFull);
    uint32_t n;
    const uint32_t ex = mj_block_scan<MJ_THREADS>(c0 + c1, tid, s_w, &n);
    hoff[2 * tid] = ex;
    hoff[2 * tid + 1] = ex + c0;
    hfirst[2 * tid] = hs0 >> 24;
    hfirst[2 * tid + 1] = hs1 >> 24;
    if (tid == 0) {
        hoff[MJ_P] = n;
        // the read that holds the group's first base (the one before the first read that starts behind it: k_mj_tile_reads)
        s_ra = max(m.r0, m.tile_r[(int64_t)g * MJ_GROUP] - 1);
        s_nrd = 0x7FFFFFFF;
    }
    for (int i = tid; i < MJ_RG_READS; i += MJ_THREADS) rcnt[i] = 0;
    for (int i = tid; i < MJ_HB_WORDS; i += MJ_THREADS) hb[i] = mb[i] = 0;
    __syncthreads();
    const int32_t ra = s_ra;
    for (int32_t i = tid;; i += MJ_THREADS) {
        const int32_t r = ra + i;
        if (r > m.r1) break;
        const int64_t o2 = B.off[r] - gb0;
        if (i <= MJ_RG_READS + 1) rsl[i] = (int32_t)std::min<int64_t>(o2, 0x7FFFFFF0);
        if (o2 >= gb1 - gb0 || r == m.r1) {  // the first boundary at or behind the group's end closes its last read
            atomicMin(&s_nrd, i);
            break;
        }
    }
    __syncthreads();
    const int32_t nrd = s_nrd;  // reads ra .. ra + nrd - 1 overlap the group; rsl[0 .. nrd] their boundaries
    if (nrd > MJ_RG_READS || n > MJ_HB_WORDS * 32u) {
        if (tid == 0) atomicOr(m.status, DH_ST_MJ_OVERFLOW);
        return;
    }
    if (n == 0) return;  // (the reads' rows of this group stay zero)
    // survivor e of the group: the list of the last partition p with hoff[p] <= e (ten steps without a branch: the searches
    // and loads of the four survivors a thread handles at a time overlap -- one survivor per iteration was a chain of an
    // LDS search, a load from the survivor pool and another search per survivor, 22 of them in a row)
    auto fetch = [&](uint32_t e, int32_t *pp) {
        int32_t lo = 0;
#pragma unroll
        for (int32_t step = MJ_P / 2; step > 0; step >>= 1)
            if (hoff[lo + step] <= e) lo += step;
        *pp = lo;
        return m.hits[hfirst[lo] + (e - hoff[lo])];
    };
    constexpr int HU = 8;
    auto read_of = [&](int32_t posg) {  // the last read with rsl[i] <= posg
        int32_t lo = 0, hi = nrd - 1;
        while (lo < hi) {
            const int32_t mid = (lo + hi + 1) >> 1;
            if (rsl[mid] <= posg)
                lo = mid;
            else
                hi = mid - 1;
        }
        return lo;
    };
    auto posg_of = [&](uint64_t sv) { return (int32_t)(((sv >> MJ_PSH) & (uint64_t)(MJ_GROUP - 1)) * (uint64_t)m.tb + (sv & ((1ull << MJ_POSBITS) - 1))); };
    for (uint32_t e0 = tid; e0 < n; e0 += MJ_THREADS * HU) {
        int32_t pv[HU];
        uint64_t svv[HU];
#pragma unroll
        for (int u = 0; u < HU; u++) {
            const uint32_t e = e0 + (uint32_t)u * MJ_THREADS;
            pv[u] = 0;
            svv[u] = e < n ? fetch(e, &pv[u]) : 0ull;
        }
#pragma unroll
        for (int u = 0; u < HU; u++) {
            const uint32_t e = e0 + (uint32_t)u * MJ_THREADS;
            const uint64_t sv = svv[u];
            const int32_t p = pv[u];
            if (sv == 0ull) continue;  // (behind the group's end, or) resolved by the filter kernel: no hit
            if (!((sv >> 62) & 1ull)) {  // its only hit, in place
                atomicOr(&hb[e >> 5], 1u << (e & 31));
                atomicAdd(&rcnt[read_of((int32_t)(sv & 0x7FFFFFull) - 1)], 1u);
                continue;
            }
            uint32_t c = 0;
            mj_lookup(ix, o, ((uint64_t)p << remsh) | ((sv >> MJ_REMSH) & remmask), (sv & MJ_ORI_BIT) != 0, (sv & MJ_PAL_BIT) != 0,
                      [&](uint64_t, int) { c++; });
            atomicOr(&hb[e >> 5], 1u << (e & 31));
            atomicOr(&mb[e >> 5], 1u << (e & 31));
            atomicAdd(&rcnt[read_of(posg_of(sv))], c);
        }
    }
    __syncthreads();
    uint32_t rc4[MJ_RG_READS / MJ_THREADS], sum = 0;
#pragma unroll
    for (int u = 0; u < MJ_RG_READS / MJ_THREADS; u++) {
        rc4[u] = rcnt[tid * (MJ_RG_READS / MJ_THREADS) + u];
        sum += rc4[u];
    }
    uint32_t tot;
    uint32_t rex = mj_block_scan<MJ_THREADS>(sum, tid, s_w, &tot);
    if (tid == 0) {
        // (the cursor keeps counting when the buffer is full: the host sizes it by the count and runs the chunk again)
        const unsigned long long base = atomicAdd((unsigned long long *)(m.ctr + 10), (unsigned long long)tot);
        if (base + tot > (unsigned long long)m.rcap) atomicOr(m.status, DH_ST_MJ_POOL);
        s_base = base;
    }
    __syncthreads();
    const unsigned long long base = s_base;
    if (tot == 0 || base + tot > (unsigned long long)m.rcap) return;
#pragma unroll
    for (int u = 0; u < MJ_RG_READS / MJ_THREADS; u++) {
        const int32_t i = tid * (MJ_RG_READS / MJ_THREADS) + u;
        rcur[i] = rex;
        if (i < nrd) {
            const int32_t r = ra + i;
            const int64_t gfirst = (B.off[r] - m.c0) / tbg;
            const int64_t jg = (int64_t)g - gfirst;
            if (jg >= 0 && jg < m.nseg)
                m.segtab[(int64_t)(r - m.r0) * m.nseg + jg] = ((base + rex) << 24) | rc4[u];
            else if (rc4[u])
                atomicOr(m.status, DH_ST_MJ_OVERFLOW);
        }
        rex += rc4[u];
    }
    __syncthreads();
    for (uint32_t e0 = tid; e0 < n; e0 += MJ_THREADS * HU) {
        int32_t pv[HU];
        uint64_t svv[HU];
        bool live[HU];
#pragma unroll
        for (int u = 0; u < HU; u++) {
            const uint32_t e = e0 + (uint32_t)u * MJ_THREADS;
            pv[u] = 0;
            live[u] = e < n && ((hb[e >> 5] >> (e & 31)) & 1u);
            svv[u] = live[u] ? fetch(e, &pv[u]) : 0ull;
        }
#pragma unroll
        for (int u = 0; u < HU; u++) {
        if (!live[u]) continue;
        const uint32_t e = e0 + (uint32_t)u * MJ_THREADS;
        const uint64_t sv = svv[u];
        const int32_t p = pv[u];
        if (!((mb[e >> 5] >> (e & 31)) & 1u)) {  // the hit itself
            const int32_t posg = (int32_t)(sv & 0x7FFFFFull) - 1, strand = (int32_t)(sv >> 63);
            const int32_t i = read_of(posg);
            const int32_t q = posg - rsl[i], blen = rsl[i + 1] - rsl[i];
            const int32_t qs = strand ? blen - k - q : q;
            const int64_t D = (int64_t)((sv >> 23) & ((1ull << 39) - 1)) + ix.sepv - qs;
            m.rhits[base + atomicAdd(&rcur[i], 1u)] = ((uint64_t)strand << 63) | ((uint64_t)D << 24) | (uint32_t)qs;
            continue;
        }
        const int32_t posg = posg_of(sv);
        const int32_t i = read_of(posg);
        const int32_t q = posg - rsl[i], blen = rsl[i + 1] - rsl[i];
        mj_lookup(ix, o, ((uint64_t)p << remsh) | ((sv >> MJ_REMSH) & remmask), (sv & MJ_ORI_BIT) != 0, (sv & MJ_PAL_BIT) != 0,
                  [&](uint64_t v, int strand) {
                      const int32_t qs = strand ? blen - k - q : q;
                      const int64_t D = (int64_t)(v & ((1ull << 40) - 1)) + ix.sepv - qs;
                      m.rhits[base + atomicAdd(&rcur[i], 1u)] = ((uint64_t)strand << 63) | ((uint64_t)D << 24) | (uint32_t)qs;
                  });
        }
    }
}

// ------------------------------------------------------------------------------------ launchers
extern "C" {

void dhk_mj_bitmap(hipStream_t st, const ulonglong2 *ent, int64_t n, int32_t k, int32_t nbbits, uint32_t *bm)
{
    if (n <= 0) return;
    const char *ev = getenv("DH_MJ_BITMAP_ATOMICS");  // development / tests: the scattered global atomics of round 5
    if (ev && atoi(ev) != 0) {
        hipLaunchKernelGGL(k_mj_bitmap, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ent, n, k, nbbits, bm);
        return;
    }
    const size_t lds = sizeof(uint32_t) << (nbbits - MJ_PBITS - 5);  // (nbbits <= MJ_MAXBITS: 128 KB)
    if (lds > 65536) (void)hipFuncSetAttribute((const void *)k_mj_bitmap_part, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_mj_bitmap_part, dim3(MJ_P), dim3(1024), lds, st, ent, n, k, nbbits, bm);
}

void dhk_mj_tile_reads(hipStream_t st, DbView B, MjView m)
{
    hipLaunchKernelGGL(k_mj_tile_reads, dim3((unsigned)((m.ntiles + 255) / 256)), dim3(256), 0, st, B, m);
}

void dhk_mj_run(hipStream_t st, DbView B, IndexView ix, DhOpts o, MjView m, int32_t ncu)
{
    (void)hipMemsetAsync(m.ctr, 0, 16 * sizeof(uint32_t), st);
    {
        const dim3 grid((unsigned)std::min<int64_t>(m.ntiles, (int64_t)ncu * 2 * 4)), block(MJ_THREADS);
        const bool masked = B.mask_bits != nullptr;
        const int samp = m.kmer_mod <= 1 ? 0 : ((m.kmer_mod & (m.kmer_mod - 1)) == 0 ? 1 : 2);
#define MJ_PART(S_, M_) hipLaunchKernelGGL((k_mj_part<S_, M_>), grid, block, 0, st, B, m)
        if (samp == 0) {
            if (masked) MJ_PART(0, true); else MJ_PART(0, false);
        } else if (samp == 1) {
            if (masked) MJ_PART(1, true); else MJ_PART(1, false);
        } else {
            if (masked) MJ_PART(2, true); else MJ_PART(2, false);
        }
#undef MJ_PART
    }
    hipLaunchKernelGGL(k_mj_transpose, dim3((unsigned)(m.ntiles_pad / 32)), dim3(256), 0, st, m);
    if (m.dbg & 2)  // development / tests (DH_MJ_DBG=2): the filter of round 5, 16 lanes per segment
        hipLaunchKernelGGL(k_mj_filter, dim3((unsigned)ncu), dim3(MJ_PROBE_THREADS), 0, st, ix, o, m);
    else
        hipLaunchKernelGGL(k_mj_filter2, dim3((unsigned)ncu), dim3(MJ_PROBE_THREADS), 0, st, ix, o, m);
    hipLaunchKernelGGL(k_mj_hits, dim3((unsigned)m.ngroups), dim3(MJ_THREADS), 0, st, B, ix, o, m);
#ifdef DH_MJ_PROF
    if (getenv("DH_TRACE")) {
        (void)hipStreamSynchronize(st);
        dhk_mj_prof_dump();
    }
#endif
}
}
