// dh_mjoin.hip -- gfx950 kernels of the radix-partitioned k-mer join that seeds a mapping pass (design: dh_mjoin.h).
// Replaces, for `damapper ref reads.block` (source/dentist/dazzler.d:6158-6170), the directory lookups of the seed filter
// (k_seed, dh_kernels.hip): the hits a read gets are the same multiset.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <algorithm>

#include "dh_kmer.h"
#include "dh_mjoin.h"

#define LANES 64

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
__device__ __forceinline__ uint64_t mj_load8(const uint64_t *p)
{
    uint64_t x;
    __builtin_memcpy(&x, p, 8);
    return x;
}

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
__global__ void __launch_bounds__(256) k_mj_bitmap(const ulonglong2 *__restrict__ ent, int64_t n, int32_t bshift, uint32_t *__restrict__ bm)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t bit = (ent[i].x & ~(1ull << 63)) >> bshift;
    atomicOr(&bm[bit >> 5], 1u << (bit & 31));
}

// ------------------------------------------------------------------------------------ partition
__global__ void __launch_bounds__(MJ_THREADS, 4)
k_mj_part(DbView B, MjView m)
{
    __shared__ uint64_t buf[MJ_CAP];
    __shared__ uint32_t cnt[MJ_P];
    __shared__ int32_t rs[MJ_RS];
    __shared__ uint32_t s_w[MJ_THREADS / LANES];
    __shared__ uint32_t s_n, s_nrs;
    __shared__ int32_t s_r;
    const int tid = threadIdx.x, lane = tid & (LANES - 1);
    const int k = m.k, mm = k - 1;
    const uint64_t mask = (1ull << (2 * k)) - 1;
    const int rcsh = 2 * (k - 1), remsh = 2 * k - MJ_PBITS;
    const uint64_t remmask = (1ull << remsh) - 1;
    const KmerSampler smp = kmer_sampler(m.kmer_mod, k);
    const int32_t per = m.tb / MJ_THREADS;
    for (int32_t t = blockIdx.x; t < m.ntiles; t += gridDim.x) {
        const int64_t tb0 = m.c0 + (int64_t)t * m.tb;
        const int32_t tlen = (int32_t)std::min<int64_t>(m.tb, m.c1 - tb0);  // k-mer starts of this tile: [0, tlen)
        for (int i = tid; i < MJ_P; i += MJ_THREADS) cnt[i] = 0;
        if (tid == 0) {
            s_n = 0;
            s_nrs = 0;
            // the first read that starts behind the tile's first base (off[r1] = c1 is the last "start": nothing crosses it)
            int32_t lo = m.r0, hi = m.r1;
            while (lo < hi) {
                const int32_t mid = (lo + hi) >> 1;
                if (B.off[mid] > tb0)
                    hi = mid;
                else
                    lo = mid + 1;
            }
            s_r = lo;
        }
        __syncthreads();
        // read starts inside (tb0, tb0 + tlen + k - 1): a k-mer may begin at one, never contain one
        for (int32_t i = tid;; i += MJ_THREADS) {
            const int32_t r = s_r + i;
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
        uint64_t w = 0;
        for (int32_t tt = 0; tt < per; tt++) {
            const int32_t pp = xr0 + mm + tt;  // the base that completes the k-mer starting at xr0 + tt
            if ((tt & 7) == 0) w = xr0 + tt < tlen ? load8(b + pp) : 0ull;
            const uint32_t c = (uint32_t)w & 3u;
            w >>= 8;
            if (pp == nxt) {
                valid = 0;
                do jn++;
                while (jn < nrs && rs[jn] == pp);
                nxt = jn < nrs ? rs[jn] : 0x7FFFFFFF;
            }
            km = ((km << 2) | c) & mask;
            rc = (rc >> 2) | ((uint64_t)(3u - c) << rcsh);
            valid++;
            const uint64_t canon = km < rc ? km : rc;
            bool em = xr0 + tt < tlen && valid >= k && kmer_sampled(canon, smp);
            if (em && B.mask_bits && mask_touch(B.mask_bits, tb0 + xr0 + tt, k)) em = false;
            const unsigned long long bal = __ballot(em);
            if (bal) {
                uint32_t base = 0;
                const int leader = __ffsll((long long)bal) - 1;
                if (lane == leader) base = atomicAdd(&s_n, (uint32_t)__popcll(bal));
                base = __shfl(base, leader, LANES);
                if (em) {
                    const uint32_t slot = base + mj_lanes_below(bal);
                    if (slot < MJ_CAP) {
                        const uint64_t p = canon >> remsh;
                        buf[slot] = (p << MJ_PSH) | ((canon & remmask) << MJ_REMSH) | (km == rc ? MJ_PAL_BIT : 0ull) |
                                    (km != canon ? MJ_ORI_BIT : 0ull) | (uint64_t)(xr0 + tt);
                        atomicAdd(&cnt[p], 1u);
                    }
                }
            }
        }
        __syncthreads();
        uint32_t n = s_n;
        if (n > MJ_CAP) {
            if (tid == 0) atomicOr(m.status, DH_ST_MJ_OVERFLOW);
            n = MJ_CAP;
        }
        // ---- counting sort by partition: offsets (two counters per thread), the tile's entries through registers
        const uint32_t c0 = cnt[2 * tid], c1 = cnt[2 * tid + 1];
        uint32_t tot;
        const uint32_t ex = mj_block_scan<MJ_THREADS>(c0 + c1, tid, s_w, &tot);
        ((uint32_t *)m.segoff)[(int64_t)t * (MJ_P / 2) + tid] = ex | ((ex + c0) << 16);
        cnt[2 * tid] = ex;
        cnt[2 * tid + 1] = ex + c0;
        uint64_t e16[MJ_CAP / MJ_THREADS];
#pragma unroll
        for (int i = 0; i < MJ_CAP / MJ_THREADS; i++) {
            const uint32_t idx = tid + i * MJ_THREADS;
            e16[i] = idx < n ? buf[idx] : 0ull;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MJ_CAP / MJ_THREADS; i++) {
            const uint32_t idx = tid + i * MJ_THREADS;
            if (idx < n) buf[atomicAdd(&cnt[e16[i] >> MJ_PSH], 1u)] = e16[i];
        }
        __syncthreads();
        uint4 *dst = (uint4 *)(m.ent + (int64_t)t * MJ_CAP);
        for (uint32_t i = tid; i < (n + 1) / 2; i += MJ_THREADS) dst[i] = ((const uint4 *)buf)[i];
        if (tid == 0) m.tile_n[t] = n;
        __syncthreads();
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

// ------------------------------------------------------------------------------------ probe
namespace {
// the lookup of one k-mer with the rules of seed_item's flush() (dh_kernels.hip): the bucket's only entry from the fat
// directory word, or the walk of a bucket with the -t cap per orientation class
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
            if (same || pal) emit(f.y, 0);
            if (!same || pal) emit(f.y, 1);
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
    for (uint32_t t = ss; t < ee; t++) {
        const ulonglong2 en = ix.ent[t];
        if ((en.x & ~ORI) != key) continue;
        const bool same = (en.x & ORI) == bori;
        if (dof && (same || pal)) emit(en.y, 0);
        if (dor && (!same || pal)) emit(en.y, 1);
    }
}
}  // namespace

__global__ void __launch_bounds__(MJ_PROBE_THREADS)
k_mj_probe(IndexView ix, DhOpts o, MjView m)
{
    __shared__ uint32_t bm[1 << (MJ_MAXBITS - MJ_PBITS - 5)];
    __shared__ int32_t s_item;
    const int tid = threadIdx.x, lane = tid & (LANES - 1), wave = tid / LANES;
    constexpr int NW = MJ_PROBE_THREADS / LANES;
    const int k = m.k, remsh = 2 * k - MJ_PBITS, bshift = 2 * k - m.nbbits;
    const uint64_t remmask = (1ull << remsh) - 1;
    const int32_t slice_words = 1 << (m.nbbits - MJ_PBITS - 5);
    const int32_t nb = m.ntiles_pad / MJ_BATCH;
    const int32_t nitems_q = (MJ_P / 8) * MJ_SLICES;
    const uint32_t xcc = mj_xcc_id();
    uint64_t page_base = 0;
    uint32_t fill = MJ_PAGE;  // the wavefront has no page yet
    int32_t curp = -1;
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
            const int32_t b0 = (int32_t)((int64_t)nb * sl / MJ_SLICES), b1 = (int32_t)((int64_t)nb * (sl + 1) / MJ_SLICES);
            const uint64_t keytop = (uint64_t)p << remsh;
            for (int32_t bb = b0 + wave; bb < b1; bb += NW) {
                const int32_t t = bb * MJ_BATCH + lane;
                const uint32_t sd = m.seg[(int64_t)p * m.ntiles_pad + t];  // (tiles beyond ntiles: zero)
                const uint32_t cnt = t < m.ntiles ? (sd & 0xFFFFu) : 0u;
                const uint64_t *ptr = m.ent + (int64_t)t * MJ_CAP + (sd >> 16);
                const uint64_t posbase = (uint64_t)(lane & (MJ_GROUP - 1)) * (uint64_t)m.tb;
                uint32_t nh = 0;
                uint64_t h0 = 0, h1 = 0;
                uint64_t *wr = nullptr;  // second pass: where hit number nh goes
                auto probe_entry = [&](uint64_t e) {
                    const uint64_t rem = (e >> MJ_REMSH) & remmask;
                    const uint32_t bit = (uint32_t)(rem >> bshift);
                    if (!((bm[bit >> 5] >> (bit & 31)) & 1u)) return;
                    const uint64_t posg = posbase + (e & ((1ull << MJ_POSBITS) - 1));
                    mj_lookup(ix, o, keytop | rem, (e & MJ_ORI_BIT) != 0, (e & MJ_PAL_BIT) != 0, [&](uint64_t v, int strand) {
                        if (!(o.strands & (1 << strand))) return;
                        const uint64_t h = ((uint64_t)strand << 63) | ((v & ((1ull << 40) - 1)) << 23) | posg;
                        if (wr) {
                            if (nh >= 2) wr[nh] = h;
                        } else if (nh == 0)
                            h0 = h;
                        else if (nh == 1)
                            h1 = h;
                        nh++;
                    });
                };
                // four entries of the lane's segment per round (independent loads)
                uint32_t cmax = cnt;
                for (int off = LANES / 2; off > 0; off >>= 1) cmax = max(cmax, (uint32_t)__shfl_xor((int)cmax, off, LANES));
                for (uint32_t i0 = 0; i0 < cmax; i0 += 4) {
                    uint64_t e[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) e[u] = i0 + u < cnt ? mj_load8(ptr + i0 + u) : 0ull;
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        if (i0 + u < cnt) probe_entry(e[u]);
                }
                // the batch's hits in lane (= tile) order: one range per tile group
                uint32_t incl = nh;
#pragma unroll
                for (int off = 1; off < LANES; off <<= 1) {
                    const uint32_t up = __shfl_up(incl, off, LANES);
                    if (lane >= off) incl += up;
                }
                const uint32_t tot = __shfl(incl, LANES - 1, LANES);
                const uint32_t pre = incl - nh;
                bool ok = true;
                if (tot > 0 && fill + tot > MJ_PAGE) {
                    uint32_t pg = 0;
                    if (lane == 0) pg = atomicAdd(&m.ctr[8], 1u);
                    pg = __shfl(pg, 0, LANES);
                    if (tot > MJ_PAGE || pg >= (uint32_t)m.npages) {
                        // (the page counter keeps counting: the host sizes the pool by it and runs the chunk again)
                        if (lane == 0) atomicOr(m.status, tot > MJ_PAGE ? DH_ST_MJ_OVERFLOW : DH_ST_MJ_POOL);
                        ok = false;
                    } else {
                        page_base = (uint64_t)pg * MJ_PAGE;
                        fill = 0;
                    }
                }
                const uint64_t first = page_base + fill + pre;
                if (ok && nh > 0) {
                    uint64_t *dst = m.hits + first;
                    dst[0] = h0;
                    if (nh > 1) dst[1] = h1;
                    if (nh > 2) {  // rare (a k-mer of a repeat): the segment once more, hits from the third on
                        wr = dst;
                        nh = 0;
                        for (uint32_t i = 0; i < cnt; i++) probe_entry(mj_load8(ptr + i));
                    }
                }
                // (shuffles need every lane: the next group's prefix is fetched by all, used by the group heads)
                const uint32_t pre_next = __shfl(pre, (lane + MJ_GROUP) & (LANES - 1), LANES);
                if ((lane & (MJ_GROUP - 1)) == 0) {
                    const uint32_t cg = ok ? ((lane + MJ_GROUP < LANES ? pre_next : tot) - pre) : 0u;
                    const int64_t g = (int64_t)bb * (MJ_BATCH / MJ_GROUP) + lane / MJ_GROUP;
                    m.hseg[g * MJ_P + p] = ((unsigned long long)first << 24) | cg;
                }
                if (ok) fill += tot;
            }
        }
    }
}

// ------------------------------------------------------------------------------------ regroup by read
// Two passes over the group's hits (no LDS buffer bounds their number: a group of low-error reads holds several times
// the hits of one at 13 % error): count per read, reserve the group's range of rhits, then finish and scatter.
__global__ void __launch_bounds__(MJ_THREADS)
k_mj_regroup(DbView B, IndexView ix, MjView m)
{
    __shared__ uint32_t hoff[MJ_P + 1];
    __shared__ uint64_t hfirst[MJ_P];
    __shared__ int32_t rsl[MJ_RG_READS + 2];
    __shared__ uint32_t rcnt[MJ_RG_READS], rcur[MJ_RG_READS];
    __shared__ uint32_t s_w[MJ_THREADS / LANES];
    __shared__ int32_t s_ra, s_nrd;
    __shared__ unsigned long long s_base;
    const int tid = threadIdx.x;
    const int32_t g = blockIdx.x;
    const int64_t tbg = (int64_t)m.tb * MJ_GROUP;
    const int64_t gb0 = m.c0 + (int64_t)g * tbg;
    if (gb0 >= m.c1) return;
    const int64_t gb1 = std::min<int64_t>(m.c1, gb0 + tbg);
    // ---- the group's hit lists, one per partition
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
        // the read that holds the group's first base, and the reads that begin inside the group
        int32_t lo = m.r0, hi = m.r1 - 1;  // largest r with off[r] <= gb0
        while (lo < hi) {
            const int32_t mid = (lo + hi + 1) >> 1;
            if (B.off[mid] <= gb0)
                lo = mid;
            else
                hi = mid - 1;
        }
        s_ra = lo;
        s_nrd = 0x7FFFFFFF;
    }
    for (int i = tid; i < MJ_RG_READS; i += MJ_THREADS) rcnt[i] = 0;
    __syncthreads();
    const int32_t ra = s_ra;
    for (int32_t i = tid;; i += MJ_THREADS) {
        const int32_t r = ra + i;
        if (r > m.r1) break;
        const int64_t o = B.off[r] - gb0;
        if (i <= MJ_RG_READS + 1) rsl[i] = (int32_t)std::min<int64_t>(o, 0x7FFFFFF0);
        if (o >= gb1 - gb0 || r == m.r1) {  // the first boundary at or behind the group's end closes its last read
            atomicMin(&s_nrd, i);
            break;
        }
    }
    __syncthreads();
    const int32_t nrd = s_nrd;  // reads ra .. ra + nrd - 1 overlap the group; rsl[0 .. nrd] their boundaries
    if (nrd > MJ_RG_READS) {
        if (tid == 0) atomicOr(m.status, DH_ST_MJ_OVERFLOW);
        return;
    }
    if (n == 0) return;  // (the reads' rows of this group stay zero)
    auto fetch = [&](uint32_t e) {  // hit e of the group: the list of the last partition p with hoff[p] <= e
        int32_t lo = 0, hi = MJ_P - 1;
        while (lo < hi) {
            const int32_t mid = (lo + hi + 1) >> 1;
            if (hoff[mid] <= e)
                lo = mid;
            else
                hi = mid - 1;
        }
        return m.hits[hfirst[lo] + (e - hoff[lo])];
    };
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
    for (uint32_t e = tid; e < n; e += MJ_THREADS) atomicAdd(&rcnt[read_of((int32_t)(fetch(e) & 0x7FFFFFull))], 1u);
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
        const unsigned long long base = atomicAdd((unsigned long long *)(m.ctr + 10), (unsigned long long)n);
        if (base + n > (unsigned long long)m.rcap) atomicOr(m.status, DH_ST_MJ_POOL);
        s_base = base;
    }
    __syncthreads();
    const unsigned long long base = s_base;
    if (base + n > (unsigned long long)m.rcap) return;
#pragma unroll
    for (int u = 0; u < MJ_RG_READS / MJ_THREADS; u++) {
        const int32_t i = tid * (MJ_RG_READS / MJ_THREADS) + u;
        rcur[i] = rex;
        if (i < nrd) {
            const int32_t r = ra + i;
            const int64_t gfirst = (B.off[r] - m.c0) / tbg;
            const int64_t j = (int64_t)g - gfirst;
            if (j >= 0 && j < m.nseg)
                m.segtab[(int64_t)(r - m.r0) * m.nseg + j] = ((base + rex) << 24) | rc4[u];
            else if (rc4[u])
                atomicOr(m.status, DH_ST_MJ_OVERFLOW);
        }
        rex += rc4[u];
    }
    __syncthreads();
    // ---- the hit as the seed filter wants it: strand << 63 | diagonal << 24 | position on the oriented read
    const int k = m.k;
    for (uint32_t e = tid; e < n; e += MJ_THREADS) {
        const uint64_t h = fetch(e);
        const int32_t posg = (int32_t)(h & 0x7FFFFFull);
        const int32_t strand = (int32_t)(h >> 63);
        const int64_t gv = (int64_t)((h >> 23) & ((1ull << 40) - 1));
        const int32_t i = read_of(posg);
        const int32_t q = posg - rsl[i];
        const int32_t blen = rsl[i + 1] - rsl[i];
        const int32_t qs = strand ? blen - k - q : q;
        const int64_t D = gv + ix.sepv - qs;
        m.rhits[base + atomicAdd(&rcur[i], 1u)] = ((uint64_t)strand << 63) | ((uint64_t)D << 24) | (uint32_t)qs;
    }
}

// ------------------------------------------------------------------------------------ launchers
extern "C" {

void dhk_mj_bitmap(hipStream_t st, const ulonglong2 *ent, int64_t n, int32_t k, int32_t nbbits, uint32_t *bm)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_mj_bitmap, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ent, n, 2 * k - nbbits, bm);
}

void dhk_mj_run(hipStream_t st, DbView B, IndexView ix, DhOpts o, MjView m, int32_t ncu)
{
    (void)hipMemsetAsync(m.ctr, 0, 16 * sizeof(uint32_t), st);
    hipLaunchKernelGGL(k_mj_part, dim3((unsigned)std::min<int64_t>(m.ntiles, (int64_t)ncu * 2 * 4)), dim3(MJ_THREADS), 0, st, B, m);
    hipLaunchKernelGGL(k_mj_transpose, dim3((unsigned)(m.ntiles_pad / 32)), dim3(256), 0, st, m);
    hipLaunchKernelGGL(k_mj_probe, dim3((unsigned)ncu), dim3(MJ_PROBE_THREADS), 0, st, ix, o, m);
    hipLaunchKernelGGL(k_mj_regroup, dim3((unsigned)m.ngroups), dim3(MJ_THREADS), 0, st, B, ix, m);
}
}
