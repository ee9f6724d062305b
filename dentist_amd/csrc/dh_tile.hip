// dh_tile.hip -- k_tile: DH-2 on gfx950, one alignment per lane (dh_tile.h holds the lane code).
//
// A wavefront = 64 independent alignments.  The only wave-level structure is time: every lane that is
// extending runs one trace tile per round, column by column in lock step (the column counter and the
// funnel-shift amounts are scalars), lanes that need bookkeeping (next candidate, end of an extension,
// a record, the next read) do it between rounds, and a lane whose read is done pulls the next one from
// the queue -- no lane waits for another lane's alignment.  Per column and lane: two 64-bit plane
// windows cut from the tile's registers with four v_alignbit, the match vector, Hyyro's eleven
// bit-vector operations on one 64-bit word and the score of the band's bottom cell; per tile: eight
// 8-byte pairs of the plane-packed read and nine dwords of the 2-bit packed contig, one 4-byte trace
// pair out.  Nothing is staged in LDS -- the working set of a lane is its registers.
#include <hip/hip_runtime.h>
#include <stdio.h>

#include "dh_tile.h"

using namespace dhtile;

#define TILE_WAVES_PER_SIMD 4

__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

__device__ __forceinline__ int32_t wave_max_i32(int32_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int32_t o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

#ifdef DH_SEED_PROF
// development: where a wavefront's time goes -- [0] bookkeeping passes (wall), [1] tile set-up, [2] column loops,
// [3] tile ends, [4] passes, [5] rounds, [6] lanes in the rounds, [7] lane-passes (lanes wanting a pass, summed)
__device__ unsigned long long g_tile_prof[20];
__device__ unsigned long long g_tile_wt[8] = {~0ull, 0, 0, 0, 0, 0, 0, 0};  // wall clock (100 MHz): min start, max end, sum of lifetimes, waves, last unit fetch (max)
__device__ unsigned long long g_tile_hist[8];  // lane-rounds by tile width: <= 8, <= 32, <= 64, < 126, >= 126; [5] first tiles, [6] partial last tiles, [7] sum of T
#define TP(i) { const unsigned long long t_ = clock64(); pacc_[i] += t_ - tp_; tp_ = t_; }
#define TPC(i, v) pacc_[i] += (unsigned long long)(v);
extern "C" void dhk_tile_prof_dump()
{
    unsigned long long h[20];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_tile_prof), sizeof(h));
    if (h[5]) fprintf(stderr, "[tile prof] columns a wavefront runs per round (max over its lanes): %.1f\n", (double)h[18] / h[5]);
    if (h[5])
        fprintf(stderr, "[tile prof] ext_end sections (G cycles, max lane per wave): rev->fwd %.2f region %.2f finish_pairs %.2f emit %.2f second %.2f tail %.2f\n",
                h[12] / 1e9, h[13] / 1e9, h[14] / 1e9, h[15] / 1e9, h[16] / 1e9, h[17] / 1e9);
    if (h[5])
        fprintf(stderr, "[tile prof] pass split (G wave cycles): ext_end %.2f next_cand %.2f fetch %.2f next_cand2 %.2f\n", h[8] / 1e9, h[9] / 1e9, h[10] / 1e9, h[11] / 1e9);
    if (h[5])
        fprintf(stderr, "[tile prof] wave cycles (G, summed over waves): book %.2f setup %.2f columns %.2f ends %.2f; passes %llu rounds %llu "
                        "lanes/round %.1f lanes/pass %.1f; per pass %.0f cycles, per round %.0f cycles\n",
                h[0] / 1e9, h[1] / 1e9, h[2] / 1e9, h[3] / 1e9, h[4], h[5], (double)h[6] / h[5], (double)h[7] / (h[4] ? h[4] : 1),
                (double)h[0] / (h[4] ? h[4] : 1), (double)(h[1] + h[2] + h[3]) / h[5]);
    {
        unsigned long long hh[8];
        (void)hipMemcpyFromSymbol(hh, HIP_SYMBOL(g_tile_hist), sizeof(hh));
        const double tot = (double)(hh[0] + hh[1] + hh[2] + hh[3] + hh[4]);
        if (tot > 0) fprintf(stderr, "[tile prof] lane-rounds by columns: <=8 %.3f <=32 %.3f <=64 %.3f <126 %.3f >=126 %.3f; first tiles %.3f, cols < T %.3f, mean T %.1f\n", hh[0] / tot, hh[1] / tot, hh[2] / tot, hh[3] / tot, hh[4] / tot, hh[5] / tot, hh[6] / tot, hh[7] / tot);
        unsigned long long zz[8] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tile_hist), zz, sizeof(zz));
    }
    {
        unsigned long long wt[8];
        (void)hipMemcpyFromSymbol(wt, HIP_SYMBOL(g_tile_wt), sizeof(wt));
        if (wt[3]) fprintf(stderr, "[tile prof] last launch: %llu wavefronts, kernel %.2f ms, mean wavefront lifetime %.2f ms, last unit handed out at %.2f ms\n", wt[3], (wt[1] - wt[0]) / 1e5, (double)wt[2] / wt[3] / 1e5, (wt[4] - wt[0]) / 1e5);
        unsigned long long zz[8] = {~0ull, 0, 0, 0, 0, 0, 0, 0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tile_wt), zz, sizeof(zz));
    }
    unsigned long long z[20] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tile_prof), z, sizeof(z));
}
#else
#define TP(i)
#define TPC(i, v)
#endif

// The column step of dh_tile.h:tile_col as the kernel runs it -- the same recurrence, two instruction-count changes:
//  * the A base arrives as the COMPLEMENTED packed word shifted to the column (nxw: bits 0..1 = ~base): one sign-extending
//    bit-field extract per bit gives ~x0 / ~x1 as 32-bit masks that serve both halves of a 64-bit vector, and
//    ~((p0 ^ x0) | (p1 ^ x1)) == (p0 ^ ~x0) & (p1 ^ ~x1)  (was: and, bfe, two moves, two 64-bit adds per column);
//  * LV = false for the columns behind the first 32: the rows-before-the-origin mask lv = ~0 << (W/2 + 1 - c) is all
//    ones from column W/2 + 1 on, so neither its shift nor the two ANDs are issued there.
template <bool TAN, bool LV, int WB>
__device__ __forceinline__ void tile_col_dev(TileT<WB> &t, typename BandVec<WB>::U p0, typename BandVec<WB>::U p1, uint32_t nxw)
{
    typedef typename BandVec<WB>::U V;
    typedef typename BandVec<WB>::S SV;
    const uint32_t n0 = (uint32_t)(((int32_t)(nxw << 31)) >> 31), n1 = (uint32_t)(((int32_t)(nxw << 30)) >> 31);
    const V nx0 = WB == 64 ? (V)(((uint64_t)n0 << 32) | n0) : (V)n0, nx1 = WB == 64 ? (V)(((uint64_t)n1 << 32) | n1) : (V)n1;
    V eq = (V)((p0 ^ nx0) & (p1 ^ nx1));
    if (LV) {
        t.lv = (V)((SV)t.lv >> 1);
        eq &= t.lv;
    }
    if (TAN) eq &= t.dm;
    t.z -= 1;
    t.wild = (V)((V)((SV)t.wild >> 1) | (V)((V)((uint32_t)t.z >> 31) << (WB - 1)));
    const V Eq = (V)(eq | t.wild);
    const V Pv = t.Pv, Mv = t.Mv;
    const V D0 = (V)((((V)((V)(Eq & Pv) + Pv)) ^ Pv) | Eq | Mv);
    const V HP = (V)(Mv | (V)~(D0 | Pv)), HN = (V)(Pv & D0);
    const V Xv = (V)(D0 >> 1);
    t.Pv = (V)(HN | (V)~(Xv | HP));
    t.Mv = (V)(HP & Xv);
    t.dbot += 1 - (int32_t)(D0 >> (WB - 1));
}

// The scan of the last column (dh_tile.h:tile_scan) as the kernel runs it: the same key -- min over the eligible rows of
// D << 16 | |row - diagonal| << 8 | W-1-i -- without a branch per row: the eligible rows [imin, imax] as a bit mask whose
// complement ORs ~0 into the keys of the others, the vertical deltas by bit-field extracts of compile-time positions
// (7 instructions per row, no exec-mask round trip; was 11 and a branch).
template <int WB>
__device__ __forceinline__ uint32_t tile_scan_dev(const TileT<WB> &t)
{
    typedef typename BandVec<WB>::U V;
    constexpr int W = WB;
    const int32_t imin = W / 2 - t.cols, imax = t.bnr + W / 2;
    const int32_t lo = imin > 0 ? imin : 0, hi = imax < W - 1 ? imax : W - 1;  // (lo <= W/2 - 1 <= hi: cols >= 1, bnr >= 0)
    const V ne = (V) ~((V)((V)(~(V)0 << lo)) & (V)(~(V)0 >> (W - 1 - hi)));
    uint32_t key = 0xFFFFFFFFu;
    int32_t d = t.dbot;
#pragma unroll
    for (int i = W - 1; i >= 0; i--) {
        const int sh = i & 31;
        const uint32_t mw = (uint32_t)(WB == 64 && i >= 32 ? (uint64_t)t.Mv >> 32 : (uint64_t)t.Mv);
        const uint32_t pw = (uint32_t)(WB == 64 && i >= 32 ? (uint64_t)t.Pv >> 32 : (uint64_t)t.Pv);
        const uint32_t nw = (uint32_t)(WB == 64 && i >= 32 ? (uint64_t)ne >> 32 : (uint64_t)ne);
        if (i < W - 1) d += (int32_t)((mw >> sh) & 1u) + (((int32_t)(pw << (31 - sh))) >> 31);
        const uint32_t off = (uint32_t)(i >= W / 2 ? i - W / 2 : W / 2 - i);
        const uint32_t kk = ((uint32_t)d << 16) | (off << 8) | (uint32_t)(W - 1 - i) | (uint32_t)(((int32_t)(nw << (31 - sh))) >> 31);
        key = kk < key ? kk : key;
    }
    return key;
}

template <bool TAN, int WB>
__global__ void __launch_bounds__(64, TILE_WAVES_PER_SIMD) k_tile(Params P)
{
    __shared__ uint32_t s_q[NTW][64];
#ifdef DH_SEED_PROF
    unsigned long long pacc_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tp_ = clock64(), pcm_ = 0;
    const unsigned long long wt0_ = wall_clock64();
    unsigned long long wtf_ = wt0_;
    unsigned lh_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    Lane l;
    const int32_t slot = (int32_t)(blockIdx.x * 64 + threadIdx.x);
    lane_init(l, slot, P.cold + slot);
#ifdef DH_SEED_PROF
    for (int i = 0; i < 6; i++) l.pt[i] = 0;
#endif
    TileT<WB> t;
    typedef typename BandVec<WB>::U V;
    uint32_t wq_base = 0, wq_rem = 0;  // the wavefront's batch of work units (uniform)
    for (;;) {
        // ---- bookkeeping until every lane extends or is out of work.  A pass costs a handful of dependent memory
        // round trips whatever the number of lanes in it, so it waits until P.book_min lanes want one (or nothing
        // else can run): short extensions (pile-up reads) would otherwise pay a pass on every round
        const unsigned long long want = __builtin_amdgcn_ballot_w64(l.st != L_RUN && l.st != L_DONE);
        if (want != 0ull) { TPC(4, 1) TPC(7, __builtin_popcountll(want)) }
#ifdef DH_SEED_PROF
        tp_ = clock64();
#endif
        if (want != 0ull && (__builtin_popcountll(want) >= P.book_min || !wave_any(l.st == L_RUN)))
        // the states are visited in the order a lane moves through them (end of an extension -> next candidate ->
        // next work unit -> its first candidate), so that one pass takes a lane all the way to its next extension:
        // as exclusive branches this chain took four passes, each paying the round trips of every branch in it
        while (wave_any(l.st != L_RUN && l.st != L_DONE)) {
#ifdef DH_SEED_PROF
            unsigned long long tq_ = clock64();
#define TQ(i) { const unsigned long long t_ = clock64(); pacc_[i] += t_ - tq_; tq_ = t_; }
#else
#define TQ(i)
#endif
            if (l.st == L_EXT_END) lane_ext_end(l, P);
            TQ(8)
            if (l.st == L_CAND) lane_next_cand(l, P);
            TQ(9)
            {
                // The queue is ONE address: a returning atomic on it costs ~11 ns chip-wide whatever else happens (92 M/s
                // measured, LABNOTES 10), and a symmetric launch of uncapped pile-ups hands out 12.6 M units -- one atomic per
                // pass and wavefront (the lanes of a pass that want a unit share it) was ~10 M of them, a tenth of a second
                // of the atomic unit's time with every wavefront's other loads queued behind its own.  A wavefront therefore
                // takes P.qbatch units per atomic and deals them to its lanes from [wq_base, wq_base + wq_rem) (uniform
                // values); what a pass needs beyond the rest of the batch comes from the next one.  The units a wavefront
                // holds at the end are at most qbatch - 1: one more unit per lane.
                const unsigned long long fm = __builtin_amdgcn_ballot_w64(l.st == L_FETCH);
                if (fm != 0ull) {
                    const uint32_t need = (uint32_t)__builtin_popcountll(fm);
                    uint32_t nb = 0, take = 0;
                    if (need > wq_rem) {
                        const uint32_t more = need - wq_rem, qb = (uint32_t)P.qbatch;
                        take = qb > more ? qb : more;
                        const int first = __builtin_ctzll(fm);
                        uint32_t b = 0;
                        if ((int)threadIdx.x == first) b = atomicAdd(P.queue, take);
                        nb = (uint32_t)__builtin_amdgcn_readlane((int)b, first);  // (uniform: the batch lives in scalar registers)
#ifdef DH_SEED_PROF
                        if (nb < (P.units ? *P.nunits : (uint32_t)P.nitems)) wtf_ = wall_clock64();
#endif
                    }
                    if (l.st == L_FETCH) {
                        const uint32_t rank = (uint32_t)__builtin_popcountll(fm & ((1ull << threadIdx.x) - 1ull));
                        const int32_t it = (int32_t)(rank < wq_rem ? wq_base + rank : nb + (rank - wq_rem));
                        if (it >= (P.units ? (int32_t)*P.nunits : P.nitems))
                            l.st = L_DONE;
                        else
                            lane_fetch(l, P, it);
                    }
                    if (need > wq_rem) {
                        wq_base = nb + (need - wq_rem);
                        wq_rem = take - (need - wq_rem);
                    } else {
                        wq_base += need;
                        wq_rem -= need;
                    }
                }
            }
            TQ(10)
            if (l.st == L_CAND) lane_next_cand(l, P);
            TQ(11)
        }
        const bool run = l.st == L_RUN;
        TP(0)
        if (!wave_any(run)) break;
        TPC(5, 1) TPC(6, __builtin_popcountll(__builtin_amdgcn_ballot_w64(run)))
        // ---- one tile of every extending lane.  Its sequence words go to LDS, [word][lane]: the column loop
        // keeps three words of either plane and two of A in registers per block of 32 columns
        if (run) {
            uint32_t q[NTW];
            tile_setup(l, P, t, q);
#pragma unroll
            for (int i = 0; i < NTW; i++) s_q[i][threadIdx.x] = q[i];
        }
        const int32_t cmax = wave_max_i32(run ? t.cols : 0);
#ifdef DH_SEED_PROF
        if (run) {
            lh_[t.cols <= 8 ? 0 : (t.cols <= 32 ? 1 : (t.cols <= 64 ? 2 : (t.cols < 126 ? 3 : 4)))]++;
            if (l.e.ntp == 0) lh_[5]++;
            if (t.cols < t.T) lh_[6]++;
            lh_[7] += (unsigned)t.T;
        }
        pcm_ += (unsigned long long)cmax;
#endif
        TP(1)
        for (int32_t blk = 0; 32 * blk < cmax; blk++) {
            const uint32_t a0 = s_q[blk][threadIdx.x], a1 = s_q[blk + 1][threadIdx.x], a2 = WB == 64 ? s_q[blk + 2][threadIdx.x] : 0u;
            const uint32_t b0 = s_q[NQ + blk][threadIdx.x], b1 = s_q[NQ + blk + 1][threadIdx.x], b2 = WB == 64 ? s_q[NQ + blk + 2][threadIdx.x] : 0u;
            // (complemented: tile_col_dev extracts ~x0 / ~x1)
            const uint64_t nab = ~((uint64_t)s_q[2 * NQ + 2 * blk][threadIdx.x] | ((uint64_t)s_q[2 * NQ + 2 * blk + 1][threadIdx.x] << 32));
            const int32_t nsh = cmax - 32 * blk < 32 ? cmax - 32 * blk : 32;
#define DH_TILE_COLUMNS(LV_)                                                                                                      \
    for (int32_t sh = 0; sh < nsh; sh++) {                                                                                        \
        const int32_t c = 32 * blk + sh + 1;                                                                                      \
        if (run && c <= t.cols) {                                                                                                 \
            V p0, p1;                                                                                                             \
            if (WB == 64) {                                                                                                       \
                p0 = (V)((uint64_t)funnel(a1, a0, (uint32_t)sh) | ((uint64_t)funnel(a2, a1, (uint32_t)sh) << 32));               \
                p1 = (V)((uint64_t)funnel(b1, b0, (uint32_t)sh) | ((uint64_t)funnel(b2, b1, (uint32_t)sh) << 32));               \
            } else {                                                                                                              \
                p0 = (V)funnel(a1, a0, (uint32_t)sh);                                                                             \
                p1 = (V)funnel(b1, b0, (uint32_t)sh);                                                                             \
            }                                                                                                                     \
            tile_col_dev<TAN, LV_, WB>(t, p0, p1, (uint32_t)(nab >> (2 * sh)));                                                   \
        }                                                                                                                         \
    }
            if (blk == 0) {
                DH_TILE_COLUMNS(true)
            } else {
                DH_TILE_COLUMNS(false)
            }
#undef DH_TILE_COLUMNS
        }
        TP(2)
        if (run) tile_end_key<WB>(l, P, t, tile_scan_dev<WB>(t));
        TP(3)
    }
#ifdef DH_SEED_PROF
    if (threadIdx.x == 0)
        for (int i = 0; i < 12; i++) atomicAdd(&g_tile_prof[i], pacc_[i]);
    if (threadIdx.x == 0) atomicAdd(&g_tile_prof[18], pcm_);
    if (threadIdx.x == 0) {
        const unsigned long long wt1_ = wall_clock64();
        atomicMin(&g_tile_wt[0], wt0_);
        atomicMax(&g_tile_wt[1], wt1_);
        atomicAdd(&g_tile_wt[2], wt1_ - wt0_);
        atomicAdd(&g_tile_wt[3], 1ull);
        atomicMax(&g_tile_wt[4], wtf_);
    }
    for (int i = 0; i < 8; i++) if (lh_[i]) atomicAdd(&g_tile_hist[i], (unsigned long long)lh_[i]);
    for (int i = 0; i < 6; i++) {  // sections of lane_ext_end: the lanes of a pass run them together, so the largest lane total ~ the wavefront's
        unsigned long long v = l.pt[i];
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o2 = __shfl_xor(v, off, 64);
            v = o2 > v ? o2 : v;
        }
        if (threadIdx.x == 0) atomicAdd(&g_tile_prof[12 + i], v);
    }
#endif
    if (l.cells) atomicAdd(&P.counters[0], (unsigned long long)l.cells);
    if (l.naln) atomicAdd(&P.counters[1], (unsigned long long)l.naln);
    if (l.err) atomicOr(P.status, l.err);
}

// plane-packed copy from the 2-bit packed one, in place: word w (32 bases, base b at bits 2b) becomes
// (low bits of the 32 bases, high bits of the 32 bases)
__device__ __forceinline__ uint32_t squeeze_even(uint64_t x)
{
    x &= 0x5555555555555555ull;
    x = (x | (x >> 1)) & 0x3333333333333333ull;
    x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
    x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
    x = (x | (x >> 16)) & 0x00000000FFFFFFFFull;
    return (uint32_t)x;
}
__global__ void __launch_bounds__(256) k_pk2planes(uint64_t *__restrict__ w, int64_t nwords)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nwords) return;
    const uint64_t x = w[i];
    w[i] = (uint64_t)squeeze_even(x) | ((uint64_t)squeeze_even(x >> 1) << 32);
}

// one thread per item: its candidates (grouped by A read by the seed kernel) become units, one per group -- only the
// candidates of one group depend on each other (a candidate inside an aligned region of the same A read is skipped).
// The cap of MAXREG attempted alignments per item (dh_tile.h: lane_next_cand; oracle/align.c) couples the groups of an item
// with MORE than MAXREG candidates; such an item used to stay one unit -- up to 64 alignments one after the other in one lane,
// ~90 ms at 166 reads per pile-up: the queue of a symmetric launch ran dry after 75 ms and the launch took 170 (round 6).  It
// is split exactly.  A group's first candidate is never skipped, so a candidate ALONE in its group means exactly one attempted
// alignment; if its index is below MAXREG fewer than MAXREG attempts precede it, whatever they were: it is always aligned, a
// unit of its own.  Everything else -- groups of several candidates (their number of attempts is not known beforehand), and
// whatever lies at index MAXREG or behind -- is ONE unit, the `rest`: it walks the item's candidates in order from its first
// one, counts one attempt for every lone candidate below index MAXREG it passes, and applies the cap to its own.
__global__ void __launch_bounds__(256)
k_units_fat(const DhCand *__restrict__ cand, const int32_t *__restrict__ ncand, const int32_t *__restrict__ candoff, int32_t item0,
            int32_t nitems, int32_t max_cand, const int64_t *__restrict__ aoff, const int64_t *__restrict__ boff,
            Unit *__restrict__ units, uint32_t *__restrict__ nunits)
{
    const int32_t it = blockIdx.x * blockDim.x + threadIdx.x;
    if (it >= nitems) return;
    const int32_t item = item0 + it;
    const int32_t nc = max(ncand[item], 0);
    if (nc == 0) return;
    const DhCand *cl = cand + (int64_t)item * max_cand;
    const int64_t bo = boff[item >> 1];
    const int32_t blen = (int32_t)(boff[(item >> 1) + 1] - bo);
    auto emit = [&](int32_t c0, int32_t c1, int32_t nd0, int32_t rest) {
        Unit u;
        u.it = it;
        u.c0 = c0;
        u.c1 = c1;
        u.blen = blen;
        u.bo = bo;
        u.aseq = cl[c0].aseq;
        u.apos = cl[c0].apos;
        u.bpos = cl[c0].bpos;
        u.ao = aoff[u.aseq];
        u.alen = (int32_t)(aoff[u.aseq + 1] - u.ao);
        u.cbase = candoff[item];
        u.nd0 = nd0;
        u.rest = rest;
        u.pad_[0] = 0;
        units[atomicAdd(nunits, 1u)] = u;
    };
    if (nc <= MAXREG) {  // the cap cannot bind: every group on its own
        int32_t c0 = 0;
        while (c0 < nc) {
            int32_t c1 = c0 + 1;
            while (c1 < nc && cl[c1].aseq == cl[c0].aseq) c1++;
            emit(c0, c1, 0, 0);
            c0 = c1;
        }
        return;
    }
    int32_t first_rest = -1;
    for (int32_t c = 0; c < nc; c++) {
        const bool alone = (c == 0 || cl[c - 1].aseq != cl[c].aseq) && (c + 1 >= nc || cl[c + 1].aseq != cl[c].aseq);
        if (alone && c < MAXREG)
            emit(c, c + 1, 0, 0);
        else if (first_rest < 0)
            first_rest = c;
    }
    // (every candidate before first_rest is a lone one below MAXREG: first_rest attempts precede it)
    if (first_rest >= 0 && first_rest < MAXREG) emit(first_rest, nc, first_rest, 1);
}

// ---- symmetric launches: the records sit in candidate-indexed slots (dh_tile.h, Params.candoff); three streaming
// kernels group them by A read for the compaction -- their atomics run at full occupancy, no lane of k_tile waits on one.
__global__ void __launch_bounds__(256) k_cand_counts(const int32_t *__restrict__ ncand, int32_t nitems, uint32_t *__restrict__ out)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > nitems) return;
    out[i] = i < nitems ? (uint32_t)max(ncand[i], 0) : 0u;  // out[nitems] receives the total from the exclusive scan
}
// item (relative to item0) that lists a record: the A read's item of the record's strand
__device__ __forceinline__ int32_t rec_item(const DhLa &la, int32_t item0) { return 2 * la.aread + (int32_t)(la.flags & 1u) - item0; }
__global__ void __launch_bounds__(256)
k_rec_count(const DhLa *__restrict__ slots, int64_t nslots, int32_t item0, uint32_t *__restrict__ nla, uint32_t *__restrict__ ntr)
{
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nslots) return;
    const DhLa la = slots[s];
    if (la.pad != 1) return;
    const int32_t it = rec_item(la, item0);
    atomicAdd(&nla[it], 1u);
    atomicAdd(&ntr[it], (uint32_t)la.tlen);
}
__global__ void __launch_bounds__(256)
k_rec_scatter(const DhLa *__restrict__ slots, int64_t nslots, int32_t item0, const uint32_t *__restrict__ la_off,
              uint32_t *__restrict__ cursor, int32_t *__restrict__ list)
{
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nslots) return;
    const DhLa la = slots[s];
    if (la.pad != 1) return;
    const int32_t it = rec_item(la, item0);
    list[la_off[it] + atomicAdd(&cursor[it], 1u)] = (int32_t)s;
}
// a wavefront per item: its records (slot ids in list[la_off[it] ..)) ordered by (bread, abpos, bbpos, aepos, slot) --
// the order the host's (A, B) pair logic expects, deterministic whatever order the scatter listed them in -- and their
// trace pairs copied in that order.  More than CS_MAX records of one (read, strand): reported (item_ovf), written unordered.
#define CS_MAX 512
#define CS_SMALL 128
// CSM: capacity of the LDS tables.  The CS_SMALL variant serves the items with at most CS_SMALL records (a capped pile-up
// read has 59) with a quarter of the LDS -- the block per item is a chain of dependent memory round trips (list -> records
// -> records again -> trace values), what helps is more items in flight: 32 per CU instead of 10 -- and leaves the others
// to the CS_MAX variant, which skips what the small one took.  Sixteen records' trace values are on their way together.
template <int CSM>
__global__ void __launch_bounds__(64)
k_compact_sym(const DhLa *__restrict__ slots, const uint16_t *__restrict__ tr_slots, int32_t trmax, const int32_t *__restrict__ list,
              const uint32_t *__restrict__ la_off, const uint32_t *__restrict__ tr_off, int64_t tr_base, DhLa *__restrict__ la_out,
              uint16_t *__restrict__ tr_out, int32_t *__restrict__ item_ovf)
{
    __shared__ uint64_t sk1[CSM], sk2[CSM];
    __shared__ int32_t stl[CSM], sso[CSM], ssl[CSM], sto[CSM];
    const int32_t it = blockIdx.x;
    const int lane = threadIdx.x;
    const uint32_t l0 = la_off[it], n = la_off[it + 1] - l0;
    if (n == 0) return;
    if (CSM == CS_SMALL ? n > (uint32_t)CS_SMALL : n <= (uint32_t)CS_SMALL) return;  // (the other variant's item)
    const uint32_t t0 = tr_off[it];
    if (n > CS_MAX) {
        if (lane == 0) item_ovf[it] = 1;
        uint32_t t = t0;
        for (uint32_t x = 0; x < n; x++) {
            const int64_t slot = list[l0 + x];
            DhLa la = slots[slot];
            const uint16_t *src = tr_slots + slot * trmax + la.toff;
            for (int32_t e = lane; e < la.tlen; e += 64) tr_out[t + e] = src[e];
            if (lane == 0) {
                la.toff = tr_base + t;
                la.pad = 0;
                la_out[l0 + x] = la;
            }
            t += la.tlen;
        }
        return;
    }
    for (uint32_t x = (uint32_t)lane; x < n; x += 64) {
        const int32_t slot = list[l0 + x];
        const DhLa la = slots[slot];
        sk1[x] = ((uint64_t)(uint32_t)la.bread << 32) | (uint32_t)la.abpos;
        sk2[x] = ((uint64_t)(uint32_t)la.bbpos << 32) | (uint32_t)la.aepos;
        stl[x] = la.tlen;
        sso[x] = (int32_t)la.toff;
        ssl[x] = slot;
    }
    __syncthreads();
    for (uint32_t x = (uint32_t)lane; x < n; x += 64) {
        const uint64_t k1 = sk1[x], k2 = sk2[x];
        const int32_t sx = ssl[x];
        int32_t rank = 0, toff = 0;
#pragma unroll 4
        for (uint32_t y = 0; y < n; y++) {
            const uint64_t y1 = sk1[y], y2 = sk2[y];
            const bool less = y1 < k1 || (y1 == k1 && (y2 < k2 || (y2 == k2 && ssl[y] < sx)));
            rank += less ? 1 : 0;
            toff += less ? stl[y] : 0;
        }
        sto[x] = toff;
        DhLa la = slots[sx];
        la.toff = tr_base + t0 + toff;
        la.pad = 0;
        la_out[l0 + rank] = la;
    }
    __syncthreads();
    // trace pairs, sixteen records at a time: their loads are in flight together (a record of the pile-up stage has ~50
    // values, so one record per step was one dependent memory round trip per record with most of it idle)
    constexpr int TU = 16;
    for (uint32_t x0 = 0; x0 < n; x0 += TU) {
        uint16_t v[TU];
#pragma unroll
        for (int j = 0; j < TU; j++) {
            const uint32_t x = x0 + j;
            v[j] = 0;
            if (x < n && lane < stl[x]) v[j] = tr_slots[(int64_t)ssl[x] * trmax + sso[x] + lane];
        }
#pragma unroll
        for (int j = 0; j < TU; j++) {
            const uint32_t x = x0 + j;
            if (x < n && lane < stl[x]) tr_out[t0 + sto[x] + lane] = v[j];
        }
        for (uint32_t x = x0; x < min(n, x0 + TU); x++) {
            const int32_t xl = stl[x];
            if (xl <= 64) continue;
            const uint16_t *src = tr_slots + (int64_t)ssl[x] * trmax + sso[x];
            uint16_t *dst = tr_out + t0 + sto[x];
            for (int32_t e = lane + 64; e < xl; e += 64) dst[e] = src[e];
        }
    }
}

extern "C" {
void dhk_tile_units(hipStream_t st, const DhCand *cand, const int32_t *ncand, const int32_t *candoff, int32_t item0, int32_t nitems,
                    int32_t max_cand, const int64_t *aoff, const int64_t *boff, Unit *units, uint32_t *nunits)
{
    if (nitems <= 0) return;
    hipLaunchKernelGGL(k_units_fat, dim3((nitems + 255) / 256), dim3(256), 0, st, cand, ncand, candoff, item0, nitems, max_cand, aoff,
                       boff, units, nunits);
}
void dhk_cand_counts(hipStream_t st, const int32_t *ncand, int32_t nitems, uint32_t *out)
{
    hipLaunchKernelGGL(k_cand_counts, dim3((nitems + 1 + 255) / 256), dim3(256), 0, st, ncand, nitems, out);
}
void dhk_rec_count(hipStream_t st, const DhLa *slots, int64_t nslots, int32_t item0, uint32_t *nla, uint32_t *ntr)
{
    if (nslots <= 0) return;
    hipLaunchKernelGGL(k_rec_count, dim3((uint32_t)((nslots + 255) / 256)), dim3(256), 0, st, slots, nslots, item0, nla, ntr);
}
void dhk_rec_scatter(hipStream_t st, const DhLa *slots, int64_t nslots, int32_t item0, const uint32_t *la_off, uint32_t *cursor,
                     int32_t *list)
{
    if (nslots <= 0) return;
    hipLaunchKernelGGL(k_rec_scatter, dim3((uint32_t)((nslots + 255) / 256)), dim3(256), 0, st, slots, nslots, item0, la_off, cursor,
                       list);
}
void dhk_compact_sym(hipStream_t st, const DhLa *slots, const uint16_t *tr_slots, int32_t trmax, const int32_t *list, int32_t nitems,
                     const uint32_t *la_off, const uint32_t *tr_off, int64_t tr_base, DhLa *la_out, uint16_t *tr_out, int32_t *item_ovf)
{
    if (nitems <= 0) return;
    hipLaunchKernelGGL(k_compact_sym<CS_SMALL>, dim3((uint32_t)nitems), dim3(64), 0, st, slots, tr_slots, trmax, list, la_off, tr_off,
                       tr_base, la_out, tr_out, item_ovf);
    hipLaunchKernelGGL(k_compact_sym<CS_MAX>, dim3((uint32_t)nitems), dim3(64), 0, st, slots, tr_slots, trmax, list, la_off, tr_off,
                       tr_base, la_out, tr_out, item_ovf);
}
void dhk_tile(hipStream_t st, int32_t nwaves, const Params *P)
{
    if (P->nitems <= 0 || nwaves <= 0) return;
#ifdef DH_SEED_PROF
    {
        unsigned long long zz[8] = {~0ull, 0, 0, 0, 0, 0, 0, 0};
        (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_tile_wt), zz, sizeof(zz), 0, hipMemcpyHostToDevice, st);
    }
#endif
    // (band = dh_align_opts.width: 64 rows on 64-bit vectors, or 32 rows on 32-bit vectors)
    if (P->o.width == 32) {
        if (P->tandem)
            hipLaunchKernelGGL((k_tile<true, 32>), dim3(nwaves), dim3(64), 0, st, *P);
        else
            hipLaunchKernelGGL((k_tile<false, 32>), dim3(nwaves), dim3(64), 0, st, *P);
    } else if (P->tandem)
        hipLaunchKernelGGL((k_tile<true, 64>), dim3(nwaves), dim3(64), 0, st, *P);
    else
        hipLaunchKernelGGL((k_tile<false, 64>), dim3(nwaves), dim3(64), 0, st, *P);
}
// resident wavefronts per CU the host launches: 12 of the 16 the registers allow measured best on configs[2]
// (mapping pass 31.3 ms against 33.6 with 16: fewer lanes share the queue's tail and the caches)
int32_t dhk_tile_waves_per_cu(void) { return 12; }
void dhk_pk2planes(hipStream_t st, void *words, int64_t nwords)
{
    if (nwords <= 0) return;
    hipLaunchKernelGGL(k_pk2planes, dim3((unsigned)((nwords + 255) / 256)), dim3(256), 0, st, (uint64_t *)words, nwords);
}
}
