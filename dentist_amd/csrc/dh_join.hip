// dh_join.hip -- gfx950 kernels of the per-pile-up k-mer join (see dh_join.h for the design).
// Replaces, for the pile-up all-vs-all of `dentist process` (processPileUps/package.d:474-485), the k-mer
// directory of the grouped DB (k_group_index) and the directory lookups of the seed filter (k_seed):
// the hits a read gets are the same multiset, produced without leaving the CU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "dh_join.h"
#include "dh_kmer.h"

#define LANES 64

#ifdef DH_SEED_PROF
__device__ unsigned long long g_join_prof[12];
#define JP(i) if (tid == 0) { const unsigned long long t_ = wall_clock64(); atomicAdd(&g_join_prof[i], t_ - tp_); tp_ = t_; }
extern "C" void dhk_join_prof_dump()
{
    unsigned long long h[12];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_join_prof), sizeof(h));
    if (h[7])
        fprintf(stderr, "[join prof] blocks %llu: gather %.1f chain %.1f count %.1f reserve %.1f emit %.1f us/block\n", h[7],
                h[0] / 100.0 / h[7], h[1] / 100.0 / h[7], h[2] / 100.0 / h[7], h[3] / 100.0 / h[7], h[4] / 100.0 / h[7]);
    unsigned long long z[12] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_join_prof), z, sizeof(z));
}
#else
#define JP(i)
#endif

namespace {

// slice of a canonical k-mer inside its group (ns slices) and bucket inside the slice's LDS table: two different
// multiplicative hashes of the k-mer, so that the entries of one slice spread over all buckets
__device__ __forceinline__ uint32_t join_slice(uint32_t canon, uint32_t ns) { return __umulhi(canon * 0x9E3779B1u, ns); }
template <int CAP>
__device__ __forceinline__ uint32_t join_bucket(uint32_t canon)
{
    return ((canon ^ (canon >> 15)) * 0x85EBCA6Bu) >> (32 - __builtin_ctz((unsigned)CAP));
}

// exclusive prefix sum over the block's threads (one value each); *total = sum.  s_w: one word per wavefront.
template <int THREADS>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, int tid, uint32_t *s_w, uint32_t *total)
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

}  // namespace

// ------------------------------------------------------------------------------------ partition
// Block pb covers the chunks [c0, c0 + JP_THREADS) of group g's chunk space (the reads of the group one after the
// other, JP_PER k-mer start positions per chunk); thread t rolls chunk c0 + t.
__global__ void __launch_bounds__(JP_THREADS)
k_join_part(JoinView jv, DbView B, int32_t k, int32_t kmer_mod)
{
    __shared__ uint64_t stage[JP_POS];
    __shared__ uint32_t scnt[JOIN_MAX_SLICES], sstart[JOIN_MAX_SLICES];
    __shared__ uint32_t cpre[JOIN_MAX_READS + 1];
    __shared__ uint32_t s_w[JP_THREADS / LANES];
    const int tid = threadIdx.x;
    const int32_t pb = blockIdx.x;
    const int32_t g = jv.pblk[pb].x, c0 = jv.pblk[pb].y;
    const int32_t r0 = jv.gfirst[g], nr = jv.gfirst[g + 1] - r0;
    const uint32_t ns = (uint32_t)jv.gns[g];
    // chunks of every read of the group (prefix sums)
    uint32_t nch = 0;
    if (tid < nr) {
        const int32_t len = (int32_t)(B.off[r0 + tid + 1] - B.off[r0 + tid]);
        const int32_t npos = len - k + 1;
        nch = npos > 0 ? (uint32_t)(npos + JP_PER - 1) / JP_PER : 0u;
    }
    uint32_t tot;
    const uint32_t ex = block_excl_scan<JP_THREADS>(nch, tid, s_w, &tot);
    if (tid < nr) cpre[tid] = ex;
    if (tid == 0) cpre[nr] = tot;
    scnt[tid] = 0;  // JP_THREADS == JOIN_MAX_SLICES
    __syncthreads();
    const uint32_t c = (uint32_t)c0 + (uint32_t)tid;
    bool live = c < tot;
    int32_t rl = 0, p0 = 0, len = 0;
    int64_t o = 0;
    if (live) {
        // read of the chunk: the last rl with cpre[rl] <= c (reads without k-mers have empty ranges)
        int32_t lo = 0, hi = nr - 1;
        while (lo < hi) {
            const int32_t mid = (lo + hi + 1) >> 1;
            if (cpre[mid] <= c)
                lo = mid;
            else
                hi = mid - 1;
        }
        rl = lo;
        p0 = (int32_t)(c - cpre[rl]) * JP_PER;
        o = B.off[r0 + rl];
        len = (int32_t)(B.off[r0 + rl + 1] - o);
    }
    const KmerSampler smp = kmer_sampler(kmer_mod, k);
    const uint32_t mask = k == 16 ? 0xFFFFFFFFu : ((1u << (2 * k)) - 1u);
    const int rcsh = 2 * (k - 1);
    const uint64_t kones = (1ull << k) - 1ull;
    uint64_t w[4] = {0, 0, 0, 0};
    uint64_t mw = 0;
    if (live) {
        const uint8_t *a = B.bases + o + p0;
#pragma unroll
        for (int u = 0; u < 4; u++) w[u] = load8(a + 8 * u);  // JP_PER + k - 1 <= 31 bases (buffers are padded)
        if (B.mask_bits) {
            const int64_t gb = o + p0;
            mw = load8(B.mask_bits + (gb >> 3)) >> (gb & 7);  // >= 57 bits of the window, JP_PER + k - 1 <= 31 are used
        }
    }
    uint32_t km = 0, rc = 0;
    int32_t valid = 0;
#pragma unroll
    for (int x = 0; x < JP_PER + 15; x++) {  // k <= 16; steps past JP_PER + k - 2 are predicated off (no break: the loop must unroll, w[] stays in registers)
        const bool act = x < JP_PER + k - 1;
        const int32_t p = p0 + x;
        uint32_t cc = (uint32_t)(w[x >> 3] >> (8 * (x & 7))) & 0xFFu;
        if (!live || p >= len) cc = 4u;
        if (cc < 4u) {
            km = ((km << 2) | cc) & mask;
            rc = (rc >> 2) | ((3u - cc) << rcsh);
            valid++;
        } else {
            km = 0;
            rc = 0;
            valid = 0;
        }
        const int j = x - (k - 1);  // k-mer start p0 + j
        if (act && j >= 0) {
            const uint32_t canon = km < rc ? km : rc;
            const bool ok = valid >= k && kmer_sampled((uint64_t)canon, smp) && ((mw >> j) & kones) == 0ull;
            uint64_t e = ~0ull;
            if (ok) {
                e = ((uint64_t)canon << 32) | (km != canon ? 1ull << 31 : 0ull) | (km == rc ? 1ull << 30 : 0ull) |
                    ((uint64_t)rl << 21) | (uint64_t)(p0 + j);
                atomicAdd(&scnt[join_slice(canon, ns)], 1u);
            }
            stage[j * JP_THREADS + tid] = e;
        }
    }
    __syncthreads();
    // the slices' ranges inside the block's region; scnt becomes the cursors
    const uint32_t mine = scnt[tid];
    uint32_t tot2;
    const uint32_t st = block_excl_scan<JP_THREADS>(mine, tid, s_w, &tot2);
    sstart[tid] = st;
    scnt[tid] = 0;
    if ((uint32_t)tid < ns) jv.psub[jv.psubrow[pb] + tid] = (st << 16) | mine;
    __syncthreads();
    uint64_t *out = jv.entries + (int64_t)pb * JP_POS;
#pragma unroll 4
    for (int j = 0; j < JP_PER; j++) {
        const uint64_t e = stage[j * JP_THREADS + tid];
        if (e == ~0ull) continue;
        const uint32_t s = join_slice((uint32_t)(e >> 32), ns);
        const uint32_t rank = atomicAdd(&scnt[s], 1u);
        out[sstart[s] + rank] = e;
    }
}

// ------------------------------------------------------------------------------------ join
template <int CAP>
__global__ void __launch_bounds__(JOIN_THREADS)
k_join(JoinView jv, DbView B, DhOpts o, const int64_t *__restrict__ goff, int32_t sepv)
{
    __shared__ uint64_t keys[CAP];
    __shared__ uint32_t head[CAP];  // bucket -> end of its range of `keys` (start = the end of the bucket before it)
    __shared__ uint16_t nhv[CAP];   // hits of an entry taken as B side (pass 1), so that pass 2 reserves them at once
    __shared__ int64_t lgoff[JOIN_MAX_READS];
    __shared__ int32_t llen[JOIN_MAX_READS];
    // which records of a pair are wanted (DbView::pflags) for the group's reads: the walks below test it for every pair of
    // entries -- from global memory that was two dependent byte loads inside the innermost loop of both passes
    __shared__ uint8_t lpf[JOIN_MAX_READS];
    // the entries whose bucket holds more than themselves (round 6): of the ~2 800 entries of a slice about 1 000 share their
    // k-mer with another entry (the intact copies of a true k-mer; 14 entries per such bucket on average) -- the others are
    // read errors, one of a kind, and most of them sit alone in their bucket: their walk is over after one step.  Dealt to the
    // lanes by index a wavefront ran as long as its longest walk with a third of its lanes at work.  The two walks go over
    // this list instead (57 -> 44 ms per step; a list of the entries WITH a partner costs its scan what it saves).
    __shared__ uint16_t act[CAP];
    __shared__ uint32_t s_nact;
    __shared__ uint32_t cnt[JOIN_MAX_READS], roff[JOIN_MAX_READS];
    __shared__ uint32_t s_w[JOIN_THREADS / LANES];
    __shared__ int32_t s_n;
    __shared__ unsigned long long s_base;
    const int tid = threadIdx.x;
    const int32_t jb = blockIdx.x;
    const int32_t g = jv.jblk[jb].x, sl = jv.jblk[jb].y;
    const int32_t r0 = jv.gfirst[g], nr = jv.gfirst[g + 1] - r0;
    for (int i = tid; i < CAP; i += JOIN_THREADS) head[i] = 0u;
    if (tid < nr) {
        lgoff[tid] = goff[r0 + tid];
        llen[tid] = (int32_t)(B.off[r0 + tid + 1] - B.off[r0 + tid]);
    }
    lpf[tid] = (tid < nr && B.pflags) ? B.pflags[r0 + tid] : (uint8_t)3;  // (no flags: every record is wanted)
    cnt[tid] = 0u;  // JOIN_THREADS == JOIN_MAX_READS
    if (tid == 0) s_n = 0;
#ifdef DH_SEED_PROF
    unsigned long long tp_ = wall_clock64();
#endif
    __syncthreads();
    // ---- gather the slice's entries from the group's part blocks.  Every thread reads the (start, count) of one part block,
    // a block-wide scan places the blocks' runs, and the entries are copied by FLAT index -- each thread finds the part
    // block of its entry in the scanned counts and up to eight loads per thread are in flight together.  (A wavefront per
    // part block -- descriptor, an LDS atomic for the place, the copy: three dependent round trips, seven part blocks per
    // wavefront one after the other -- was 11.6 of the 64 us a block took.)
    {
        __shared__ uint32_t gsrc[JOIN_THREADS], gend[JOIN_THREADS];  // start inside the part block; end of its run in `keys`
        const int32_t pb0 = jv.pfirst[g], npb = jv.pfirst[g + 1] - pb0;
        uint32_t ntot = 0;
        for (int32_t c0 = 0; c0 < npb; c0 += JOIN_THREADS) {
            const int32_t m = min(JOIN_THREADS, npb - c0);
            const uint32_t u = tid < m ? jv.psub[jv.psubrow[pb0 + c0 + tid] + sl] : 0u;
            const uint32_t c = u & 0xFFFFu;
            uint32_t tot;
            const uint32_t ex = block_excl_scan<JOIN_THREADS>(c, tid, s_w, &tot);
            gsrc[tid] = u >> 16;
            gend[tid] = ex + c;
            __syncthreads();
            if (ntot + tot <= (uint32_t)CAP) {
                int32_t top = 1;
                while (top < m) top <<= 1;
                constexpr int GU = 8;
                for (uint32_t f0 = (uint32_t)tid; f0 < tot; f0 += JOIN_THREADS * GU) {
                    int32_t lo[GU];
#pragma unroll
                    for (int q = 0; q < GU; q++) lo[q] = 0;
                    // the part block of flat index f: the first j with gend[j] > f (empty runs repeat a value and are skipped)
                    for (int32_t step = top >> 1; step > 0; step >>= 1) {
#pragma unroll
                        for (int q = 0; q < GU; q++) {
                            const int32_t idx = lo[q] + step;
                            if (idx < m && gend[idx - 1] <= f0 + (uint32_t)q * JOIN_THREADS) lo[q] = idx;
                        }
                    }
                    uint64_t v[GU];
#pragma unroll
                    for (int q = 0; q < GU; q++) {
                        const uint32_t f = f0 + (uint32_t)q * JOIN_THREADS;
                        v[q] = 0;
                        if (f < tot) {
                            const int32_t j = lo[q];
                            const uint32_t first = j ? gend[j - 1] : 0u;
                            v[q] = jv.entries[(int64_t)(pb0 + c0 + j) * JP_POS + gsrc[j] + (f - first)];
                        }
                    }
#pragma unroll
                    for (int q = 0; q < GU; q++) {
                        const uint32_t f = f0 + (uint32_t)q * JOIN_THREADS;
                        if (f < tot) keys[ntot + f] = v[q];
                    }
                }
            }
            ntot += tot;
            __syncthreads();  // gsrc / gend are rewritten by the next round
        }
        if (tid == 0) s_n = (int32_t)min(ntot, 0x7FFFFFFFu);
    }
    __syncthreads();
    JP(0)
    const int32_t n = s_n;
    if (n > CAP) {  // a k-mer with thousands of copies landed here: the whole call takes the directory path
        if (tid == 0) atomicOr(jv.status, DH_ST_JOIN_OVERFLOW);
        return;
    }
    // ---- the entries bucket by bucket (counts, scan, scatter through registers): a bucket is then a contiguous range of
    // `keys` whose loads do not depend on each other -- as chains (entry -> previous entry of its bucket) every step of a
    // walk was an LDS round trip, and a wavefront walks as long as its longest bucket (the true k-mers' ~coverage copies)
    for (int32_t i = tid; i < n; i += JOIN_THREADS) atomicAdd(&head[join_bucket<CAP>((uint32_t)(keys[i] >> 32))], 1u);
    __syncthreads();
    {
        constexpr int PER = CAP / JOIN_THREADS;
        uint32_t c[PER], sum = 0;
#pragma unroll
        for (int u = 0; u < PER; u++) {
            c[u] = head[tid * PER + u];
            sum += c[u];
        }
        uint32_t tot;
        uint32_t at = block_excl_scan<JOIN_THREADS>(sum, tid, s_w, &tot);
#pragma unroll
        for (int u = 0; u < PER; u++) {
            head[tid * PER + u] = at;
            at += c[u];
        }
        uint64_t ke[PER];
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const int32_t i = tid + u * JOIN_THREADS;
            ke[u] = i < n ? keys[i] : 0ull;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PER; u++)
            if (tid + u * JOIN_THREADS < n) keys[atomicAdd(&head[join_bucket<CAP>((uint32_t)(ke[u] >> 32))], 1u)] = ke[u];
    }
    __syncthreads();
    {
        // thread t owns the entries [t * PER, (t + 1) * PER): the list keeps the entries in index (= bucket) order, so the
        // lanes of a wavefront walk buckets of similar length
        constexpr int PER = CAP / JOIN_THREADS;
        uint32_t fl = 0, c = 0;
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const int32_t i = tid * PER + u;
            if (i < n) {
                // an entry alone in its bucket has no partner (without any skip_self rule it still meets itself)
                const uint32_t hb = join_bucket<CAP>((uint32_t)(keys[i] >> 32));
                const bool partner = o.skip_self == 0 || head[hb] - (hb ? head[hb - 1] : 0u) >= 2u;
                if (partner) {
                    fl |= 1u << u;
                    c++;
                }
            }
        }
        uint32_t tot;
        uint32_t at = block_excl_scan<JOIN_THREADS>(c, tid, s_w, &tot);
#pragma unroll
        for (int u = 0; u < PER; u++)
            if (fl & (1u << u)) act[at++] = (uint16_t)(tid * PER + u);
        if (tid == 0) s_nact = tot;
    }
    __syncthreads();
    const int32_t nact = (int32_t)s_nact;
    JP(1)
    // the entries of bucket hb, four at a time (loads past the end read the last entry again and are masked: a branch
    // per load made every one of them a round trip of its own); fn(key of an entry)
#define JOIN_WALK(canon_, BODY)                                                                      \
    {                                                                                                \
        const uint32_t hb_ = join_bucket<CAP>(canon_);                                               \
        const uint32_t b1_ = head[hb_];                                                              \
        for (uint32_t t_ = hb_ ? head[hb_ - 1] : 0u; t_ < b1_; t_ += 4) {                            \
            uint64_t kk_[4];                                                                         \
            _Pragma("unroll") for (int j_ = 0; j_ < 4; j_++) kk_[j_] = keys[min(t_ + j_, b1_ - 1)];  \
            _Pragma("unroll") for (int j_ = 0; j_ < 4; j_++) {                                       \
                const uint64_t ka = kk_[j_];                                                         \
                if (t_ + j_ >= b1_ || (uint32_t)(ka >> 32) != canon_) continue;                      \
                BODY                                                                                 \
            }                                                                                        \
        }                                                                                            \
    }
    const int32_t tcap = o.tcap;
    const int32_t k = o.k;
    // walk of entry i taken as the B side: fn(a key) for every entry of the same canonical k-mer (itself included)
    // ---- pass 1: orientation classes of every entry's k-mer (-t cap), hits per B read
    uint32_t dmask = 0;  // (forward ok, reverse ok) of this thread's entries, two bits each
    {
        int slot = 0;
        for (int32_t a = tid; a < nact; a += JOIN_THREADS, slot++) {
            const int32_t i = act[a];
            const uint64_t key = keys[i];
            const uint32_t canon = (uint32_t)(key >> 32);
            const uint32_t ori = (uint32_t)(key >> 31) & 1u;
            const bool pal = ((key >> 30) & 1ull) != 0;
            const int32_t brl = (int32_t)(key >> 21) & (JOIN_MAX_READS - 1);
            const int32_t R = r0 + brl;
            const uint32_t pfb = lpf[brl];
            int32_t n_same = 0, n_opp = 0, k_same = 0, k_opp = 0;
            JOIN_WALK(canon, {
                const bool same = ((uint32_t)(ka >> 31) & 1u) == ori;
                const int32_t arl_ = (int32_t)(ka >> 21) & (JOIN_MAX_READS - 1);
                const int32_t Aq = r0 + arl_;
                const uint32_t pfa = lpf[arl_];
                bool keep = true;
                if (o.skip_self == 1) keep = Aq != R;
                if (o.skip_self == 2) keep = Aq != R && ((Aq < R) == (((Aq + R) & 1) == 0));
                // neither record of the pair is wanted (dh_pair_seeded: (a, b) wanted iff f[a] & 1 and f[b] & 2)
                if (o.skip_self == 2 && !(((pfa & 1u) && (pfb & 2u)) || ((pfb & 1u) && (pfa & 2u)))) keep = false;
                n_same += same ? 1 : 0;
                k_same += same && keep ? 1 : 0;
                n_opp += same ? 0 : 1;
                k_opp += !same && keep ? 1 : 0;
            })
            // a k-mer occurring more than tcap times in an orientation class yields no hits of that class (its own
            // occurrence counts, as an index entry does)
            const bool dof = n_same <= tcap && (o.strands & 1);
            const bool dor = (pal ? n_same <= tcap : (n_opp >= 1 && n_opp <= tcap)) && (o.strands & 2);
            const int32_t nh = (dof ? k_same : 0) + (dor ? (pal ? k_same : k_opp) : 0);
            dmask |= ((dof ? 1u : 0u) | (dor ? 2u : 0u)) << (2 * slot);
            nhv[i] = (uint16_t)nh;
            if (nh) atomicAdd(&cnt[brl], (uint32_t)nh);
        }
    }
    __syncthreads();
    JP(2)
    // ---- the block's range of the hit buffer, one segment per B read
    const uint32_t mine = cnt[tid];
    uint32_t total;
    const uint32_t ro = block_excl_scan<JOIN_THREADS>(mine, tid, s_w, &total);
    roff[tid] = ro;
    if (tid == 0) s_base = total ? atomicAdd(jv.cursor, (unsigned long long)total) : 0ull;
    __syncthreads();
    const unsigned long long base = s_base;
    if (tid < nr) jv.segtab[jv.segrow[r0 + tid] + sl] = ((uint64_t)(base + ro) << 24) | (uint64_t)mine;
    if (total == 0) return;
    if (base + total > (unsigned long long)jv.hits_cap) {  // the host reruns this kernel with a buffer of cursor entries
        if (tid == 0) atomicOr(jv.status, DH_ST_JOIN_HITCAP);
        return;
    }
    cnt[tid] = 0u;  // now the cursors of the segments
    __syncthreads();
    JP(3)
    // ---- pass 2: the hits
    {
        int slot = 0;
        for (int32_t a = tid; a < nact; a += JOIN_THREADS, slot++) {
            const uint32_t dm = (dmask >> (2 * slot)) & 3u;
            if (!dm) continue;
            const int32_t i = act[a];
            const bool dof = (dm & 1u) != 0, dor = (dm & 2u) != 0;
            const uint64_t key = keys[i];
            const uint32_t canon = (uint32_t)(key >> 32);
            const uint32_t ori = (uint32_t)(key >> 31) & 1u;
            const bool pal = ((key >> 30) & 1ull) != 0;
            const int32_t brl = (int32_t)(key >> 21) & (JOIN_MAX_READS - 1);
            const int32_t q = (int32_t)(key & (JOIN_MAX_LEN - 1));
            const int32_t R = r0 + brl;
            const uint32_t pfb = lpf[brl];
            const int32_t qrev = llen[brl] - k - q;  // position on the reverse-complemented read
            // (one returning atomic per entry, not one per hit: each was a round trip in front of its store)
            uint64_t *dst = jv.hits + base + roff[brl] + atomicAdd(&cnt[brl], (uint32_t)nhv[i]);
            JOIN_WALK(canon, {
                const bool same = ((uint32_t)(ka >> 31) & 1u) == ori;
                const int32_t arl = (int32_t)(ka >> 21) & (JOIN_MAX_READS - 1);
                const int32_t Aq = r0 + arl;
                if (o.skip_self == 1 && Aq == R) continue;
                if (o.skip_self == 2 && (Aq == R || ((Aq < R) != (((Aq + R) & 1) == 0)))) continue;
                const uint32_t pfa = lpf[arl];
                if (o.skip_self == 2 && !(((pfa & 1u) && (pfb & 2u)) || ((pfb & 1u) && (pfa & 2u)))) continue;
                const int64_t gv = lgoff[arl] + (int64_t)(ka & (JOIN_MAX_LEN - 1));
                if (dof && (same || pal)) {
                    const int64_t D = gv + sepv - q;
                    *dst++ = ((uint64_t)D << 24) | (uint32_t)q;
                }
                if (dor && (!same || pal)) {
                    const int64_t D = gv + sepv - qrev;
                    *dst++ = (1ull << 63) | ((uint64_t)D << 24) | (uint32_t)qrev;
                }
            })
        }
    }
#ifdef DH_SEED_PROF
    __syncthreads();
    JP(4)
    if (tid == 0) atomicAdd(&g_join_prof[7], 1ull);
#endif
}

#undef JOIN_WALK

template __global__ void k_join<JOIN_CAP>(JoinView, DbView, DhOpts, const int64_t *, int32_t);

// reads whose hits exceed 2048 / 4096 / 8192 entries (out[0..2]) and the largest count (out[3]): the host picks the LDS
// capacity of the seed filter's back end from them
__global__ void __launch_bounds__(256)
k_join_hist(JoinView jv, const int32_t *__restrict__ group, int32_t nreads, unsigned int *__restrict__ out)
{
    const int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nreads) return;
    const int32_t ns = jv.gns[group[r]];
    const uint64_t *row = jv.segtab + jv.segrow[r];
    uint32_t n = 0;
    for (int32_t s = 0; s < ns; s++) n += (uint32_t)(row[s] & 0xFFFFFFull);
    if (n > 2048u) atomicAdd(&out[0], 1u);
    if (n > 4096u) atomicAdd(&out[1], 1u);
    if (n > 8192u) atomicAdd(&out[2], 1u);
    atomicMax(&out[3], n);
}

extern "C" void dhk_join_hist(hipStream_t st, JoinView jv, const int32_t *group, int32_t nreads, unsigned int *out4)
{
    (void)hipMemsetAsync(out4, 0, 4 * sizeof(unsigned int), st);
    if (nreads <= 0) return;
    hipLaunchKernelGGL(k_join_hist, dim3((uint32_t)((nreads + 255) / 256)), dim3(256), 0, st, jv, group, nreads, out4);
}

extern "C" void dhk_join_part(hipStream_t st, JoinView jv, DbView B, int32_t k, int32_t kmer_mod)
{
    if (jv.npart <= 0) return;
    hipLaunchKernelGGL(k_join_part, dim3((uint32_t)jv.npart), dim3(JP_THREADS), 0, st, jv, B, k, kmer_mod);
}

extern "C" void dhk_join(hipStream_t st, JoinView jv, DbView B, DhOpts o, const int64_t *goff, int32_t sepv)
{
    if (jv.njoin <= 0) return;
    hipLaunchKernelGGL(k_join<JOIN_CAP>, dim3((uint32_t)jv.njoin), dim3(JOIN_THREADS), 0, st, jv, B, o, goff, sepv);
}
