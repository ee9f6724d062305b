// dh_consensus.hip -- gfx950 kernels of the pile-up consensus path.
//
// K1b k_gather_slices   build a DB from slices of another DB (cropped reads, flank windows,
//                       reference reads) without leaving HBM
// K7  k_tile_qv         intrinsic QV per trace tile of every pile-up read (DASqv role)
// K8a k_seg_vote        one thread per (overlap, trace tile): Needleman-Wunsch of the tile
//                       (findAlignment semantics, util/string.d:478-520, 775-831) with the score
//                       matrix interleaved in HBM, canonical indel placement, column votes
// K8b k_emit            run-length aware majority emission of the new consensus
//
// Arithmetic spec: DESIGN.md "Algorithm DH-1 / consensus"; reference call sites
// source/dentist/dazzler.d:6142-6156 (DASqv), 6185-6231 (daccord).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "dh_device.h"

#define MAXINS 4
// The op list of a tile (the path of its alignment, back to front: 0 = pair, 1 = deletion, 2 = insertion), eight ops to an
// 8-byte word, the words interleaved over the tiles of the launch: word g of tile dp = opbuf64[g * nseg + dp].  The
// producers collect eight ops in a register and store them with one (coalesced) store, the consumer fetches a word per
// eight ops and has the next one on its way -- one op per byte apart by nseg bytes, every op was a store / a load of its own.
#define OP_WORD(g) ((uint64_t *)opbuf)[(int64_t)(g)*NDP + dp]
#define OP_PUT(op_)                                         \
    {                                                       \
        oacc |= (uint64_t)(uint8_t)(op_) << (8 * (nops & 7)); \
        nops++;                                             \
        if ((nops & 7) == 0) {                              \
            OP_WORD((nops >> 3) - 1) = oacc;                \
            oacc = 0;                                       \
        }                                                   \
    }
#define OP_FLUSH \
    if (nops & 7) OP_WORD(nops >> 3) = oacc;
#define VSTRIDE (6 + 4 * MAXINS)
#define MAXQV 50
#define SEG_MAX 250 /* longest tile side the u8 score matrix supports */

// ------------------------------------------------------------------------------------ K1b

__global__ void __launch_bounds__(256)
k_gather_slices(const uint8_t *__restrict__ src, const int64_t *__restrict__ src_off,
                const int32_t *__restrict__ sidx, const int32_t *__restrict__ sbeg,
                const int64_t *__restrict__ dst_off, int32_t n, uint8_t *__restrict__ dst)
{
    const int32_t s = blockIdx.y;
    if (s >= n) return;
    const int64_t so = src_off[sidx[s]] + sbeg[s];
    const int64_t d0 = dst_off[s], len = dst_off[s + 1] - d0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len;
         i += (int64_t)gridDim.x * blockDim.x)
        dst[d0 + i] = src[so + i];
}

// multi-part gather: every output sequence is the concatenation of parts taken from one of two
// source DBs, optionally reverse-complemented (cropped read + contig support patches)
struct PartDesc {
    int32_t src, sidx, sbeg, len, rc, pad;
    int64_t dst;
};

__global__ void __launch_bounds__(256)
k_gather_parts(const uint8_t *__restrict__ src0, const int64_t *__restrict__ off0,
               const uint8_t *__restrict__ src1, const int64_t *__restrict__ off1,
               const PartDesc *__restrict__ parts, int32_t n, uint8_t *__restrict__ dst)
{
    const int32_t p = blockIdx.y;
    if (p >= n) return;
    const PartDesc d = parts[p];
    const uint8_t *s = (d.src ? src1 + off1[d.sidx] : src0 + off0[d.sidx]) + d.sbeg;
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < d.len; i += gridDim.x * blockDim.x) {
        uint8_t c = d.rc ? s[d.len - 1 - i] : s[i];
        if (d.rc && c < 4) c = (uint8_t)(3 - c);
        dst[d.dst + i] = c;
    }
}

// ------------------------------------------------------------------------------------ K6b
// The alignment funnel of computeQVs (processPileUps/package.d:474-516) on the records the pile-up all-vs-all left on
// the device -- averageErrorRate <= maxAlignmentError, chainLocalAlignments per (A, B) pair (chaining.d:122-334, the
// arithmetic of dh_process.cpp:chain_pair), isValidPileUpAlignment (dazzler.d:4126-4141) -- so that 3.5 M records
// (configs[2]) neither travel to the host nor back for the tile QVs.  One wavefront per A read; its records are one
// range of the compacted array (item_off: exclusive prefix sums over the items (read, strand)), the records of a pair
// in ascending index order are what the host sees after its stable merge by B read.  Flags are updated in place:
// DISABLED (0x20), START | BEST (0x4 | 0x10) / NEXT (0x8) of the kept chains, IMPROPER (0x40: still counted by the tile
// QVs, dropped afterwards).  la_first[r] = first record of A read r (nreads + 1 entries), live[r] = records of r that
// stay after the proper-overlap filter.  status |= 1: a read with more than PF_MAXN records, a pair with more than
// PF_MAXG enabled records, or a pair with two best chains that share records -- the caller redoes the batch on the host
// (never met on pile-ups up to 250 reads).  The kernel serves the default --min-relative-score 1.0 only.
#define PF_MAXN 1024
#define PF_MAXG 8
__device__ __forceinline__ int32_t pf_score(int32_t ab, int32_t ae, int32_t bb, int32_t be) { return ((ae - ab) + (be - bb)) / 2; }

__global__ void __launch_bounds__(64)
k_pile_funnel(DhLa *__restrict__ las, const uint32_t *__restrict__ item_off, int32_t nreads, const int64_t *__restrict__ roff,
              int32_t max_err_ppm, int32_t tsp, int32_t *__restrict__ la_first, int32_t *__restrict__ live,
              int32_t *__restrict__ status)
{
    __shared__ int32_t s_b[PF_MAXN], s_ab[PF_MAXN], s_ae[PF_MAXN], s_bb[PF_MAXN], s_be[PF_MAXN];
    __shared__ uint32_t s_fl[PF_MAXN];
    const int32_t r = blockIdx.x;
    if (r >= nreads) return;
    const int lane = threadIdx.x;
    const int32_t l0 = (int32_t)item_off[2 * r], l1 = (int32_t)item_off[2 * r + 2], n = l1 - l0;
    if (lane == 0) {
        la_first[r] = l0;
        if (r == nreads - 1) la_first[nreads] = l1;
    }
    if (n > PF_MAXN) {
        if (lane == 0) {
            atomicOr(status, 1);
            live[r] = 0;
        }
        return;
    }
    for (int32_t i = lane; i < n; i += 64) {
        const DhLa la = las[l0 + i];
        uint32_t fl = la.flags;
        if ((int64_t)la.diffs * 1000000 > (int64_t)max_err_ppm * (la.aepos - la.abpos)) fl |= 0x20u;
        s_b[i] = la.bread;
        s_ab[i] = la.abpos;
        s_ae[i] = la.aepos;
        s_bb[i] = la.bbpos;
        s_be[i] = la.bepos;
        s_fl[i] = fl;
    }
    __syncthreads();
    // ---- chains per (A, B) pair: the lane of a pair's first record does the pair
    for (int32_t i = lane; i < n; i += 64) {
        const int32_t b = s_b[i];
        bool leader = true;
        for (int32_t j = 0; j < i && leader; j++) leader = s_b[j] != b;
        if (!leader) continue;
        int32_t idx[PF_MAXG], nen = 0;
        for (int32_t j = i; j < n; j++)
            if (s_b[j] == b && !(s_fl[j] & 0x20u)) {
                if (nen < PF_MAXG) idx[nen] = j;
                nen++;
            }
        if (nen == 0) continue;
        if (nen > PF_MAXG) {
            atomicOr(status, 1);
            continue;
        }
        if (nen == 1) {  // a single enabled record is its own best chain
            const int32_t x = idx[0];
            if (pf_score(s_ab[x], s_ae[x], s_bb[x], s_be[x]) < tsp)
                s_fl[x] |= 0x20u;
            else
                s_fl[x] = (s_fl[x] & ~(0x4u | 0x8u | 0x10u)) | 0x4u | 0x10u;
            continue;
        }
        // order by (abpos, bbpos, index): insertion sort, stable
        for (int32_t u = 1; u < nen; u++) {
            const int32_t x = idx[u];
            int32_t v = u - 1;
            while (v >= 0 && (s_ab[idx[v]] > s_ab[x] || (s_ab[idx[v]] == s_ab[x] && s_bb[idx[v]] > s_bb[x]))) {
                idx[v + 1] = idx[v];
                v--;
            }
            idx[v + 1] = x;
        }
        int32_t dist[PF_MAXG], pred[PF_MAXG];
        for (int32_t v = 0; v < nen; v++) {
            dist[v] = -pf_score(s_ab[idx[v]], s_ae[idx[v]], s_bb[idx[v]], s_be[idx[v]]);
            pred[v] = -1;
        }
        for (int32_t u = 0; u < nen; u++)
            for (int32_t v = u + 1; v < nen; v++) {
                const int32_t x = idx[u], y = idx[v];
                if ((s_fl[x] & 1u) != (s_fl[y] & 1u)) continue;
                const int32_t ga = s_ab[y] - s_ae[x], gb = s_bb[y] - s_be[x];
                if (!(s_ab[x] < s_ab[y] && s_bb[x] < s_bb[y])) continue;
                const int32_t aga = ga < 0 ? -ga : ga, agb = gb < 0 ? -gb : gb, dg = ga - gb < 0 ? gb - ga : ga - gb;
                if (dg > 1000 || (aga > agb ? aga : agb) > 10000) continue;
                const int32_t lax = s_ae[x] - s_ab[x], lay = s_ae[y] - s_ab[y], lbx = s_be[x] - s_bb[x], lby = s_be[y] - s_bb[y];
                const int32_t mla = lax < lay ? lax : lay, mlb = lbx < lby ? lbx : lby;
                if (!((double)(ga < 0 ? -ga : 0) <= 0.3 * (double)mla && (double)(gb < 0 ? -gb : 0) <= 0.3 * (double)mlb)) continue;
                const int32_t d = dist[u] + dg + (aga > agb ? aga : agb) / 10 - pf_score(s_ab[y], s_ae[y], s_bb[y], s_be[y]);
                if (dist[v] > d) {
                    dist[v] = d;
                    pred[v] = u;
                }
            }
        int32_t mind = dist[0];
        for (int32_t v = 1; v < nen; v++) mind = dist[v] < mind ? dist[v] : mind;
        const int32_t best = -mind;
        const double thr_d = (double)tsp > 1.0 * (double)best ? (double)tsp : 1.0 * (double)best;
        const int32_t thr = (int32_t)thr_d;
        int32_t ends[PF_MAXG];  // end nodes by ascending distance, stable
        for (int32_t v = 0; v < nen; v++) ends[v] = v;
        for (int32_t u = 1; u < nen; u++) {
            const int32_t x = ends[u];
            int32_t v = u - 1;
            while (v >= 0 && dist[ends[v]] > dist[x]) {
                ends[v + 1] = ends[v];
                v--;
            }
            ends[v + 1] = x;
        }
        // (a chain that runs into records of a better chain of the pair -- an alternate chain, two chains of the best score
        // with a common prefix -- is written with its whole path, the shared records twice, chaining.d:247-285: records are
        // inserted on the host only, the pair is left as it is and reported)
        uint32_t keep = 0, first = 0;
        bool shared = false;
        for (int32_t q = 0; q < nen; q++) {
            const int32_t e = ends[q];
            if (-dist[e] < thr || (keep >> e & 1u)) continue;
            int32_t last = e;
            for (int32_t v = e; v >= 0; v = pred[v]) {
                shared = shared || (keep >> v & 1u);
                keep |= 1u << v;
                last = v;
            }
            first |= 1u << last;
        }
        if (shared) {
            atomicOr(status, 1);
            continue;
        }
        for (int32_t v = 0; v < nen; v++)
            s_fl[idx[v]] = (keep >> v & 1u) ? (s_fl[idx[v]] & ~(0x4u | 0x8u | 0x10u)) | ((first >> v & 1u) ? (0x4u | 0x10u) : 0x8u)
                                            : s_fl[idx[v]] | 0x20u;
    }
    __syncthreads();
    // ---- isValidPileUpAlignment with allowance = trace spacing; flags back, live records counted
    const int32_t alen = (int32_t)(roff[r + 1] - roff[r]);
    int32_t cnt = 0;
    for (int32_t i = lane; i < n; i += 64) {
        uint32_t fl = s_fl[i];
        if (!(fl & 0x20u)) {
            const int32_t b = s_b[i];
            const int32_t blen = (int32_t)(roff[b + 1] - roff[b]);
            const bool ab = s_ab[i] <= tsp, bb = s_bb[i] <= tsp;
            const bool ae = s_ae[i] + tsp >= alen, be = s_be[i] + tsp >= blen;
            const bool ok = b != r && (((ab && bb) && (ae || be)) || ((ae && be) && (ab || bb)));
            if (!ok)
                fl |= 0x40u;
            else
                cnt++;
        }
        las[l0 + i].flags = fl;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
    if (lane == 0) live[r] = cnt;
}

// records of selected A reads gathered into one array (the overlaps of the reference reads: all the host needs);
// IMPROPER becomes DISABLED on the way (filterPileUpAlignments runs after the tile QVs, dazzler.d:4043-4094)
__global__ void __launch_bounds__(64)
k_gather_read_records(const DhLa *__restrict__ las, const int32_t *__restrict__ la_first, const int32_t *__restrict__ sel,
                      const int32_t *__restrict__ dst_off, int32_t nsel, DhLa *__restrict__ out)
{
    const int32_t s = blockIdx.x;
    if (s >= nsel) return;
    const int32_t r = sel[s], l0 = la_first[r], n = la_first[r + 1] - l0, d0 = dst_off[s];
    for (int32_t i = threadIdx.x; i < n; i += 64) {
        DhLa la = las[l0 + i];
        if (la.flags & 0x40u) la.flags = (la.flags & ~0x40u) | 0x20u;
        out[d0 + i] = la;
    }
}

// ------------------------------------------------------------------------------------ K7

// las sorted by aread; la_first[r] .. la_first[r+1] are the LAs with aread == r.
// qv[r * maxtiles + t]; tile value = floor(200 * diffs / (tile_len + bbases)), the QV is the mean
// of the lowest min(cov, m) values, capped at MAXQV; MAXQV when no overlap covers the tile.
// One wavefront per read.  Per tile the lanes share the overlaps of the read (lane l takes the overlaps
// l, l + 64, ...), the tile values -- at most 200, since a tile's diffs cannot exceed the longer of its
// two sides -- are counted into 256 LDS bins, and the sum of the lowest `cov` of them is one wave-wide
// scan over the bins (four bins per lane).  No per-thread arrays (the round-2 kernel kept 64 sorted values
// per thread in scratch: 15 GB of traffic for 1 MB of output), any number of overlaps, any cov.
__global__ void __launch_bounds__(64)
k_tile_qv(const DhLa *__restrict__ las, const uint16_t *__restrict__ trace,
          const int32_t *__restrict__ la_first, const int64_t *__restrict__ roff, int32_t nreads,
          int32_t tspace, const int32_t *__restrict__ cov_of, int32_t maxtiles, uint8_t *__restrict__ qv)
{
    __shared__ int32_t hist[256];
    // what a tile asks of an overlap -- its A interval, the number of its trace pairs, where they lie --, staged once per
    // read: every tile of the read walks all of its overlaps, and re-reading the 48-byte records from memory tile after
    // tile was 1.57 GB of fetches per launch for 170 MB of records (round-4 verdict).  A read with more overlaps than fit
    // reads the records as before.
    constexpr int32_t QCAP = 512;
    __shared__ int32_t s_ab[QCAP], s_ae[QCAP], s_np[QCAP];
    __shared__ int64_t s_to[QCAP];
    const int32_t r = blockIdx.x;
    if (r >= nreads) return;
    const int lane = threadIdx.x;
    const int32_t cov = cov_of[r];
    const int32_t rlen = (int32_t)(roff[r + 1] - roff[r]);
    const int32_t nt = (rlen + tspace - 1) / tspace;
    const int32_t l0 = la_first[r], l1 = la_first[r + 1];
    const bool staged = l1 - l0 <= QCAP;
    if (staged)
        for (int32_t i = lane; i < l1 - l0; i += 64) {
            const DhLa la = las[l0 + i];
            s_ab[i] = (la.flags & 0x20u) ? 0x7FFFFFFF : la.abpos;  // (a disabled overlap covers no tile)
            s_ae[i] = la.aepos;
            s_np[i] = la.tlen / 2;
            s_to[i] = la.toff;
        }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    for (int32_t t = 0; t < nt && t < maxtiles; t++) {
        const int32_t t0 = t * tspace, t1 = (t0 + tspace < rlen) ? t0 + tspace : rlen;
#pragma unroll
        for (int x = 0; x < 4; x++) hist[4 * lane + x] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        for (int32_t i = lane; i < l1 - l0; i += 64) {
            int32_t abpos, aepos, np;
            int64_t toff;
            if (staged) {
                abpos = s_ab[i];
                aepos = s_ae[i];
                np = s_np[i];
                toff = s_to[i];
            } else {
                const DhLa la = las[l0 + i];
                if (la.flags & 0x20u) continue;
                abpos = la.abpos;
                aepos = la.aepos;
                np = la.tlen / 2;
                toff = la.toff;
            }
            if (abpos > t0 || aepos < t1) continue;
            const int32_t e = t - abpos / tspace;
            const int32_t seg0 = e == 0 ? abpos : t0;
            const int32_t seg1 = (e == np - 1) ? aepos : t1;
            if (seg0 != t0 || seg1 != t1) continue;
            const uint16_t *tr = trace + toff;
            const int32_t val = 200 * (int32_t)tr[2 * e] / ((t1 - t0) + (int32_t)tr[2 * e + 1]);
            atomicAdd(&hist[val < 255 ? val : 255], 1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        int32_t c[4], mine = 0;
#pragma unroll
        for (int x = 0; x < 4; x++) {
            c[x] = hist[4 * lane + x];
            mine += c[x];
        }
        // exclusive scan of the lanes' counts
        int32_t incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int32_t v = __shfl_up(incl, off, 64);
            if (lane >= off) incl += v;
        }
        const int32_t m = __shfl(incl, 63, 64);
        int32_t q = MAXQV;
        if (m > 0) {
            const int32_t use = m < cov ? m : cov;
            int32_t before = incl - mine, part = 0;
#pragma unroll
            for (int x = 0; x < 4; x++) {
                int32_t take = use - before;
                take = take < 0 ? 0 : (take > c[x] ? c[x] : take);
                part += take * (4 * lane + x);
                before += c[x];
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
            q = part / use;
            if (q > MAXQV) q = MAXQV;
        }
        if (lane == 0) qv[(int64_t)r * maxtiles + t] = (uint8_t)q;
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------ K8a

struct SegDesc {
    int32_t tmpl;        // template (A) sequence index in the template DB
    int32_t a0, a1;      // A interval of the tile
    int32_t bseq;        // B sequence index in the read DB
    int32_t b0, b1;      // B interval (in the orientation of the overlap)
    int32_t comp;        // 1: B is reverse-complemented
    int32_t band;        // half-width of the DP band: tile diffs of the trace + 1 (see k_seg_vote)
};

__global__ void __launch_bounds__(64)
k_seg_vote(const SegDesc *__restrict__ segs, int32_t nseg, DbView T, DbView R,
           const uint8_t *__restrict__ rrc, const int64_t *__restrict__ voff,
           uint32_t *__restrict__ dmat, int32_t bandmax, uint8_t *__restrict__ opbuf,
           uint16_t *__restrict__ nops_out, int32_t *__restrict__ status)
{
    extern __shared__ __align__(16) uint8_t smem[];
    const int32_t dp = blockIdx.x * blockDim.x + threadIdx.x;
    if (dp >= nseg) return;
    const SegDesc sg = segs[dp];
    const int32_t rl = sg.a1 - sg.a0, ql = sg.b1 - sg.b0;
    if (rl > SEG_MAX || ql > SEG_MAX || sg.band > bandmax || rl - ql >= sg.band || ql - rl >= sg.band) {
        atomicOr(status, DH_ST_POOL_OVERFLOW);
        nops_out[dp] = 0;
        return;
    }
    const uint8_t *ref = T.bases + T.off[sg.tmpl] + sg.a0;
    const uint8_t *qry = (sg.comp ? rrc : R.bases) + R.off[sg.bseq] + sg.b0;
    const int64_t NDP = nseg;
    // ---- banded fill (unit mismatch, indel 1, no free shift).  The trace already holds an
    // alignment of this tile with `diffs` differences, so the optimum D <= diffs.  A cell whose
    // true score is <= w is exact inside a band |i - j| <= w (its optimal path never leaves the
    // band); the traceback only ever selects neighbours with score <= D and w = diffs + 1 > D, so
    // fill + traceback restricted to the band give exactly the path of the full matrix
    // (findAlignment, util/string.d:478-520).  Cells outside the band count as 255.
    //
    // The traceback rule (tracebackScoringMatrix, util/string.d:775-831: smallest neighbour,
    // diagonal > insertion > deletion) depends only on the three neighbours the fill has in
    // registers, so the fill stores the 2-bit decision of every cell (16 per dword, interleaved
    // over the threads) instead of the scores.
    //
    // LDS: the rolling row in band coordinates c = j - i + w ([c][lane], one byte per cell:
    // conflict-free, updated in place because F[i-1][j] sits at c + 1), and the query packed
    // 8 bases per dword ([word][lane]).
    const int32_t w = sg.band < 1 ? 1 : sg.band;
    const int32_t WD = (2 * bandmax + 16) >> 4;  // decision dwords per matrix row
#define DM(i, cw) dmat[((int64_t)(i) * WD + (cw)) * NDP + dp]
    uint8_t *row = smem + threadIdx.x;
    uint32_t *qw = (uint32_t *)(smem + (size_t)(2 * bandmax + 2) * 64) + threadIdx.x;
    for (int32_t wd = 0; wd * 8 < ql; wd++) {
        uint32_t x = 0;
        for (int32_t b = 0; b < 8 && wd * 8 + b < ql; b++) x |= (uint32_t)(qry[wd * 8 + b] & 15u) << (4 * b);
        qw[wd * 64] = x;
    }
    for (int32_t c = 0; c <= 2 * w + 1; c++) {
        const int32_t j = c - w;  // F[0][j] = j
        row[c * 64] = (uint8_t)((j >= 0 && j <= ql && j < 255) ? j : 255);
    }
    for (int32_t i = 1; i <= rl; i++) {
        const uint32_t rc = ref[i - 1];
        int32_t cst, left;
        if (i <= w) {
            cst = w - i + 1;  // j = 1
            left = i;         // F[i][0]
        } else {
            cst = 0;
            left = 255;  // F[i][i-w-1] lies outside the band
        }
        int32_t diag = row[cst * 64];                // F[i-1][j-1] of the first cell
        if (i <= w) row[(w - i) * 64] = (uint8_t)i;  // F[i][0]: the next row's first diagonal
        const int32_t cend = (ql - i + w) < 2 * w ? (ql - i + w) : 2 * w;
        const int32_t jq = i - w + cst - 1;  // query index of the first cell
        int32_t widx = jq >> 3, inword = 8 - (jq & 7);
        uint32_t curw = qw[widx * 64] >> (4 * (jq & 7));
        uint32_t acc = 0;
        for (int32_t c = cst; c <= cend; c++) {
            const int32_t up = c == 2 * w ? 255 : (int32_t)row[(c + 1) * 64];
            const uint32_t qb = curw & 15u;
            curw >>= 4;
            if (--inword == 0) {
                widx++;
                curw = qw[widx * 64];
                inword = 8;
            }
            const int32_t m = diag + (rc == qb ? 0 : 1);
            int32_t v = m < up + 1 ? m : up + 1;
            v = v < left + 1 ? v : left + 1;
            v = v < 255 ? v : 255;
            const uint32_t op = (diag <= left && diag <= up) ? 0u : (left <= up ? 2u : 1u);
            acc |= op << (2 * (c & 15));
            if ((c & 15) == 15) {
                DM(i, c >> 4) = acc;
                acc = 0;
            }
            row[c * 64] = (uint8_t)v;
            diag = up;
            left = v;
        }
        if (cst <= cend && (cend & 15) != 15) DM(i, cend >> 4) = acc;
    }
    // ---- traceback: ops are produced back to front into the interleaved op buffer; opbuf row
    // t = op number t counted from the END of the path.
    const int32_t opcap = 2 * SEG_MAX;
    int32_t i = rl, j = ql, nops = 0;
    uint64_t oacc = 0;
    while (i > 0 && j > 0) {
        const int32_t c = j - i + w;
        const uint8_t op = (uint8_t)((DM(i, c >> 4) >> (2 * (c & 15))) & 3u);
        if (op == 0) {
            --i;
            --j;
        } else if (op == 2) {
            --j;
        } else {
            --i;
        }
        OP_PUT(op)
    }
    while (i > 0) {
        OP_PUT(1)
        --i;
    }
    while (j > 0) {
        OP_PUT(2)
        --j;
    }
    OP_FLUSH
    (void)opcap;
    nops_out[dp] = (uint16_t)nops;
#undef DM
}

// K8a, bit-parallel fill: the same Needleman-Wunsch tile and the same traceback rule as k_seg_vote, for tiles whose band
// (trace diffs + 1) is at most 31 (NW = 1) or 63 (NW = 2).  The banded fill is Hyyro's diagonal-band form of Myers'
// bit-vector recurrence -- the column step of DH-2 (dh_tile.h: tile_col) with the same conventions: band row R of matrix
// row i is query index j = i - 32 NW + R, D[0][j] = |j| and rows j <= 0 never match (which yields F[i][0] = i), the row
// above the band is missing, the row below it is a never-matching virtual row.  Computed scores are never below the true
// ones and exact wherever the true score is at most the band's half-width, which is all the argument in k_seg_vote needs:
// the traceback only ever selects neighbours scoring <= D <= diffs < band.  The 2-bit decision of every band cell
// (smallest neighbour; diagonal > insertion > deletion, util/string.d:775-831) depends on the three neighbours only
// through differences the step has as bit vectors:
//     left - diag = h(R - 1)   horizontal delta (HP / HN) of the row above         [left  = F[i][j-1]]
//     up   - diag = dd(R) - h(R)   with dd = 1 - D0                                 [up    = F[i-1][j]]
// so  op0 = (diag <= left) & (diag <= up) = ~HN' & ~(D0 & HP)   (X' = X << 1, the top row has no left: op0 bit 0 = B)
//     L   = (left <= up)  = [h(R - 1) + h(R) <= dd(R)]          (top row: 0)
// Two words per matrix row and plane are stored (op0, L), 64 NW cells each; about 90 (NW = 1) / 170 (NW = 2)
// instructions per matrix row instead of ~15 per band cell.
#ifdef DH_SEED_PROF
__device__ unsigned long long g_vote_prof[16];
#define VP(i)                                          \
    if (threadIdx.x == 0) {                            \
        const unsigned long long t_ = wall_clock64();  \
        atomicAdd(&g_vote_prof[i], t_ - tp_);          \
        tp_ = t_;                                      \
    }
#else
#define VP(i)
#endif

template <int NW>
struct BV {
    uint64_t w[NW];
};
template <int NW>
__device__ __forceinline__ BV<NW> bv_shl1(const BV<NW> &a)
{
    BV<NW> r;
#pragma unroll
    for (int k = NW - 1; k > 0; k--) r.w[k] = (a.w[k] << 1) | (a.w[k - 1] >> 63);
    r.w[0] = a.w[0] << 1;
    return r;
}
template <int NW>
__device__ __forceinline__ BV<NW> bv_shr1(const BV<NW> &a, uint64_t top)  // logical; `top` enters at the highest bit
{
    BV<NW> r;
#pragma unroll
    for (int k = 0; k < NW - 1; k++) r.w[k] = (a.w[k] >> 1) | (a.w[k + 1] << 63);
    r.w[NW - 1] = (a.w[NW - 1] >> 1) | (top << 63);
    return r;
}

template <int NW>
__global__ void __launch_bounds__(64)
k_seg_vote_bp(const SegDesc *__restrict__ segs, int32_t nseg, DbView T, DbView R, const uint8_t *__restrict__ rrc,
              uint64_t *__restrict__ dmat, uint8_t *__restrict__ opbuf, uint16_t *__restrict__ nops_out,
              int32_t *__restrict__ status)
{
    constexpr int HALF = 32 * NW;         // band rows above the diagonal
    constexpr int QW = (SEG_MAX + 63) / 64 + 1 + NW;  // plane words per tile: HALF bits of lead-in + the query
    __shared__ uint64_t s_pl[3][QW][64];  // planes of the query codes (bits 0, 1, 2), [word][lane]
    const int32_t dp = blockIdx.x * blockDim.x + threadIdx.x;
    if (dp >= nseg) return;
    const SegDesc sg = segs[dp];
    const int32_t rl = sg.a1 - sg.a0, ql = sg.b1 - sg.b0;
    if (rl > SEG_MAX || ql > SEG_MAX || sg.band > HALF - 1 || rl - ql >= sg.band || ql - rl >= sg.band) {
        atomicOr(status, DH_ST_POOL_OVERFLOW);
        nops_out[dp] = 0;
        return;
    }
    const uint8_t *ref = T.bases + T.off[sg.tmpl] + sg.a0;
    const uint8_t *qry = (sg.comp ? rrc : R.bases) + R.off[sg.bseq] + sg.b0;
    const int64_t NDP = nseg;
    const int lane = threadIdx.x;
#ifdef DH_SEED_PROF
    unsigned long long tp_ = wall_clock64();
#endif
    // ---- planes: bit (HALF + j - 1) of the plane string = bit of the code of query base j (1-based)
    {
        uint64_t a0 = 0, a1 = 0, a2 = 0;
        int32_t wi = HALF / 64, bi = HALF % 64;  // (NW = 1: word 0, bit 32; NW = 2: word 1, bit 0)
#pragma unroll
        for (int k = 0; k < QW; k++) s_pl[0][k][lane] = s_pl[1][k][lane] = s_pl[2][k][lane] = 0;
        // eight query bases per (unaligned) load, the next eight on their way: one byte at a time the loop waited for a
        // global load per base (the DBs carry 64 bytes of padding)
        uint64_t qnext = 0;
        if (ql > 0) __builtin_memcpy(&qnext, qry, 8);
        for (int32_t j0 = 0; j0 < ql; j0 += 8) {
            const uint64_t qw = qnext;
            if (j0 + 8 < ql) __builtin_memcpy(&qnext, qry + j0 + 8, 8);
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int32_t j = j0 + u;
                if (j >= ql) break;
                const uint64_t c = (qw >> (8 * u)) & 7u;
                a0 |= (c & 1u) << bi;
                a1 |= ((c >> 1) & 1u) << bi;
                a2 |= ((c >> 2) & 1u) << bi;
                if (++bi == 64 || j + 1 == ql) {
                    s_pl[0][wi][lane] = a0;
                    s_pl[1][wi][lane] = a1;
                    s_pl[2][wi][lane] = a2;
                    a0 = a1 = a2 = 0;
                    bi = 0;
                    wi++;
                }
            }
        }
    }
    VP(8)
    // window of matrix row i: bit R <-> j = i - HALF + R <-> plane string bit (i + R - 1); row 0 would start at bit -1,
    // the loop starts with row 1 at bit 0: the first NW words, and one new bit per row from the feed words
    BV<NW> p0, p1, p2;
#pragma unroll
    for (int k = 0; k < NW; k++) {
        p0.w[k] = s_pl[0][k][lane];
        p1.w[k] = s_pl[1][k][lane];
        p2.w[k] = s_pl[2][k][lane];
    }
    uint64_t f0 = s_pl[0][NW][lane], f1 = s_pl[1][NW][lane], f2 = s_pl[2][NW][lane];
    int32_t fword = NW, fleft = 64;
    // column 0 (dh_tile.h: tile_setup): vertical deltas aligned for row 1
    BV<NW> Pv, Mv, lv;
#pragma unroll
    for (int k = 0; k < NW; k++) {
        const uint64_t hi = k * 64 >= HALF ? ~0ull : (k * 64 + 64 <= HALF ? 0ull : ~0ull << (HALF - k * 64));
        Pv.w[k] = hi;
        Mv.w[k] = ~hi;
        const int hb = HALF + 1;  // lv of column 0: rows j >= 1
        lv.w[k] = k * 64 >= hb ? ~0ull : (k * 64 + 64 <= hb ? 0ull : ~0ull << (hb - k * 64));
    }
    // (measured, round 6: the decision words of a block's tiles as one contiguous piece, [block][row][word][lane], instead
    // of interleaved over the launch -- no difference: the fill is VALU-bound, ~150 instructions per matrix row)
#define DMW(i, k) dmat[((int64_t)(i) * (2 * NW) + (k)) * NDP + dp]
    uint64_t refw = 0;  // template bases i - 1 .. i + 6: one unaligned 8-byte load per 8 matrix rows (the DBs are padded)
    for (int32_t i = 1; i <= rl; i++) {
        if (((i - 1) & 7) == 0) __builtin_memcpy(&refw, ref + (i - 1), 8);
        const uint32_t rc = (uint32_t)(refw >> (8 * ((i - 1) & 7))) & 7u;
        const uint64_t x0 = 0ull - (uint64_t)(rc & 1u), x1 = 0ull - (uint64_t)((rc >> 1) & 1u), x2 = 0ull - (uint64_t)((rc >> 2) & 1u);
        // rows j >= 1 of this matrix row: arithmetic shift of the 64 NW-bit mask
        {
            const uint64_t top = lv.w[NW - 1] >> 63;
            lv = bv_shr1<NW>(lv, top);
        }
        BV<NW> Eq, D0, HP, HN;
        uint64_t carry = 0;
#pragma unroll
        for (int k = 0; k < NW; k++) {
            Eq.w[k] = ~((p0.w[k] ^ x0) | (p1.w[k] ^ x1) | (p2.w[k] ^ x2)) & lv.w[k];
            const uint64_t x = Eq.w[k] & Pv.w[k];
            const uint64_t s1 = x + Pv.w[k];
            const uint64_t c1 = s1 < x ? 1ull : 0ull;
            const uint64_t s2 = s1 + carry;
            const uint64_t c2 = s2 < s1 ? 1ull : 0ull;
            carry = c1 | c2;
            D0.w[k] = (s2 ^ Pv.w[k]) | Eq.w[k] | Mv.w[k];
            HP.w[k] = Mv.w[k] | ~(D0.w[k] | Pv.w[k]);
            HN.w[k] = Pv.w[k] & D0.w[k];
        }
        // decisions of the band cells of this row
        const BV<NW> HPs = bv_shl1<NW>(HP), HNs = bv_shl1<NW>(HN);
#pragma unroll
        for (int k = 0; k < NW; k++) {
            uint64_t A = ~HNs.w[k];
            const uint64_t B = ~(D0.w[k] & HP.w[k]);
            const uint64_t t2 = HPs.w[k] & HP.w[k];
            const uint64_t t1 = (HPs.w[k] & ~HP.w[k] & ~HN.w[k]) | (~HPs.w[k] & ~HNs.w[k] & HP.w[k]);
            uint64_t L = ~t2 & ~(t1 & D0.w[k]);
            if (k == 0) {
                A |= 1ull;
                L &= ~1ull;
            }
            DMW(i, k) = A & B;
            DMW(i, NW + k) = L;
        }
        // next row: vertical deltas one band row further down
        const BV<NW> Xv = bv_shr1<NW>(D0, 0ull);
#pragma unroll
        for (int k = 0; k < NW; k++) {
            Pv.w[k] = HN.w[k] | ~(Xv.w[k] | HP.w[k]);
            Mv.w[k] = HP.w[k] & Xv.w[k];
        }
        // slide the query window by one base
        p0 = bv_shr1<NW>(p0, f0 & 1ull);
        p1 = bv_shr1<NW>(p1, f1 & 1ull);
        p2 = bv_shr1<NW>(p2, f2 & 1ull);
        f0 >>= 1;
        f1 >>= 1;
        f2 >>= 1;
        if (--fleft == 0) {
            fword++;
            f0 = fword < QW ? s_pl[0][fword][lane] : 0ull;
            f1 = fword < QW ? s_pl[1][fword][lane] : 0ull;
            f2 = fword < QW ? s_pl[2][fword][lane] : 0ull;
            fleft = 64;
        }
    }
    VP(9)
    // ---- traceback (as k_seg_vote): ops back to front into the interleaved op buffer
    int32_t i = rl, j = ql, nops = 0;
    uint64_t oacc = 0;
    // (the decision words of a matrix row lie nseg * 8 bytes from the next row's: read one step at a time the walk was a
    // chain of dependent cache misses -- the words of the next PB rows are fetched together; a step stays in its row (op 2)
    // or moves to the next one)
    constexpr int PB = NW == 1 ? 8 : 4;
    while (i > 0 && j > 0) {
        uint64_t zr[PB][NW], lr[PB][NW];
        const int32_t i0 = i;
#pragma unroll
        for (int u = 0; u < PB; u++)
#pragma unroll
            for (int k = 0; k < NW; k++) {
                zr[u][k] = i0 - u > 0 ? DMW(i0 - u, k) : 0ull;
                lr[u][k] = i0 - u > 0 ? DMW(i0 - u, NW + k) : 0ull;
            }
#pragma unroll
        for (int u = 0; u < PB; u++) {
            while (i == i0 - u && i > 0 && j > 0) {
                const int32_t Rr = j - i + HALF;
                uint64_t z = zr[u][0], l = lr[u][0];
                if (NW > 1 && (Rr >> 6)) {
                    z = zr[u][NW - 1];
                    l = lr[u][NW - 1];
                }
                const uint8_t op = ((z >> (Rr & 63)) & 1ull) ? 0 : (((l >> (Rr & 63)) & 1ull) ? 2 : 1);
                if (op == 0) {
                    --i;
                    --j;
                } else if (op == 2) {
                    --j;
                } else {
                    --i;
                }
                OP_PUT(op)
            }
        }
    }
    while (i > 0) {
        OP_PUT(1)
        --i;
    }
    while (j > 0) {
        OP_PUT(2)
        --j;
    }
    OP_FLUSH
    nops_out[dp] = (uint16_t)nops;
    VP(10)
#ifdef DH_SEED_PROF
    if (threadIdx.x == 0) atomicAdd(&g_vote_prof[15], 1ull);
#endif
#undef DMW
}

// K8a (second half): per-column view of every tile from its op list, canonical indel placement,
// votes.  The per-thread column arrays live in LDS (one byte per entry) -- as private arrays they sat in scratch
// memory and every one of the ~700 dependent accesses of a tile paid an HBM-backed round trip.  A lane's three arrays
// are a row of the block's LDS ([lane][array][column], row stride an odd number of 8-byte words: lanes on the same
// column hit different banks), so that the passes after the op list read EIGHT columns with one load and go to single
// bytes only where a column holds an indel: column by column ([column][lane]) each of the ~400 reads of a tile was
// an LDS round trip in front of a branch (6 wavefronts per CU, 70 % of the wave cycles waiting).
// bytes of a lane's row: 3 arrays of seg_vote2_arr(ncolmax) bytes + padding
__host__ __device__ inline int32_t seg_vote2_arr(int32_t ncolmax) { return (ncolmax + 2 + 7) & ~7; }
__host__ __device__ inline int32_t seg_vote2_row(int32_t ncolmax)
{
    const int32_t s = 3 * seg_vote2_arr(ncolmax);
    return ((s >> 3) & 1) ? s : s + 8;
}
__device__ __forceinline__ bool has_byte5(uint64_t w)
{
    const uint64_t t = w ^ 0x0505050505050505ull;
    return (((t - 0x0101010101010101ull) & ~t) & 0x8080808080808080ull) != 0ull;
}
// bit i of the result = byte i of `t` is zero
__device__ __forceinline__ uint32_t zero_bytes8(uint64_t t)
{
    const uint64_t m = ~(((t & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | t) & 0x8080808080808080ull;
    // the four flag bits of a half (at 7, 15, 23, 31) land at 24 .. 27 of one 32-bit product, no two terms on one bit
    const uint32_t lo = (((uint32_t)m >> 7) * 0x01020408u) >> 24, hi = (((uint32_t)(m >> 32) >> 7) * 0x01020408u) >> 24;
    return (lo & 15u) | ((hi & 15u) << 4);
}
// bit i of the result = bit `k` of byte i
__device__ __forceinline__ uint32_t bit_plane8(uint64_t t, int k)
{
    const uint64_t m = (t >> k) & 0x0101010101010101ull;
    const uint32_t lo = ((uint32_t)m * 0x01020408u) >> 24, hi = ((uint32_t)(m >> 32) * 0x01020408u) >> 24;
    return (lo & 15u) | ((hi & 15u) << 4);
}
// a set of columns 0 .. 127 in two registers
struct ColSet {
    uint64_t lo, hi;
    __device__ __forceinline__ void put8(int32_t xb, uint32_t m8)  // xb a multiple of 8
    {
        if (xb < 64)
            lo |= (uint64_t)m8 << xb;
        else
            hi |= (uint64_t)m8 << (xb - 64);
    }
    __device__ __forceinline__ bool test(int32_t x) const { return ((x < 64 ? lo >> x : hi >> (x - 64)) & 1ull) != 0ull; }
    __device__ __forceinline__ void set(int32_t x)
    {
        if (x < 64)
            lo |= 1ull << x;
        else
            hi |= 1ull << (x - 64);
    }
    __device__ __forceinline__ void clear(int32_t x)
    {
        if (x < 64)
            lo &= ~(1ull << x);
        else
            hi &= ~(1ull << (x - 64));
    }
    __device__ __forceinline__ bool any() const { return (lo | hi) != 0ull; }
    __device__ __forceinline__ int32_t pop_lowest()
    {
        if (lo) {
            const int32_t x = __builtin_ctzll(lo);
            lo &= lo - 1;
            return x;
        }
        const int32_t x = __builtin_ctzll(hi);
        hi &= hi - 1;
        return 64 + x;
    }
    __device__ __forceinline__ void keep_below(int32_t n)  // columns 0 .. n - 1, n <= 128
    {
        if (n <= 64) {
            hi = 0ull;
            if (n < 64) lo &= (1ull << n) - 1ull;
        } else if (n < 128)
            hi &= (1ull << (n - 64)) - 1ull;
    }
};
__device__ __forceinline__ ColSet operator&(const ColSet &a, const ColSet &b) { return ColSet{a.lo & b.lo, a.hi & b.hi}; }
// where a walk to the left from column x ends that may step from st to st - 1 while column st - 1 is in `w`
// (`while (st > 0 && w[st - 1]) st--`): one more than the highest column below x that is not in w
__device__ __forceinline__ int32_t walk_left(const ColSet &w, int32_t x)
{
    ColSet t{~w.lo, ~w.hi};
    t.keep_below(x);
    if (t.hi) return 128 - __builtin_clzll(t.hi);
    if (t.lo) return 64 - __builtin_clzll(t.lo);
    return 0;
}

// NW8 > 0: tiles of up to 8 * NW8 - 2 <= 126 columns, the two canonical-placement passes on column sets in registers;
// NW8 == 0: any tile length, the passes byte by byte (the formulation the oracle has; DH_VOTE_BYTEWISE=1 selects it)
template <int NW8>
__global__ void __launch_bounds__(64)
k_seg_vote2(const SegDesc *__restrict__ segs, int32_t nseg, DbView T, DbView R,
            const uint8_t *__restrict__ rrc, const int64_t *__restrict__ voff,
            const uint8_t *__restrict__ opbuf, const uint16_t *__restrict__ nops_in, int32_t ncolmax,
            uint32_t *__restrict__ votes, uint32_t *__restrict__ cdiff, uint32_t *__restrict__ vother)
{
    extern __shared__ __align__(16) uint8_t smem[];
    const int32_t dp = blockIdx.x * blockDim.x + threadIdx.x;
    if (dp >= nseg) return;
    const SegDesc sg = segs[dp];
    const int32_t rl = sg.a1 - sg.a0;
    const int32_t nops = nops_in[dp];
    if (rl > ncolmax || nops == 0) return;  // flagged by k_seg_vote
    const uint8_t *ref = T.bases + T.off[sg.tmpl] + sg.a0;
    const uint8_t *qry = (sg.comp ? rrc : R.bases) + R.off[sg.bseq] + sg.b0;
    const int64_t NDP = nseg;
    // colst[x]: base aligned to column x (5 = deleted); ins[x]: bases inserted before column x --
    // count (0..5, 5 = more than MAXINS) in bits 0-2, bits 3-6 = "base t is one of ACGT";
    // ibp[x]: the first MAXINS inserted bases, 2 bits each.  3 bytes per column keep 8 blocks of 64
    // tiles resident per CU.
    const int32_t ARR = seg_vote2_arr(ncolmax);
    uint8_t *colst = smem + (size_t)threadIdx.x * seg_vote2_row(ncolmax);
    uint8_t *ins = colst + ARR;
    uint8_t *ibp = ins + ARR;
#ifdef DH_SEED_PROF
    unsigned long long tp_ = wall_clock64();
#endif
#define CS(x) colst[(x)]
#define IN(x) ins[(x)]
#define IB(x) ibp[(x)]
#define CS8(x) (*(const uint64_t *)(colst + (x)))
#define IN8(x) (*(const uint64_t *)(ins + (x)))
#define IB8(x) (*(const uint64_t *)(ibp + (x)))
    // the template bases of the tile, eight to a word (the DBs carry 64 bytes of padding): every load in flight before the
    // op list is walked
    uint64_t rwv[NW8 > 0 ? NW8 : 1];
    if (NW8 > 0) {
#pragma unroll
        for (int w = 0; w < NW8; w++) __builtin_memcpy(&rwv[w], ref + min(8 * w, rl & ~7), 8);  // (words behind the tile: unused)
    }
    for (int32_t x = 0; x <= rl; x += 8) {
        *(uint64_t *)(ins + x) = 0ull;
        *(uint64_t *)(ibp + x) = 0ull;
    }
    VP(0)
    {
        // eight ops per word (OP_WORD), the next word on its way while this one is applied; the query bases the eight ops
        // consume come from one unaligned 8-byte load (the DBs carry 64 bytes of padding)
        const uint64_t *opw = (const uint64_t *)opbuf;
        int32_t x = 0, y = 0;
        int32_t g = (nops - 1) >> 3;
        uint64_t wnext = opw[(int64_t)g * NDP + dp];
        uint64_t qnext;
        __builtin_memcpy(&qnext, qry, 8);
        // the insertion in front of column x is gathered in registers and stored when the column comes (its ops are
        // consecutive): read-modify-write of ins[x] / ibp[x] per op was two dependent LDS round trips at nearly every op
        // position of a wavefront
        uint32_t rn = 0, rv = 0, rb = 0;  // bases so far (0 .. 5), their "is one of ACGT" bits, the first MAXINS bases
        for (; g >= 0; g--) {
            const uint64_t ow = wnext, qw = qnext;
            const int32_t y0 = y;
            if (g > 0) {
                // the query bases of the next word start behind the ones this word consumes (every op but a deletion):
                // known from the word alone, so that the load does not wait for the eight ops
                const uint32_t nval = (uint32_t)min(8, nops - 8 * g);
                const uint32_t dels = zero_bytes8(ow ^ 0x0101010101010101ull) & ((1u << nval) - 1u);
                wnext = opw[(int64_t)(g - 1) * NDP + dp];
                __builtin_memcpy(&qnext, qry + y + ((int32_t)nval - __builtin_popcount(dels)), 8);
            }
#pragma unroll
            for (int u = 7; u >= 0; u--) {
                if (8 * g + u >= nops) continue;
                const uint8_t op = (uint8_t)(ow >> (8 * u));
                if (op <= 1) {
                    if (rn) {
                        IN(x) = (uint8_t)(rn | (rv << 3));
                        IB(x) = (uint8_t)rb;
                        rn = rv = rb = 0;
                    }
                    CS(x) = op == 0 ? (uint8_t)(qw >> (8 * (y - y0))) : (uint8_t)5;
                    y += op == 0 ? 1 : 0;
                    x++;
                } else {
                    const uint32_t q = (uint8_t)(qw >> (8 * (y - y0)));
                    if (rn < MAXINS) {
                        rb |= (q & 3u) << (2 * rn);
                        if (q < 4) rv |= 1u << rn;
                    }
                    if (rn < 5) rn++;
                    y++;
                }
            }
        }
        if (rn) {  // inserted behind the last column
            IN(x) = (uint8_t)(rn | (rv << 3));
            IB(x) = (uint8_t)rb;
        }
    }
    VP(1)
    // ---- canonical (leftmost) placement of indels inside homopolymer runs of the template.
    // A deleted column x of template base c moves to the start of the run of columns before it that hold c in the template
    // AND in the read, with nothing inserted in between; an insertion of n equal bases c before column x moves to the left
    // over columns that hold c in both, with nothing inserted.  Columns are visited in increasing order and a move only
    // changes columns before the visited one, so the set of visited columns is the set of indels the op list left.
    bool bytewise = NW8 == 0;
    ColSet cm{0, 0}, zi{0, 0};
    if (NW8 > 0) {
        // Round 6: byte by byte (below) a wavefront visited nearly every column -- some lane of the 64 has an indel there
        // -- and every visit and every step of its walk was a chain of dependent LDS / global byte loads: 91 of the 180 us of
        // a wavefront.  Here a lane holds the columns as sets in registers: cm = read base equals template base, zi = nothing
        // inserted before the column, the two bit planes of the template bases; a walk is one count-leading-zeros
        ColSet del{0, 0}, pl{0, 0}, ph{0, 0}, pn{0, 0};
#pragma unroll
        for (int w = 0; w < NW8; w++) {
            const int32_t xw = min(8 * w, ARR - 8);  // (ARR < 8 * NW8: the words behind the arrays are not read)
            const uint64_t cw = CS8(xw), iw = IN8(xw), rw = rwv[w];
            del.put8(8 * w, zero_bytes8(cw ^ 0x0505050505050505ull));
            cm.put8(8 * w, zero_bytes8(cw ^ rw));
            zi.put8(8 * w, zero_bytes8(iw & 0x0707070707070707ull));
            pl.put8(8 * w, bit_plane8(rw, 0));
            ph.put8(8 * w, bit_plane8(rw, 1));
            pn.put8(8 * w, zero_bytes8(rw & 0xFCFCFCFCFCFCFCFCull) ^ 0xFFu);
        }
        ColSet ntmpl = pn;
        ntmpl.keep_below(rl);
        // (a template base that is none of a / c / g / t inside the tile: the byte-wise passes)
        bytewise = ntmpl.any();
        if (!bytewise) {
            // e[c], the columns whose template base is c, from the two bit planes of the template bases
            auto tmpl_is = [&](int32_t c) {
                const uint64_t xl = (c & 1) ? 0ull : ~0ull, xh = (c & 2) ? 0ull : ~0ull;
                return ColSet{(pl.lo ^ xl) & (ph.lo ^ xh), (pl.hi ^ xl) & (ph.hi ^ xh)};
            };
            del.keep_below(rl);
            const ColSet zs{(zi.lo >> 1) | (zi.hi << 63), zi.hi >> 1};  // zs[p] = nothing inserted before column p + 1
            while (del.any()) {
                const int32_t x = del.pop_lowest();
                const int32_t c = (pl.test(x) ? 1 : 0) | (ph.test(x) ? 2 : 0);
                const int32_t st = walk_left(tmpl_is(c) & cm & zs, x);
                if (st < x) {
                    CS(st) = 5;
                    CS(x) = (uint8_t)c;
                    cm.clear(st);
                    cm.set(x);
                }
            }
            VP(2)
            ColSet vis{~zi.lo & ~1ull, ~zi.hi};
            vis.keep_below(rl + 1);
            while (vis.any()) {
                const int32_t x = vis.pop_lowest();
                const uint8_t v = IN(x), bits = IB(x), c = bits & 3;
                const int32_t n = v & 7;
                if (n > MAXINS) continue;
                // all n inserted bases are the same ACGT base
                const uint8_t want_valid = (uint8_t)(((1u << n) - 1u) << 3);
                bool same = (v & want_valid) == want_valid;
                for (int32_t t = 1; t < n; t++) same = same && ((bits >> (2 * t)) & 3) == c;
                if (!same) continue;
                const int32_t st = walk_left(tmpl_is(c) & cm & zi, x);
                if (st < x) {
                    IB(st) = bits;
                    IN(st) = v;
                    IN(x) = 0;
                    IB(x) = 0;
                    zi.clear(st);
                    zi.set(x);
                }
            }
        }
    }
    if (bytewise) {
        // A column only ever changes columns before it (and itself), so the eight bytes of a block stay valid while its
        // columns are visited.
        for (int32_t xb = 0; xb < rl; xb += 8) {
            const uint64_t w = CS8(xb);
            if (!has_byte5(w)) continue;
            for (int32_t j = 0; j < 8; j++) {
                const int32_t x = xb + j;
                if (x >= rl || (uint8_t)(w >> (8 * j)) != 5) continue;
                const uint8_t c = ref[x];
                int32_t st = x;
                while (st > 0 && CS(st - 1) == c && ref[st - 1] == c && (IN(st) & 7) == 0) st--;
                if (st < x) {
                    CS(st) = 5;
                    CS(x) = c;
                }
            }
        }
        for (int32_t xb = 0; xb <= rl; xb += 8) {
            const uint64_t w = IN8(xb);
            if ((w & 0x0707070707070707ull) == 0ull) continue;
            for (int32_t j = 0; j < 8; j++) {
                const int32_t x = xb + j;
                const uint8_t v = (uint8_t)(w >> (8 * j));
                const int32_t n = v & 7;
                if (x < 1 || x > rl || n == 0 || n > MAXINS) continue;
                const uint8_t bits = IB(x), c = bits & 3;
                // all n inserted bases are the same ACGT base
                const uint8_t want_valid = (uint8_t)(((1u << n) - 1u) << 3);
                bool same = (v & want_valid) == want_valid;
                for (int32_t t = 1; t < n; t++) same = same && ((bits >> (2 * t)) & 3) == c;
                if (!same) continue;
                int32_t st = x;
                while (st > 0 && CS(st - 1) == c && ref[st - 1] == c && (IN(st - 1) & 7) == 0) st--;
                if (st < x) {
                    IB(st) = bits;
                    IN(st) = v;
                    IN(x) = 0;
                    IB(x) = 0;
                }
            }
        }
    }
    VP(3)
    // ---- votes, sparse: a column whose read base equals the template base casts no atomic at
    // all -- the cover of a column comes from a difference array (+1 at the first column of the
    // tile, -1 behind its last one; k_votes_finish scans it) and the template-base count is what
    // is left of the cover after the explicit votes.  ~10x fewer atomics at 13 % error.
    const int64_t c0 = voff[sg.tmpl] + sg.a0;
    uint32_t *v = votes + c0 * VSTRIDE;
    atomicAdd(&cdiff[c0], 1u);
    atomicSub(&cdiff[c0 + rl], 1u);
    if (!bytewise) {
        // the columns that vote are known as sets (round 6): one atomic per lane and step instead of a walk over all the
        // columns with three divergent atomic sites -- 290 -> ~60 atomic instructions per wavefront
        ColSet mm{~cm.lo, ~cm.hi};
        mm.keep_below(rl);
        while (mm.any()) {
            const int32_t x = mm.pop_lowest();
            const uint8_t cs = CS(x);
            uint32_t *col = v + (int64_t)x * VSTRIDE;
            atomicAdd(cs == 5 ? &col[4] : (cs < 4 ? &col[cs] : &vother[c0 + x]), 1u);
        }
        ColSet iv{~zi.lo, ~zi.hi};
        iv.keep_below(rl + 1);
        while (iv.any()) {
            const int32_t x = iv.pop_lowest();
            const uint8_t ivb = IN(x), bits = IB(x);
            uint32_t *col = v + (int64_t)x * VSTRIDE;
            const int32_t ic = ivb & 7;
            const int32_t n = ic < MAXINS ? ic : MAXINS;
            for (int32_t t = 0; t < n; t++)
                if (ivb & (8u << t)) atomicAdd(&col[6 + 4 * t + ((bits >> (2 * t)) & 3)], 1u);
        }
    } else
    for (int32_t xb = 0; xb <= rl; xb += 8) {
        const uint64_t wi = IN8(xb), wb = IB8(xb), wc = CS8(xb);
        uint64_t rw;  // template bases xb .. xb + 7
        __builtin_memcpy(&rw, ref + xb, 8);
        // nothing inserted, every read base the template's a / c / g / t: no vote in these eight columns
        if (xb + 8 <= rl && wi == 0ull && wc == rw && (rw & 0xFCFCFCFCFCFCFCFCull) == 0ull) continue;
#pragma unroll
        for (int32_t j = 0; j < 8; j++) {
            const int32_t x = xb + j;
            if (x > rl) break;
            uint32_t *col = v + (int64_t)x * VSTRIDE;
            const uint8_t iv = (uint8_t)(wi >> (8 * j)), bits = (uint8_t)(wb >> (8 * j));
            const int32_t ic = iv & 7;
            const int32_t n = ic < MAXINS ? ic : MAXINS;
            for (int32_t t = 0; t < n; t++)
                if (iv & (8u << t)) atomicAdd(&col[6 + 4 * t + ((bits >> (2 * t)) & 3)], 1u);
            if (x == rl) break;
            const uint8_t cs = (uint8_t)(wc >> (8 * j)), rc = (uint8_t)(rw >> (8 * j));
            if (cs == rc && rc < 4) continue;
            if (cs == 5)
                atomicAdd(&col[4], 1u);
            else if (cs < 4)
                atomicAdd(&col[cs], 1u);
            else
                atomicAdd(&vother[c0 + x], 1u);
        }
    }
    VP(4)
#ifdef DH_SEED_PROF
    if (threadIdx.x == 0) atomicAdd(&g_vote_prof[7], 1ull);
#endif
#undef CS
#undef IN
#undef IB
#undef CS8
#undef IN8
#undef IB8
}

// column -> template map of the vote space (-1 for the spare column after each template): binary
// search of the column in the per-template offsets
__global__ void __launch_bounds__(256)
k_col_tmpl(const int64_t *__restrict__ voff, int32_t ntmpl, int64_t ncols_total, int32_t *__restrict__ col_tmpl)
{
    const int64_t gc = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gc >= ncols_total) return;
    int32_t lo = 0, hi = ntmpl;  // voff[lo] <= gc < voff[hi]
    while (hi - lo > 1) {
        const int32_t mid = (lo + hi) >> 1;
        if (voff[mid] <= gc)
            lo = mid;
        else
            hi = mid;
    }
    col_tmpl[gc] = gc < voff[lo + 1] - 1 ? lo : -1;
}

// completes the sparse votes of k_seg_vote2: cexcl = exclusive scan of the cover difference array
// (cover of column x = cexcl[x + 1]); the template base gets the votes of all covering tiles that
// voted for nothing else in this column
__global__ void __launch_bounds__(256)
k_votes_finish(DbView T, const int64_t *__restrict__ voff, const int32_t *__restrict__ col_tmpl,
               int64_t ncols_total, const uint32_t *__restrict__ cexcl, const uint32_t *__restrict__ vother,
               uint32_t *__restrict__ votes)
{
    const int64_t gc = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gc >= ncols_total) return;
    const int32_t t = col_tmpl[gc];
    if (t < 0) return;
    uint32_t *col = votes + gc * VSTRIDE;
    const uint32_t cover = cexcl[gc + 1];
    col[5] = cover;
    const uint8_t rc = T.bases[T.off[t] + (gc - voff[t])];
    if (rc < 4) col[rc] = cover - (col[0] + col[1] + col[2] + col[3]) - col[4] - vother[gc];
}

// ------------------------------------------------------------------------------------ K8b

// Emission in two kernels.  k_emit_runs: one thread per column; the first column of every
// homopolymer run of the template decides the whole run (run-length vote) and stages the
// symbols of each of its columns (at most ESTR per column).  k_emit_pack: one block per template
// scans the per-column counts and packs the staged symbols.
#define ESTR (1 + 2 * MAXINS)

__global__ void __launch_bounds__(256)
k_emit_runs(DbView T, int32_t ntmpl, const int64_t *__restrict__ voff, const uint32_t *__restrict__ votes,
            const int32_t *__restrict__ col_tmpl, int64_t ncols_total, uint8_t *__restrict__ stage,
            uint8_t *__restrict__ cnt)
{
    const int64_t gc = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // column in vote space
    if (gc >= ncols_total) return;
    const int32_t t = col_tmpl[gc];
    if (t < 0) return;  // the spare column after each template
    const int32_t rs = (int32_t)(gc - voff[t]);
    const uint8_t *ref = T.bases + T.off[t];
    const int32_t rlen = (int32_t)(T.off[t + 1] - T.off[t]);
    if (rs > 0 && ref[rs - 1] == ref[rs]) return;  // not the first column of its run
    const uint32_t *v = votes + voff[t] * VSTRIDE;
    int32_t re = rs + 1;
    while (re < rlen && ref[re] == ref[rs]) re++;
    const uint8_t c = ref[rs];
    const int64_t den = (int64_t)v[(int64_t)rs * VSTRIDE + 5] + 1;
    int64_t net = 0;
    int32_t ncols = 0;
    for (int32_t x = rs; x < re; x++) {
        const uint32_t *col = v + (int64_t)x * VSTRIDE;
        net += col[4];
        if (c < 4)
            for (int k = 0; k < MAXINS; k++) net -= col[6 + 4 * k + c];
        int best = c < 4 ? c : 0;
        uint32_t bv[4];
        for (int k = 0; k < 4; k++) bv[k] = col[k] + ((c == k) ? 1u : 0u);
        for (int k = 0; k < 4; k++)
            if (bv[k] > bv[best]) best = k;
        if (best == c) ncols++;
    }
    if (c < 4 && re < rlen)
        for (int k = 0; k < MAXINS; k++) net -= v[(int64_t)re * VSTRIDE + 6 + 4 * k + c];
    const int64_t adj = net >= 0 ? (2 * net + den) / (2 * den) : -((2 * (-net) + den) / (2 * den));
    int64_t target = (int64_t)ncols - adj;
    if (target < 0) target = 0;
    if (target > ncols + MAXINS) target = ncols + MAXINS;
    int64_t extra = target > ncols ? target - ncols : 0, keep = target < ncols ? target : ncols;
    for (int32_t x = rs; x < re; x++) {
        const uint32_t *col = v + (int64_t)x * VSTRIDE;
        uint8_t *o = stage + (voff[t] + x) * ESTR;
        int32_t n = 0;
        const uint32_t cover = col[5];
        const uint8_t pc = (x == rs && rs > 0) ? ref[rs - 1] : 255;
        for (int k = 0; k < MAXINS; k++) {
            const uint32_t *iv = col + 6 + 4 * k;
            uint32_t tot = 0;
            int best = -1;
            for (int b = 0; b < 4; b++) {
                if (b == c || b == pc) continue;
                tot += iv[b];
                if (best < 0 || iv[b] > iv[best]) best = b;
            }
            if (best < 0 || 2 * tot <= cover + 1) break;
            o[n++] = (uint8_t)best;
        }
        int best = c < 4 ? c : 0;
        uint32_t bv[4];
        for (int k = 0; k < 4; k++) bv[k] = col[k] + ((c == k) ? 1u : 0u);
        for (int k = 0; k < 4; k++)
            if (bv[k] > bv[best]) best = k;
        if (best != c) {
            if (2 * col[4] <= cover + 1) o[n++] = (uint8_t)best;
        } else {
            if (x == rs)
                for (int64_t e = 0; e < extra; e++) o[n++] = c;
            if (keep > 0) {
                o[n++] = c;
                keep--;
            }
        }
        cnt[voff[t] + x] = (uint8_t)n;
    }
}

__global__ void __launch_bounds__(256)
k_emit_pack(DbView T, int32_t ntmpl, const int64_t *__restrict__ voff, const uint8_t *__restrict__ stage,
            const uint8_t *__restrict__ cnt, const int64_t *__restrict__ out_off,
            uint8_t *__restrict__ out, int32_t *__restrict__ out_len)
{
    __shared__ int32_t part[256];
    __shared__ int32_t carry;
    const int32_t t = blockIdx.x;
    if (t >= ntmpl) return;
    const int32_t rlen = (int32_t)(T.off[t + 1] - T.off[t]);
    const uint8_t *cn = cnt + voff[t];
    const uint8_t *st = stage + voff[t] * ESTR;
    uint8_t *o = out + out_off[t];
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int32_t base = 0; base < rlen; base += 256) {
        const int32_t x = base + threadIdx.x;
        const int32_t my = x < rlen ? cn[x] : 0;
        part[threadIdx.x] = my;
        __syncthreads();
        for (int s = 1; s < 256; s <<= 1) {
            const int32_t add = (int)threadIdx.x >= s ? part[threadIdx.x - s] : 0;
            __syncthreads();
            part[threadIdx.x] += add;
            __syncthreads();
        }
        const int32_t off = carry + part[threadIdx.x] - my;
        for (int32_t e = 0; e < my; e++) o[off + e] = st[(int64_t)x * ESTR + e];
        __syncthreads();
        if (threadIdx.x == 255) carry += part[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) out_len[t] = carry;
}

// ------------------------------------------------------------------------------------ launchers

extern "C" {

void dhk_gather_slices(hipStream_t st, const uint8_t *src, const int64_t *src_off, const int32_t *sidx,
                       const int32_t *sbeg, const int64_t *dst_off, int32_t n, int32_t max_len,
                       uint8_t *dst)
{
    if (n <= 0) return;
    int gx = (max_len + 255) / 256;
    gx = gx < 1 ? 1 : (gx > 64 ? 64 : gx);
    for (int32_t s0 = 0; s0 < n; s0 += 65535) {
        const int32_t cnt = n - s0 < 65535 ? n - s0 : 65535;
        hipLaunchKernelGGL(k_gather_slices, dim3(gx, cnt), dim3(256), 0, st, src, src_off, sidx + s0,
                           sbeg + s0, dst_off + s0, cnt, dst);
    }
}

// ranges of 16-bit trace values copied into one compact array: a wavefront per range (desc = src offset, dst offset, length)
__global__ void __launch_bounds__(256)
k_gather_ranges16(const uint16_t *__restrict__ src, const int64_t *__restrict__ desc, int32_t n, uint16_t *__restrict__ dst)
{
    const int32_t r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const int64_t so = desc[3 * (int64_t)r], d0 = desc[3 * (int64_t)r + 1], len = desc[3 * (int64_t)r + 2];
    for (int64_t e = threadIdx.x & 63; e < len; e += 64) dst[d0 + e] = src[so + e];
}

void dhk_gather_ranges16(hipStream_t st, const uint16_t *src, const int64_t *desc, int32_t n, uint16_t *dst)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_gather_ranges16, dim3((uint32_t)((n + 3) / 4)), dim3(256), 0, st, src, desc, n, dst);
}

void dhk_gather_parts(hipStream_t st, const uint8_t *src0, const int64_t *off0, const uint8_t *src1,
                      const int64_t *off1, const void *parts, int32_t n, int32_t max_len, uint8_t *dst)
{
    if (n <= 0) return;
    int gx = (max_len + 255) / 256;
    gx = gx < 1 ? 1 : (gx > 64 ? 64 : gx);
    for (int32_t s0 = 0; s0 < n; s0 += 65535) {
        const int32_t cnt = n - s0 < 65535 ? n - s0 : 65535;
        hipLaunchKernelGGL(k_gather_parts, dim3(gx, cnt), dim3(256), 0, st, src0, off0, src1, off1,
                           (const PartDesc *)parts + s0, cnt, dst);
    }
}

void dhk_pile_funnel(hipStream_t st, DhLa *las, const uint32_t *item_off, int32_t nreads, const int64_t *roff,
                     int32_t max_err_ppm, int32_t tsp, int32_t *la_first, int32_t *live, int32_t *status)
{
    if (nreads <= 0) return;
    hipLaunchKernelGGL(k_pile_funnel, dim3(nreads), dim3(64), 0, st, las, item_off, nreads, roff, max_err_ppm, tsp, la_first,
                       live, status);
}

void dhk_gather_read_records(hipStream_t st, const DhLa *las, const int32_t *la_first, const int32_t *sel,
                             const int32_t *dst_off, int32_t nsel, DhLa *out)
{
    if (nsel <= 0) return;
    hipLaunchKernelGGL(k_gather_read_records, dim3(nsel), dim3(64), 0, st, las, la_first, sel, dst_off, nsel, out);
}

void dhk_tile_qv(hipStream_t st, const DhLa *las, const uint16_t *trace, const int32_t *la_first,
                 const int64_t *roff, int32_t nreads, int32_t tspace, const int32_t *cov, int32_t maxtiles,
                 uint8_t *qv)
{
    if (nreads <= 0) return;
    hipLaunchKernelGGL(k_tile_qv, dim3(nreads), dim3(64), 0, st, las, trace, la_first, roff, nreads,
                       tspace, cov, maxtiles, qv);
}

// ncolmax: longest tile on the template side (trace spacing of the pile-up alignments);
// cdiff / vother: zeroed uint32 arrays over the vote columns (+ 2); call dhk_votes_finish after the
// last launch of a round
void dhk_seg_vote(hipStream_t st, const void *segs, int32_t nseg, DbView T, DbView R,
                  const uint8_t *rrc, const int64_t *voff, uint32_t *dmat, int32_t bandmax, int32_t qmax,
                  int32_t ncolmax, uint8_t *opbuf, uint16_t *nops, uint32_t *votes, uint32_t *cdiff,
                  uint32_t *vother, int32_t *status, int32_t mode)
{
    if (nseg <= 0) return;
    // mode 1 / 2: bit-parallel fill with one / two 64-cell words per matrix row (bands up to 31 / 63); 0: scalar fill
    if (mode == 1)
        hipLaunchKernelGGL(k_seg_vote_bp<1>, dim3((nseg + 63) / 64), dim3(64), 0, st, (const SegDesc *)segs, nseg, T, R, rrc,
                           (uint64_t *)dmat, opbuf, nops, status);
    else if (mode == 2)
        hipLaunchKernelGGL(k_seg_vote_bp<2>, dim3((nseg + 63) / 64), dim3(64), 0, st, (const SegDesc *)segs, nseg, T, R, rrc,
                           (uint64_t *)dmat, opbuf, nops, status);
    else {
        const size_t lds = (size_t)(2 * bandmax + 2) * 64 + ((size_t)(qmax + 7) / 8 + 1) * 256;
        hipLaunchKernelGGL(k_seg_vote, dim3((nseg + 63) / 64), dim3(64), lds, st, (const SegDesc *)segs, nseg,
                           T, R, rrc, voff, dmat, bandmax, opbuf, nops, status);
    }
    const size_t lds2 = (size_t)seg_vote2_row(ncolmax) * 64;
    // the canonical-placement passes on column sets in registers: tiles up to 102 / 126 columns (the trace spacing of the
    // pile-up alignments is 100); longer tiles, or DH_VOTE_BYTEWISE=1, byte by byte
    const char *ev = getenv("DH_VOTE_BYTEWISE");
    const bool bytewise = ev && atoi(ev) != 0;
    const int32_t nw8 = seg_vote2_arr(ncolmax) / 8;
    auto launch = [&](auto kern) {
        if (lds2 > 65536) (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
        hipLaunchKernelGGL(kern, dim3((nseg + 63) / 64), dim3(64), lds2, st, (const SegDesc *)segs, nseg, T, R, rrc, voff,
                           (const uint8_t *)opbuf, (const uint16_t *)nops, ncolmax, votes, cdiff, vother);
    };
    if (bytewise || nw8 > 16)
        launch(k_seg_vote2<0>);
    else if (nw8 <= 13)
        launch(k_seg_vote2<13>);
    else
        launch(k_seg_vote2<16>);
#ifdef DH_SEED_PROF
    if (getenv("DH_TRACE") && nseg > 100000) {
        (void)hipStreamSynchronize(st);
        unsigned long long h[16], z[16] = {0};
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_vote_prof), sizeof h);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_vote_prof), z, sizeof z);
        const double w = h[7] ? (double)h[7] : 1.0, wb = h[15] ? (double)h[15] : 1.0;
        fprintf(stderr, "[vote prof] fill mode %d, %.0f wavefronts timed: planes %.1f fill %.1f traceback %.1f us/wavefront\n", mode, wb,
                h[8] / wb / 100.0, h[9] / wb / 100.0, h[10] / wb / 100.0);
        fprintf(stderr, "[vote prof] %d tiles, %.0f wavefronts timed: init %.1f ops %.1f canon-del %.1f canon-ins %.1f votes %.1f us/wavefront\n",
                nseg, w, h[0] / w / 100.0, h[1] / w / 100.0, h[2] / w / 100.0, h[3] / w / 100.0, h[4] / w / 100.0);
    }
#endif
}

void dhk_col_tmpl(hipStream_t st, const int64_t *voff, int32_t ntmpl, int64_t ncols_total, int32_t *col_tmpl)
{
    if (ncols_total <= 0 || ntmpl <= 0) return;
    hipLaunchKernelGGL(k_col_tmpl, dim3((unsigned)((ncols_total + 255) / 256)), dim3(256), 0, st, voff, ntmpl,
                       ncols_total, col_tmpl);
}

void dhk_votes_finish(hipStream_t st, DbView T, const int64_t *voff, const int32_t *col_tmpl, int64_t ncols_total,
                      const uint32_t *cexcl, const uint32_t *vother, uint32_t *votes)
{
    if (ncols_total <= 0) return;
    hipLaunchKernelGGL(k_votes_finish, dim3((unsigned)((ncols_total + 255) / 256)), dim3(256), 0, st, T, voff,
                       col_tmpl, ncols_total, cexcl, vother, votes);
}

void dhk_emit(hipStream_t st, DbView T, int32_t ntmpl, const int64_t *voff, const uint32_t *votes,
              const int32_t *col_tmpl, int64_t ncols_total, uint8_t *stage, uint8_t *cnt,
              const int64_t *out_off, uint8_t *out, int32_t *out_len)
{
    if (ntmpl <= 0) return;
    hipLaunchKernelGGL(k_emit_runs, dim3((unsigned)((ncols_total + 255) / 256)), dim3(256), 0, st, T, ntmpl,
                       voff, votes, col_tmpl, ncols_total, stage, cnt);
    hipLaunchKernelGGL(k_emit_pack, dim3(ntmpl), dim3(256), 0, st, T, ntmpl, voff, stage, cnt, out_off, out,
                       out_len);
}

}  // extern "C"
