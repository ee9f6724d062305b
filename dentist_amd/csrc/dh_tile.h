// dh_tile.h -- DH-2, the tile-by-tile banded bit-parallel extension: one alignment per LANE.
//
// Role: the local-alignment arithmetic of `damapper` / `daligner -A` (call sites
// source/dentist/dazzler.d:6121-6170; argv source/dentist/commandline.d:2918-2955), selected with
// dh_align_opts.algo = 1.  DESIGN.md section 3 ("DH-2") states the algorithm; oracle/align.c
// (extend_tiled) is its plain-DP restatement and the parity fence.
//
// This header is the whole per-lane state machine -- work fetch, candidate loop, tile set-up, the
// column step (Hyyro's diagonal-band form of Myers' bit-vector recurrence on one 64-bit word), the
// choice of the next tile's origin, trace pairs, records -- written once as host/device code:
//   * k_tile (dh_tile.hip) runs it with 64 lanes per wavefront, the column loop in lock step;
//   * tests/native/tile_host.cpp compiles the same functions for the CPU, so that `-m "not gpu"`
//     tests compare this very code with the oracle (no GPU needed to find an arithmetic slip).
#ifndef DH_TILE_H
#define DH_TILE_H

#include <stdint.h>

#include "dh_device.h"

#if defined(__HIPCC__)
#define DH_HD __host__ __device__ __forceinline__
#else
#define DH_HD inline
#endif

namespace dhtile {

constexpr int W = 64;        // band rows = bits of one vector (the default band; a band of 32 rows runs on 32-bit vectors:
                             // half the instructions per column, +- 16 diagonals of drift per tile instead of +- 32)
constexpr int TS_MAX = 128;  // longest tile (trace spacing) the per-tile buffers hold
constexpr int NQ = 7;        // plane words of a tile: window bits [0, 224) >= (TS_MAX - 1) + 64 + 31
template <int WB> struct BandVec;
template <> struct BandVec<64> {
    typedef uint64_t U;
    typedef int64_t S;
};
template <> struct BandVec<32> {
    typedef uint32_t U;
    typedef int32_t S;
};
constexpr int NAW = 9;       // raw packed-A words of a tile: 2 * TS_MAX bits + 30 bits of misalignment
constexpr int MAXREG = 64;   // aligned regions remembered per (read, strand) item (as DH-1)
constexpr int REGF = 8;      // ints per region (7 used)

enum { L_FETCH = 0, L_CAND = 1, L_RUN = 2, L_EXT_END = 3, L_DONE = 4 };

struct PlanePair {  // 32 bases of a plane-packed copy: bit g & 31 of .x / .y = low / high bit of base g
    uint32_t x, y;
};

struct Cold;

// work unit of a symmetric launch (k_units_fat): one group of candidates of an item, with everything the lane needs to
// start on the first of them -- fetching a unit is ONE memory round trip (four 16-byte loads) instead of the chain
// unit -> item -> candidate -> A offsets
struct Unit {
    int32_t it, c0, c1, blen;      // item - item0, candidates [c0, c1) of it, length of the item's read
    int64_t bo, ao;                // base offsets of the read and of the first candidate's A sequence
    int32_t aseq, apos, bpos, alen;  // the first candidate and the length of its A sequence
    int32_t cbase;                   // first candidate slot of the item (Params.candoff[item])
    int32_t nd0;                     // alignments of the item attempted before this unit (the cap of MAXREG attempts per item
                                     // counts them: k_units_fat splits an item of more than MAXREG candidates)
    int32_t rest;                    // 1: the unit is the REST of such an item -- its candidates in order, minus the candidates
                                     // below index MAXREG that are alone in their group (units of their own; each counts as
                                     // one attempted alignment when the walk passes it)
    int32_t pad_[1];
};

struct Params {
    const int64_t *aoff, *boff;       // offsets of the A / B sequences (bases)
    const uint32_t *apk, *arcpk;      // 2-bit packed A, forward / reverse complement: base g in dword g >> 4, bits 2 (g & 15)
    const PlanePair *bpp, *brcpp;     // plane-packed B, forward / reverse complement (indexed by absolute base >> 5)
    // the transposed pair (mode 1: A'' = a B sequence, B'' = an A sequence): 2-bit packed B, plane-packed A.  A symmetric
    // launch (A == B) passes the same copies here; a mapping with the transposed file (`damapper -C`) passes the second
    // set of copies and the second set of outputs below; otherwise NULL
    const uint32_t *apk1, *arcpk1;
    const PlanePair *bpp1, *brcpp1;
    DhOpts o;
    int32_t item0, nitems;            // items (read * 2 + strand) [item0, item0 + nitems)
    const DhCand *cand;               // max_cand per item, indexed by absolute item
    const int32_t *ncand;
    uint32_t *queue;                  // work counter
    const Unit *units;                // optional work units (symmetric launches), else NULL: the items are the units
    const uint32_t *nunits;           // their number (device side)
    // symmetric mode: RECORDS IN PLACE.  Candidate c of item i owns the two record slots 2 (candoff[i] + c) + mode
    // (mode 0: the pair as seeded, 1: the transposed pair) of out_la / out_trace: the trace pairs are written where
    // the compaction reads them while the alignment runs, the record is one plain store (pad = 1 marks it valid,
    // the slots are zeroed before the launch).  No slot is claimed, nothing is copied, no atomic sits in a lane's
    // way -- the claims were returning device-scope atomics on the counter of one read, hit by every lane that
    // aligned a partner of that read at about the same time, and with in-order vmcnt every later load of the
    // wavefront queued behind them: 64 % of the wave cycles of the pile-up launch (clocks: profiles/r04_*).
    // The records are counted and grouped by A read afterwards (k_rec_count / k_rec_scatter / k_compact_sym).
    const int32_t *candoff;           // per item (absolute): exclusive prefix sum of max(ncand, 0); NULL otherwise
    int32_t *regs;                    // nlanes * MAXREG * REGF
    Cold *cold;                       // nlanes: the lanes' cold state
    int32_t book_min;                 // lanes that must be waiting before a wavefront does a bookkeeping pass (k_tile)
    int32_t qbatch;                   // work units a wavefront takes from the queue per atomic (k_tile; 0 or 1: what its lanes need)
    int32_t nbmax, trmax;             // pairs a direction can yield; u16 values per output slot (>= 2 * (2 * nbmax + 2))
    DhLa *out_la;                     // max_la records per item (absolute item index); symmetric mode: two per candidate
    uint16_t *out_trace;              // trmax values per record slot
    int32_t *out_nla, *out_ntr;       // per item (not used by symmetric launches)
    DhLa *out_la2;                    // transposed records of a mapping (not symmetric mode), laid out like out_la; or NULL
    uint16_t *out_trace2;
    int32_t *out_nla2, *out_ntr2;
    unsigned long long *counters;     // [0] band cells computed, [1] candidates aligned
    int32_t *status;
    const uint8_t *pflags;            // symmetric mode, optional (DbView::pflags): which records are wanted
    int32_t tandem;                   // dh_align_opts.skip_self == 3: a read against itself below the main diagonal (Tile::dm)
};

// one running extension (registers)
struct Ext {
    int64_t ga, gb;           // absolute base index of A'[0] / B'[0] in the copies of this direction
    int32_t src;              // bit 0: A' from the reverse-complement copy, bit 1: B' from the reverse-complement copy, bit 2: transposed pair
    int32_t an, bn, tp_first;
    int32_t a0, b0, dsum, ntp;
    int32_t best_s, best_a, best_b, best_d, best_nseg;  // best end so far; nseg = trace segments up to it
    int32_t klo, khi, bklo, bkhi;                       // diagonal excursion: path so far / up to the best end
    uint32_t pair1, best_pair1;                         // first segment (diffs << 16 | bbases): as computed / of the best end
    uint16_t *pairs;                                    // where the trace pairs of the alignment in flight go
};

// what a lane touches only between extensions (next candidate, end of an extension, records): it lives
// in memory -- one record per lane slot -- so that the tile loop keeps its registers
struct Cold {
    // item
    int32_t item, strand, nc, c, blen, nd, nacc, ntr;
    int64_t bo;
    // candidate
    int32_t c_aseq, as, bs, alen;
    int64_t ao;
    // the alignment in flight: mode 0 = the candidate as seeded, 1 = the transposed pair (symmetric mode);
    // g_*: its A / B sequences (offset, length) and seed point
    int32_t mode, g_alen, g_as, g_blen, g_bs;
    int64_t g_ao, g_bo;
    // result of the reverse extension
    int32_t rv_i, rv_j, rv_d, rv_nseg, rv_klo, rv_khi;
    uint32_t rv_pair1;
    int32_t nacc2, ntr2;  // transposed records of the item (mapping with the transposed file)
    int32_t cbase;        // symmetric mode: first candidate slot of the item
    int32_t nd0, rest;    // work units: alignments of the item attempted before this unit / outside it (Unit::nd0, Unit::rest)
};

struct Lane {
#if defined(DH_SEED_PROF) && defined(__HIPCC__)
    unsigned long long pt[6];  // development: cycles inside the sections of lane_ext_end
#endif
#if defined(DH_SEED_PROF) && defined(__HIP_DEVICE_COMPILE__)
#define LP_BEGIN unsigned long long lp_ = clock64();
#define LP(i) { const unsigned long long t_ = clock64(); l.pt[i] += t_ - lp_; lp_ = t_; }
#else
#define LP_BEGIN
#define LP(i)
#endif
    int32_t st;
    int32_t dir;   // 1 = reverse extension (runs first), 0 = forward
    int32_t roff;  // 1: the seed is not on a trace boundary (the two first tiles share a trace interval)
    int32_t slot;  // lane slot (scratch index)
    int32_t err;
    uint32_t naln;
    uint64_t cells;
    Ext e;
    Cold *c;
};

// the tile in flight
template <int WB>
struct TileT {
    typename BandVec<WB>::U Pv, Mv, lv, wild;
    typename BandVec<WB>::U dm;  // tandem mode only: the rows of the band that may match (B's base before A's on the read)
    int32_t z, dbot, cols, bnr, T;
};
typedef TileT<64> Tile;
// the sequence words of a tile, NTW dwords per lane (registers on the host, LDS on the device):
//   [0, NQ)        low-bit plane of B': bit x of the 224-bit string = B'[b0 - 32 + x]
//   [NQ, 2 NQ)     high-bit plane
//   [2 NQ, NTW)    A'[a0 + x] at bits 2x of the 256-bit string
constexpr int NTW = 2 * NQ + NAW - 1;

DH_HD uint32_t funnel(uint32_t hi, uint32_t lo, uint32_t sh)  // ({hi, lo} >> sh)[31:0], sh in [0, 31]
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, sh);
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (sh & 31u));
#endif
}

DH_HD int32_t nbound(int32_t x, int32_t first, int32_t ts) { return x >= first ? (x - first) / ts + 1 : 0; }

// ------------------------------------------------------------------------------ extension

DH_HD uint16_t *lane_pairs(const Lane &l, const Params &P);

DH_HD void ext_begin(Lane &l, const Params &P, int32_t dir)
{
    Ext &e = l.e;
    const Cold &c = *l.c;
    l.dir = dir;
    const int32_t ts = P.o.tspace, as = c.g_as, bs = c.g_bs;
    // the reverse extension is a forward extension over the reverse-complemented copies
    e.ga = c.g_ao + (dir ? c.g_alen - as : as);
    e.gb = c.g_bo + (dir ? c.g_blen - bs : bs);
    e.an = dir ? as : c.g_alen - as;
    e.bn = dir ? bs : c.g_blen - bs;
    const bool brc_side = (c.strand != 0) != (dir != 0);
    e.src = (dir ? 1 : 0) | (brc_side ? 2 : 0) | (c.mode ? 4 : 0);
    e.tp_first = dir ? ((as % ts) ? (as % ts) : ts) : ts - (as % ts);
    e.a0 = e.b0 = e.dsum = e.ntp = 0;
    e.best_s = e.best_a = e.best_b = e.best_d = e.best_nseg = 0;
    e.klo = e.khi = e.bklo = e.bkhi = 0;
    e.pair1 = e.best_pair1 = 0;
    e.pairs = lane_pairs(l, P);
    l.st = (e.an > 0 && e.bn > 0) ? L_RUN : L_EXT_END;
}

template <int WB>
DH_HD void tile_setup(const Lane &l, const Params &P, TileT<WB> &t, uint32_t *q)
{
    typedef typename BandVec<WB>::U V;
    constexpr int W = WB;
    const Ext &e = l.e;
    t.T = e.ntp == 0 ? e.tp_first : P.o.tspace;
    const int32_t anr = e.an - e.a0;
    t.bnr = e.bn - e.b0;
    t.cols = t.T < anr ? t.T : anr;
    // column 0: D[0][j] = |j| for the band rows j = i - W/2; vectors aligned for column 1
    t.Pv = (V)(~(V)0 << (W / 2));
    t.Mv = (V)~t.Pv;
    t.dbot = W / 2 - 1;
    t.lv = (V)(~(V)0 << (W / 2 + 1));  // rows j >= 1 of column 0
    const int32_t th = t.bnr + W / 2 + 1;  // first bit of column 0 past the end of B'
    t.wild = th >= W ? (V)0 : (V)(~(V)0 << th);
    t.z = t.bnr - W / 2 + 1;
    // tandem mode (dh_align_opts.skip_self == 3, datander: A' and B' are the same read): a cell in which B's base is not
    // BEFORE A's on the read never matches, so that the alignment of a read with itself stays below the main diagonal.  Row
    // i of any column of the tile is base b = a + i - W/2 - delta with delta = (ga + a0) - (gb + b0) (read-relative), the (signed) distance
    // of the tile's origin from the main diagonal in the copies at hand: forward the rows i < delta + W/2 are allowed;
    // backward the copies are mirrored (delta < 0, B's base must come AFTER A's in them): the rows i > W/2 + delta.
    t.dm = (V)~(V)0;
    if (P.tandem) {
        const int64_t delta = (e.ga - l.c->g_ao + e.a0) - (e.gb - l.c->g_bo + e.b0);
        const int64_t nlow = l.dir ? W / 2 + delta + 1 : delta + W / 2;
        const V low = nlow <= 0 ? (V)0 : (nlow >= W ? (V)~(V)0 : (V)(((V)1 << nlow) - (V)1));
        t.dm = l.dir ? (V)~low : low;
    }
    // B planes: bit x of the window string = base (gb + b0 - W/2 + x)
    {
        const int64_t g = e.gb + e.b0 - W / 2;
        const PlanePair *p = ((e.src & 4) ? ((e.src & 2) ? P.brcpp1 : P.bpp1) : ((e.src & 2) ? P.brcpp : P.bpp)) + (g >> 5);
        const uint32_t s = (uint32_t)(g & 31);
        PlanePair r[NQ + 1];
#pragma unroll
        for (int i = 0; i <= NQ; i++) r[i] = p[i];
#pragma unroll
        for (int i = 0; i < NQ; i++) {
            q[i] = funnel(r[i + 1].x, r[i].x, s);
            q[NQ + i] = funnel(r[i + 1].y, r[i].y, s);
        }
    }
    // A: base x of the tile = base (ga + a0 + x)
    {
        const int64_t g = e.ga + e.a0;
        const uint32_t *p = ((e.src & 4) ? ((e.src & 1) ? P.arcpk1 : P.apk1) : ((e.src & 1) ? P.arcpk : P.apk)) + (g >> 4);
        const uint32_t s = (uint32_t)(g & 15) << 1;
        uint32_t r[NAW];
#pragma unroll
        for (int i = 0; i < NAW; i++) r[i] = p[i];
#pragma unroll
        for (int i = 0; i < NAW - 1; i++) q[2 * NQ + i] = funnel(r[i + 1], r[i], s);
    }
}

// one column of the band: c = 1 .. cols.  (p0, p1) = plane windows of the column: bit i = low / high bit
// of base B'[b0 + c + i - W/2 - 1], the base a path consumes to reach row i of the column; x = A'[a0 + c - 1]
template <bool TAN = false, int WB = 64>
DH_HD void tile_col(TileT<WB> &t, typename BandVec<WB>::U p0, typename BandVec<WB>::U p1, uint32_t x)
{
    typedef typename BandVec<WB>::U V;
    typedef typename BandVec<WB>::S SV;
    const V x0 = (V)0 - (V)(x & 1u), x1 = (V)0 - (V)((x >> 1) & 1u);
    // rows before the tile's origin never match (lv), rows past the end of B' match everything (wild)
    t.lv = (V)((SV)t.lv >> 1);
    t.z -= 1;
    t.wild = (V)((V)((SV)t.wild >> 1) | (V)((V)((uint32_t)t.z >> 31) << (WB - 1)));
    const V Eq = (V)(((V)~((p0 ^ x0) | (p1 ^ x1)) & (TAN ? (V)(t.lv & t.dm) : t.lv)) | t.wild);
    const V Pv = t.Pv, Mv = t.Mv;
    const V D0 = (V)((((V)((V)(Eq & Pv) + Pv)) ^ Pv) | Eq | Mv);
    const V HP = (V)(Mv | (V)~(D0 | Pv)), HN = (V)(Pv & D0);
    const V Xv = (V)(D0 >> 1);
    t.Pv = (V)(HN | (V)~(Xv | HP));
    t.Mv = (V)(HP & Xv);
    t.dbot += 1 - (int32_t)(D0 >> (WB - 1));
}

// the columns of a tile in sequence (host; the device kernel runs the same steps in lock step)
template <int WB = 64>
DH_HD void tile_window(const uint32_t *q, int32_t c, typename BandVec<WB>::U &p0, typename BandVec<WB>::U &p1, uint32_t &x)
{
    typedef typename BandVec<WB>::U V;
    const int32_t cm = c - 1, k = cm >> 5;
    const uint32_t sh = (uint32_t)(cm & 31);
    const uint32_t *q0 = q, *q1 = q + NQ, *aw = q + 2 * NQ;
    if (WB == 64) {
        p0 = (V)((uint64_t)funnel(q0[k + 1], q0[k], sh) | ((uint64_t)funnel(q0[k + 2], q0[k + 1], sh) << 32));
        p1 = (V)((uint64_t)funnel(q1[k + 1], q1[k], sh) | ((uint64_t)funnel(q1[k + 2], q1[k + 1], sh) << 32));
    } else {
        p0 = (V)funnel(q0[k + 1], q0[k], sh);
        p1 = (V)funnel(q1[k + 1], q1[k], sh);
    }
    x = (aw[cm >> 4] >> ((cm & 15) << 1)) & 3u;
}

// the last column: the row to go on from / to end at.  Returns the key (D << 16 | |row - diagonal| << 8 | W-1-i)
template <int WB>
DH_HD uint32_t tile_scan(const TileT<WB> &t)
{
    constexpr int W = WB;
    // eligible rows: j = cols - W/2 + i >= 0 and j - bnr <= cols
    const int32_t imin = W / 2 - t.cols, imax = t.bnr + W / 2;
    uint32_t key = 0xFFFFFFFFu;
    int32_t d = t.dbot;
#pragma unroll 8
    for (int i = W - 1; i >= 0; i--) {
        if (i < W - 1) d += (int32_t)((t.Mv >> i) & 1u) - (int32_t)((t.Pv >> i) & 1u);
        const uint32_t off = (uint32_t)(i >= W / 2 ? i - W / 2 : W / 2 - i);
        uint32_t kk = ((uint32_t)d << 16) | (off << 8) | (uint32_t)(W - 1 - i);
        kk = (i >= imin && i <= imax) ? kk : 0xFFFFFFFFu;
        key = kk < key ? kk : key;
    }
    return key;
}

// a tile is done: trace pair, next origin or end of the extension
// trace pair of a tile: (diffs, B bases) as two u16 values -- one 4-byte store on the device (the slots start on
// multiples of trmax = 4 (nbmax + 1) values of 4-byte aligned buffers; two 2-byte stores per tile and lane doubled the
// scattered partial writes the bookkeeping loads queue behind)
DH_HD void store_pair(uint16_t *pairs, int32_t pidx, uint32_t d, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    *(uint32_t *)(pairs + 2 * pidx) = (d & 0xFFFFu) | (b << 16);
#else
    pairs[2 * pidx] = (uint16_t)d;
    pairs[2 * pidx + 1] = (uint16_t)b;
#endif
}

// (key = tile_scan(t): the kernel passes the key of its own scan, dh_tile.hip:tile_scan_dev)
template <int WB>
DH_HD void tile_end_key(Lane &l, const Params &P, const TileT<WB> &t, const uint32_t key)
{
    constexpr int W = WB;
    Ext &e = l.e;
    uint16_t *pairs = e.pairs;
    const int32_t ts = P.o.tspace, pen = P.o.pen, nbmax = P.nbmax;
    l.cells += (uint64_t)t.cols * W;
    const int32_t ci = W - 1 - (int32_t)(key & 255u), dt = (int32_t)(key >> 16);
    const int32_t j = t.cols - W / 2 + ci, tw = j > t.bnr ? j - t.bnr : 0;
    // pair index of this tile in the candidate's slot: interval (floor(as / ts) +- ...) -> nbmax +- ...
    const int32_t kt = e.ntp + 1;  // 1-based tile number
    const int32_t pidx = l.dir ? nbmax - kt + l.roff : nbmax + kt - 1;
    if (pidx < 0 || 2 * pidx + 1 >= P.trmax) {
        l.err |= DH_ST_POOL_OVERFLOW;
        l.st = L_EXT_END;
        return;
    }
    if (tw > 0 || t.cols < t.T) {
        // the end of B' (a row past it stands for the cell (cols - tw, bnr)) or of A'
        const int32_t ea = e.a0 + t.cols - tw, eb = e.b0 + j - tw, ed = e.dsum + dt;
        const int32_t sc = ea + eb - pen * ed;
        if (sc > e.best_s) {
            const int32_t k = ea - eb;
            e.best_s = sc;
            e.best_a = ea;
            e.best_b = eb;
            e.best_d = ed;
            e.bklo = k < e.klo ? k : e.klo;
            e.bkhi = k > e.khi ? k : e.khi;
            const uint32_t pr = ((uint32_t)dt << 16) | (uint32_t)(j - tw);
            // a last segment without A bases (the end sits on the previous boundary) adds no pair
            const bool seg = ea > e.a0;
            e.best_nseg = e.ntp + (seg ? 1 : 0);
            e.best_pair1 = kt == 1 ? (seg ? pr : 0u) : e.pair1;
            if (seg && kt > 1) {
                store_pair(pairs, pidx, (uint32_t)dt, (uint32_t)(j - tw));
            }
        }
        l.st = L_EXT_END;
        return;
    }
    e.a0 += t.T;
    e.b0 += j;
    e.dsum += dt;
    e.ntp = kt;
    if (kt == 1)
        e.pair1 = ((uint32_t)dt << 16) | (uint32_t)j;
    else {
        store_pair(pairs, pidx, (uint32_t)dt, (uint32_t)j);
    }
    const int32_t k = e.a0 - e.b0;
    e.klo = k < e.klo ? k : e.klo;
    e.khi = k > e.khi ? k : e.khi;
    const int32_t sc = e.a0 + e.b0 - pen * e.dsum;
    if (sc > e.best_s) {
        e.best_s = sc;
        e.best_a = e.a0;
        e.best_b = e.b0;
        e.best_d = e.dsum;
        e.best_nseg = kt;
        e.best_pair1 = e.pair1;
        e.bklo = e.klo;
        e.bkhi = e.khi;
    } else if (sc < e.best_s - P.o.xdrop) {
        l.st = L_EXT_END;
        return;
    }
    if (e.a0 >= e.an || e.b0 >= e.bn) l.st = L_EXT_END;
    (void)ts;
}

template <int WB>
DH_HD void tile_end(Lane &l, const Params &P, const TileT<WB> &t)
{
    tile_end_key<WB>(l, P, t, tile_scan(t));
}

// ------------------------------------------------------------------------------ bookkeeping

#if defined(__HIP_DEVICE_COMPILE__)
#define DH_ATOMIC_ADD(p, v) atomicAdd((p), (v))
#else
static inline int32_t dh_host_fetch_add(int32_t *p, int32_t v)
{
    const int32_t old = *p;
    *p = old + v;
    return old;
}
#define DH_ATOMIC_ADD(p, v) dh_host_fetch_add((p), (v))
#endif

DH_HD void lane_init(Lane &l, int32_t slot, Cold *cold)
{
    l.st = L_FETCH;
    l.cells = 0;
    l.naln = 0;
    l.slot = slot;
    l.err = 0;
    l.dir = 0;
    l.roff = 0;
    l.c = cold;
    l.e.pairs = nullptr;
}

// where the trace pairs of the running candidate go: straight into its output slot (symmetric mode: the slot of
// (candidate, mode))
DH_HD uint16_t *lane_pairs(const Lane &l, const Params &P)
{
    if (P.o.skip_self == 2) return P.out_trace + (2 * ((int64_t)l.c->cbase + l.c->c) + l.c->mode) * P.trmax;
    if (l.c->mode) return P.out_trace2 + ((int64_t)l.c->item * P.o.max_la + l.c->nacc2) * P.trmax;
    return P.out_trace + ((int64_t)l.c->item * P.o.max_la + l.c->nacc) * P.trmax;
}

// geometry of the alignment to run for the current candidate.  mode 0: A = the candidate's A sequence,
// B = the item's read in the item's orientation.  mode 1 (symmetric mode, second record): the
// transposed pair through the same seed -- A'' = the item's read on its forward strand, B'' = the A
// sequence, complemented when the item is (both axes mirrored then)
DH_HD void cand_geometry(Lane &l, const Params &P, int32_t mode)
{
    Cold &c = *l.c;
    c.mode = mode;
    if (!mode) {
        c.g_ao = c.ao;
        c.g_alen = c.alen;
        c.g_as = c.as;
        c.g_bo = c.bo;
        c.g_blen = c.blen;
        c.g_bs = c.bs;
    } else {
        c.g_ao = c.bo;
        c.g_alen = c.blen;
        c.g_as = c.strand ? c.blen - c.bs : c.bs;
        c.g_bo = c.ao;
        c.g_blen = c.alen;
        c.g_bs = c.strand ? c.alen - c.as : c.as;
    }
    l.roff = (c.g_as % P.o.tspace) != 0 ? 1 : 0;
}

// `it` = work unit index: an item, or (P.units) one group of candidates of an item
DH_HD void lane_fetch(Lane &l, const Params &P, int32_t it)
{
    Cold &c = *l.c;
    if (P.units) {
        // a unit has at least one candidate and the lane no aligned region yet: the first candidate runs as it is
        // (what lane_next_cand would decide), from the unit record alone
        const Unit u = P.units[it];
        const int32_t item = P.item0 + u.it;
        c.item = item;
        c.strand = item & 1;
        c.nc = u.c1;
        c.bo = u.bo;
        c.blen = u.blen;
        c.nd = c.nacc = c.ntr = c.nacc2 = c.ntr2 = 0;
        c.c = u.c0;
        c.cbase = u.cbase;
        c.nd0 = u.nd0;
        c.rest = u.rest;
        c.c_aseq = u.aseq;
        c.as = u.apos;
        c.bs = u.bpos;
        c.ao = u.ao;
        c.alen = u.alen;
        cand_geometry(l, P, 0);
        ext_begin(l, P, 1);
        return;
    }
    const int32_t item = P.item0 + it;
    c.item = item;
    c.strand = item & 1;
    const int32_t nc = P.ncand[item];
    c.nc = nc > 0 ? nc : 0;
    const int64_t bo = P.boff[item >> 1];
    c.bo = bo;
    c.blen = (int32_t)(P.boff[(item >> 1) + 1] - bo);
    c.nd = c.nacc = c.ntr = c.nacc2 = c.ntr2 = 0;
    c.c = 0;
    c.cbase = P.candoff ? P.candoff[item] : 0;
    c.nd0 = 0;
    c.rest = 0;
    l.st = L_CAND;
}

// next candidate of the item that no aligned region covers -> its reverse extension; none -> item done
DH_HD void lane_next_cand(Lane &l, const Params &P)
{
    Cold &c = *l.c;
    const int32_t *rg = P.regs + (int64_t)l.slot * MAXREG * REGF;
    const bool sym = P.o.skip_self == 2;
    int32_t ci = c.c;
    const int32_t nc = c.nc, nd = c.nd, item = c.item;
    int32_t nd0 = c.nd0;
    while (ci < nc && (sym || c.nacc < P.o.max_la) && nd + nd0 < MAXREG) {
        const DhCand cd = P.cand[(int64_t)item * P.o.max_cand + ci];
        if (c.rest && ci < MAXREG) {
            // a candidate alone in its group is a unit of its own (k_units_fat): one attempted alignment, whatever comes of it
            const DhCand *cl = P.cand + (int64_t)item * P.o.max_cand;
            const bool alone = (ci == 0 || cl[ci - 1].aseq != cd.aseq) && (ci + 1 >= nc || cl[ci + 1].aseq != cd.aseq);
            if (alone) {
                nd0++;
                c.nd0 = nd0;
                ci++;
                continue;
            }
        }
        const int32_t sdc = cd.apos - cd.bpos;
        bool covd = false;
        for (int32_t x = 0; x < nd; x++) {
            const int32_t *g = rg + x * REGF;
            covd = covd || (g[0] == cd.aseq && cd.apos >= g[1] && cd.apos < g[2] && cd.bpos >= g[3] && cd.bpos < g[4] &&
                            sdc >= g[5] - 64 && sdc <= g[6] + 64);
        }
        if (covd) {
            ci++;
            continue;
        }
        c.c = ci;
        c.c_aseq = cd.aseq;
        c.as = cd.apos;
        c.bs = cd.bpos;
        const int64_t ao = P.aoff[cd.aseq];
        c.ao = ao;
        c.alen = (int32_t)(P.aoff[cd.aseq + 1] - ao);
        cand_geometry(l, P, 0);
        ext_begin(l, P, 1);
        return;
    }
    c.c = ci;
    if (!sym) {
        P.out_nla[item] = c.nacc;
        P.out_ntr[item] = c.ntr;
        if (P.out_la2) {
            P.out_nla2[item] = c.nacc2;
            P.out_ntr2[item] = c.ntr2;
        }
    }
    l.st = L_FETCH;
}

// the record of the alignment that just ended (both extensions), its pairs completed in `pairs`;
// returns the number of pairs and where they start inside the slot
DH_HD int32_t finish_pairs(const Lane &l, const Params &P, uint16_t *pairs, int32_t *first_out)
{
    const Ext &e = l.e;
    const Cold &c = *l.c;
    const int32_t nbmax = P.nbmax, nr = c.rv_nseg, nf = e.best_nseg;
    // the pair of the seed's interval: the forward tile 1 plus, when the seed is not on a boundary,
    // the reverse tile 1 (the two are the halves of one trace interval)
    const uint32_t seedp = (nf >= 1 ? e.best_pair1 : 0u) + ((l.roff && nr >= 1) ? c.rv_pair1 : 0u);
    const bool seedslot = nf >= 1 || (l.roff && nr >= 1);
    if (seedslot) {
        store_pair(pairs, nbmax, seedp >> 16, seedp & 0xFFFFu);
    }
    if (!l.roff && nr >= 1) {  // the reverse tile 1 is an interval of its own
        store_pair(pairs, nbmax - 1, c.rv_pair1 >> 16, c.rv_pair1 & 0xFFFFu);
    }
    const int32_t lo_idx = nbmax - nr + l.roff;
    const int32_t hi_idx = nbmax + (nf > (seedslot ? 1 : 0) ? nf : (seedslot ? 1 : 0));
    const int32_t first = (nr == 0 && l.roff) ? nbmax : lo_idx;  // no reverse segment: the range starts at the seed slot
    *first_out = first;
    return hi_idx - first;
}

// an extension ended: after the reverse one the forward one starts, after the forward one the candidate
// becomes a region and, if it passes, a record (symmetric mode: then the transposed pair is aligned
// for the second record)
DH_HD void lane_ext_end(Lane &l, const Params &P)
{
    Ext &e = l.e;
    Cold &c = *l.c;
    LP_BEGIN
    if (l.err) {
        l.st = L_DONE;
        return;
    }
    if (l.dir == 1) {
        c.rv_i = e.best_a;
        c.rv_j = e.best_b;
        c.rv_d = e.best_d;
        c.rv_nseg = e.best_nseg;
        c.rv_klo = e.bklo;
        c.rv_khi = e.bkhi;
        c.rv_pair1 = e.best_pair1;
        ext_begin(l, P, 0);
        LP(0)
        return;
    }
    const bool sym = P.o.skip_self == 2;
    const int32_t mode = c.mode, item = c.item, c_aseq = c.c_aseq, strand = c.strand;
    const int32_t as = c.g_as, bs = c.g_bs, sd = as - bs;
    const int32_t abpos = as - c.rv_i, bbpos = bs - c.rv_j, aepos = as + e.best_a, bepos = bs + e.best_b;
    const int32_t diffs = c.rv_d + e.best_d;
    if (mode == 0) {
        l.naln += 1;
        int32_t lo = sd + e.bklo, hi = sd + e.bkhi;
        lo = (sd - c.rv_khi) < lo ? (sd - c.rv_khi) : lo;
        hi = (sd - c.rv_klo) > hi ? (sd - c.rv_klo) : hi;
        int32_t *g = P.regs + ((int64_t)l.slot * MAXREG + c.nd) * REGF;
        g[0] = c_aseq;
        g[1] = abpos;
        g[2] = aepos;
        g[3] = bbpos;
        g[4] = bepos;
        g[5] = lo;
        g[6] = hi;
        c.nd += 1;
    }
    LP(1)
    const int64_t al = aepos - abpos, bl = bepos - bbpos;
    const bool accept = al >= P.o.min_len && (int64_t)2 * diffs * 1000000ll <= (int64_t)P.o.max_err_ppm * (al + bl);
    if (accept) {
        uint16_t *pairs = e.pairs;
        int32_t first;
        const int32_t npairs = finish_pairs(l, P, pairs, &first);
        LP(2)
        DhLa la;
        la.tlen = 2 * npairs;
        la.diffs = diffs;
        la.abpos = abpos;
        la.bbpos = bbpos;
        la.aepos = aepos;
        la.bepos = bepos;
        la.flags = strand ? 1u : 0u;
        la.aread = mode ? (item >> 1) : c_aseq;
        la.bread = mode ? c_aseq : (item >> 1);
        la.pad = 0;
        la.toff = 2 * (int64_t)first;  // where the pairs start inside the slot (k_compact honours it)
        if (!sym && mode) {  // the transposed record of a mapping: at most one per accepted record, so nacc2 <= nacc
            P.out_la2[(int64_t)item * P.o.max_la + c.nacc2] = la;
            c.ntr2 += 2 * npairs;
            c.nacc2 += 1;
        } else if (!sym) {
            P.out_la[(int64_t)item * P.o.max_la + c.nacc] = la;
            c.ntr += 2 * npairs;
        } else if (!P.pflags || dh_rec_wanted(P.pflags, la.aread, la.bread)) {
            la.pad = 1;  // valid (the slots start zeroed); grouped by A read after the launch
            P.out_la[2 * ((int64_t)c.cbase + c.c) + mode] = la;
        }
        LP(3)
        if (mode == 0) c.nacc += 1;
        // the second record: the transposed pair, aligned on its own (symmetric mode: when its A read's records are wanted)
        if ((sym || P.out_la2) && mode == 0 && (!sym || !P.pflags || dh_rec_wanted(P.pflags, item >> 1, c_aseq))) {
            cand_geometry(l, P, 1);
            ext_begin(l, P, 1);
            LP(4)
            return;
        }
    }
    c.c += 1;
    l.st = L_CAND;
    LP(5)
}

}  // namespace dhtile

#ifdef __cplusplus
extern "C" {
#endif
// nwaves wavefronts (64 lanes = 64 alignments each) over the items of P; regs holds nwaves * 64 lane slots
void dhk_tile(hipStream_t st, int32_t nwaves, const dhtile::Params *P);
int32_t dhk_tile_waves_per_cu(void);
// 2-bit packed words (32 bases each) -> plane-packed words, in place
void dhk_pk2planes(hipStream_t st, void *words, int64_t nwords);
// work units of a symmetric launch: the candidates of every item grouped by A read (items with more than 64
// candidates stay whole); units must hold nitems * max_cand records, *nunits counts them (zeroed by the caller)
void dhk_tile_units(hipStream_t st, const DhCand *cand, const int32_t *ncand, const int32_t *candoff, int32_t item0, int32_t nitems,
                    int32_t max_cand, const int64_t *aoff, const int64_t *boff, dhtile::Unit *units, uint32_t *nunits);
// symmetric launches (records in candidate-indexed slots): out[i] = max(ncand[i], 0) for the exclusive scan that yields
// candoff; records per A-read item; their slot ids grouped by item; ordered compaction
void dhk_cand_counts(hipStream_t st, const int32_t *ncand, int32_t nitems, uint32_t *out);
void dhk_rec_count(hipStream_t st, const DhLa *slots, int64_t nslots, int32_t item0, uint32_t *nla, uint32_t *ntr);
void dhk_rec_scatter(hipStream_t st, const DhLa *slots, int64_t nslots, int32_t item0, const uint32_t *la_off, uint32_t *cursor,
                     int32_t *list);
void dhk_compact_sym(hipStream_t st, const DhLa *slots, const uint16_t *tr_slots, int32_t trmax, const int32_t *list, int32_t nitems,
                     const uint32_t *la_off, const uint32_t *tr_off, int64_t tr_base, DhLa *la_out, uint16_t *tr_out, int32_t *item_ovf);
#ifdef __cplusplus
}
#endif
#endif
