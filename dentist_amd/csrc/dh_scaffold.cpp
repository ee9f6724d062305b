// dh_scaffold.cpp -- the scaffold-graph pile-up builder of `dentist collect`
// (collectPileUps/pileups.d:173-208 build): read alignments of every read (pileups.d:821-888
// collectReadAlignments over base.d:1964-2050 SeededAlignment), one join per read alignment
// (base.d:2680-2722 makeJoin), the scaffold graph with its four nodes per contig (scaffold.d:75-115,
// 237-244), multi-edges merged (pileups.d:626-636), forks resolved by read support
// (pileups.d:1592-1657, 1754-1804), min-spanning-reads (pileups.d:1807-1838), input-gap marks removed
// (pileups.d:1840-1852), extensions merged into their gap (scaffold.d:789-816) and the pile-ups read
// off the edges in edge order (pileups.d:435-444).
//
// Host code: per-read work runs on the thread pool, the graph itself has 4 nodes per contig and is
// handled serially.  Where the reference merges equal edges after an unstable sort, the order here is
// the stable one (read id, then position on the read).  resolveBubbles (pileups.d:1100-1590) is the Resolver below
// (Paton's cycle base as util/math.d:2380-2480 walks it, simple bubbles, skipped path, collectFixedSimpleBubbles,
// graph surgery); the re-mapping of the skipping reads is a callback -- dh_remap_skipping_reads on the device
// (dh_scaffold_pileups_resolved) or the caller's (dh_scaffold_pileups_cb).  dh_scaffold_pileups itself runs without
// it: the forks of a bubble then go through the read-support rule like any other fork.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <array>
#include <atomic>
#include <cstring>
#include <functional>
#include <vector>

#include "dh_internal.h"
#include "dh_parallel.h"

namespace {
enum { FRONT = 0, BACK = 1 };
enum { PRE = 0, BEGIN = 1, END = 2, POST = 3 };
enum { T_PILEUP = 1, T_INPUTGAP = 2 };

struct Node {
    int32_t contig, part;
    bool operator<(const Node &o) const { return contig != o.contig ? contig < o.contig : part < o.part; }
    bool operator==(const Node &o) const { return contig == o.contig && part == o.part; }
};
inline bool real_part(int32_t p) { return p == BEGIN || p == END; }

struct Edge {
    Node s, e;
    uint32_t types = 0;
    std::vector<dh_read_alignment> ras;
    bool is_default() const { return s.part == BEGIN && e.part == END && s.contig == e.contig; }
    bool is_gap() const { return s.contig != e.contig && real_part(s.part) && real_part(e.part); }
    bool empty_payload() const { return types == 0 && ras.empty(); }
};
inline bool key_less(const Edge &a, const Edge &b) { return a.s == b.s ? a.e < b.e : a.s < b.s; }
inline bool key_eq(const Edge &a, const Edge &b) { return a.s == b.s && a.e == b.e; }
inline Edge make_edge(Node a, Node b)
{
    Edge x;
    if (b < a) std::swap(a, b);  // undirected: start <= end (math.d:385-398)
    x.s = a;
    x.e = b;
    return x;
}

void remove_none_joins(std::vector<Edge> &g)
{
    g.erase(std::remove_if(g.begin(), g.end(), [](const Edge &x) { return !x.is_default() && x.empty_payload(); }),
            g.end());
}

struct SA {
    int64_t la;
    int32_t seed, b, e, srel;
};

struct Ctx {
    const dh_la *las;
    const int64_t *coff, *roff;
    // the bubble resolver adds alignments: LA i >= n is (*extra)[i - n]
    int64_t n = INT64_MAX;
    const std::vector<dh_la> *extra = nullptr;
    const dh_la &L(int64_t i) const { return i < n ? las[i] : (*extra)[(size_t)(i - n)]; }
    int32_t alen(const dh_la &l) const { return (int32_t)(coff[l.aread + 1] - coff[l.aread]); }
    int32_t blen(const dh_la &l) const { return (int32_t)(roff[l.bread + 1] - roff[l.bread]); }
};

// one read alignment with the edge it contributes to: plain data, so that a run of reads yields one flat array
// (no allocation per join); runs are turned into edges by raw_to_edges
struct RawJoin {
    Node s, e;
    dh_read_alignment ra;
};

// collectReadAlignments for the enabled LAs idx[0..cnt) of one read, appended to `out` as raw joins
// (sa, on return: the seeded alignments of the read in read order -- what the bubble resolver checks against the skipped path)
void read_joins(const Ctx &c, const int64_t *idx, int64_t cnt, std::vector<RawJoin> &out, std::vector<SA> &sa,
                std::vector<std::pair<size_t, size_t>> &sl)
{
    sa.clear();
    for (int64_t x = 0; x < cnt; x++) {
        const dh_la &l = c.L(idx[x]);
        const bool comp = (l.flags & DH_FLAG_COMP) != 0;
        const int32_t bl = c.blen(l);
        const int32_t b = comp ? bl - l.bepos : l.bbpos, e = comp ? bl - l.bbpos : l.bepos;
        if (l.bbpos > l.abpos) sa.push_back(SA{idx[x], FRONT, b, e, comp ? -FRONT : FRONT});                 // isFrontExtension
        if (bl - l.bepos > c.alen(l) - l.aepos) sa.push_back(SA{idx[x], BACK, b, e, comp ? -BACK : BACK});  // isBackExtension
    }
    if (sa.empty()) return;
    std::stable_sort(sa.begin(), sa.end(), [](const SA &p, const SA &q) {
        if (p.b != q.b) return p.b < q.b;
        if (p.e != q.e) return p.e < q.e;
        return p.srel < q.srel;
    });
    for (size_t i = 0; i + 1 < sa.size(); i++)
        if (sa[i].e > sa[i + 1].b && !(sa[i].la == sa[i + 1].la && sa[i].seed != sa[i + 1].seed)) return;  // a region of the read used twice
    const bool start_ext = sa[0].b > 0;
    // slices [0,1) if the read starts with an extension, then pairs
    sl.clear();
    if (start_ext) sl.emplace_back(0, 1);
    for (size_t i = start_ext ? 1 : 0; i < sa.size(); i += 2) sl.emplace_back(i, std::min(i + 2, sa.size()));
    for (auto &p : sl)  // one invalid read alignment discards the read
        if (p.second - p.first == 2 && c.L(sa[p.first].la).aread == c.L(sa[p.first + 1].la).aread) return;
    for (auto &p : sl) {
        dh_read_alignment ra;
        memset(&ra, 0, sizeof(ra));
        Edge e;
        if (p.second - p.first == 2) {
            SA a = sa[p.first], b = sa[p.first + 1];
            if (!(c.L(a.la).aread < c.L(b.la).aread)) std::swap(a, b);  // getInOrder
            ra.n = 2;
            ra.la0 = (int32_t)a.la;
            ra.la1 = (int32_t)b.la;
            ra.seed0 = (uint8_t)a.seed;
            ra.seed1 = (uint8_t)b.seed;
            e = make_edge(Node{c.L(a.la).aread, a.seed == FRONT ? BEGIN : END},
                          Node{c.L(b.la).aread, b.seed == FRONT ? BEGIN : END});
        } else {
            const SA a = sa[p.first];
            ra.n = 1;
            ra.la0 = (int32_t)a.la;
            ra.la1 = -1;
            ra.seed0 = (uint8_t)a.seed;
            const int32_t ct = c.L(a.la).aread;
            e = a.seed == FRONT ? make_edge(Node{ct, PRE}, Node{ct, BEGIN}) : make_edge(Node{ct, END}, Node{ct, POST});
        }
        ra.read = c.L(ra.la0).bread;
        out.push_back(RawJoin{e.s, e.e, ra});
    }
}

// the raw joins of a run of reads as edges: equal edges merged, their read alignments in input (= read) order
void raw_to_edges(std::vector<RawJoin> &raw, std::vector<Edge> &out)
{
    std::stable_sort(raw.begin(), raw.end(), [](const RawJoin &a, const RawJoin &b) { return a.s == b.s ? a.e < b.e : a.s < b.s; });
    for (size_t i = 0; i < raw.size();) {
        size_t j = i + 1;
        while (j < raw.size() && raw[j].s == raw[i].s && raw[j].e == raw[i].e) j++;
        Edge m;
        m.s = raw[i].s;
        m.e = raw[i].e;
        m.types = T_PILEUP;
        m.ras.reserve(j - i);
        for (size_t x = i; x < j; x++) m.ras.push_back(raw[x].ra);
        out.push_back(std::move(m));
        i = j;
    }
}

// sort stably and merge equal edges: types OR-ed, read alignments concatenated (bulkAdd!mergeJoins)
void merge_multi_edges(std::vector<Edge> &g)
{
    std::stable_sort(g.begin(), g.end(), key_less);
    std::vector<Edge> out;
    for (size_t i = 0; i < g.size();) {
        size_t j = i + 1;
        while (j < g.size() && key_eq(g[i], g[j])) j++;
        Edge m = std::move(g[i]);
        if (j - i > 1) {
            size_t tot = m.ras.size();
            for (size_t x = i + 1; x < j; x++) tot += g[x].ras.size();
            m.ras.reserve(tot);
            for (size_t x = i + 1; x < j; x++) {
                m.types |= g[x].types;
                m.ras.insert(m.ras.end(), g[x].ras.begin(), g[x].ras.end());
            }
        }
        out.push_back(std::move(m));
        i = j;
    }
    g.swap(out);
}

// edges incident to every node: inc[4 * contig + part] = edge indices in edge order (two flat arrays: a vector per node
// was 4 000 small allocations per call, half a millisecond of the plan every rank of a sharded run derives)
struct Incidence {
    std::vector<int32_t> off, idx;
    struct Span {
        const int32_t *b, *e;
        const int32_t *begin() const { return b; }
        const int32_t *end() const { return e; }
        size_t size() const { return (size_t)(e - b); }
        bool empty() const { return b == e; }
    };
    Span operator[](size_t node) const { return Span{idx.data() + off[node], idx.data() + off[node + 1]}; }
    size_t size() const { return off.size() - 1; }  // nodes
};
Incidence incidence(const std::vector<Edge> &g, int32_t ncontigs)
{
    Incidence inc;
    inc.off.assign((size_t)ncontigs * 4 + 1, 0);
    for (const Edge &x : g) {
        inc.off[(size_t)x.s.contig * 4 + x.s.part + 1]++;
        if (!(x.e == x.s)) inc.off[(size_t)x.e.contig * 4 + x.e.part + 1]++;
    }
    for (size_t v = 0; v + 1 < inc.off.size(); v++) inc.off[v + 1] += inc.off[v];
    inc.idx.resize((size_t)inc.off.back());
    std::vector<int32_t> cur(inc.off.begin(), inc.off.end() - 1);
    for (size_t i = 0; i < g.size(); i++) {
        inc.idx[(size_t)cur[(size_t)g[i].s.contig * 4 + g[i].s.part]++] = (int32_t)i;
        if (!(g[i].e == g[i].s)) inc.idx[(size_t)cur[(size_t)g[i].e.contig * 4 + g[i].e.part]++] = (int32_t)i;
    }
    return inc;
}
}  // namespace

struct dh_scaffold {
    std::vector<dh_join> joins;
    std::vector<dh_read_alignment> entries;
};

extern "C" void dh_default_scaffold_opts(dh_scaffold_opts *o)
{
    if (!o) return;
    memset(o, 0, sizeof(*o));
    o->min_spanning_reads = 3;     // commandline.d:2125, 2187
    o->merge_extensions = 1;       // commandline.d:2217-2222
    o->best_pile_up_margin = 3.0;  // commandline.d:1345
    o->existing_gap_bonus = 6.0;   // commandline.d:1688
}

namespace {
// blob of the sharded collector (dh_shard_read_joins): JoinHead, then one JoinRec per raw join with copies of the FIRST
// record of the alignment chain of each flank, then the other members of those chains (x0 of flank 0, then x1 of flank 1,
// join by join) -- an entry of a pile-up names the first record of its chain and the cropper finds the members behind it
#pragma pack(push, 1)
struct JoinHead {
    int64_t njoins, nextra;
};
struct JoinRec {
    Node s, e;
    int32_t read;
    uint8_t seed0, seed1, n, pad;
    int32_t x0, x1;  // chain members after the first, per flank
    dh_la la0, la1;  // la1 zeroed for an extension
};
#pragma pack(pop)
static_assert(sizeof(JoinRec) == 32 + 2 * sizeof(dh_la) && sizeof(JoinHead) == 16, "join blob layout");

// The raw joins of the reads [read_first, read_first + nreads) named by `las` (bread = global read id), in read
// order: one flat array per run of reads (a run per host thread).
int collect_raw_joins(const char *who, const dh_la *las, int64_t n, const int64_t *contig_off, int32_t ncontigs,
                      const int64_t *read_off, int32_t read_first, int32_t nreads, std::vector<std::vector<RawJoin>> *raws)
{
    Ctx c{las, contig_off, read_off - read_first};
    auto T0_ = std::chrono::steady_clock::now();
    auto LAP_ = [&](const char *w) { if (getenv("DH_TRACE")) { auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[scaffold] %-24s %.2f ms\n", w, std::chrono::duration<double, std::milli>(t - T0_).count()); T0_ = t; } };
    // ---- the enabled LAs grouped by read, input order inside a read
    std::atomic<int> bad{0};
    const int64_t lgrain = 1 << 14, lchunks = (n + lgrain - 1) / lgrain;  // (a mapping of configs[2]: 70 runs for up to 64 threads)
    std::vector<std::vector<std::pair<int32_t, int32_t>>> live((size_t)std::max<int64_t>(lchunks, 1));
    dh_parallel_for(lchunks, 1, [&](int64_t clo, int64_t chi) {
        for (int64_t ch = clo; ch < chi; ch++) {
            auto &v = live[(size_t)ch];
            const int64_t i1 = std::min(n, (ch + 1) * lgrain);
            for (int64_t i = ch * lgrain; i < i1; i++) {
                const int64_t rd = (int64_t)las[i].bread - read_first;
                if (rd < 0 || rd >= nreads || las[i].aread < 0 || las[i].aread >= ncontigs) bad = 1;
                else if (!(las[i].flags & DH_FLAG_DISABLED)) v.emplace_back((int32_t)rd, (int32_t)i);
            }
        }
    });
    if (bad) return dh_fail(DH_EINVAL, std::string(who) + ": read or contig id out of range");
    std::vector<int64_t> first((size_t)nreads + 1, 0);
    int64_t nlive = 0;
    std::vector<int64_t> loff(live.size() + 1, 0);
    for (size_t ch = 0; ch < live.size(); ch++) loff[ch + 1] = loff[ch] + (int64_t)live[ch].size();
    nlive = loff.back();
    std::vector<int64_t> order((size_t)nlive);
    // records in read order (a mapping as the device hands it over, or LAsort order inside one contig): the live
    // records are already grouped -- offsets by a merge walk per run of reads instead of a serial counting sort
    std::atomic<int> unsorted{0};
    dh_parallel_for((int64_t)live.size(), 1, [&](int64_t clo, int64_t chi) {
        for (int64_t ch = clo; ch < chi; ch++) {
            const auto &v = live[(size_t)ch];
            // (the last live record before this chunk: the nearest earlier chunk that has one -- a run of 16 384 disabled
            // records, e.g. a fully filtered repeat contig in A-major input, must not hide a descent in read id)
            int32_t prev = -1;
            for (int64_t pc = ch - 1; pc >= 0; pc--)
                if (!live[(size_t)pc].empty()) {
                    prev = live[(size_t)pc].back().first;
                    break;
                }
            for (size_t k = 0; k < v.size(); k++) {
                if (v[k].first < prev) unsorted = 1;
                prev = v[k].first;
                order[(size_t)loff[(size_t)ch] + k] = v[k].second;
            }
        }
    });
    if (!unsorted.load()) {
        const int64_t rgrain = 1 << 13, rchunks = ((int64_t)nreads + 1 + rgrain - 1) / rgrain;
        dh_parallel_for(rchunks, 1, [&](int64_t clo, int64_t chi) {
            for (int64_t ch = clo; ch < chi; ch++) {
                const int64_t r0 = ch * rgrain, r1 = std::min<int64_t>((int64_t)nreads + 1, r0 + rgrain);
                int64_t lo = 0, hi = nlive;  // first live record of a read >= r0
                while (lo < hi) {
                    const int64_t mid = (lo + hi) >> 1;
                    if ((int64_t)las[order[(size_t)mid]].bread - read_first < r0)
                        lo = mid + 1;
                    else
                        hi = mid;
                }
                for (int64_t r = r0; r < r1; r++) {
                    while (lo < nlive && (int64_t)las[order[(size_t)lo]].bread - read_first < r) lo++;
                    first[(size_t)r] = lo;
                }
            }
        });
    } else {
        for (const auto &v : live)
            for (const auto &e : v) first[(size_t)e.first + 1]++;
        for (int32_t r = 0; r < nreads; r++) first[(size_t)r + 1] += first[(size_t)r];
        std::vector<int64_t> cur(first.begin(), first.end() - 1);
        for (const auto &v : live)
            for (const auto &e : v) order[(size_t)cur[(size_t)e.first]++] = e.second;
    }
    live.clear();
    LAP_("group by read");
    // ---- raw joins of the reads, in read order (collectScaffoldJoins, pileups.d:650-667)
    // (64 runs: with 16 the step ran on 16 of the host's cores; the runs stay raw -- scaffold_from_runs sorts them by
    // edge, all runs at once)
    const int64_t grain = std::max<int64_t>(4096, ((int64_t)nreads + 63) / 64), nchunks = ((int64_t)nreads + grain - 1) / grain;
    raws->assign((size_t)std::max<int64_t>(nchunks, 1), {});
    dh_parallel_for(nchunks, 1, [&](int64_t clo, int64_t chi) {
        std::vector<SA> sa;
        std::vector<std::pair<size_t, size_t>> sl;
        for (int64_t ch = clo; ch < chi; ch++) {
            std::vector<RawJoin> &raw = (*raws)[(size_t)ch];
            const int32_t r1 = (int32_t)std::min<int64_t>(nreads, (ch + 1) * grain);
            raw.clear();
            for (int32_t rd = (int32_t)(ch * grain); rd < r1; rd++) {
                const int64_t cnt = first[(size_t)rd + 1] - first[(size_t)rd];
                if (cnt > 0) read_joins(c, order.data() + first[(size_t)rd], cnt, raw, sa, sl);
            }
        }
    });
    LAP_("read joins");
    return DH_OK;
}

struct Resolver;
int scaffold_from_runs(std::vector<std::vector<RawJoin>> &runs, int32_t ncontigs, const int32_t *input_gaps, int32_t ngaps,
                       const dh_scaffold_opts *opts, dh_scaffold **out, Resolver *rs = nullptr);
}  // namespace

extern "C" int dh_scaffold_pileups(const dh_la *las, int64_t n, const int64_t *contig_off, int32_t ncontigs,
                                   const int64_t *read_off, int32_t nreads, const int32_t *input_gaps, int32_t ngaps,
                                   const dh_scaffold_opts *opts, dh_scaffold **out)
{
    if ((n > 0 && !las) || !contig_off || !read_off || !opts || !out || ncontigs < 0 || nreads < 0 || n < 0 ||
        n >= (1ll << 31) || (ngaps > 0 && !input_gaps) || ngaps < 0)
        return dh_fail(DH_EINVAL, "dh_scaffold_pileups: bad argument");
    for (int32_t g = 0; g < ngaps; g++)
        if (input_gaps[2 * g] < 0 || input_gaps[2 * g] >= ncontigs || input_gaps[2 * g + 1] < 0 || input_gaps[2 * g + 1] >= ncontigs)
            return dh_fail(DH_EINVAL, "dh_scaffold_pileups: input gap names a contig out of range");
    // alignment chains are the unit (SeededAlignment.from(AlignmentChain), base.d:1964-2050: first.begin .. last.end): the
    // builder runs on one pseudo record per chain, the read alignments then name the chain's first record
    dh_chain_view cv;
    dh_chain_view_build(las, n, cv);
    const dh_la *u = cv.trivial ? las : cv.unit.data();
    const int64_t nu = cv.trivial ? n : (int64_t)cv.unit.size();
    std::vector<std::vector<RawJoin>> found;
    if (int rc = collect_raw_joins("dh_scaffold_pileups", u, nu, contig_off, ncontigs, read_off, 0, nreads, &found)) return rc;
    if (int rc = scaffold_from_runs(found, ncontigs, input_gaps, ngaps, opts, out)) return rc;
    if (!cv.trivial)
        for (dh_read_alignment &ra : (*out)->entries) {
            ra.la0 = (int32_t)cv.first[(size_t)ra.la0];
            if (ra.n == 2) ra.la1 = (int32_t)cv.first[(size_t)ra.la1];
        }
    return DH_OK;
}

// ---- the sharded collector: every rank turns the alignments of ITS reads into raw joins (a per-read computation),
// the joins are all-gathered in rank order (= read order), and every rank builds the same scaffold from them
extern "C" int dh_shard_read_joins(const dh_la *las, int64_t n, const int64_t *contig_off, int32_t ncontigs,
                                   const int64_t *read_off, int32_t read_first, int32_t nreads, uint8_t **blob, int64_t *nbytes)
{
    if ((n > 0 && !las) || !contig_off || !read_off || !blob || !nbytes || ncontigs < 0 || nreads < 0 || n < 0 || n >= (1ll << 31) ||
        read_first < 0)
        return dh_fail(DH_EINVAL, "dh_shard_read_joins: bad argument");
    // alignment chains are the unit, as in dh_scaffold_pileups: the joins are collected on one pseudo record per chain,
    // the blob carries every member (a read with a long indel next to a gap is cropped through the member that covers
    // the crop point, dh_crop_pileups)
    dh_chain_view cv;
    dh_chain_view_build(las, n, cv);
    const dh_la *u = cv.trivial ? las : cv.unit.data();
    const int64_t nu = cv.trivial ? n : (int64_t)cv.unit.size();
    auto mfirst = [&](int32_t ci) { return cv.trivial ? (int64_t)ci : cv.first[(size_t)ci]; };
    auto mend = [&](int32_t ci) { return cv.trivial ? (int64_t)ci + 1 : cv.first[(size_t)ci + 1]; };
    std::vector<std::vector<RawJoin>> raws;
    if (int rc = collect_raw_joins("dh_shard_read_joins", u, nu, contig_off, ncontigs, read_off, read_first, nreads, &raws)) return rc;
    size_t tot = 0, totx = 0;
    std::vector<size_t> at(raws.size()), atx(raws.size());
    for (size_t i = 0; i < raws.size(); i++) {
        at[i] = tot;
        atx[i] = totx;
        tot += raws[i].size();
        if (!cv.trivial)
            for (const RawJoin &r : raws[i])
                totx += (size_t)(mend(r.ra.la0) - mfirst(r.ra.la0) - 1) + (r.ra.la1 >= 0 ? (size_t)(mend(r.ra.la1) - mfirst(r.ra.la1) - 1) : 0);
    }
    const size_t bytes = sizeof(JoinHead) + tot * sizeof(JoinRec) + totx * sizeof(dh_la);
    uint8_t *blk = (uint8_t *)malloc(bytes);
    if (!blk) return dh_fail(DH_EINVAL, "dh_shard_read_joins: out of memory");
    JoinHead *head = (JoinHead *)blk;
    head->njoins = (int64_t)tot;
    head->nextra = (int64_t)totx;
    JoinRec *out = (JoinRec *)(blk + sizeof(JoinHead));
    dh_la *extra = (dh_la *)(blk + sizeof(JoinHead) + tot * sizeof(JoinRec));
    dh_parallel_for((int64_t)raws.size(), 1, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; i++) {
            size_t xat = atx[(size_t)i];
            for (size_t x = 0; x < raws[(size_t)i].size(); x++) {
                const RawJoin &r = raws[(size_t)i][x];
                JoinRec &j = out[at[(size_t)i] + x];
                j.s = r.s;
                j.e = r.e;
                j.read = r.ra.read;
                j.seed0 = r.ra.seed0;
                j.seed1 = r.ra.seed1;
                j.n = r.ra.n;
                j.pad = 0;
                const int64_t f0 = mfirst(r.ra.la0), e0 = mend(r.ra.la0);
                j.la0 = las[f0];
                j.x0 = (int32_t)(e0 - f0 - 1);
                for (int64_t m = f0 + 1; m < e0; m++) extra[xat++] = las[m];
                if (r.ra.la1 >= 0) {
                    const int64_t f1 = mfirst(r.ra.la1), e1 = mend(r.ra.la1);
                    j.la1 = las[f1];
                    j.x1 = (int32_t)(e1 - f1 - 1);
                    for (int64_t m = f1 + 1; m < e1; m++) extra[xat++] = las[m];
                } else {
                    memset(&j.la1, 0, sizeof(dh_la));
                    j.x1 = 0;
                }
            }
        }
    });
    *blob = blk;
    *nbytes = (int64_t)bytes;
    return DH_OK;
}

// the gathered join blobs -> the scaffold every rank derives; glas = the LA records of the joins (two per join, the
// second unused for an extension), which the entries of the scaffold index
int dh_scaffold_from_join_blobs(const uint8_t *const *blobs, const int64_t *sizes, int32_t world, int32_t ncontigs,
                                const int32_t *input_gaps, int32_t ngaps, const dh_scaffold_opts *opts,
                                dh_la_vec &glas, dh_scaffold **out)
{
    if (!blobs || !sizes || world < 1 || !opts || !out || ncontigs < 0 || (ngaps > 0 && !input_gaps) || ngaps < 0)
        return dh_fail(DH_EINVAL, "dh_shard_graph_plan_create: bad argument");
    auto T0_ = std::chrono::steady_clock::now();
    auto LAP_ = [&](const char *w) { if (getenv("DH_TRACE")) { auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[scaffold] %-24s %.2f ms\n", w, std::chrono::duration<double, std::milli>(t - T0_).count()); T0_ = t; } };
    int64_t tot = 0, totx = 0;
    std::vector<int64_t> start((size_t)world + 1, 0);
    std::vector<const JoinRec *> recs((size_t)world, nullptr);
    std::vector<const dh_la *> extras((size_t)world, nullptr);
    for (int32_t r = 0; r < world; r++) {
        if (sizes[r] < (int64_t)sizeof(JoinHead) || !blobs[r]) return dh_fail(DH_EINVAL, "dh_shard_graph_plan_create: blob size");
        JoinHead h;
        memcpy(&h, blobs[r], sizeof(h));
        if (h.njoins < 0 || h.nextra < 0 || h.njoins > sizes[r] / (int64_t)sizeof(JoinRec) || h.nextra > sizes[r] / (int64_t)sizeof(dh_la) ||
            (int64_t)sizeof(JoinHead) + h.njoins * (int64_t)sizeof(JoinRec) + h.nextra * (int64_t)sizeof(dh_la) != sizes[r])
            return dh_fail(DH_EINVAL, "dh_shard_graph_plan_create: blob size");
        recs[(size_t)r] = (const JoinRec *)(blobs[r] + sizeof(JoinHead));
        extras[(size_t)r] = (const dh_la *)(blobs[r] + sizeof(JoinHead) + h.njoins * sizeof(JoinRec));
        start[(size_t)r + 1] = start[(size_t)r] + h.njoins;
        tot += h.njoins;
        totx += h.nextra;
    }
    if (4 * tot + totx >= (1ll << 31)) return dh_fail(DH_EINVAL, "dh_shard_graph_plan_create: too many joins");
    for (int32_t g = 0; g < ngaps; g++)
        if (input_gaps[2 * g] < 0 || input_gaps[2 * g] >= ncontigs || input_gaps[2 * g + 1] < 0 || input_gaps[2 * g + 1] >= ncontigs)
            return dh_fail(DH_EINVAL, "dh_shard_graph_plan_create: input gap names a contig out of range");
    // where the records of every join go: the members of a chain behind its first record (la0 chain, then la1 chain).  Two
    // copies that would read as one chain (a record flagged NEXT without START behind a copy of its own read on the same
    // contig) are kept apart by a separator record that continues nothing.  One host thread per rank's blob lays its
    // records out from zero (the record in front of a blob's first one is the last record of the nearest non-empty blob
    // before it); the blobs' sizes are then summed up
    std::vector<int32_t> pos0((size_t)tot), pos1((size_t)tot);
    std::vector<int64_t> xoff((size_t)tot);
    std::vector<int64_t> nrec((size_t)world + 1, 0);
    std::vector<std::vector<int32_t>> seps((size_t)world);  // where a blob's separator records go (blob-relative)
    std::atomic<int> malformed{0};
    auto last_record = [&](int32_t r) -> const dh_la * {  // of blob r, nullptr: no joins (call only on checked blobs' tails)
        const int64_t nj = start[(size_t)r + 1] - start[(size_t)r];
        if (!nj) return nullptr;
        const JoinRec &j = recs[(size_t)r][nj - 1];
        const int64_t nx = (sizes[r] - (int64_t)sizeof(JoinHead) - nj * (int64_t)sizeof(JoinRec)) / (int64_t)sizeof(dh_la);
        const int32_t tail = j.n == 2 ? j.x1 : j.x0;
        const int64_t last = j.n == 2 ? nx - 1 : nx - 1 - j.x1;
        if (j.x0 < 0 || j.x1 < 0 || (j.n != 1 && j.n != 2) || (tail > 0 && (last < 0 || last >= nx))) return nullptr;  // (reported as malformed by its own pass)
        return tail ? &extras[(size_t)r][last] : (j.n == 2 ? &j.la1 : &j.la0);
    };
    dh_parallel_for(world, 1, [&](int64_t rlo, int64_t rhi) {
        auto may_continue = [](const dh_la &x) { return (x.flags & DH_FLAG_NEXT) && !(x.flags & DH_FLAG_START); };
        for (int32_t r = (int32_t)rlo; r < (int32_t)rhi; r++) {
            const dh_la *prev = nullptr;  // the record that will precede the next one placed
            for (int32_t q = r - 1; q >= 0 && !prev; q--) prev = last_record(q);
            int64_t xr = 0, cur = 0;
            const int64_t nj = start[(size_t)r + 1] - start[(size_t)r];
            const int64_t nx = (sizes[r] - (int64_t)sizeof(JoinHead) - nj * (int64_t)sizeof(JoinRec)) / (int64_t)sizeof(dh_la);
            int64_t at = start[(size_t)r];
            bool bad = false;
            for (int64_t i = 0; i < nj; i++, at++) {
                const JoinRec &j = recs[(size_t)r][i];
                if (j.x0 < 0 || j.x1 < 0 || (j.n != 1 && j.n != 2) || xr + j.x0 + j.x1 > nx) {
                    bad = true;
                    break;
                }
                xoff[(size_t)at] = xr;
                if (prev && may_continue(j.la0) && dh_continues_chain(*prev, j.la0)) seps[(size_t)r].push_back((int32_t)cur++);
                pos0[(size_t)at] = (int32_t)cur;
                cur += 1 + j.x0;
                prev = j.x0 ? &extras[(size_t)r][xr + j.x0 - 1] : &j.la0;
                if (j.n == 2) {
                    if (may_continue(j.la1) && dh_continues_chain(*prev, j.la1)) seps[(size_t)r].push_back((int32_t)cur++);
                    pos1[(size_t)at] = (int32_t)cur;
                    cur += 1 + j.x1;
                    prev = j.x1 ? &extras[(size_t)r][xr + j.x0 + j.x1 - 1] : &j.la1;
                } else
                    pos1[(size_t)at] = -1;
                xr += j.x0 + j.x1;
            }
            if (bad || xr != nx) malformed = 1;
            nrec[(size_t)r + 1] = cur;
        }
    });
    if (malformed) return dh_fail(DH_EINVAL, "dh_shard_graph_plan_create: malformed join record");
    for (int32_t r = 0; r < world; r++) nrec[(size_t)r + 1] += nrec[(size_t)r];
    LAP_("blobs: positions");
    dh_la sep;
    memset(&sep, 0, sizeof(sep));
    sep.aread = sep.bread = -1;
    sep.flags = DH_FLAG_DISABLED;
    glas.resize((size_t)nrec[(size_t)world]);  // (unwritten: every record is stored below, by the thread that touches its page first)
    for (int32_t r = 0; r < world; r++)
        for (int32_t x : seps[(size_t)r]) glas[(size_t)(nrec[(size_t)r] + x)] = sep;
    LAP_("blobs: records array");
    // runs of the concatenation (rank order = read order), as in the single-rank builder
    const int64_t grain = 4096, nruns = (tot + grain - 1) / grain;
    std::vector<std::vector<RawJoin>> found((size_t)std::max<int64_t>(nruns, 1));
    std::atomic<int> bad{0};
    dh_parallel_for(nruns, 1, [&](int64_t lo, int64_t hi) {
        for (int64_t run = lo; run < hi; run++) {
            std::vector<RawJoin> &raw = found[(size_t)run];
            const int64_t a0 = run * grain, a1 = std::min(tot, a0 + grain);
            raw.resize((size_t)(a1 - a0));
            int32_t r = (int32_t)(std::upper_bound(start.begin(), start.end(), a0) - start.begin()) - 1;
            for (int64_t at = a0; at < a1; at++) {
                while (at >= start[(size_t)r + 1]) r++;
                const JoinRec &j = recs[(size_t)r][at - start[(size_t)r]];
                if (j.s.contig < 0 || j.s.contig >= ncontigs || j.e.contig < 0 || j.e.contig >= ncontigs || j.s.part < 0 || j.s.part > 3 ||
                    j.e.part < 0 || j.e.part > 3)
                    bad = 1;
                const dh_la *x = extras[(size_t)r] + xoff[(size_t)at];
                const size_t p0 = (size_t)(nrec[(size_t)r] + pos0[(size_t)at]), p1 = j.n == 2 ? (size_t)(nrec[(size_t)r] + pos1[(size_t)at]) : 0;
                glas[p0] = j.la0;
                for (int32_t m = 0; m < j.x0; m++) glas[p0 + 1 + (size_t)m] = x[m];
                if (j.n == 2) {
                    glas[p1] = j.la1;
                    for (int32_t m = 0; m < j.x1; m++) glas[p1 + 1 + (size_t)m] = x[j.x0 + m];
                }
                RawJoin &q = raw[(size_t)(at - a0)];
                q.s = j.s;
                q.e = j.e;
                memset(&q.ra, 0, sizeof(q.ra));
                q.ra.read = j.read;
                q.ra.la0 = (int32_t)p0;
                q.ra.la1 = j.n == 2 ? (int32_t)p1 : -1;
                q.ra.seed0 = j.seed0;
                q.ra.seed1 = j.seed1;
                q.ra.n = j.n;
            }
        }
    });
    if (bad) return dh_fail(DH_EINVAL, "dh_shard_graph_plan_create: malformed join record");
    LAP_("blobs: records, raw joins");
    return scaffold_from_runs(found, ncontigs, input_gaps, ngaps, opts, out);
}

namespace {
// ---- resolveBubbles (pileups.d:1100-1590).  A bubble: a cycle of at most max_bubble_size nodes with exactly two nodes of
// degree >= 3 (extension joins not counted) that are joined by an edge carrying a pile-up -- the reads of that pile-up
// SKIP the contigs on the other side of the cycle (their alignments there were masked or filtered away).  The skipping
// reads are mapped onto the intermediate contigs once more, without any mask (remap = getReadAlignmentsOnContigs,
// :1316-1385: dh_remap_skipping_reads on the device), their read alignments are collected again from the old and the
// new alignments together, kept iff they walk the skipped path in order (collectFixedSimpleBubbles :1414-1491), and
// replace the pile-up of the skipping edge.
struct Resolver {
    Ctx c;                       // LA access incl. the alignments added so far
    std::vector<dh_la> *extra;   // the added alignments (LA index n + i)
    // contig ids, read ids (ascending, distinct) -> alignments with ids of the full DBs, DISABLED unless they cover their contig
    std::function<int(const std::vector<int32_t> &, const std::vector<int32_t> &, std::vector<dh_la> &)> remap;
    int32_t max_bubble_size = 8, max_iterations = 4;  // commandline.d:1826-1837
    int32_t resolved = 0;
};

inline bool is_ext_join(const Edge &e)
{
    return e.s.contig == e.e.contig && ((e.s.part == PRE && e.e.part == BEGIN) || (e.s.part == END && e.e.part == POST));
}

// Paton's cycle base exactly as util/math.d:2380-2480 walks it (roots in node order, LIFO, incident edges in edge order)
std::vector<std::vector<int32_t>> cycle_base(const std::vector<Edge> &g, const Incidence &inc)
{
    const size_t nn = inc.size();
    std::vector<std::vector<int32_t>> used(nn), cycles;
    std::vector<int32_t> parent(nn, -1), stack;
    auto has = [&](int32_t node, int32_t x) { return std::find(used[(size_t)node].begin(), used[(size_t)node].end(), x) != used[(size_t)node].end(); };
    for (int32_t root = 0; root < (int32_t)nn; root++) {
        if (parent[(size_t)root] >= 0 || inc[(size_t)root].empty()) continue;  // (isolated nodes yield nothing)
        parent[(size_t)root] = root;
        used[(size_t)root].push_back(root);
        stack.push_back(root);
        while (!stack.empty()) {
            const int32_t cur = stack.back();
            stack.pop_back();
            const Node me{cur / 4, cur % 4};
            for (int32_t ei : inc[(size_t)cur]) {
                const Edge &e = g[(size_t)ei];
                const Node o = e.s == me ? e.e : e.s;
                const int32_t nb = o.contig * 4 + o.part;
                if (used[(size_t)nb].empty()) {
                    parent[(size_t)nb] = cur;
                    used[(size_t)nb].push_back(cur);
                    stack.push_back(nb);
                } else if (nb == cur)
                    cycles.push_back({cur});
                else if (!has(cur, nb)) {
                    std::vector<int32_t> cyc{nb, cur};
                    int32_t p = parent[(size_t)cur];
                    for (; !has(nb, p); p = parent[(size_t)p]) cyc.push_back(p);
                    cyc.push_back(p);
                    cycles.push_back(std::move(cyc));
                    used[(size_t)nb].push_back(cur);
                }
            }
        }
    }
    return cycles;
}

int resolve_bubbles(std::vector<Edge> &g, int32_t ncontigs, Resolver &rs)
{
    auto find_edge = [&](Node a, Node b) -> Edge * {
        const Edge key = make_edge(a, b);
        auto it = std::lower_bound(g.begin(), g.end(), key, key_less);
        return it != g.end() && key_eq(*it, key) ? &*it : nullptr;
    };
    std::vector<SA> sa;
    std::vector<std::pair<size_t, size_t>> sl;
    for (int32_t iter = 0; iter < rs.max_iterations; iter++) {
        const auto inc = incidence(g, ncontigs);
        std::vector<int32_t> deg(inc.size(), 0);
        for (size_t v = 0; v < inc.size(); v++)
            for (int32_t ei : inc[v]) deg[v] += is_ext_join(g[(size_t)ei]) ? 0 : 1;
        auto node_of = [](int32_t v) { return Node{v / 4, v % 4}; };
        std::vector<std::vector<int32_t>> bubbles;
        for (auto &cyc : cycle_base(g, inc)) {
            if ((int32_t)cyc.size() > rs.max_bubble_size) continue;
            int32_t esc[2], ne = 0;
            bool ok = true;
            for (int32_t v : cyc) {
                if (deg[(size_t)v] >= 3) {
                    if (ne < 2) esc[ne] = v;
                    ne++;
                } else if (deg[(size_t)v] < 2)
                    ok = false;
            }
            if (!ok || ne != 2) continue;
            const Edge *sk = find_edge(node_of(esc[0]), node_of(esc[1]));
            if (sk && (sk->types & T_PILEUP)) bubbles.push_back(cyc);
        }
        if (bubbles.empty()) break;
        for (const auto &cyc : bubbles) {
            int32_t esc[2], ne = 0;
            for (int32_t v : cyc)
                if (deg[(size_t)v] >= 3 && ne < 2) esc[ne++] = v;
            Edge *sk = find_edge(node_of(esc[0]), node_of(esc[1]));
            if (!sk || !(sk->types & T_PILEUP)) continue;  // resolved through another bubble of this iteration
            // the skipping reads and the contigs they skip
            std::vector<int32_t> inter, rids;
            for (int32_t v : cyc)
                if (deg[(size_t)v] == 2) inter.push_back(v / 4);
            std::sort(inter.begin(), inter.end());
            inter.erase(std::unique(inter.begin(), inter.end()), inter.end());
            for (const dh_read_alignment &ra : sk->ras) rids.push_back(ra.read);
            std::sort(rids.begin(), rids.end());
            rids.erase(std::unique(rids.begin(), rids.end()), rids.end());
            std::vector<dh_la> fresh;
            if (int rc = rs.remap(inter, rids, fresh)) return rc;
            const int64_t first_new = rs.c.n + (int64_t)rs.extra->size();
            rs.extra->insert(rs.extra->end(), fresh.begin(), fresh.end());
            // the skipped path: from the skipping edge's start around the cycle to its end (the long way)
            const int32_t v0 = sk->s.contig * 4 + sk->s.part, v1 = sk->e.contig * 4 + sk->e.part;
            const int32_t L = (int32_t)cyc.size();
            const int32_t i0 = (int32_t)(std::find(cyc.begin(), cyc.end(), v0) - cyc.begin()),
                          i1 = (int32_t)(std::find(cyc.begin(), cyc.end(), v1) - cyc.begin());
            auto walk = [&](int32_t a, int32_t b) {
                std::vector<Node> p;
                for (int32_t x = 0; x <= ((b - a) % L + L) % L; x++) p.push_back(node_of(cyc[(size_t)((a + x) % L)]));
                return p;
            };
            std::vector<Node> path = walk(i0, i1);
            if (path.size() == 2) path = walk(i1, i0);
            if (path.size() <= 2) return dh_fail(DH_EINVAL, "resolveBubbles: skipped path is too short");
            // old and new alignments of every skipping read, in that order (pileups.d:1262-1266), enabled ones only
            std::vector<std::pair<int32_t, int64_t>> al;  // (read, LA)
            for (const dh_read_alignment &ra : sk->ras) {
                al.emplace_back(ra.read, (int64_t)ra.la0);
                if (ra.n == 2) al.emplace_back(ra.read, (int64_t)ra.la1);
            }
            for (size_t x = 0; x < fresh.size(); x++)
                if (!(fresh[x].flags & DH_FLAG_DISABLED)) al.emplace_back(fresh[x].bread, first_new + (int64_t)x);
            std::stable_sort(al.begin(), al.end(), [](const auto &p, const auto &q) { return p.first < q.first; });
            std::vector<RawJoin> raw;
            for (size_t i = 0; i < al.size();) {
                size_t j = i;
                std::vector<int64_t> idx;
                while (j < al.size() && al[j].first == al[i].first) idx.push_back(al[j++].second);
                const size_t r0 = raw.size();
                read_joins(rs.c, idx.data(), (int64_t)idx.size(), raw, sa, sl);
                if (raw.size() > r0) {
                    // collectFixedSimpleBubbles: the seeded alignments (read order) must walk the skipped path
                    auto matches = [&](const Node &nd, const SA &x) {
                        return nd.contig == rs.c.L(x.la).aread && ((nd.part == BEGIN && x.seed == FRONT) || (nd.part == END && x.seed == BACK));
                    };
                    const bool rev = path[0].contig != rs.c.L(sa[0].la).aread;
                    bool good = true;
                    size_t at = 0;
                    auto pnode = [&](size_t x) { return rev ? path[path.size() - 1 - x] : path[x]; };
                    while (at < sa.size() && !matches(pnode(0), sa[at])) at++;
                    if (at == sa.size() || path.size() > sa.size() - at) good = false;
                    for (size_t x = 0; good && x < path.size(); x++) good = matches(pnode(x), sa[at + x]);
                    if (!good) raw.resize(r0);
                }
                i = j;
            }
            sk->types &= ~(uint32_t)T_PILEUP;
            sk->ras.clear();
            std::vector<Edge> add;
            // one edge per read alignment, merged by the stable multi-edge merge (bulkAdd!mergeJoins)
            for (const RawJoin &r : raw) {
                Edge e;
                e.s = r.s;
                e.e = r.e;
                e.types = T_PILEUP;
                e.ras.push_back(r.ra);
                add.push_back(std::move(e));
            }
            for (Edge &e : add) g.push_back(std::move(e));
            merge_multi_edges(g);
            rs.resolved++;
        }
        remove_none_joins(g);
    }
    return DH_OK;
}

int scaffold_from_runs(std::vector<std::vector<RawJoin>> &found, int32_t ncontigs, const int32_t *input_gaps, int32_t ngaps,
                       const dh_scaffold_opts *opts, dh_scaffold **out, Resolver *rs)
{
    auto T0_ = std::chrono::steady_clock::now();
    auto LAP_ = [&](const char *w) { if (getenv("DH_TRACE")) { auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[scaffold] %-24s %.2f ms\n", w, std::chrono::duration<double, std::milli>(t - T0_).count()); T0_ = t; } };
    // ---- the scaffold: default edges, read joins, input gaps (buildScaffold, scaffold.d:237-244)
    std::vector<Edge> g;
    for (int32_t ct = 0; ct < ncontigs; ct++) g.push_back(make_edge(Node{ct, BEGIN}, Node{ct, END}));
    // the runs' raw joins (their concatenation is in read order) become edges by ONE stable sort by edge, spread over the
    // host threads: the key space is cut into buckets by the start contig, every run counts and then scatters its joins
    // into the buckets' slices (bucket-major, run order inside: the concatenation's order), every bucket is sorted stably
    // by edge on its own and yields its edges with their read alignments in read order -- an edge's array is allocated
    // once, at its size.  (Edges per run first, then a merge of the runs' edge lists: reads lie anywhere, so a run of
    // 4 096 joins of configs[2] held 2 500 edges of 1.6 read alignments -- 100 000 small arrays allocated, merged and freed,
    // 3.3 of the plan's 8 ms at N = 8; before that one serial stable sort over all edges: 7.7 of the collect stage's 14 ms.)
    {
        const int32_t nbuck = (int32_t)std::max<int64_t>(1, std::min<int64_t>(256, ncontigs / 4));
        const int64_t nruns = (int64_t)found.size();
        auto bucket_of = [&](int32_t contig) {  // (monotone in the contig; ids are checked by the callers, the clamp keeps a stray one inside)
            const int64_t b = (int64_t)contig * nbuck / std::max(ncontigs, 1);
            return (int32_t)std::min<int64_t>(std::max<int64_t>(b, 0), nbuck - 1);
        };
        std::vector<int64_t> at((size_t)nruns * (size_t)nbuck + 1, 0);  // [bucket][run] -> first slot
        dh_parallel_for(nruns, 1, [&](int64_t lo, int64_t hi) {
            for (int64_t run = lo; run < hi; run++)
                for (const RawJoin &r : found[(size_t)run]) at[(size_t)bucket_of(r.s.contig) * (size_t)nruns + (size_t)run]++;
        });
        int64_t total = 0;
        for (size_t i = 0; i < (size_t)nruns * (size_t)nbuck; i++) {
            const int64_t c = at[i];
            at[i] = total;
            total += c;
        }
        at[(size_t)nruns * (size_t)nbuck] = total;
        std::vector<const RawJoin *> item((size_t)total);
        dh_parallel_for(nruns, 1, [&](int64_t lo, int64_t hi) {
            std::vector<int64_t> cur((size_t)nbuck);
            for (int64_t run = lo; run < hi; run++) {
                for (int32_t bk = 0; bk < nbuck; bk++) cur[(size_t)bk] = at[(size_t)bk * (size_t)nruns + (size_t)run];
                for (const RawJoin &r : found[(size_t)run]) item[(size_t)cur[(size_t)bucket_of(r.s.contig)]++] = &r;
            }
        });
        std::vector<std::vector<Edge>> merged((size_t)nbuck);
        dh_parallel_for(nbuck, 1, [&](int64_t blo, int64_t bhi) {
            for (int64_t bk = blo; bk < bhi; bk++) {
                const RawJoin **i0 = item.data() + at[(size_t)bk * (size_t)nruns], **i1 = item.data() + at[(size_t)(bk + 1) * (size_t)nruns];
                std::stable_sort(i0, i1, [](const RawJoin *a, const RawJoin *b) { return a->s == b->s ? a->e < b->e : a->s < b->s; });
                std::vector<Edge> &out = merged[(size_t)bk];
                for (const RawJoin **i = i0; i < i1;) {
                    const RawJoin **j = i + 1;
                    while (j < i1 && (*j)->s == (*i)->s && (*j)->e == (*i)->e) j++;
                    Edge m;
                    m.s = (*i)->s;
                    m.e = (*i)->e;
                    m.types = T_PILEUP;
                    m.ras.reserve((size_t)(j - i));
                    for (const RawJoin **x = i; x < j; x++) m.ras.push_back((*x)->ra);
                    out.push_back(std::move(m));
                    i = j;
                }
            }
        });
        for (auto &v : merged)
            for (auto &e : v) g.push_back(std::move(e));
    }
    found.clear();
    for (int32_t x = 0; x < ngaps; x++) {
        Edge e = make_edge(Node{input_gaps[2 * x], END}, Node{input_gaps[2 * x + 1], BEGIN});
        e.types = T_INPUTGAP;
        g.push_back(std::move(e));
    }
    merge_multi_edges(g);
    remove_none_joins(g);
    LAP_("graph merge");
    if (rs) {
        if (int rc = resolve_bubbles(g, ncontigs, *rs)) return rc;
        LAP_("resolve bubbles");
    }
    // ---- discardAmbiguousJoins (pileups.d:1592-1657)
    {
        const auto inc = incidence(g, ncontigs);
        std::vector<char> drop(g.size(), 0);
        for (int32_t ct = 0; ct < ncontigs; ct++)
            for (int32_t part : {BEGIN, END}) {
                const auto in = inc[(size_t)ct * 4 + part];
                if (in.size() <= 2) continue;
                std::vector<int32_t> gj;
                for (int32_t ei : in)
                    if (g[(size_t)ei].is_gap() && (g[(size_t)ei].types & T_PILEUP)) gj.push_back(ei);
                if (gj.size() <= 1) continue;
                // findCorrectGapJoin (pileups.d:1754-1804): the best-supported join wins if it beats the
                // runner-up by the margin; joins of the input assembly get a bonus
                std::vector<std::pair<double, size_t>> val;
                for (size_t x = 0; x < gj.size(); x++) {
                    const Edge &e = g[(size_t)gj[x]];
                    val.emplace_back((double)e.ras.size() * ((e.types & T_INPUTGAP) ? opts->existing_gap_bonus : 1.0), x);
                }
                std::stable_sort(val.begin(), val.end(), [](const auto &p, const auto &q) { return p.first > q.first; });
                const size_t keep = val[1].first * opts->best_pile_up_margin < val[0].first ? val[0].second : gj.size();
                for (size_t x = 0; x < gj.size(); x++)
                    if (x != keep) drop[(size_t)gj[x]] = 1;
            }
        for (size_t i = 0; i < g.size(); i++)
            if (drop[i]) {
                g[i].types &= ~(uint32_t)T_PILEUP;
                g[i].ras.clear();
            }
        remove_none_joins(g);
    }
    LAP_("discard ambiguous");
    // ---- enforceMinSpanningReads, removeInputGaps (pileups.d:1807-1852)
    for (Edge &e : g)
        if ((e.types & T_PILEUP) && e.is_gap() && (int64_t)e.ras.size() < opts->min_spanning_reads) {
            e.types &= ~(uint32_t)T_PILEUP;
            e.ras.clear();
        }
    remove_none_joins(g);
    for (Edge &e : g) e.types &= ~(uint32_t)T_INPUTGAP;
    remove_none_joins(g);
    // ---- mergeExtensionsWithGaps (scaffold.d:789-816): emptied edges stay until the end
    if (opts->merge_extensions) {
        const auto inc = incidence(g, ncontigs);
        for (int32_t ct = 0; ct < ncontigs; ct++)
            for (int32_t part : {BEGIN, END}) {
                const auto in = inc[(size_t)ct * 4 + part];
                if (in.size() > 3) return dh_fail(DH_EINVAL, "dh_scaffold_pileups: node degree must be <= 3");
                if (in.size() != 3) continue;
                int32_t nd[2], k = 0;
                for (int32_t ei : in)
                    if (!g[(size_t)ei].is_default() && k < 2) nd[k++] = ei;
                if (k != 2) continue;
                const Node me{ct, part};
                auto other = [&](const Edge &e) { return e.s == me ? e.e : e.s; };
                const int gi = real_part(other(g[(size_t)nd[0]]).part) ? 0 : 1;
                Edge &gap = g[(size_t)nd[gi]], &ext = g[(size_t)nd[1 - gi]];
                gap.types |= ext.types;
                gap.ras.insert(gap.ras.end(), ext.ras.begin(), ext.ras.end());
                ext.types = 0;
                ext.ras.clear();
            }
        remove_none_joins(g);
    }
    LAP_("min reads, extensions");
    // ---- collectPileUps (pileups.d:435-444): valid pile-ups in edge order
    dh_scaffold *res = new dh_scaffold();
    for (const Edge &e : g) {
        if (!(e.types & T_PILEUP) || e.ras.empty()) continue;
        bool any_gap = false, all_front = true, all_back = true;
        for (const dh_read_alignment &ra : e.ras) {
            any_gap = any_gap || ra.n == 2;
            all_front = all_front && ra.n == 1 && ra.seed0 == FRONT;
            all_back = all_back && ra.n == 1 && ra.seed0 == BACK;
        }
        const bool is_ext = all_front || all_back;
        if (is_ext == any_gap) continue;  // PileUp.isValid, base.d:2755-2790
        dh_join j;
        j.contig0 = e.s.contig;
        j.part0 = e.s.part;
        j.contig1 = e.e.contig;
        j.part1 = e.e.part;
        j.type = any_gap ? 1 : (all_front ? 0 : 2);  // ReadAlignmentType: front, gap, back
        j.count = (int32_t)e.ras.size();
        j.first = (int64_t)res->entries.size();
        res->joins.push_back(j);
        res->entries.insert(res->entries.end(), e.ras.begin(), e.ras.end());
    }
    LAP_("collect pile-ups");
    *out = res;
    return DH_OK;
}
}  // namespace

extern "C" int32_t dh_scaffold_npiles(const dh_scaffold *s) { return s ? (int32_t)s->joins.size() : 0; }
extern "C" int64_t dh_scaffold_nentries(const dh_scaffold *s) { return s ? (int64_t)s->entries.size() : 0; }
extern "C" const dh_join *dh_scaffold_joins(const dh_scaffold *s) { return s ? s->joins.data() : nullptr; }
extern "C" const dh_read_alignment *dh_scaffold_entries(const dh_scaffold *s) { return s ? s->entries.data() : nullptr; }
extern "C" void dh_scaffold_destroy(dh_scaffold *s) { delete s; }

// The gap pile-ups the process path handles: joins (c, end) -- (c + 1, begin) whose spanning reads see
// both contigs in the same orientation; of every such pile-up the spanning read alignments (left LA
// seeded at the back, right LA at the front) become (read, left LA, right LA) triples ordered by read.
// Everything else (extension pile-ups, joins between other contig ends, extension reads merged into a
// gap) is counted in *skipped.
extern "C" int dh_scaffold_spanning(const dh_scaffold *s, const dh_la *las, int64_t n, dh_pileups **out, int32_t *skipped)
{
    if (!s || !out || (n > 0 && !las)) return dh_fail(DH_EINVAL, "dh_scaffold_spanning: bad argument");
    std::vector<int32_t> cl, cnt, tri;
    int32_t skip = 0;
    for (const dh_join &j : s->joins) {
        if (!(j.type == 1 && j.part0 == END && j.part1 == BEGIN && j.contig1 == j.contig0 + 1)) {
            skip++;
            continue;
        }
        std::vector<std::array<int32_t, 3>> t;
        for (int64_t x = j.first; x < j.first + j.count; x++) {
            const dh_read_alignment &ra = s->entries[(size_t)x];
            if (ra.n != 2 || ra.la0 < 0 || ra.la0 >= n || ra.la1 < 0 || ra.la1 >= n) continue;
            if ((las[ra.la0].flags & DH_FLAG_COMP) != (las[ra.la1].flags & DH_FLAG_COMP)) continue;
            t.push_back({ra.read, ra.la0, ra.la1});
        }
        if (t.empty()) {
            skip++;
            continue;
        }
        std::stable_sort(t.begin(), t.end(), [](const auto &a, const auto &b) { return a[0] < b[0]; });
        cl.push_back(j.contig0);
        cnt.push_back((int32_t)t.size());
        for (auto &x : t) tri.insert(tri.end(), x.begin(), x.end());
    }
    if (skipped) *skipped = skip;
    return dh_pileups_create(cl.data(), cnt.data(), (int32_t)cl.size(), tri.data(), out);
}

// The pile-ups of the gap joins (c, end)--(c + 1, begin) with EVERY read alignment the builder put into
// them: reads spanning the gap as (read, left LA, right LA) and the extension-type read alignments that
// mergeExtensionsWithGaps (scaffold.d:789-816) moved into the gap -- (read, LA, -1) for a back extension
// of the left contig, (read, -1, LA) for a front extension of the right one -- which the reference crops
// and aligns like any other member of the pile-up (cropper.d:113-175, 339-361).  Entries are ordered by
// read, then by their position in the builder's list.  *skipped = pile-ups of any other kind.
extern "C" int dh_scaffold_gap_pileups(const dh_scaffold *s, const dh_la *las, int64_t n, dh_pileups **out, int32_t *skipped)
{
    if (!s || !out || (n > 0 && !las)) return dh_fail(DH_EINVAL, "dh_scaffold_gap_pileups: bad argument");
    // joins are independent: host threads fill one triple list per join, the lists are concatenated in join order
    const int64_t nj = (int64_t)s->joins.size();
    std::vector<std::vector<std::array<int32_t, 3>>> per((size_t)nj);
    dh_parallel_for(nj, 16, [&](int64_t lo, int64_t hi) {
        for (int64_t ji = lo; ji < hi; ji++) {
            const dh_join &j = s->joins[(size_t)ji];
            if (!(j.type == 1 && j.part0 == END && j.part1 == BEGIN && j.contig1 == j.contig0 + 1)) continue;
            std::vector<std::array<int32_t, 3>> &t = per[(size_t)ji];
            t.reserve((size_t)j.count);
            for (int64_t x = j.first; x < j.first + j.count; x++) {
                const dh_read_alignment &ra = s->entries[(size_t)x];
                if (ra.la0 < 0 || ra.la0 >= n) continue;
                if (ra.n == 2) {
                    if (ra.la1 < 0 || ra.la1 >= n) continue;
                    if ((las[ra.la0].flags & DH_FLAG_COMP) != (las[ra.la1].flags & DH_FLAG_COMP)) continue;
                    t.push_back({ra.read, ra.la0, ra.la1});
                } else if (las[ra.la0].aread == j.contig0 && ra.seed0 == BACK)
                    t.push_back({ra.read, ra.la0, -1});
                else if (las[ra.la0].aread == j.contig1 && ra.seed0 == FRONT)
                    t.push_back({ra.read, -1, ra.la0});
            }
            std::stable_sort(t.begin(), t.end(), [](const auto &a, const auto &b) { return a[0] < b[0]; });
        }
    });
    std::vector<int32_t> cl, cnt, tri;
    int32_t skip = 0;
    size_t total = 0;
    for (const auto &t : per) total += t.size();
    tri.reserve(3 * total);
    for (int64_t ji = 0; ji < nj; ji++) {
        const auto &t = per[(size_t)ji];
        if (t.empty()) {
            skip++;
            continue;
        }
        cl.push_back(s->joins[(size_t)ji].contig0);
        cnt.push_back((int32_t)t.size());
        for (const auto &x : t) tri.insert(tri.end(), x.begin(), x.end());
    }
    if (skipped) *skipped = skip;
    return dh_pileups_create(cl.data(), cnt.data(), (int32_t)cl.size(), tri.data(), out);
}

// Every pile-up of the scaffold the process stage can take (`dentist process` is handed all of them; `--only` of
// `dentist output`, commandline.d:2230-2250, decides later which insertions are used): only & 1 = the gap joins of
// any two contig ends -- same orientation, anti-parallel ((c, end) -> (d, end), (c, begin) -> (d, begin)), contig-
// skipping --, only & 2 = the extension joins ((c, pre) -> (c, begin), (c, end) -> (c, post)).  Pile-ups come with
// their nodes (dh_pileups_create_joins); entries are (read, LA on flank 0, LA on flank 1), extension-type read
// alignments merged into a gap have one of the two.  A read alignment whose two complement flags do not fit the
// join (parallel: equal, anti-parallel: different) is left out, as dh_scaffold_gap_pileups does.  *skipped = joins
// without entries, joins of a contig with itself.
extern "C" int dh_scaffold_all_pileups(const dh_scaffold *s, const dh_la *las, int64_t n, int32_t only, dh_pileups **out,
                                       int32_t *skipped)
{
    if (!s || !out || (n > 0 && !las) || (only & ~3) || !only) return dh_fail(DH_EINVAL, "dh_scaffold_all_pileups: bad argument");
    const int64_t nj = (int64_t)s->joins.size();
    std::vector<std::vector<std::array<int32_t, 3>>> per((size_t)nj);
    std::vector<std::array<int32_t, 4>> node((size_t)nj);
    dh_parallel_for(nj, 16, [&](int64_t lo, int64_t hi) {
        for (int64_t ji = lo; ji < hi; ji++) {
            const dh_join &j = s->joins[(size_t)ji];
            const bool gap = (j.part0 == BEGIN || j.part0 == END) && (j.part1 == BEGIN || j.part1 == END) && j.contig0 != j.contig1;
            const bool fext = j.contig0 == j.contig1 && j.part0 == PRE && j.part1 == BEGIN;
            const bool bext = j.contig0 == j.contig1 && j.part0 == END && j.part1 == POST;
            if (!((gap && (only & 1)) || ((fext || bext) && (only & 2)))) continue;
            const int32_t c0 = j.contig0, c1 = gap ? j.contig1 : -1;
            const int32_t s0 = gap ? (j.part0 == BEGIN ? FRONT : BACK) : (fext ? FRONT : BACK), s1 = gap ? (j.part1 == BEGIN ? FRONT : BACK) : 0;
            node[(size_t)ji] = {c0, s0 == FRONT ? DH_SEED_FRONT : DH_SEED_BACK, c1, gap && s1 == BACK ? DH_SEED_BACK : DH_SEED_FRONT};
            std::vector<std::array<int32_t, 3>> &t = per[(size_t)ji];
            t.reserve((size_t)j.count);
            auto flank_of = [&](int32_t la, int32_t seed) {
                if (las[la].aread == c0 && seed == s0) return 0;
                if (gap && las[la].aread == c1 && seed == s1) return 1;
                return -1;
            };
            for (int64_t x = j.first; x < j.first + j.count; x++) {
                const dh_read_alignment &ra = s->entries[(size_t)x];
                if (ra.la0 < 0 || ra.la0 >= n) continue;
                const int f0 = flank_of(ra.la0, ra.seed0);
                if (ra.n == 2) {
                    if (ra.la1 < 0 || ra.la1 >= n) continue;
                    const int f1 = flank_of(ra.la1, ra.seed1);
                    if (f0 < 0 || f1 < 0 || f0 == f1) continue;
                    const bool same = (las[ra.la0].flags & DH_FLAG_COMP) == (las[ra.la1].flags & DH_FLAG_COMP);
                    if (same != (s0 != s1)) continue;
                    t.push_back({ra.read, f0 == 0 ? ra.la0 : ra.la1, f0 == 0 ? ra.la1 : ra.la0});
                } else if (f0 == 0)
                    t.push_back({ra.read, ra.la0, -1});
                else if (f0 == 1)
                    t.push_back({ra.read, -1, ra.la0});
            }
            std::stable_sort(t.begin(), t.end(), [](const auto &a, const auto &b) { return a[0] < b[0]; });
        }
    });
    // dh_pileups_create_joins wants the pile-ups in node order: (contig0, seed0, contig1 or "none" last, seed1)
    std::vector<int64_t> order;
    int32_t skip = 0;
    for (int64_t ji = 0; ji < nj; ji++) {
        if (per[(size_t)ji].empty()) {
            skip++;
            continue;
        }
        order.push_back(ji);
    }
    auto key = [&](int64_t ji) {
        const auto &q = node[(size_t)ji];
        return std::array<int64_t, 4>{q[0], q[1], q[2] < 0 ? INT32_MAX : q[2], q[3]};
    };
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return key(a) < key(b); });
    std::vector<int32_t> nd, cnt, tri;
    for (int64_t ji : order) {
        nd.insert(nd.end(), node[(size_t)ji].begin(), node[(size_t)ji].end());
        cnt.push_back((int32_t)per[(size_t)ji].size());
        for (const auto &x : per[(size_t)ji]) tri.insert(tri.end(), x.begin(), x.end());
    }
    if (skipped) *skipped = skip;
    return dh_pileups_create_joins(nd.data(), cnt.data(), (int32_t)cnt.size(), tri.data(), out);
}

// dh_scaffold_pileups with resolveBubbles (pileups.d:1124-1315) between the raw scaffold and discardAmbiguousJoins, as
// build() runs it (pileups.d:186).  The alignments the resolver adds are returned in *extra: LA index n + i of the
// result's read alignments = record i of *extra (ids of the full DBs; traces for the cropper when the device maps).
static int scaffold_resolved(const dh_la *las, int64_t n, const int64_t *contig_off, int32_t ncontigs, const int64_t *read_off,
                             int32_t nreads, const int32_t *input_gaps, int32_t ngaps, const dh_scaffold_opts *opts,
                             int32_t max_bubble_size, int32_t max_iterations,
                             std::function<int(const std::vector<int32_t> &, const std::vector<int32_t> &, std::vector<dh_la> &)> remap,
                             std::vector<dh_la> &extra, dh_scaffold **out, int32_t *resolved)
{
    for (int32_t g = 0; g < ngaps; g++)
        if (input_gaps[2 * g] < 0 || input_gaps[2 * g] >= ncontigs || input_gaps[2 * g + 1] < 0 || input_gaps[2 * g + 1] >= ncontigs)
            return dh_fail(DH_EINVAL, "dh_scaffold_pileups: input gap names a contig out of range");
    // (chains as units for the mapping's alignments, as in dh_scaffold_pileups; the re-mapped alignments are taken
    // record by record: they cover a whole intermediate contig)
    dh_chain_view cv;
    dh_chain_view_build(las, n, cv);
    const dh_la *u = cv.trivial ? las : cv.unit.data();
    const int64_t nu = cv.trivial ? n : (int64_t)cv.unit.size();
    std::vector<std::vector<RawJoin>> found;
    if (int rc = collect_raw_joins("dh_scaffold_pileups", u, nu, contig_off, ncontigs, read_off, 0, nreads, &found)) return rc;
    Resolver rs;
    rs.c = Ctx{u, contig_off, read_off};
    rs.c.n = nu;
    rs.c.extra = &extra;
    rs.extra = &extra;
    rs.remap = [&](const std::vector<int32_t> &cids, const std::vector<int32_t> &rids, std::vector<dh_la> &fresh) -> int {
        if (int rc = remap(cids, rids, fresh)) return rc;
        for (const dh_la &l : fresh)
            if (l.aread < 0 || l.aread >= ncontigs || l.bread < 0 || l.bread >= nreads)
                return dh_fail(DH_EINVAL, "resolveBubbles: the re-mapping returned an id outside the DBs");
        return DH_OK;
    };
    rs.max_bubble_size = max_bubble_size > 0 ? max_bubble_size : 8;
    rs.max_iterations = max_iterations > 0 ? max_iterations : 4;
    const int rc = scaffold_from_runs(found, ncontigs, input_gaps, ngaps, opts, out, &rs);
    if (resolved) *resolved = rs.resolved;
    if (!rc) {
        // unit index -> record index: the first record of the chain; the added alignments follow the n records
        auto rec_of = [&](int32_t x) { return x < nu ? (int32_t)(cv.trivial ? x : cv.first[(size_t)x]) : (int32_t)(n + (x - nu)); };
        for (dh_read_alignment &ra : (*out)->entries) {
            ra.la0 = rec_of(ra.la0);
            if (ra.n == 2) ra.la1 = rec_of(ra.la1);
        }
    }
    return rc;
}

// host only: the re-mapping is the caller's (tests feed the oracle's alignments; a D host could spawn damapper)
extern "C" int dh_scaffold_pileups_cb(const dh_la *las, int64_t n, const int64_t *contig_off, int32_t ncontigs, const int64_t *read_off,
                                      int32_t nreads, const int32_t *input_gaps, int32_t ngaps, const dh_scaffold_opts *opts,
                                      int32_t max_bubble_size, int32_t max_iterations, dh_remap_fn remap, void *user,
                                      dh_scaffold **out, dh_la_set **extra, int32_t *resolved)
{
    if ((n > 0 && !las) || !contig_off || !read_off || !opts || !out || !extra || !remap || ncontigs < 0 || nreads < 0 || n < 0 ||
        n >= (1ll << 30) || (ngaps > 0 && !input_gaps) || ngaps < 0)
        return dh_fail(DH_EINVAL, "dh_scaffold_pileups_cb: bad argument");
    dh_la_set *ex = new dh_la_set();
    std::vector<dh_la> added;
    const int rc = scaffold_resolved(las, n, contig_off, ncontigs, read_off, nreads, input_gaps, ngaps, opts, max_bubble_size,
                                     max_iterations,
                                     [&](const std::vector<int32_t> &cids, const std::vector<int32_t> &rids, std::vector<dh_la> &fresh) -> int {
                                         dh_la *p = nullptr;
                                         int64_t cnt = 0;
                                         if (int rc2 = remap(user, cids.data(), (int32_t)cids.size(), rids.data(), (int32_t)rids.size(), &p, &cnt)) {
                                             free(p);
                                             return dh_fail(rc2, "resolveBubbles: the re-mapping callback failed");
                                         }
                                         if (cnt < 0 || (cnt > 0 && !p)) return dh_fail(DH_EINVAL, "resolveBubbles: bad callback result");
                                         fresh.assign(p, p + cnt);
                                         free(p);
                                         return DH_OK;
                                     },
                                     added, out, resolved);
    if (rc) {
        delete ex;
        return rc;
    }
    ex->la.assign(added.begin(), added.end());
    *extra = ex;
    return DH_OK;
}

// the device maps: dh_remap_skipping_reads with the mapping options of the caller, no mask
extern "C" int dh_scaffold_pileups_resolved(dh_ctx *ctx, dh_db *contigs, dh_db *reads, const dh_la *las, int64_t n,
                                            const int32_t *input_gaps, int32_t ngaps, const dh_scaffold_opts *opts,
                                            const dh_align_opts *map_opts, int32_t allowance, int32_t max_bubble_size,
                                            int32_t max_iterations, dh_scaffold **out, dh_la_set **extra, int32_t *resolved)
{
    if (!ctx || !contigs || !reads || (n > 0 && !las) || !opts || !map_opts || !out || !extra || n < 0 || n >= (1ll << 30) ||
        (ngaps > 0 && !input_gaps) || ngaps < 0 || allowance < 0)
        return dh_fail(DH_EINVAL, "dh_scaffold_pileups_resolved: bad argument");
    dh_la_set *ex = new dh_la_set();
    ex->tspace = map_opts->tspace;
    std::vector<dh_la> added;
    const int rc = scaffold_resolved(
        las, n, contigs->h_off.data(), contigs->n, reads->h_off.data(), reads->n, input_gaps, ngaps, opts, max_bubble_size, max_iterations,
        [&](const std::vector<int32_t> &cids, const std::vector<int32_t> &rids, std::vector<dh_la> &fresh) -> int {
            dh_la_set *set = nullptr;
            if (int rc2 = dh_remap_skipping_reads(ctx, contigs, reads, cids.data(), (int32_t)cids.size(), rids.data(), (int32_t)rids.size(),
                                                  map_opts, allowance, &set))
                return rc2;
            const int64_t t0 = (int64_t)ex->trace.size();
            fresh.assign(set->la.begin(), set->la.end());
            for (dh_la &l : fresh) l.toff += t0;
            ex->trace.insert(ex->trace.end(), set->trace.begin(), set->trace.end());
            dh_la_set_destroy(set);
            return DH_OK;
        },
        added, out, resolved);
    if (rc) {
        delete ex;
        return rc;
    }
    ex->la.assign(added.begin(), added.end());
    *extra = ex;
    return DH_OK;
}

