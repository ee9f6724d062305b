// dh_process.cpp -- host side of the pile-up consensus path: the call sequence of
// `dentist process` (source/dentist/commands/processPileUps/package.d:283-374) over a BATCH of
// pile-ups, with every tool spawn of the reference replaced by kernels on the context's stream:
//   crop (cropper.d:113-175, 446-550)            -> k_gather_slices
//   daligner pile-up all-vs-all (package.d:478)  -> dh_align_db on the grouped pile-up DB
//   filters (dazzler.d:3879-3899, 4043-4141)     -> host flags (same predicates as the reference)
//   DASqv (dazzler.d:6142-6156)                  -> k_tile_qv
//   reference-read ranking (package.d:518-568)   -> host (the reference's own D logic)
//   daccord (dazzler.d:6185-6231)                -> k_seg_vote + k_emit, `rounds` times
//   daligner -A flanks vs consensus (:655-667)   -> dh_align_db
//   insertion (package.d:699-805, insertions.d:110-146) -> host
#include <array>
#include <chrono>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <numeric>

#include "dh_internal.h"
#include "dh_parallel.h"

extern "C" {
void dhk_gather_slices(hipStream_t st, const uint8_t *src, const int64_t *src_off, const int32_t *sidx,
                       const int32_t *sbeg, const int64_t *dst_off, int32_t n, int32_t max_len,
                       uint8_t *dst);
void dhk_gather_parts(hipStream_t st, const uint8_t *src0, const int64_t *off0, const uint8_t *src1,
                      const int64_t *off1, const void *parts, int32_t n, int32_t max_len, uint8_t *dst);
void dhk_gather_ranges16(hipStream_t st, const uint16_t *src, const int64_t *desc, int32_t n, uint16_t *dst);
void dhk_tile_qv(hipStream_t st, const DhLa *las, const uint16_t *trace, const int32_t *la_first,
                 const int64_t *roff, int32_t nreads, int32_t tspace, const int32_t *cov, int32_t maxtiles,
                 uint8_t *qv);
void dhk_pile_funnel(hipStream_t st, DhLa *las, const uint32_t *item_off, int32_t nreads, const int64_t *roff,
                     int32_t max_err_ppm, int32_t tsp, int32_t *la_first, int32_t *live, int32_t *status);
void dhk_gather_read_records(hipStream_t st, const DhLa *las, const int32_t *la_first, const int32_t *sel,
                             const int32_t *dst_off, int32_t nsel, DhLa *out);
void dhk_seg_vote(hipStream_t st, const void *segs, int32_t nseg, DbView T, DbView R,
                  const uint8_t *rrc, const int64_t *voff, uint32_t *dmat, int32_t bandmax, int32_t qmax,
                  int32_t ncolmax, uint8_t *opbuf, uint16_t *nops, uint32_t *votes, uint32_t *cdiff,
                  uint32_t *vother, int32_t *status, int32_t mode);
void dhk_votes_finish(hipStream_t st, DbView T, const int64_t *voff, const int32_t *col_tmpl, int64_t ncols_total,
                      const uint32_t *cexcl, const uint32_t *vother, uint32_t *votes);
void dhk_scan(hipStream_t st, uint32_t *v, int64_t n, uint32_t *sums);
void dhk_col_tmpl(hipStream_t st, const int64_t *voff, int32_t ntmpl, int64_t ncols_total, int32_t *col_tmpl);
void dhk_emit(hipStream_t st, DbView T, int32_t ntmpl, const int64_t *voff, const uint32_t *votes,
              const int32_t *col_tmpl, int64_t ncols_total, uint8_t *stage, uint8_t *cnt,
              const int64_t *out_off, uint8_t *out, int32_t *out_len);
}

#define MAXINS 4
#define VSTRIDE (6 + 4 * MAXINS)
#define MAXQV 50
#define SEG_MAX 250
#define FLAG_IMPROPER 0x40u /* internal: fails isValidPileUpAlignment, dropped after the tile QVs */

struct PartDescH {
    int32_t src, sidx, sbeg, len, rc, pad;
    int64_t dst;
};

struct SegDescH {
    int32_t tmpl, a0, a1, bseq, b0, b1, comp, band;
};

extern "C" void dh_default_process_opts(dh_process_opts *o)
{
    memset(o, 0, sizeof(*o));
    o->tspace_map = 100;
    o->allowance = 100;
    o->min_anchor = 500;
    o->min_reads = 3;
    o->max_reads = 60;
    o->tspace_pile = 126;
    o->rounds = 3;
    o->flank_window = 20000;
    o->max_align_err_ppm = 300000;
    o->max_ins_err_ppm = 100000;
    o->bad_fraction_ppm = 80000;
    o->width = 30;
    o->dust = 1;
    o->min_relative_score_ppm = 1000000;
}

// ------------------------------------------------------------------------------------ DB helpers

int dh_db_adopt(dh_ctx *ctx, uint8_t *d_alloc, uint8_t *d_bases, const std::vector<int64_t> &off,
                const std::vector<int32_t> &group, dh_db **out)
{
    dh_db *db = new dh_db();
    db->ctx = ctx;
    db->n = (int32_t)off.size() - 1;
    db->h_off = off;
    db->total = off.back();
    db->d_bases = d_bases;
    db->d_bases_alloc = d_alloc;
    for (int32_t i = 0; i < db->n; i++)
        db->max_len = std::max<int32_t>(db->max_len, (int32_t)(off[(size_t)i + 1] - off[(size_t)i]));
    HIPCHK(dh_dev_alloc(&db->d_off, sizeof(int64_t) * off.size()));
    HIPCHK(hipMemcpyAsync(db->d_off, off.data(), sizeof(int64_t) * off.size(), hipMemcpyHostToDevice,
                          ctx->stream));
    if (!group.empty()) {
        db->h_group = group;
        for (int32_t g : group) db->ngroups = std::max(db->ngroups, g + 1);
        HIPCHK(dh_dev_alloc(&db->d_group, sizeof(int32_t) * group.size()));
        HIPCHK(hipMemcpyAsync(db->d_group, group.data(), sizeof(int32_t) * group.size(),
                              hipMemcpyHostToDevice, ctx->stream));
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    *out = db;
    return DH_OK;
}

int dh_db_from_slices(dh_ctx *ctx, const dh_db *src, const std::vector<int32_t> &sidx,
                      const std::vector<int32_t> &sbeg, const std::vector<int32_t> &slen,
                      const std::vector<int32_t> &group, dh_db **out, bool inherit_mask)
{
    const int32_t n = (int32_t)sidx.size();
    std::vector<int64_t> off((size_t)n + 1, 0);
    int32_t max_len = 0;
    for (int32_t i = 0; i < n; i++) {
        off[(size_t)i + 1] = off[(size_t)i] + slen[(size_t)i];
        max_len = std::max(max_len, slen[(size_t)i]);
    }
    uint8_t *d_alloc = nullptr, *d_bases = nullptr;
    if (int rc = dh_alloc_bases(ctx->stream, off.back(), &d_alloc, &d_bases)) return rc;
    if (int rc = dh_db_adopt(ctx, d_alloc, d_bases, off, group, out)) {
        dh_dev_free(d_alloc);
        return rc;
    }
    if (n > 0) {
        DevBuf<int32_t> d_sidx, d_sbeg;
        HIPCHK(d_sidx.alloc((size_t)n));
        HIPCHK(d_sbeg.alloc((size_t)n));
        HIPCHK(hipMemcpyAsync(d_sidx.p, sidx.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice,
                              ctx->stream));
        HIPCHK(hipMemcpyAsync(d_sbeg.p, sbeg.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice,
                              ctx->stream));
        dhk_gather_slices(ctx->stream, src->d_bases, src->d_off, d_sidx.p, d_sbeg.p, (*out)->d_off, n,
                          max_len, d_bases);
        if (inherit_mask && src->d_mask_bits) {  // slices keep the soft mask of their source (the flank DB's -mrep)
            uint8_t *layer;
            if (int rc = dh_ensure_mask_layer(*out, 0, &layer)) return rc;
            dhk_mask_slices(ctx->stream, (const uint32_t *)src->d_mask_bits, src->d_off, d_sidx.p, d_sbeg.p, (*out)->d_off, n,
                            max_len, (uint32_t *)layer);
            if (int rc = dh_mask_recompose(*out)) return rc;
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(ctx->stream));
    }
    return DH_OK;
}

// ------------------------------------------------------------------------------------ collect

struct dh_pileups {
    std::vector<int32_t> contig_left;
    std::vector<std::vector<int32_t>> triples;  // read, left LA, right LA
    // general joins (dh_pileups_create_joins): (contig0, seed0, contig1, seed1) per pile-up, contig1 = -1 for an
    // extension pile-up; empty = every pile-up is the gap (contig_left, BACK) -> (contig_left + 1, FRONT)
    std::vector<std::array<int32_t, 4>> join;
    std::array<int32_t, 4> join_of(size_t i) const
    {
        return join.empty() ? std::array<int32_t, 4>{contig_left[i], DH_SEED_BACK, contig_left[i] + 1, DH_SEED_FRONT} : join[i];
    }
};
static int refuse_general(const dh_pileups *p, const char *who)
{
    return p && !p->join.empty() ? dh_fail(DH_EINVAL, std::string(who) + ": pile-ups of general joins are not handled here") : DH_OK;
}

// Candidates: for every read and every gap the read spans, ONE (read, left LA, right LA) entry --
// the qualifying pair with the longest anchors (ties: lowest LA indices) -- grouped by gap, ordered
// by read id.  No min/max-reads cut yet (the sharded path applies it after the exchange).
static int collect_candidates(const dh_la *las, int64_t n, const int64_t *contig_off, int32_t ncontigs,
                              const dh_process_opts &o, dh_pileups **out)
{
    if (n >= (1ll << 31)) return dh_fail(DH_EINVAL, "dh_collect_spanning: more than 2^31 - 1 local alignments");
    // the enabled LAs (dh_collect_filter leaves most of a mapping disabled) as (read, LA index), listed
    // by the host threads over runs of the input and grouped by read with a counting sort that keeps
    // the LA order inside a read
    const int64_t lgrain = 1 << 16, lchunks = (n + lgrain - 1) / lgrain;
    std::vector<std::vector<std::pair<int32_t, int32_t>>> live((size_t)std::max<int64_t>(lchunks, 1));
    std::atomic<int> bad{0};
    dh_parallel_for(lchunks, 1, [&](int64_t clo, int64_t chi) {
        for (int64_t c = clo; c < chi; c++) {
            auto &v = live[(size_t)c];
            const int64_t i1 = std::min(n, (c + 1) * lgrain);
            for (int64_t i = c * lgrain; i < i1; i++) {
                if (las[i].bread < 0 || las[i].aread < 0 || las[i].aread >= ncontigs) bad = 1;
                else if (!(las[i].flags & DH_FLAG_DISABLED)) v.emplace_back(las[i].bread, (int32_t)i);
            }
        }
    });
    if (bad) return dh_fail(DH_EINVAL, "dh_collect_spanning: read or contig id out of range");
    int32_t nreads = 0;
    int64_t nlive = 0;
    for (const auto &v : live) {
        nlive += (int64_t)v.size();
        for (const auto &e : v) nreads = std::max(nreads, e.first + 1);
    }
    std::vector<int64_t> first((size_t)nreads + 1, 0), order((size_t)nlive);
    for (const auto &v : live)
        for (const auto &e : v) first[(size_t)e.first + 1]++;
    for (int32_t r = 0; r < nreads; r++) first[(size_t)r + 1] += first[(size_t)r];
    {
        std::vector<int64_t> cur(first.begin(), first.end() - 1);
        for (const auto &v : live)
            for (const auto &e : v) order[(size_t)cur[(size_t)e.first]++] = e.second;
    }
    live.clear();
    // reads are independent: host threads take runs of reads and list their entries (gap, read, iL,
    // iR) in read order; the runs are concatenated in order and split by gap afterwards
    struct Ent {
        int32_t gap, rd, iL, iR;
    };
    const int64_t grain = 16384, nchunks = ((int64_t)nreads + grain - 1) / grain;
    std::vector<std::vector<Ent>> found((size_t)std::max<int64_t>(nchunks, 1));
    dh_parallel_for(nchunks, 1, [&](int64_t clo, int64_t chi) {
        for (int64_t c = clo; c < chi; c++) {
            std::vector<Ent> &out_c = found[(size_t)c];
            const int32_t r1 = (int32_t)std::min<int64_t>(nreads, (c + 1) * grain);
            std::vector<std::pair<int32_t, std::pair<int64_t, std::pair<int64_t, int64_t>>>> best;  // gap -> (anchors, (iL, iR))
            for (int32_t rd = (int32_t)(c * grain); rd < r1; rd++) {
                const int64_t *idx = order.data() + first[(size_t)rd], cnt = first[(size_t)rd + 1] - first[(size_t)rd];
                if (cnt < 2) continue;
                best.clear();
                for (int64_t x = 0; x < cnt; x++) {
                    const int64_t iL = idx[x];
                    const dh_la &L = las[iL];
                    if (L.flags & DH_FLAG_DISABLED) continue;  // dropped by dh_collect_filter
                    if (L.aread + 1 >= ncontigs) continue;
                    const int64_t cl = contig_off[L.aread + 1] - contig_off[L.aread];
                    if (L.aepos + o.allowance < cl || L.aepos - L.abpos < o.min_anchor) continue;
                    for (int64_t y = 0; y < cnt; y++) {
                        const int64_t iR = idx[y];
                        const dh_la &R = las[iR];
                        if (R.flags & DH_FLAG_DISABLED) continue;
                        if (R.aread != L.aread + 1 || (R.flags & DH_FLAG_COMP) != (L.flags & DH_FLAG_COMP)) continue;
                        if (R.abpos > o.allowance || R.aepos - R.abpos < o.min_anchor) continue;
                        if (R.bbpos + o.allowance < L.bepos - o.allowance) continue;
                        const int64_t anchors = (int64_t)(L.aepos - L.abpos) + (R.aepos - R.abpos);
                        size_t k = 0;
                        while (k < best.size() && best[k].first != L.aread) k++;
                        if (k == best.size()) best.push_back(std::make_pair(L.aread, std::make_pair((int64_t)-1, std::make_pair(iL, iR))));
                        if (anchors > best[k].second.first) best[k].second = std::make_pair(anchors, std::make_pair(iL, iR));
                    }
                }
                for (auto &b : best)
                    out_c.push_back(Ent{b.first, rd, (int32_t)b.second.second.first, (int32_t)b.second.second.second});
            }
        }
    });
    std::map<int32_t, std::vector<int32_t>> piles;
    for (const std::vector<Ent> &v : found)
        for (const Ent &e : v) {
            std::vector<int32_t> &t = piles[e.gap];
            t.push_back(e.rd);
            t.push_back(e.iL);
            t.push_back(e.iR);
        }
    dh_pileups *p = new dh_pileups();
    for (auto &kv : piles) {
        p->contig_left.push_back(kv.first);
        p->triples.push_back(std::move(kv.second));
    }
    *out = p;
    return DH_OK;
}

// min-reads / max-reads cut of one candidate list (ordered by read id): fewer than min_reads
// distinct reads -> dropped; more than max_reads -> the max_reads entries with the lowest error
// rate of their two anchoring LAs stay (ties: lower read id), still ordered by read id.
static bool select_pile(std::vector<int32_t> &v, const dh_la *las, const dh_process_opts &o)
{
    const int32_t cnt = (int32_t)v.size() / 3;
    if (cnt < o.min_reads) return false;
    if (o.max_reads <= 0 || cnt <= o.max_reads) return true;  // max_reads 0 = no cap (the reference has none)
    // key = (class, error rate, entry); class = 2 x (extension entry) + (a further entry of its read).  A spanning read that opens with an extension enters the
    // pile-up as TWO extension entries (pileups.d:870) cropped from the same bases -- read[cropL, end) and read[0, cropR)
    // overlap in the gap -- so both of them in the vote count that read's errors twice.  With more entries than the cap
    // there are enough distinct reads: a read's second entry is considered only after every read's best one
    // (configs[2]: consensus error 0.091 % -> the spanning collector's level with the same 60 entries).
    std::vector<std::array<int64_t, 3>> key((size_t)cnt);
    for (int32_t e = 0; e < cnt; e++) {
        // an extension entry (one index is -1) is judged by the one alignment it has
        const int32_t iL = v[(size_t)e * 3 + 1], iR = v[(size_t)e * 3 + 2];
        int64_t len = 0, diffs = 0;
        if (iL >= 0) {
            len += las[iL].aepos - las[iL].abpos;
            diffs += las[iL].diffs;
        }
        if (iR >= 0) {
            len += las[iR].aepos - las[iR].abpos;
            diffs += las[iR].diffs;
        }
        key[(size_t)e] = {0, diffs * 1000000 / std::max<int64_t>(len, 1), e};
    }
    for (int32_t e = 0; e < cnt;) {  // entries of one read are adjacent: all but its best one rank behind
        int32_t f = e + 1, best = e;
        while (f < cnt && v[(size_t)f * 3] == v[(size_t)e * 3]) f++;
        for (int32_t x = e + 1; x < f; x++)
            if (key[(size_t)x][1] < key[(size_t)best][1]) best = x;
        for (int32_t x = e; x < f; x++) key[(size_t)x][0] = x == best ? 0 : 1;
        // ... and an extension entry (it covers the gap as far as its read goes) only after the reads that span the gap
        for (int32_t x = e; x < f; x++)
            if (v[(size_t)x * 3 + 1] < 0 || v[(size_t)x * 3 + 2] < 0) key[(size_t)x][0] += 2;
        e = f;
    }
    std::sort(key.begin(), key.end());  // entries are in read-id order, so e breaks ties by read id
    std::vector<int32_t> keep((size_t)o.max_reads);
    for (int32_t x = 0; x < o.max_reads; x++) keep[(size_t)x] = (int32_t)key[(size_t)x][2];
    std::sort(keep.begin(), keep.end());
    std::vector<int32_t> w;
    w.reserve((size_t)o.max_reads * 3);
    for (int32_t e : keep) w.insert(w.end(), v.begin() + (size_t)e * 3, v.begin() + (size_t)e * 3 + 3);
    v.swap(w);
    return true;
}

extern "C" int dh_collect_candidates(const dh_la *las, int64_t n, const int64_t *contig_off, int32_t ncontigs,
                                     const dh_process_opts *opts, dh_pileups **out)
{
    if ((n > 0 && !las) || !contig_off || !opts || !out || ncontigs < 0)
        return dh_fail(DH_EINVAL, "dh_collect_candidates: bad argument");
    return collect_candidates(las, n, contig_off, ncontigs, *opts, out);
}

extern "C" int dh_pileups_select(const dh_pileups *cands, const dh_la *las, int64_t n,
                                 const dh_process_opts *opts, dh_pileups **out)
{
    if (!cands || !opts || !out || (n > 0 && !las)) return dh_fail(DH_EINVAL, "dh_pileups_select: bad argument");
    // pile-ups are independent (the cut reads the anchoring LAs: cache misses into the mapping's records)
    const size_t np = cands->contig_left.size();
    std::vector<std::vector<int32_t>> sel(np);
    std::vector<char> keep(np, 0);
    std::atomic<int> bad{0};
    dh_parallel_for((int64_t)np, 4, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; i++) {
            std::vector<int32_t> v = cands->triples[(size_t)i];
            bool ok = true;
            for (size_t e = 0; e < v.size() && ok; e += 3)  // -1 = no alignment on that side (extension entry)
                ok = v[e + 1] >= -1 && v[e + 1] < n && v[e + 2] >= -1 && v[e + 2] < n && (v[e + 1] >= 0 || v[e + 2] >= 0);
            if (!ok) {
                bad = 1;
                continue;
            }
            if (!select_pile(v, las, *opts)) continue;
            keep[(size_t)i] = 1;
            sel[(size_t)i] = std::move(v);
        }
    });
    if (bad) return dh_fail(DH_EINVAL, "dh_pileups_select: LA index out of range");
    dh_pileups *p = new dh_pileups();
    for (size_t i = 0; i < np; i++)
        if (keep[i]) {
            p->contig_left.push_back(cands->contig_left[i]);
            if (!cands->join.empty()) p->join.push_back(cands->join[i]);
            p->triples.push_back(std::move(sel[i]));
        }
    *out = p;
    return DH_OK;
}

// internal helpers of dh_map_reads: LA indices shifted by a constant; pile-ups of several parts (ascending
// read ranges) concatenated gap by gap
void dh_pileups_shift(dh_pileups *p, int32_t by)
{
    for (auto &t : p->triples)
        for (size_t e = 0; e + 2 < t.size(); e += 3) {
            if (t[e + 1] >= 0) t[e + 1] += by;
            if (t[e + 2] >= 0) t[e + 2] += by;
        }
}
int dh_pileups_concat(dh_pileups *const *parts, int32_t nparts, dh_pileups **out)
{
    for (int32_t i = 0; i < nparts; i++)
        if (int rc = refuse_general(parts[i], "dh_pileups_concat")) return rc;
    std::map<int32_t, std::vector<int32_t>> m;
    for (int32_t i = 0; i < nparts; i++) {
        if (!parts[i]) continue;
        for (size_t g = 0; g < parts[i]->contig_left.size(); g++) {
            std::vector<int32_t> &t = m[parts[i]->contig_left[g]];
            t.insert(t.end(), parts[i]->triples[g].begin(), parts[i]->triples[g].end());
        }
    }
    dh_pileups *p = new dh_pileups();
    for (auto &kv : m) {
        p->contig_left.push_back(kv.first);
        p->triples.push_back(std::move(kv.second));
    }
    *out = p;
    return DH_OK;
}

extern "C" int dh_pileups_create(const int32_t *contig_left, const int32_t *count, int32_t npiles,
                                 const int32_t *triples, dh_pileups **out)
{
    if (npiles < 0 || !out || (npiles > 0 && (!contig_left || !count || !triples)))
        return dh_fail(DH_EINVAL, "dh_pileups_create: bad argument");
    dh_pileups *p = new dh_pileups();
    int64_t at = 0;
    for (int32_t i = 0; i < npiles; i++) {
        if (count[i] < 0 || contig_left[i] < 0 || (i > 0 && contig_left[i] <= contig_left[i - 1])) {
            delete p;
            return dh_fail(DH_EINVAL, "dh_pileups_create: pile-ups must be ordered by contig and counts >= 0");
        }
        p->contig_left.push_back(contig_left[i]);
        p->triples.emplace_back(triples + at * 3, triples + (at + count[i]) * 3);
        at += count[i];
    }
    *out = p;
    return DH_OK;
}

extern "C" int dh_pileups_create_joins(const int32_t *nodes4, const int32_t *count, int32_t npiles, const int32_t *triples,
                                       dh_pileups **out)
{
    if (npiles < 0 || !out || (npiles > 0 && (!nodes4 || !count || !triples)))
        return dh_fail(DH_EINVAL, "dh_pileups_create_joins: bad argument");
    dh_pileups *p = new dh_pileups();
    int64_t at = 0;
    for (int32_t i = 0; i < npiles; i++) {
        const int32_t *q = nodes4 + 4 * (size_t)i;
        const bool ext = q[2] < 0;
        // node order of the scaffold graph: (contig, part) with begin < end, i.e. seed front < seed back
        auto key = [](const int32_t *x) { return std::array<int64_t, 4>{x[0], x[1], x[2] < 0 ? INT32_MAX : x[2], x[3]}; };
        if (count[i] < 0 || q[0] < 0 || (q[1] != DH_SEED_FRONT && q[1] != DH_SEED_BACK) || (!ext && (q[3] != DH_SEED_FRONT && q[3] != DH_SEED_BACK)) ||
            (!ext && q[2] <= q[0]) || (i > 0 && !(key(q - 4) < key(q)))) {
            delete p;
            return dh_fail(DH_EINVAL, "dh_pileups_create_joins: joins must be ordered by their nodes, contig0 < contig1, seeds 0 / 1, counts >= 0");
        }
        p->contig_left.push_back(q[0]);
        p->join.push_back({q[0], q[1], ext ? -1 : q[2], ext ? 0 : q[3]});
        p->triples.emplace_back(triples + at * 3, triples + (at + count[i]) * 3);
        at += count[i];
    }
    *out = p;
    return DH_OK;
}

extern "C" int dh_pileups_get_join(const dh_pileups *p, int32_t i, int32_t *nodes4)
{
    if (!p || !nodes4 || i < 0 || i >= (int32_t)p->contig_left.size()) return dh_fail(DH_EINVAL, "dh_pileups_get_join: bad argument");
    const std::array<int32_t, 4> j = p->join_of((size_t)i);
    memcpy(nodes4, j.data(), sizeof(int32_t) * 4);
    return DH_OK;
}

extern "C" int dh_collect_spanning(const dh_la *las, int64_t n, const int64_t *contig_off,
                                   int32_t ncontigs, const dh_process_opts *opts, dh_pileups **out)
{
    if ((n > 0 && !las) || !contig_off || !opts || !out || ncontigs < 0)
        return dh_fail(DH_EINVAL, "dh_collect_spanning: bad argument");
    dh_pileups *c = nullptr;
    if (int rc = collect_candidates(las, n, contig_off, ncontigs, *opts, &c)) return rc;
    const int rc = dh_pileups_select(c, las, n, opts, out);
    delete c;
    return rc;
}

// pile-ups.db of a collect result (what `dentist collect` hands to `dentist process`,
// collectPileUps/package.d:88-96 writePileUpsDb): every read of a pile-up is a ReadAlignment of two
// SeededAlignments -- its chain on the left contig seeded at the back, its chain on the right contig
// seeded at the front (pileups.d:821-888); chains hold one local alignment with its trace points.
extern "C" int dh_pileups_write_db(const dh_pileups *p, const dh_la *las, int64_t n, const uint16_t *trace,
                                   const int64_t *contig_off, int32_t ncontigs, const int64_t *read_off, int32_t nreads,
                                   int32_t tspace, const char *path)
{
    if (!p || !contig_off || !read_off || !path || (n > 0 && (!las || !trace)))
        return dh_fail(DH_EINVAL, "dh_pileups_write_db: bad argument");
    std::vector<int32_t> nra, nsa;
    std::vector<dh_seeded> sa;
    std::vector<dh_chain_la> la;
    std::vector<uint16_t> tp;
    for (size_t i = 0; i < p->contig_left.size(); i++) {
        const std::vector<int32_t> &t = p->triples[i];
        const std::array<int32_t, 4> jn = p->join_of(i);
        nra.push_back((int32_t)t.size() / 3);
        for (size_t e = 0; e + 2 < t.size(); e += 3) {
            nsa.push_back((t[e + 1] >= 0 ? 1 : 0) + (t[e + 2] >= 0 ? 1 : 0));
            for (int side = 0; side < 2; side++) {
                const int32_t li = t[e + 1 + (size_t)side];
                if (li == -1 && t[e + 2 - (size_t)side] >= 0) continue;  // extension entry: one seeded alignment
                if (li < 0 || li >= n) return dh_fail(DH_EINVAL, "dh_pileups_write_db: LA index out of range");
                const dh_la &x = las[li];
                if (x.aread < 0 || x.aread >= ncontigs || x.bread < 0 || x.bread >= nreads)
                    return dh_fail(DH_EINVAL, "dh_pileups_write_db: id out of range");
                dh_seeded s;
                memset(&s, 0, sizeof(s));
                s.id = li;
                s.contig_a_id = (uint32_t)(x.aread + 1);
                s.contig_a_len = (uint32_t)(contig_off[x.aread + 1] - contig_off[x.aread]);
                s.contig_b_id = (uint32_t)(x.bread + 1);
                s.contig_b_len = (uint32_t)(read_off[x.bread + 1] - read_off[x.bread]);
                s.flags = (x.flags & DH_FLAG_COMP) ? 1 : 0;
                s.seed = (uint8_t)jn[1 + 2 * (size_t)side];  // AlignmentLocationSeed of the flank (plain gap: back, front)
                s.tspace = (uint16_t)tspace;
                s.nla = 1;
                sa.push_back(s);
                la.push_back(dh_chain_la{(uint32_t)x.abpos, (uint32_t)x.aepos, (uint32_t)x.bbpos, (uint32_t)x.bepos,
                                         (uint32_t)x.diffs, x.tlen / 2});
                tp.insert(tp.end(), trace + x.toff, trace + x.toff + x.tlen);
            }
        }
    }
    return dh_pileupdb_write(path, (int32_t)nra.size(), nra.data(), nsa.data(), sa.data(), la.data(), tp.data());
}

// all pile-ups at once: contig_left[npiles], count[npiles], triples[3 * total]; arrays may be NULL to
// size; returns the total number of triples
extern "C" int64_t dh_pileups_flat(const dh_pileups *p, int32_t *contig_left, int32_t *count, int32_t *triples)
{
    if (!p) return 0;
    int64_t at = 0;
    for (size_t i = 0; i < p->contig_left.size(); i++) {
        const std::vector<int32_t> &t = p->triples[i];
        if (contig_left) contig_left[i] = p->contig_left[i];
        if (count) count[i] = (int32_t)t.size() / 3;
        if (triples && !t.empty()) memcpy(triples + 3 * at, t.data(), sizeof(int32_t) * t.size());
        at += (int64_t)t.size() / 3;
    }
    return at;
}

extern "C" void dh_pileups_destroy(dh_pileups *p) { delete p; }
extern "C" int32_t dh_pileups_count(const dh_pileups *p) { return p ? (int32_t)p->contig_left.size() : 0; }
extern "C" int32_t dh_pileups_get(const dh_pileups *p, int32_t i, int32_t *contig_left,
                                  const int32_t **triples)
{
    if (!p || i < 0 || i >= (int32_t)p->contig_left.size()) return -1;
    if (contig_left) *contig_left = p->contig_left[(size_t)i];
    if (triples) *triples = p->triples[(size_t)i].data();
    return (int32_t)p->triples[(size_t)i].size() / 3;
}

// ------------------------------------------------------------------------------------ trace maths

static int32_t ceil_to(int32_t x, int32_t m) { return (x + m - 1) / m * m; }

// Alignment chains (base.d:306-421) in the cropper: an entry names the FIRST record of its chain, the members follow it
// (dh_continues_chain).  to!(ReferenceRegion, "contigA") of a chain = the union of its members' A intervals
// (common/package.d:228-241); the common alignment region of a flank = the intersection of the entries' regions.
typedef std::vector<std::pair<int32_t, int32_t>> Region;
static int64_t chain_end(const dh_la *las, int64_t n, int64_t i)
{
    int64_t j = i + 1;
    while (j < n && dh_continues_chain(las[j - 1], las[j])) j++;
    return j;
}
static void intersect_chain(Region &reg, const dh_la *las, int64_t n, int64_t i)
{
    const int64_t j = chain_end(las, n, i);
    Region mine;
    for (int64_t x = i; x < j; x++) mine.emplace_back(las[x].abpos, las[x].aepos);
    if (j - i > 1) {
        std::sort(mine.begin(), mine.end());
        Region m2;
        for (const auto &iv : mine)
            if (!m2.empty() && iv.first <= m2.back().second)
                m2.back().second = std::max(m2.back().second, iv.second);
            else
                m2.push_back(iv);
        mine.swap(m2);
    }
    Region out;
    for (const auto &a : reg)
        for (const auto &b : mine) {
            const int32_t lo = std::max(a.first, b.first), hi = std::min(a.second, b.second);
            if (lo < hi) out.emplace_back(lo, hi);
        }
    reg.swap(out);
}
// the first member of the chain at record i that covers apos (AlignmentChain.translateTracePoint, base.d:866-880)
static int64_t covering_member(const dh_la *las, int64_t n, int64_t i, int32_t apos)
{
    const int64_t j = chain_end(las, n, i);
    for (int64_t x = i; x < j; x++)
        if (las[x].abpos <= apos && apos <= las[x].aepos) return x;
    return -1;
}

// getCommonTracePoint, cropper.d:446-500: candidates are the trace points of the region (plus the contig end),
// innermost first for `front` seeds; the common A region minus the repeat mask is tried first, then the region itself.
static int32_t common_trace_point_in(const Region &reg, int32_t contig_len, int32_t ts, bool seed_front)
{
    if (reg.empty()) return -1;
    const int32_t lo = reg.front().first, hi = reg.back().second;
    const int32_t tp_min = ceil_to(lo, ts), tp_sup = ceil_to(hi, ts);
    std::vector<int32_t> cands;
    for (int32_t c = tp_min; c < tp_sup; c += ts) cands.push_back(c);
    if (tp_sup > contig_len) cands.push_back(contig_len);
    if (seed_front) std::reverse(cands.begin(), cands.end());
    for (int32_t c : cands) {
        bool in = c == hi;
        for (size_t x = 0; x < reg.size() && !in; x++) in = reg[x].first <= c && c < reg[x].second;
        if (in) return c;
    }
    return -1;
}
// mask: sorted disjoint (begin, end) pairs of this contig, nmask of them (may be 0 / NULL)
static int32_t common_trace_point(const Region &reg, int32_t contig_len, int32_t ts, bool seed_front,
                                  const int32_t *mask = nullptr, int64_t nmask = 0)
{
    if (nmask > 0 && !reg.empty()) {
        Region un;  // reg - mask
        for (const auto &iv : reg) {
            int32_t b = iv.first;
            for (int64_t m = 0; m < nmask && b < iv.second; m++) {
                const int32_t mb = mask[2 * m], me = mask[2 * m + 1];
                if (me <= b) continue;
                if (mb >= iv.second) break;
                if (mb > b) un.emplace_back(b, mb);
                b = std::max(b, me);
            }
            if (b < iv.second) un.emplace_back(b, iv.second);
        }
        const int32_t c = common_trace_point_in(un, contig_len, ts, seed_front);
        if (c >= 0) return c;
    }
    return common_trace_point_in(reg, contig_len, ts, seed_front);
}

// the cropper's common trace point as an entry of its own: first[] names the first record of each alignment chain of
// one flank (all on the same contig, all with the same seed)
extern "C" int dh_common_trace_point(const dh_la *las, int64_t n, const int32_t *first, int32_t count, int32_t contig_len,
                                     int32_t tspace, int32_t seed_front, const int32_t *mask_iv, int64_t nmask, int32_t *out)
{
    if (!out || count < 0 || (count > 0 && (!las || !first)) || tspace < 1 || nmask < 0 || (nmask > 0 && !mask_iv))
        return dh_fail(DH_EINVAL, "dh_common_trace_point: bad argument");
    Region reg{{0, INT32_MAX}};
    for (int32_t x = 0; x < count; x++) {
        if (first[x] < 0 || first[x] >= n) return dh_fail(DH_EINVAL, "dh_common_trace_point: record index out of range");
        intersect_chain(reg, las, n, first[x]);
    }
    *out = count > 0 ? common_trace_point(reg, contig_len, tspace, seed_front != 0, mask_iv, nmask) : -1;
    return DH_OK;
}

static int32_t trace_points_up_to_a(const dh_la &la, int32_t ts, int32_t apos, int32_t mode)
{
    const int32_t ntp = la.tlen / 2;
    const int32_t second = la.abpos / ts * ts + ts;
    if (mode == 0) {
        if (apos < second) return 0;
        if (apos < la.aepos) return 1 + (apos - second) / ts;
        return ntp;
    }
    const int32_t second_from_last = (la.aepos - 1) / ts * ts;
    if (apos == la.abpos) return 0;
    if (apos <= second) return 1;
    if (apos <= second_from_last) return 1 + (apos - second + ts - 1) / ts;
    return ntp;
}

// Trace.translateTracePoint!"contigA"(pos, mode), base.d:185-203: the position is assigned to a trace
// point of the LA; returns its coordinates on A and on B
static void translate_trace_point(const dh_la &la, const uint16_t *tr, int32_t ts, int32_t apos, int32_t mode,
                                  int32_t *outa, int32_t *outb)
{
    const int32_t ntp = la.tlen / 2;
    const int32_t idx = trace_points_up_to_a(la, ts, apos, mode);
    int32_t b = la.bbpos;
    for (int32_t i = 0; i < idx; i++) b += tr[2 * i + 1];
    *outb = b;
    *outa = idx == 0 ? la.abpos : (idx < ntp ? la.abpos / ts * ts + idx * ts : la.aepos);
}

static int32_t translate_floor_b(const dh_la &la, const uint16_t *tr, int32_t ts, int32_t apos)
{
    int32_t a, b;
    translate_trace_point(la, tr, ts, apos, 0, &a, &b);
    return b;
}

// the same through the C ABI (the cropper of `dentist process` is built on it: cropper.d:503-550)
extern "C" int dh_translate_trace_point(const dh_la *la, const uint16_t *trace, int32_t tspace, int32_t apos,
                                        int32_t mode, int32_t *out_a, int32_t *out_b)
{
    if (!la || !trace || !out_a || !out_b || tspace < 1 || (mode != 0 && mode != 1) || la->tlen < 0 || la->tlen % 2)
        return dh_fail(DH_EINVAL, "dh_translate_trace_point: bad argument");
    if (apos < la->abpos || apos > la->aepos)  // the reference asserts contigA.begin <= pos <= contigA.end
        return dh_fail(DH_EINVAL, "dh_translate_trace_point: position outside the local alignment");
    if (la->tlen / 2 != (la->aepos + tspace - 1) / tspace - la->abpos / tspace)
        return dh_fail(DH_EINVAL, "dh_translate_trace_point: trace length does not fit the A interval");
    translate_trace_point(*la, trace + la->toff, tspace, apos, mode, out_a, out_b);
    return DH_OK;
}

// isValidPileUpAlignment (flat), dazzler.d:4126-4141
static bool valid_pileup_alignment(const dh_la &la, bool same, int32_t alen, int32_t blen, int32_t allow)
{
    const bool ab = la.abpos <= allow, bb = la.bbpos <= allow;
    const bool ae = la.aepos + allow >= alen, be = la.bepos + allow >= blen;
    return !same && (((ab && bb) && (ae || be)) || ((ae && be) && (ab || bb)));
}

// chainLocalAlignments / buildAlignmentChains (common/alignments/chaining.d:122-334) with the
// defaults of commandline.d:1819, 1982, 2014, 2165-2173 and minRelativeScore = min_rel (--min-relative-score, :2141-2153).
// `la` is grouped by (aread, bread) [first, last):
//  * the pair's enabled LAs are split into the connected components of the undirected chainability relation (:182);
//  * a shortest-path problem rates the chains (:227-233; relaxations over the LAs ordered by (abpos, bbpos, index), a
//    topological order -- no edge joins two components, so one pass serves all of them);
//  * per component the end nodes within effectiveMinScore of the component's best chain are taken best first (:236-266):
//    a node already on a taken chain is no end node, a chain that runs into nodes of a better chain is an ALTERNATE chain
//    and is composed of its whole path (:269-285) -- the LAs it shares are written once per chain: their further
//    occurrences go to `dups` (record index, flags) and are inserted behind the first one by the caller;
//  * the chains scoring >= max(minScore, minRelativeScore * best of the pair) are accepted (:305-312).
// First LA of a chain: START (+ BEST unless alternate, dazzler.d:2063-2068), the others NEXT; every other enabled LA of the
// pair gets DISABLED.  Ties: the lower position in the (abpos, bbpos, index) order first (oracle/pile.c:chain_pair).
struct ChainDup {
    size_t i;
    uint32_t flags;
};
static void chain_pair(LaVec &la, size_t first, size_t last, int32_t min_score, double min_rel_score, std::vector<ChainDup> &dups)
{
    const int32_t max_indel = 1000, max_gap = 10000;
    const double max_rel_overlap = 0.3;
    const uint32_t cmask = DH_FLAG_START | DH_FLAG_NEXT | DH_FLAG_BEST;
    // fast path (the common case): a single enabled LA is its own best chain
    size_t nen = 0, only = first;
    for (size_t i = first; i < last; i++)
        if (!(la[i].flags & DH_FLAG_DISABLED)) {
            nen++;
            only = i;
        }
    if (nen == 0) return;
    if (nen == 1) {
        dh_la &l = la[only];
        const int32_t sc = ((l.aepos - l.abpos) + (l.bepos - l.bbpos)) / 2;
        if (sc < (int32_t)std::max<double>(min_score, min_rel_score * sc))
            l.flags |= DH_FLAG_DISABLED;
        else
            l.flags = (l.flags & ~cmask) | DH_FLAG_START | DH_FLAG_BEST;
        return;
    }
    std::vector<size_t> order;
    for (size_t i = first; i < last; i++)
        if (!(la[i].flags & DH_FLAG_DISABLED)) order.push_back(i);
    const size_t n = order.size();
    std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) {
        if (la[x].abpos != la[y].abpos) return la[x].abpos < la[y].abpos;
        if (la[x].bbpos != la[y].bbpos) return la[x].bbpos < la[y].bbpos;
        return x < y;
    });
    auto score = [&](const dh_la &x) { return ((x.aepos - x.abpos) + (x.bepos - x.bbpos)) / 2; };
    auto chainable = [&](const dh_la &x, const dh_la &y) {
        if ((x.flags & DH_FLAG_COMP) != (y.flags & DH_FLAG_COMP)) return false;
        const int32_t ga = y.abpos - x.aepos, gb = y.bbpos - x.bepos;
        if (!(x.abpos < y.abpos && x.bbpos < y.bbpos)) return false;
        if (std::abs(ga - gb) > max_indel || std::max(std::abs(ga), std::abs(gb)) > max_gap) return false;
        const int32_t mla = std::min(x.aepos - x.abpos, y.aepos - y.abpos);
        const int32_t mlb = std::min(x.bepos - x.bbpos, y.bepos - y.bbpos);
        return std::max(0, -ga) <= max_rel_overlap * mla && std::max(0, -gb) <= max_rel_overlap * mlb;
    };
    auto chain_score = [&](const dh_la &x, const dh_la &y) {
        const int32_t ga = y.abpos - x.aepos, gb = y.bbpos - x.bepos;
        return std::abs(ga - gb) + std::max(std::abs(ga), std::abs(gb)) / 10 - score(y);
    };
    std::vector<int32_t> dist(n), pred(n, -1), comp(n);
    for (size_t v = 0; v < n; v++) {
        dist[v] = -score(la[order[v]]);
        comp[v] = (int32_t)v;
    }
    for (size_t u = 0; u < n; u++)
        for (size_t v = u + 1; v < n; v++)
            if (chainable(la[order[u]], la[order[v]])) {
                const int32_t d = dist[u] + chain_score(la[order[u]], la[order[v]]);
                if (dist[v] > d) {
                    dist[v] = d;
                    pred[v] = (int32_t)u;
                }
                const int32_t cu = comp[u], cv = comp[v];
                if (cu != cv)
                    for (size_t w = 0; w < n; w++)
                        if (comp[w] == cv) comp[w] = cu;
            }
    // components in the order of their smallest record index (util/graphalgo.d:43-66)
    std::vector<size_t> cmin(n, SIZE_MAX), cord;
    for (size_t v = 0; v < n; v++) cmin[(size_t)comp[v]] = std::min(cmin[(size_t)comp[v]], order[v]);
    for (size_t v = 0; v < n; v++)
        if (cmin[v] != SIZE_MAX) cord.push_back(v);
    std::sort(cord.begin(), cord.end(), [&](size_t x, size_t y) { return cmin[x] < cmin[y]; });
    struct Sel {
        size_t end;
        bool alt;
        int32_t score;
    };
    std::vector<Sel> sel;
    std::vector<uint8_t> forbidden(n, 0);
    std::vector<size_t> ends;
    for (size_t c : cord) {
        ends.clear();
        for (size_t v = 0; v < n; v++)
            if ((size_t)comp[v] == c) ends.push_back(v);
        std::stable_sort(ends.begin(), ends.end(), [&](size_t x, size_t y) { return dist[x] < dist[y]; });
        const int32_t cbest = -dist[ends[0]];
        const int32_t cthr = (int32_t)std::max<double>(min_score, min_rel_score * cbest);
        for (size_t e : ends) {
            if (forbidden[e] || -dist[e] < cthr) continue;
            bool alt = false;
            for (int32_t v = (int32_t)e; v >= 0; v = pred[(size_t)v]) {
                alt = alt || forbidden[(size_t)v];
                forbidden[(size_t)v] = 1;
            }
            sel.push_back({e, alt, -dist[e]});
        }
    }
    int32_t best = 0;
    for (size_t x = 0; x < sel.size(); x++)
        if (x == 0 || sel[x].score > best) best = sel[x].score;
    const int32_t thr = (int32_t)std::max<double>(min_score, min_rel_score * best);
    std::vector<uint8_t> occ(n, 0);
    std::vector<size_t> path;
    for (const Sel &c : sel) {
        if (c.score < thr) continue;
        path.clear();
        for (int32_t v = (int32_t)c.end; v >= 0; v = pred[(size_t)v]) path.push_back((size_t)v);
        std::reverse(path.begin(), path.end());
        for (size_t k = 0; k < path.size(); k++) {
            const size_t v = path[k];
            dh_la &l = la[order[v]];
            const uint32_t f = k == 0 ? (DH_FLAG_START | (c.alt ? 0u : DH_FLAG_BEST)) : DH_FLAG_NEXT;
            if (!occ[v]) {
                occ[v] = 1;
                l.flags = (l.flags & ~cmask) | f;
            } else
                dups.push_back({order[v], (l.flags & ~cmask) | f});
        }
    }
    for (size_t v = 0; v < n; v++)
        if (!occ[v]) la[order[v]].flags |= DH_FLAG_DISABLED;
}

// ------------------------------------------------------------------------------------ results

// (struct dh_insertions: dh_internal.h)

extern "C" void dh_insertions_destroy(dh_insertions *r) { delete r; }
extern "C" int32_t dh_insertions_count(const dh_insertions *r) { return r ? (int32_t)r->rec.size() : 0; }
extern "C" const dh_insertion *dh_insertions_records(const dh_insertions *r) { return r ? r->rec.data() : nullptr; }
extern "C" const uint8_t *dh_insertions_bases(const dh_insertions *r) { return r ? r->bases.data() : nullptr; }
extern "C" int64_t dh_insertions_bases_len(const dh_insertions *r) { return r ? (int64_t)r->bases.size() : 0; }
// read ids (0-based) of every record's pile-up: ids[off[i] .. off[i + 1]); off has count + 1 entries (all 0 when the
// result carries no ids)
extern "C" const int32_t *dh_insertions_read_ids(const dh_insertions *r) { return r ? r->ids.data() : nullptr; }
extern "C" const int32_t *dh_insertions_read_ids_off(const dh_insertions *r)
{
    return r && r->ids_off.size() == r->rec.size() + 1 ? r->ids_off.data() : nullptr;
}

// insertions.db of a result (what `dentist process` hands to `dentist output`,
// processPileUps/package.d:156-158, 789-805): one insertion per closed gap -- start = (left contig,
// end), end = (right contig, begin), the whole consensus as sequence, the two flank overlaps
// (contig = A, consensus = B, seeds back / front) and the sorted 1-based read ids of the pile-up.
extern "C" int dh_insertions_write_db(const dh_insertions *r, const int64_t *contig_off, int32_t ncontigs,
                                      int32_t tspace, const char *path)
{
    if (!r || !contig_off || !path || ncontigs < 0) return dh_fail(DH_EINVAL, "dh_insertions_write_db: bad argument");
    std::vector<dh_insertion_rec> ins;
    std::vector<uint8_t> bases;
    std::vector<uint32_t> ids;
    std::vector<dh_seeded> sa;
    std::vector<dh_chain_la> la;
    std::vector<uint16_t> tp;
    for (size_t i = 0; i < r->rec.size(); i++) {
        const dh_insertion &x = r->rec[i];
        if (x.status != DH_PILE_OK || r->flank_of[i] < 0) continue;
        const bool ext = (x.join & DH_JOIN_EXTENSION) != 0;
        const int32_t nf = ext ? 1 : 2;
        const int32_t fcontig[2] = {x.contig_left, ext ? x.contig_left : (x.join == 0 && x.contig_right == 0 ? x.contig_left + 1 : x.contig_right)};
        const bool front[2] = {(x.join & DH_JOIN_FLANK0_FRONT) != 0, (x.join & DH_JOIN_FLANK1_BACK) == 0};
        if (fcontig[0] < 0 || fcontig[0] >= ncontigs || fcontig[1] < 0 || fcontig[1] >= ncontigs)
            return dh_fail(DH_EINVAL, "dh_insertions_write_db: gap outside the contigs");
        dh_insertion_rec q;
        memset(&q, 0, sizeof(q));
        // makeJoin (base.d:2680-2722): a gap joins the seeded parts of its two contigs (begin = 1, end = 2); a front
        // extension is (contig, pre = 0) -> (contig, begin), a back extension (contig, end) -> (contig, post = 3)
        q.start_contig = fcontig[0] + 1;
        q.end_contig = fcontig[1] + 1;
        if (ext) {
            q.start_part = front[0] ? 0 : 2;
            q.end_part = front[0] ? 1 : 3;
        } else {
            q.start_part = front[0] ? 1 : 2;
            q.end_part = front[1] ? 1 : 2;
        }
        q.seq_len = x.cons_len;
        q.contig_len = 0;
        q.noverlaps = nf;
        q.nread_ids = r->ids_off[i + 1] - r->ids_off[i];
        ins.push_back(q);
        bases.insert(bases.end(), r->bases.begin() + x.cons_off, r->bases.begin() + x.cons_off + x.cons_len);
        std::vector<uint32_t> my(r->ids.begin() + r->ids_off[i], r->ids.begin() + r->ids_off[i + 1]);
        for (uint32_t &v : my) v += 1;
        std::sort(my.begin(), my.end());
        ids.insert(ids.end(), my.begin(), my.end());
        for (int side = 0; side < nf; side++) {
            const dh_la &f = r->flank[(size_t)r->flank_of[i] + (size_t)side];
            const int32_t c = fcontig[side];
            dh_seeded s;
            memset(&s, 0, sizeof(s));
            s.id = (int64_t)sa.size();
            s.contig_a_id = (uint32_t)(c + 1);
            s.contig_a_len = (uint32_t)(contig_off[c + 1] - contig_off[c]);
            s.contig_b_id = 1;
            s.contig_b_len = (uint32_t)x.cons_len;
            s.flags = (f.flags & DH_FLAG_COMP) ? 1 : 0;
            s.seed = front[side] ? 0 : 1;  // AlignmentLocationSeed: front = 0, back = 1 (plain gap: the back of the left contig, the front of the right one)
            s.tspace = (uint16_t)tspace;
            s.nla = 1;
            sa.push_back(s);
            la.push_back(dh_chain_la{(uint32_t)f.abpos, (uint32_t)f.aepos, (uint32_t)f.bbpos, (uint32_t)f.bepos, (uint32_t)f.diffs, f.tlen / 2});
            tp.insert(tp.end(), r->flank_tr.begin() + f.toff, r->flank_tr.begin() + f.toff + f.tlen);
        }
    }
    return dh_insertiondb_write(path, (int32_t)ins.size(), ins.data(), bases.data(), ids.data(), sa.data(), la.data(), tp.data());
}

struct ProcStats {
    float ms[7] = {0, 0, 0, 0, 0, 0, 0};
    int64_t counters[3] = {0, 0, 0};
    // the work of the call: [0] pile-ups processed, [1] their entries (cropped reads), [2] cropped bases,
    // [3] algorithmic bytes = sum over pile-ups of (n^2 + 2) L, n entries of mean cropped length L
    int64_t work[4] = {0, 0, 0, 0};
};
static thread_local ProcStats g_pstats;

extern "C" int dh_get_process_work(dh_ctx *ctx, int64_t *work4)
{
    if (!ctx || !work4) return dh_fail(DH_EINVAL, "dh_get_process_work: NULL argument");
    memcpy(work4, g_pstats.work, sizeof(g_pstats.work));
    return DH_OK;
}

extern "C" int dh_get_process_stats(dh_ctx *ctx, float *ms7, int64_t *counters3)
{
    if (!ctx) return dh_fail(DH_EINVAL, "ctx is NULL");
    if (ms7) memcpy(ms7, g_pstats.ms, sizeof(g_pstats.ms));
    if (counters3) memcpy(counters3, g_pstats.counters, sizeof(g_pstats.counters));
    return DH_OK;
}

// ------------------------------------------------------------------------------------ consensus round

// One voting + emission round.  T: templates (one per active pile-up), R: pile-up reads.
// las: overlaps with A = a template coordinate system; tmpl_of[i] = template of LA i or -1.
static int consensus_round(dh_ctx *ctx, dh_db *T, dh_db *R, const LaVec &las,
                           const TraceVec &trace, const std::vector<int32_t> &tmpl_of,
                           int32_t ts, dh_db **newT, int64_t *nseg_out, int64_t *ncell_out)
{
    hipStream_t st = ctx->stream;
    std::vector<SegDescH, PinnedAlloc<SegDescH>> segs;  // page-locked: uploaded every round
    int32_t wmax = 1, bandmax = 1;
    int64_t ncell = 0;
    size_t class_end[3] = {0, 0, 0};  // tiles of the overlaps of each band class end here (classes are contiguous)
    {
        // the selected overlaps and where their tiles go; host threads then fill the tiles
        std::vector<size_t> sel;
        std::vector<size_t> soff(1, 0);
        {
            // selection in input order: host threads scan runs of the LAs, the runs are concatenated
            const int64_t grain = 1 << 16, nch = ((int64_t)las.size() + grain - 1) / grain;
            // overlaps are grouped by the widest band of their tiles (tile diffs + 1): up to 31 / up to 63 cells take
            // the bit-parallel fill with one / two words per matrix row, wider ones the scalar fill (dhk_seg_vote)
            std::vector<std::array<std::vector<size_t>, 3>> part((size_t)std::max<int64_t>(nch, 1));
            dh_parallel_for(nch, 1, [&](int64_t clo, int64_t chi) {
                for (int64_t c = clo; c < chi; c++) {
                    const size_t i1 = std::min(las.size(), (size_t)(c + 1) * (size_t)grain);
                    for (size_t i = (size_t)c * (size_t)grain; i < i1; i++)
                        if (tmpl_of[i] >= 0 && !(las[i].flags & DH_FLAG_DISABLED)) {
                            // an overlap with a tile spanning more than SEG_MAX B bases (a > 100 % local
                            // indel rate) takes no part in the vote
                            const uint16_t *tr = trace.data() + las[i].toff;
                            bool too_long = false;
                            int32_t dmax = 0;
                            for (int32_t e = 0; e < las[i].tlen / 2; e++) {
                                too_long = too_long || tr[2 * e + 1] > SEG_MAX;
                                dmax = std::max<int32_t>(dmax, tr[2 * e]);
                            }
                            if (!too_long) part[(size_t)c][dmax + 1 <= 31 ? 0 : (dmax + 1 <= 63 ? 1 : 2)].push_back(i);
                        }
                }
            });
            for (int cls = 0; cls < 3; cls++) {
                for (const auto &v : part)
                    for (size_t i : v[(size_t)cls]) {
                        sel.push_back(i);
                        soff.push_back(soff.back() + (size_t)(las[i].tlen / 2));
                    }
                class_end[cls] = soff.back();
            }
        }
        segs.resize(soff.back());
        std::mutex red;
        dh_parallel_for((int64_t)sel.size(), 256, [&](int64_t lo_, int64_t hi_) {
            int32_t wm = 1, bm = 1;
            int64_t nc = 0;
            for (int64_t q = lo_; q < hi_; q++) {
                const size_t i = sel[(size_t)q];
                const int32_t t = tmpl_of[i];
                const dh_la &la = las[i];
                const uint16_t *tr = trace.data() + la.toff;
                SegDescH *out = segs.data() + soff[(size_t)q];
                int32_t a0 = la.abpos, b0 = la.bbpos;
                for (int32_t e = 0; e < la.tlen / 2; e++) {
                    int32_t a1 = (a0 / ts + 1) * ts;
                    if (a1 > la.aepos) a1 = la.aepos;
                    const int32_t b1 = b0 + tr[2 * e + 1];
                    // DP band: the trace's own path through the tile bounds the optimum (k_seg_vote)
                    const int32_t band = std::min<int32_t>((int32_t)tr[2 * e], std::max(a1 - a0, b1 - b0)) + 1;
                    out[e] = SegDescH{t, a0, a1, la.bread, b0, b1, (int32_t)(la.flags & DH_FLAG_COMP), band};
                    wm = std::max(wm, b1 - b0);
                    bm = std::max(bm, band);
                    nc += (int64_t)(a1 - a0) * std::min(b1 - b0, 2 * band + 1);
                    a0 = a1;
                    b0 = b1;
                }
            }
            std::lock_guard<std::mutex> lk(red);
            wmax = std::max(wmax, wm);
            bandmax = std::max(bandmax, bm);
            ncell += nc;
        });
    }
    *nseg_out = (int64_t)segs.size();
    *ncell_out = ncell;
    const int32_t nt = T->n;
    std::vector<int64_t> voff((size_t)nt + 1, 0), ooff((size_t)nt + 1, 0);
    for (int32_t t = 0; t < nt; t++) {
        const int64_t len = T->h_off[(size_t)t + 1] - T->h_off[(size_t)t];
        voff[(size_t)t + 1] = voff[(size_t)t] + len + 1;
        ooff[(size_t)t + 1] = ooff[(size_t)t] + len * (1 + 2 * MAXINS) + 8;
    }
    // big per-round buffers come from the context's grow-only scratch arena (slots 17..23)
    struct P {
        void *p = nullptr;
    };
    struct { int64_t *p; } d_voff, d_ooff;
    struct { uint32_t *p; } d_votes;
    struct { uint8_t *p; } d_out, d_stage, d_cnt;
    struct { int32_t *p; } d_status, d_outlen, d_coltmpl;
#define SCRP(id, buf, count)                                                                     \
    if (int rc_ = dh_scratch(ctx, id, sizeof(*buf.p) * std::max<size_t>((size_t)(count), 1), (void **)&buf.p)) return rc_;
    SCRP(17, d_voff, voff.size() + ooff.size())
    d_ooff.p = d_voff.p + voff.size();
    SCRP(18, d_votes, (size_t)voff.back() * VSTRIDE)
    SCRP(19, d_out, (size_t)ooff.back())
    SCRP(20, d_status, 2 + (size_t)nt + (size_t)voff.back())
    d_outlen.p = d_status.p + 2;
    d_coltmpl.p = d_outlen.p + nt;
    SCRP(21, d_stage, (size_t)voff.back() * (2 + 2 * MAXINS))
    d_cnt.p = d_stage.p + (size_t)voff.back() * (1 + 2 * MAXINS);
    HIPCHK(hipMemcpyAsync(d_voff.p, voff.data(), sizeof(int64_t) * voff.size(), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_ooff.p, ooff.data(), sizeof(int64_t) * ooff.size(), hipMemcpyHostToDevice, st));
    HIPCHK(dhk_memset(st, d_votes.p, 0, sizeof(uint32_t) * (size_t)voff.back() * VSTRIDE));
    HIPCHK(hipMemsetAsync(d_status.p, 0, sizeof(int32_t), st));
    // sparse votes: cover difference array and "other code" counts per column, scan partial sums
    const size_t ncolp = (size_t)voff.back() + 2;
    struct { uint32_t *p; } d_cdiff;
    SCRP(25, d_cdiff, 2 * ncolp + ncolp / 2048 + 8)
    uint32_t *d_vother = d_cdiff.p + ncolp, *d_csums = d_vother + ncolp;
    HIPCHK(dhk_memset(st, d_cdiff.p, 0, sizeof(uint32_t) * 2 * ncolp));
    if (int rc = dh_ensure_rc(R)) return rc;
    // the decision matrices of one launch live interleaved in HBM: bound the launch to ~6 GB
    for (int cls = 0; cls < 3; cls++) {
        const size_t c0 = cls ? class_end[cls - 1] : 0, c1 = class_end[cls];
        if (c1 <= c0) continue;
        const int32_t mode = getenv("DH_CONS_SCALAR") ? 0 : (cls == 0 ? 1 : (cls == 1 ? 2 : 0));  // (development: scalar fill for everything)
        // bytes of decisions per matrix row: two bit planes of 64 cells per word, or 2 bits per band cell
        const size_t mrow = mode ? (size_t)16 * (size_t)mode : 4 * (size_t)((2 * bandmax + 16) >> 4);
        const int64_t per_dp = (int64_t)(ts + 1) * (int64_t)mrow + 2 * SEG_MAX;
        const int64_t max_dp = std::max<int64_t>(4096, (6ll << 30) / per_dp);
        for (size_t s0 = c0; s0 < c1; s0 += (size_t)max_dp) {
            const int32_t cnt = (int32_t)std::min<size_t>((size_t)max_dp, c1 - s0);
            struct { SegDescH *p; } ds;
            struct { uint8_t *p; } fm, ob;
            SCRP(22, ds, (size_t)cnt)
            SCRP(23, fm, (size_t)cnt * (size_t)(ts + 1) * mrow + (size_t)cnt * 2 * SEG_MAX + (size_t)cnt * 2 + 32)
            ob.p = fm.p + (((size_t)cnt * (size_t)(ts + 1) * mrow + 7) & ~(size_t)7);  // op words: 8 ops each, 8-byte aligned
            uint16_t *d_nops = (uint16_t *)(ob.p + (((size_t)cnt * 2 * SEG_MAX + 7) & ~(size_t)7));
            HIPCHK(hipMemcpyAsync(ds.p, segs.data() + s0, sizeof(SegDescH) * (size_t)cnt, hipMemcpyHostToDevice, st));
            dhk_seg_vote(st, ds.p, cnt, T->view(), R->view(), R->d_rc, d_voff.p, (uint32_t *)fm.p, bandmax, wmax, ts,
                         ob.p, d_nops, d_votes.p, d_cdiff.p, d_vother, d_status.p, mode);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(st));
        }
    }
    {
        // column -> template map of the vote space (-1 for the spare column after each template)
        dhk_col_tmpl(st, d_voff.p, nt, voff.back(), d_coltmpl.p);
        HIPCHK(dhk_memset(st, d_cnt.p, 0, (size_t)voff.back()));
        dhk_scan(st, d_cdiff.p, (int64_t)ncolp, d_csums);  // exclusive: cover of column x = [x + 1]
        dhk_votes_finish(st, T->view(), d_voff.p, d_coltmpl.p, voff.back(), d_cdiff.p, d_vother, d_votes.p);
        dhk_emit(st, T->view(), nt, d_voff.p, d_votes.p, d_coltmpl.p, voff.back(), d_stage.p, d_cnt.p, d_ooff.p,
                 d_out.p, d_outlen.p);
        HIPCHK(hipStreamSynchronize(st));
    }
    HIPCHK(hipGetLastError());
    std::vector<int32_t> outlen((size_t)nt);
    int32_t status = 0;
    HIPCHK(hipMemcpyAsync(outlen.data(), d_outlen.p, sizeof(int32_t) * (size_t)nt, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(&status, d_status.p, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (status) return dh_fail(DH_EOVERFLOW, "consensus: tile exceeds the score-matrix capacity");
    // compact the emitted sequences into the next template DB (device to device)
    std::vector<int64_t> noff((size_t)nt + 1, 0);
    int32_t max_len = 0;
    for (int32_t t = 0; t < nt; t++) {
        noff[(size_t)t + 1] = noff[(size_t)t] + outlen[(size_t)t];
        max_len = std::max(max_len, outlen[(size_t)t]);
    }
    uint8_t *d_alloc = nullptr, *d_bases = nullptr;
    if (int rc = dh_alloc_bases(st, noff.back(), &d_alloc, &d_bases)) return rc;
    if (int rc = dh_db_adopt(ctx, d_alloc, d_bases, noff, T->h_group, newT)) {
        dh_dev_free(d_alloc);
        return rc;
    }
    std::vector<int32_t> ident((size_t)nt), zero((size_t)nt, 0);
    std::iota(ident.begin(), ident.end(), 0);
    DevBuf<int32_t> d_id, d_zero;
    HIPCHK(d_id.alloc((size_t)nt));
    HIPCHK(d_zero.alloc((size_t)nt));
    HIPCHK(hipMemcpyAsync(d_id.p, ident.data(), sizeof(int32_t) * (size_t)nt, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_zero.p, zero.data(), sizeof(int32_t) * (size_t)nt, hipMemcpyHostToDevice, st));
    dhk_gather_slices(st, d_out.p, d_ooff.p, d_id.p, d_zero.p, (*newT)->d_off, nt, max_len, d_bases);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    return DH_OK;
}

// ------------------------------------------------------------------------------------ process

struct DbGuard {
    std::vector<dh_db *> dbs;
    ~DbGuard()
    {
        for (dh_db *d : dbs) dh_db_destroy(d);
    }
};
struct SetGuard {
    std::vector<dh_la_set *> sets;
    ~SetGuard()
    {
        for (dh_la_set *s : sets) dh_la_set_destroy(s);
    }
};

// ------------------------------------------------------------------------------------ stage entry points

static int64_t trace_extent(const dh_la *las, int64_t n)
{
    int64_t m = 0;
    for (int64_t i = 0; i < n; i++) m = std::max<int64_t>(m, las[i].toff + las[i].tlen);
    return m;
}

// DAScover + DASqv for a pile-up DB (dazzler.d:3782-3792, 6142-6156): intrinsic QV of every
// tspace tile of every read from the overlaps of that read (las grouped by aread, ascending).
extern "C" int dh_tile_qv(dh_ctx *ctx, dh_db *db, const dh_la *las, int64_t n, const uint16_t *trace,
                          int32_t tspace, int32_t cov, uint8_t *qv, int32_t maxtiles)
{
    if (!ctx || !db || !qv || (n > 0 && (!las || !trace)) || tspace < 1 || maxtiles < 1 || cov < 1)
        return dh_fail(DH_EINVAL, "dh_tile_qv: bad argument");
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int32_t npr = db->n;
    std::vector<int32_t> la_first((size_t)npr + 1, 0);
    for (int64_t i = 0; i < n; i++) {
        if (las[i].aread < 0 || las[i].aread >= npr || (i > 0 && las[i].aread < las[i - 1].aread))
            return dh_fail(DH_EINVAL, "dh_tile_qv: overlaps must be grouped by aread (ascending) and inside the DB");
        la_first[(size_t)las[i].aread + 1]++;
    }
    for (int32_t r = 0; r < npr; r++) la_first[(size_t)r + 1] += la_first[(size_t)r];
    const int64_t nt = trace_extent(las, n);
    DevBuf<DhLa> d_las;
    DevBuf<uint16_t> d_tr;
    DevBuf<int32_t> d_first, d_cov;
    DevBuf<uint8_t> d_qv;
    std::vector<int32_t> cov_of((size_t)npr, cov);
    HIPCHK(d_las.alloc((size_t)n));
    HIPCHK(d_tr.alloc((size_t)nt));
    HIPCHK(d_first.alloc(la_first.size()));
    HIPCHK(d_cov.alloc(cov_of.size()));
    HIPCHK(d_qv.alloc((size_t)npr * maxtiles));
    if (n > 0) {
        HIPCHK(hipMemcpyAsync(d_las.p, las, sizeof(dh_la) * (size_t)n, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(d_tr.p, trace, sizeof(uint16_t) * (size_t)nt, hipMemcpyHostToDevice, st));
    }
    HIPCHK(hipMemcpyAsync(d_first.p, la_first.data(), sizeof(int32_t) * la_first.size(), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_cov.p, cov_of.data(), sizeof(int32_t) * cov_of.size(), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(d_qv.p, 255, (size_t)npr * maxtiles, st));
    dhk_tile_qv(st, d_las.p, d_tr.p, d_first.p, db->d_off, npr, tspace, d_cov.p, maxtiles, d_qv.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(qv, d_qv.p, (size_t)npr * maxtiles, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return DH_OK;
}

// computeintrinsicqv + daccord -f -I<i>,<i> (dazzler.d:4213-4255, 6172-6231): consensus of read
// ref_read of the DB from its overlaps (the records with aread == ref_read).  rounds > 1 re-aligns
// every read of the DB to the consensus and votes again, as dh_process_pileups does.
extern "C" int dh_consensus(dh_ctx *ctx, dh_db *db, const dh_la *las, int64_t n, const uint16_t *trace,
                            int32_t tspace, int32_t ref_read, int32_t rounds, uint8_t *out, int64_t cap,
                            int64_t *out_len)
{
    if (!ctx || !db || !out || !out_len || (n > 0 && (!las || !trace)) || ref_read < 0 || ref_read >= db->n ||
        rounds < 1 || rounds > 8 || tspace < 16 || tspace > SEG_MAX)
        return dh_fail(DH_EINVAL, "dh_consensus: bad argument");
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    DbGuard dbg;
    SetGuard sg;
    LaVec pl(las, las + n);
    const int64_t nt = trace_extent(las, n);
    TraceVec tr(trace, trace + nt);
    dh_db *T = nullptr;
    const int32_t rlen = (int32_t)(db->h_off[(size_t)ref_read + 1] - db->h_off[(size_t)ref_read]);
    const int32_t grp = db->h_group.empty() ? 0 : db->h_group[(size_t)ref_read];
    if (int rc = dh_db_from_slices(ctx, db, {ref_read}, {0}, {rlen}, {grp}, &T)) return rc;
    dbg.dbs.push_back(T);
    {
        std::vector<int32_t> tmpl_of(pl.size(), -1);
        for (size_t i = 0; i < pl.size(); i++)
            if (pl[i].aread == ref_read) tmpl_of[i] = 0;
        dh_db *nT = nullptr;
        int64_t nseg = 0, ncell = 0;
        if (int rc = consensus_round(ctx, T, db, pl, tr, tmpl_of, tspace, &nT, &nseg, &ncell)) return rc;
        dbg.dbs.push_back(nT);
        T = nT;
    }
    for (int32_t round = 1; round < rounds; round++) {
        dh_align_opts ro;
        dh_default_align_opts(&ro);
        ro.tspace = tspace;
        ro.min_len = 500;
        ro.max_la = 4;
        ro.max_cand = 32;
        dh_la_set *rset = nullptr;
        if (int rc = dh_align_db_ex(ctx, T, db, &ro, 0, 0, &rset)) return rc;
        sg.sets.push_back(rset);
        std::vector<int32_t> tmpl_of(rset->la.size(), 0);
        const int32_t alen = (int32_t)(T->h_off[1] - T->h_off[0]);
        for (size_t i = 0; i < rset->la.size(); i++) {
            dh_la &la = rset->la[i];
            const int32_t blen = (int32_t)(db->h_off[(size_t)la.bread + 1] - db->h_off[(size_t)la.bread]);
            if (!valid_pileup_alignment(la, false, alen, blen, tspace)) la.flags |= DH_FLAG_DISABLED;
        }
        dh_db *nT = nullptr;
        int64_t nseg = 0, ncell = 0;
        if (int rc = consensus_round(ctx, T, db, rset->la, rset->trace, tmpl_of, tspace, &nT, &nseg, &ncell)) return rc;
        dbg.dbs.push_back(nT);
        T = nT;
    }
    *out_len = T->total;
    if (T->total > cap) return dh_fail(DH_EOVERFLOW, "dh_consensus: output buffer too small");
    if (T->total > 0) HIPCHK(hipMemcpyAsync(out, T->d_bases, (size_t)T->total, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return DH_OK;
}

// ------------------------------------------------------------------------------------ crop stage
struct dh_cropped;
extern "C" int dh_process_pileups_masked(dh_ctx *ctx, dh_db *contigs, dh_db *reads, const dh_la *las, int64_t n,
                                         const uint16_t *trace, const dh_pileups *piles, const int64_t *rep_ptr,
                                         const int32_t *rep_iv, const dh_process_opts *opts, dh_insertions **out);
extern "C" int dh_crop_pileups_masked(dh_ctx *ctx, dh_db *contigs, dh_db *reads, int32_t read_first, const dh_la *las,
                                      int64_t n, const uint16_t *trace, const dh_pileups *piles, const int64_t *rep_ptr,
                                      const int32_t *rep_iv, const dh_process_opts *opts, dh_cropped **out);
extern "C" int dh_process_cropped(dh_ctx *ctx, dh_db *contigs, dh_cropped *crop, const dh_process_opts *opts,
                                  dh_insertions **out);

// What `dentist process` holds after cropPileUp (cropper.d:113-175): per pile-up the common trace
// points, per cropped read its pile-up, its position in the pile-up's read list, its read id and
// its bases ([support patch] + read slice + [support patch]).  The bases live on the device
// (`dev`, an ungrouped DB in (pile, entry) order) and/or on the host.
struct dh_cropped {
    dh_ctx *ctx = nullptr;
    std::vector<dh_insertion> rec;
    std::vector<int32_t> pile, entry, read_id;
    std::vector<uint8_t> kind;  // per cropped read, bits 0-1: 0 = alignments on both flanks (spans the gap), 1 = on flank 0 only, 2 = on flank 1 only;
                                // bit 2 / 3: its alignment on flank 0 / 1 is a complement one
    std::vector<int64_t> off{0};
    // page-locked and not zero-filled on resize(): the cropped reads travel device -> host -> (collective) -> host -> device
    // in the sharded path, 21 MB per rank at N = 8
    std::vector<uint8_t, PinnedAlloc<uint8_t>> bases;
    bool host_valid = false;
    bool comp_known = true;  // false: made without kinds (dh_cropped_create): the complement bits are not there
    int32_t batch_most = 0;  // largest pile-up of the batch this crop is a part of (dh_process_pileups splits a batch): the
                             // record slots of the pile-up alignment are sized by it, so that the split does not show
    dh_db *dev = nullptr;
    float ms_crop = 0;
};

extern "C" void dh_cropped_destroy(dh_cropped *c)
{
    if (!c) return;
    if (c->dev) dh_db_destroy(c->dev);
    delete c;
}
extern "C" int32_t dh_cropped_npiles(const dh_cropped *c) { return c ? (int32_t)c->rec.size() : 0; }
extern "C" const dh_insertion *dh_cropped_records(const dh_cropped *c) { return c ? c->rec.data() : nullptr; }
extern "C" int32_t dh_cropped_nreads(const dh_cropped *c) { return c ? (int32_t)c->pile.size() : 0; }
extern "C" const int32_t *dh_cropped_pile(const dh_cropped *c) { return c ? c->pile.data() : nullptr; }
extern "C" const int32_t *dh_cropped_entry(const dh_cropped *c) { return c ? c->entry.data() : nullptr; }
extern "C" const int32_t *dh_cropped_read_id(const dh_cropped *c) { return c ? c->read_id.data() : nullptr; }
extern "C" const uint8_t *dh_cropped_kind(const dh_cropped *c) { return c ? c->kind.data() : nullptr; }
extern "C" const int64_t *dh_cropped_offsets(const dh_cropped *c) { return c ? c->off.data() : nullptr; }
extern "C" const uint8_t *dh_cropped_bases(dh_cropped *c)
{
    if (!c) return nullptr;
    if (!c->host_valid) {
        c->bases.resize((size_t)std::max<int64_t>(c->off.back(), 1));
        if (c->dev && c->off.back() > 0) {
            if (hipSetDevice(c->ctx->device) != hipSuccess ||
                hipMemcpyAsync(c->bases.data(), c->dev->d_bases, (size_t)c->off.back(), hipMemcpyDeviceToHost,
                               c->ctx->stream) != hipSuccess ||
                hipStreamSynchronize(c->ctx->stream) != hipSuccess) {
                dh_fail(DH_EHIP, "dh_cropped_bases: device to host copy failed");
                return nullptr;
            }
        }
        c->host_valid = true;
    }
    return c->bases.data();
}

extern "C" int dh_cropped_create2(const dh_insertion *rec, int32_t npiles, int32_t nreads, const int32_t *pile,
                                  const int32_t *entry, const int32_t *read_id, const uint8_t *kind, const int64_t *off,
                                  const uint8_t *bases, dh_cropped **out);
extern "C" int dh_cropped_create(const dh_insertion *rec, int32_t npiles, int32_t nreads, const int32_t *pile,
                                 const int32_t *entry, const int32_t *read_id, const int64_t *off,
                                 const uint8_t *bases, dh_cropped **out)
{
    return dh_cropped_create2(rec, npiles, nreads, pile, entry, read_id, nullptr, off, bases, out);
}

// kind: per read 0 / 1 / 2 (see dh_cropped_kind), NULL = every read spans its gap
extern "C" int dh_cropped_create2(const dh_insertion *rec, int32_t npiles, int32_t nreads, const int32_t *pile,
                                  const int32_t *entry, const int32_t *read_id, const uint8_t *kind, const int64_t *off,
                                  const uint8_t *bases, dh_cropped **out)
{
    if (npiles < 0 || nreads < 0 || !out || (npiles > 0 && !rec) ||
        (nreads > 0 && (!pile || !entry || !read_id || !off || !bases)))
        return dh_fail(DH_EINVAL, "dh_cropped_create: bad argument");
    dh_cropped *c = new dh_cropped();
    c->rec.assign(rec, rec + npiles);
    if (nreads > 0) {
        if (off[0] != 0) {
            delete c;
            return dh_fail(DH_EINVAL, "dh_cropped_create: off[0] must be 0");
        }
        for (int32_t i = 0; i < nreads; i++)
            if (pile[i] < 0 || pile[i] >= npiles || off[i + 1] < off[i] ||
                (i > 0 && (pile[i] < pile[i - 1] || (pile[i] == pile[i - 1] && entry[i] <= entry[i - 1])))) {
                delete c;
                return dh_fail(DH_EINVAL, "dh_cropped_create: reads must be ordered by (pile, entry)");
            }
        c->pile.assign(pile, pile + nreads);
        c->entry.assign(entry, entry + nreads);
        c->read_id.assign(read_id, read_id + nreads);
        if (kind)
            c->kind.assign(kind, kind + nreads);
        else {
            c->kind.assign((size_t)nreads, 0);
            c->comp_known = false;
        }
        for (uint8_t k : c->kind)
            if ((k & 3) > 2 || k > 15) {
                delete c;
                return dh_fail(DH_EINVAL, "dh_cropped_create: kind must be 0, 1 or 2 (| 4, 8: complement alignment on flank 0, 1)");
            }
        c->off.assign(off, off + nreads + 1);
        c->bases.assign(bases, bases + off[nreads]);
    }
    c->host_valid = true;
    *out = c;
    return DH_OK;
}

// cropPileUp for a batch (cropper.d:113-175, 446-550): common trace point per flank from ALL entries
// of a pile-up; bases are cut for the entries whose read is in `reads` -- read ids in the triples
// are ids of the whole reads DB, `reads` holds [read_first, read_first + reads->n) of it (one rank's
// share when the mapping is sharded; read_first = 0 and the whole DB otherwise).  LAs of reads that
// are not held here only need their A intervals (no trace).
extern "C" int dh_crop_pileups(dh_ctx *ctx, dh_db *contigs, dh_db *reads, int32_t read_first, const dh_la *las,
                               int64_t n, const uint16_t *trace, const dh_pileups *piles,
                               const dh_process_opts *opts, dh_cropped **out)
{
    return dh_crop_pileups_masked(ctx, contigs, reads, read_first, las, n, trace, piles, nullptr, nullptr, opts, out);
}

// rep_ptr[ncontigs + 1] / rep_iv: the repeat mask (sorted disjoint (begin, end) pairs per contig) the common trace points
// keep out of when they can (cropper.d:446-500); NULL = no mask
extern "C" int dh_crop_pileups_masked(dh_ctx *ctx, dh_db *contigs, dh_db *reads, int32_t read_first, const dh_la *las,
                                      int64_t n, const uint16_t *trace, const dh_pileups *piles, const int64_t *rep_ptr,
                                      const int32_t *rep_iv, const dh_process_opts *opts, dh_cropped **out)
{
    if (!ctx || !contigs || !reads || !piles || !opts || !out || (n > 0 && !las) || (rep_ptr && !rep_iv && rep_ptr[contigs->n] > 0))
        return dh_fail(DH_EINVAL, "dh_crop_pileups: NULL argument");
    const dh_process_opts &o = *opts;
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    hipEvent_t ev[2];
    for (auto &e : ev) HIPCHK(hipEventCreate(&e));
    struct EvGuard {
        hipEvent_t *e;
        ~EvGuard()
        {
            for (int i = 0; i < 2; i++) (void)hipEventDestroy(e[i]);
        }
    } evg{ev};
    HIPCHK(hipEventRecord(ev[0], st));
    dh_cropped *c = new dh_cropped();
    c->ctx = ctx;
    struct CGuard {
        dh_cropped *&c;
        bool ok = false;
        ~CGuard()
        {
            if (!ok) dh_cropped_destroy(c);
        }
    } cg{c};
    const int32_t np = (int32_t)piles->contig_left.size();
    c->rec.resize((size_t)np);
    const int32_t tsm = o.tspace_map;
    // every pile-up read = [support patch] + read slice + [support patch]; parts are gathered on
    // the device from the reads DB (src 0) and the contigs DB (src 1)
    std::vector<PartDescH> parts;
    int32_t pile_max_len = 0;
    // pile-ups are independent: host threads compute crop points and read slices (the trace walks are
    // cache misses into the mapping's trace array), the parts are laid out serially afterwards
    struct Slice {
        int32_t e, rd, lrd, b0, b1, kind;  // kind: 0 / 1 / 2 | complement of the alignment on flank 0 << 2 | on flank 1 << 3
    };
    struct PileCrop {
        int32_t pc[2] = {-1, -1}, p0[2] = {0, 0}, p1[2] = {0, 0};  // support patch of flank f: contig pc[f], [p0, p1)
        std::vector<Slice> sl;
    };
    std::vector<PileCrop> pc((size_t)np);
    std::atomic<int> err{0};  // 1 gap outside, 2 LA index, 3 trace NULL, 4 trace does not fit
    dh_parallel_for(np, 4, [&](int64_t plo, int64_t phi) {
        for (int64_t p = plo; p < phi; p++) {
            dh_insertion &r = c->rec[(size_t)p];
            memset(&r, 0, sizeof(r));
            // the two flanks (cropper.d:113-175 treats every pile-up alike: one common trace point per involved
            // contig, taken from the alignments seeded there)
            const std::array<int32_t, 4> jn = piles->join_of((size_t)p);
            const int32_t nf = jn[2] < 0 ? 1 : 2;
            const int32_t fc[2] = {jn[0], jn[2]};
            const bool front[2] = {jn[1] == DH_SEED_FRONT, jn[3] == DH_SEED_FRONT};
            if (fc[0] < 0 || fc[0] >= contigs->n || (nf == 2 && fc[1] >= contigs->n)) {
                err = 1;
                continue;
            }
            r.contig_left = fc[0];
            r.contig_right = nf == 2 ? fc[1] : -1;
            r.join = (front[0] ? DH_JOIN_FLANK0_FRONT : 0) | (nf == 2 && !front[1] ? DH_JOIN_FLANK1_BACK : 0) | (nf == 1 ? DH_JOIN_EXTENSION : 0);
            r.ref_read = r.ref_read_id = -1;
            r.crop_left = r.crop_right = -1;
            if (nf == 2 && fc[0] == fc[1]) {
                r.status = DH_PILE_UNSUPPORTED_JOIN;
                continue;
            }
            const std::vector<int32_t> &tr3 = piles->triples[(size_t)p];
            const int32_t ne = (int32_t)tr3.size() / 3;
            Region reg[2] = {Region{{0, INT32_MAX}}, Region{{0, INT32_MAX}}};
            bool bad = false;
            for (int32_t e = 0; e < ne && !bad; e++) {
                // an entry is a read spanning the gap (two alignments) or an extension over one contig end
                // merged into the gap's pile-up (one alignment, the other index is -1: scaffold.d:789-816)
                const int32_t ix[2] = {tr3[(size_t)e * 3 + 1], tr3[(size_t)e * 3 + 2]};
                if (ix[0] < -1 || ix[0] >= n || ix[1] < -1 || ix[1] >= n || (ix[0] < 0 && ix[1] < 0) || (nf == 1 && ix[1] >= 0)) {
                    bad = true;
                    break;
                }
                for (int f = 0; f < nf; f++)
                    if (ix[f] >= 0) {
                        if (las[ix[f]].aread != fc[f])
                            bad = true;
                        else
                            intersect_chain(reg[f], las, n, ix[f]);
                    }
            }
            if (bad) {
                err = 2;
                continue;
            }
            int32_t clen[2] = {0, 0}, crop[2] = {-1, -1};
            for (int f = 0; f < nf; f++) {
                clen[f] = (int32_t)(contigs->h_off[(size_t)fc[f] + 1] - contigs->h_off[(size_t)fc[f]]);
                const int64_t m0 = rep_ptr ? rep_ptr[fc[f]] : 0, m1 = rep_ptr ? rep_ptr[fc[f] + 1] : 0;
                crop[f] = common_trace_point(reg[f], clen[f], tsm, front[f], rep_iv ? rep_iv + 2 * m0 : nullptr, m1 - m0);
            }
            r.crop_left = crop[0];
            r.crop_right = crop[1];
            if (crop[0] < 0 || (nf == 2 && crop[1] < 0)) {
                r.status = DH_PILE_NO_COMMON_TRACE_POINT;
                continue;
            }
            PileCrop &q = pc[(size_t)p];
            // fetchSupportPatches, cropper.d:224-262
            for (int f = 0; f < nf; f++) {
                q.pc[f] = fc[f];
                if (front[f]) {
                    if (crop[f] < o.min_anchor) {
                        q.p0[f] = crop[f];
                        q.p1[f] = std::min(clen[f], o.min_anchor);
                    }
                } else if (clen[f] - crop[f] < o.min_anchor) {
                    q.p0[f] = std::max(0, clen[f] - o.min_anchor);
                    q.p1[f] = crop[f];
                }
            }
            for (int32_t e = 0; e < ne; e++) {
                const int32_t rd = tr3[(size_t)e * 3];
                const int64_t lrd = (int64_t)rd - read_first;
                if (lrd < 0 || lrd >= reads->n) continue;  // held by another rank
                if (!trace) {
                    err = 3;
                    break;
                }
                const int32_t ix[2] = {tr3[(size_t)e * 3 + 1], tr3[(size_t)e * 3 + 2]};
                const int32_t rl = (int32_t)(reads->h_off[(size_t)lrd + 1] - reads->h_off[(size_t)lrd]);
                // getCroppingSlice per alignment, intersected (cropper.d:339-348, 503-550): a back-seeded alignment
                // keeps [crop point, read end), a front-seeded one [0, crop point) of the read as the alignment sees it
                // -- mirrored for a complement alignment (:533-538)
                // (a chain translates through the first of its members that covers the crop point)
                int32_t b0 = 0, b1 = rl, kind = ix[1] < 0 ? 1 : (ix[0] < 0 ? 2 : 0);
                bool fail = false;
                for (int f = 0; f < nf && !fail; f++) {
                    if (ix[f] < 0) continue;
                    const int64_t m = covering_member(las, n, ix[f], crop[f]);
                    if (m < 0) {
                        fail = true;
                        break;
                    }
                    const int32_t b = translate_floor_b(las[m], trace + las[m].toff, tsm, crop[f]);
                    int32_t lo = front[f] ? 0 : b, hi = front[f] ? b : rl;
                    if (las[ix[f]].flags & DH_FLAG_COMP) {
                        const int32_t t = lo;
                        lo = rl - hi;
                        hi = rl - t;
                        kind |= 4 << f;
                    }
                    b0 = std::max(b0, lo);
                    b1 = std::min(b1, hi);
                }
                if (fail) {
                    err = 4;
                    break;
                }
                if (b1 - b0 < 14) continue;  // records shorter than 14 bp are dropped (dazzler.d:150)
                if (b0 < 0 || b1 > rl) {
                    err = 4;
                    break;
                }
                q.sl.push_back(Slice{e, rd, (int32_t)lrd, b0, b1, kind});
            }
        }
    });
    switch (err.load()) {
        case 1: return dh_fail(DH_EINVAL, "dh_crop_pileups: gap outside the contigs DB");
        case 2: return dh_fail(DH_EINVAL, "dh_crop_pileups: LA index out of range, or an alignment that is not on its flank's contig");
        case 3: return dh_fail(DH_EINVAL, "dh_crop_pileups: trace is NULL");
        case 4: return dh_fail(DH_EINVAL, "dh_crop_pileups: trace does not fit its read");
        default: break;
    }
    for (int32_t p = 0; p < np; p++) {
        const PileCrop &q = pc[(size_t)p];
        const std::array<int32_t, 4> jn = piles->join_of((size_t)p);
        const bool front[2] = {jn[1] == DH_SEED_FRONT, jn[3] == DH_SEED_FRONT};
        dh_insertion &r = c->rec[(size_t)p];
        for (const Slice &x : q.sl) {
            int64_t dst = c->off.back();
            // getSingleReadPatch / getReadPatches, cropper.d:351-378: the patch of an alignment goes to the read's front
            // when (contig seed == front) == complement, else to its back, reverse-complemented for a complement
            // alignment; an extension entry gets the patch of its own contig only
            int pre = -1, post = -1;
            for (int f = 0; f < 2; f++) {
                const bool has = f == 0 ? (x.kind & 3) != 2 : ((x.kind & 3) != 1 && q.pc[1] >= 0);
                if (!has || q.p1[f] <= q.p0[f]) continue;
                const bool comp = (x.kind & (4 << f)) != 0;
                if (front[f] == comp)
                    pre = f;
                else
                    post = f;
            }
            if (pre >= 0) {
                parts.push_back(PartDescH{1, q.pc[pre], q.p0[pre], q.p1[pre] - q.p0[pre], (x.kind & (4 << pre)) ? 1 : 0, 0, dst});
                dst += q.p1[pre] - q.p0[pre];
            }
            parts.push_back(PartDescH{0, x.lrd, x.b0, x.b1 - x.b0, 0, 0, dst});
            dst += x.b1 - x.b0;
            if (post >= 0) {
                parts.push_back(PartDescH{1, q.pc[post], q.p0[post], q.p1[post] - q.p0[post], (x.kind & (4 << post)) ? 1 : 0, 0, dst});
                dst += q.p1[post] - q.p0[post];
            }
            pile_max_len = std::max<int32_t>(pile_max_len, (int32_t)(dst - c->off.back()));
            c->off.push_back(dst);
            c->pile.push_back(p);
            c->entry.push_back(x.e);
            c->read_id.push_back(x.rd);
            c->kind.push_back((uint8_t)x.kind);
            r.nreads++;
        }
    }
    {
        uint8_t *d_alloc = nullptr, *d_bases = nullptr;
        if (int rc = dh_alloc_bases(st, c->off.back(), &d_alloc, &d_bases)) return rc;
        if (int rc = dh_db_adopt(ctx, d_alloc, d_bases, c->off, std::vector<int32_t>(), &c->dev)) {
            dh_dev_free(d_alloc);
            return rc;
        }
        if (!parts.empty()) {
            DevBuf<PartDescH> d_parts;
            HIPCHK(d_parts.alloc(parts.size()));
            HIPCHK(hipMemcpyAsync(d_parts.p, parts.data(), sizeof(PartDescH) * parts.size(), hipMemcpyHostToDevice, st));
            dhk_gather_parts(st, reads->d_bases, reads->d_off, contigs->d_bases, contigs->d_off, d_parts.p,
                             (int32_t)parts.size(), pile_max_len, d_bases);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(st));
        }
    }
    HIPCHK(hipEventRecord(ev[1], st));
    HIPCHK(hipEventSynchronize(ev[1]));
    HIPCHK(hipEventElapsedTime(&c->ms_crop, ev[0], ev[1]));
    cg.ok = true;
    *out = c;
    return DH_OK;
}

extern "C" int dh_process_pileups(dh_ctx *ctx, dh_db *contigs, dh_db *reads, const dh_la *las, int64_t n,
                                  const uint16_t *trace, const dh_pileups *piles,
                                  const dh_process_opts *opts, dh_insertions **out)
{
    return dh_process_pileups_masked(ctx, contigs, reads, las, n, trace, piles, nullptr, nullptr, opts, out);
}

// The same on a mapping result whose trace values were left on the device (dh_map_reads, want_sorted & 8): the cropper reads
// the trace of the pile-up reads' records only -- one record in ten at configs[2] --, so those ranges are gathered on the
// device (k_gather_ranges16), brought over in one copy and laid out at their offsets in a host array nothing else of
// which is touched; 330 MB of trace values per step of configs[2] no longer cross PCIe.  rep_ptr / rep_iv may be NULL.
extern "C" int dh_process_pileups_set(dh_ctx *ctx, dh_db *contigs, dh_db *reads, dh_la_set *set, const dh_pileups *piles,
                                      const int64_t *rep_ptr, const int32_t *rep_iv, const dh_process_opts *opts,
                                      dh_insertions **out)
{
    if (!ctx || !reads || !set || !piles || !out) return dh_fail(DH_EINVAL, "dh_process_pileups_set: NULL argument");
    const int64_t n = (int64_t)set->la.size();
    if (!(set->trace.empty() && set->d_trace_own_len > 0))
        return dh_process_pileups_masked(ctx, contigs, reads, set->la.data(), n, set->trace.data(), piles, rep_ptr, rep_iv, opts, out);
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    std::vector<uint8_t> need((size_t)reads->n, 0);
    for (const auto &t : piles->triples)
        for (size_t x = 0; x + 2 < t.size(); x += 3) {
            if (t[x] < 0 || t[x] >= reads->n) return dh_fail(DH_EINVAL, "dh_process_pileups_set: read id out of range");
            need[(size_t)t[x]] = 1;
        }
    const dh_la *la = set->la.data();
    // the records of the needed reads (all of them: chain members follow their first record), found by the host threads
    const int64_t grain = 1 << 15, nch = (n + grain - 1) / grain;
    std::vector<std::vector<int64_t>> part((size_t)std::max<int64_t>(nch, 1));
    std::atomic<int> bad{0};
    dh_parallel_for(nch, 1, [&](int64_t clo, int64_t chi) {
        for (int64_t c = clo; c < chi; c++) {
            auto &v = part[(size_t)c];
            for (int64_t i = c * grain; i < std::min(n, (c + 1) * grain); i++) {
                if (la[i].bread < 0 || la[i].bread >= reads->n) continue;
                if (!need[(size_t)la[i].bread] || la[i].tlen <= 0) continue;
                if (la[i].toff < 0 || la[i].toff + la[i].tlen > set->d_trace_own_len) bad = 1;
                v.push_back(i);
            }
        }
    });
    if (bad.load()) return dh_fail(DH_EINVAL, "dh_process_pileups_set: a record's trace lies outside the set's trace");
    std::vector<int64_t> desc;
    int64_t total = 0;
    for (const auto &v : part)
        for (int64_t i : v) {
            desc.push_back(la[i].toff);
            desc.push_back(total);
            desc.push_back(la[i].tlen);
            total += la[i].tlen;
        }
    const int64_t nsel = (int64_t)desc.size() / 3;
    // (from the pool of page-locked result buffers, as the whole trace would have been: no page is faulted in here -- a
    // malloc'd array cost 30 ms of first-touch faults per call -- and nothing but the gathered ranges is written)
    TraceVec sparse_v((size_t)std::max<int64_t>(set->d_trace_own_len, 1));
    uint16_t *sparse = sparse_v.data();
    if (nsel > 0) {
        if (nsel > INT32_MAX) return dh_fail(DH_EOVERFLOW, "dh_process_pileups_set: too many records");
        DevBuf<int64_t> d_desc;
        DevBuf<uint16_t> d_tt;
        HIPCHK(d_desc.alloc(desc.size()));
        HIPCHK(d_tt.alloc((size_t)total));
        TraceVec tmp((size_t)total);
        HIPCHK(hipMemcpyAsync(d_desc.p, desc.data(), sizeof(int64_t) * desc.size(), hipMemcpyHostToDevice, st));
        dhk_gather_ranges16(st, set->d_trace_own, d_desc.p, (int32_t)nsel, d_tt.p);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(tmp.data(), d_tt.p, sizeof(uint16_t) * (size_t)total, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        const int64_t *dp = desc.data();
        const uint16_t *tp = tmp.data();
        dh_parallel_for(nsel, 4096, [&](int64_t lo, int64_t hi) {
            for (int64_t r = lo; r < hi; r++) memcpy(sparse + dp[3 * r], tp + dp[3 * r + 1], sizeof(uint16_t) * (size_t)dp[3 * r + 2]);
        });
    }
    return dh_process_pileups_masked(ctx, contigs, reads, la, n, sparse, piles, rep_ptr, rep_iv, opts, out);
}

// with the repeat mask of the contigs (--mask of `dentist process`: the cropper keeps its trace points out of it)
extern "C" int dh_process_pileups_masked(dh_ctx *ctx, dh_db *contigs, dh_db *reads, const dh_la *las, int64_t n,
                                         const uint16_t *trace, const dh_pileups *piles, const int64_t *rep_ptr,
                                         const int32_t *rep_iv, const dh_process_opts *opts, dh_insertions **out)
{
    if (!ctx || !contigs || !reads || !piles || !opts || !out || (n > 0 && (!las || !trace)))
        return dh_fail(DH_EINVAL, "dh_process_pileups: NULL argument");
    int32_t batch_most = 0;
    for (const auto &t : piles->triples) batch_most = std::max(batch_most, (int32_t)(t.size() / 3));
    auto one = [&](dh_ctx *cx, const dh_pileups *pl, dh_insertions **res) -> int {
        dh_cropped *c = nullptr;
        if (int rc = dh_crop_pileups_masked(cx, contigs, reads, 0, las, n, trace, pl, rep_ptr, rep_iv, opts, &c)) return rc;
        c->batch_most = batch_most;
        const int rc = dh_process_cropped(cx, contigs, c, opts, res);
        dh_cropped_destroy(c);
        return rc;
    };
    const size_t np = piles->contig_left.size();
    if (np < 64 || getenv("DH_PROCESS_SERIAL")) return one(ctx, piles, out);
    // Parts of the batch run concurrently, each on its own context (streams, scratch) and host thread: between its
    // kernels a part has host work -- device-to-host copies of 3.5 M overlap records, LAsort, filters and chains, the
    // per-tile descriptors of the consensus rounds -- during which the device served nobody (configs[2]: one call 188 ms,
    // two concurrent halves 160 ms).  Pile-ups are independent and keep their order; the parts balance n^2.
    // (three parts: with two, both tend to sit in their host phases at the same time -- measured at configs[2] on one
    // MI355X, two runs each: 2 parts 116.9 / 127.3 ms, 3 parts 109.3 / 110.9 ms, 4 parts 112.1 ms of process wall)
    int32_t nparts = 3;
    if (const char *e = getenv("DH_PROCESS_PARTS")) nparts = std::max(1, std::min(4, atoi(e)));
    nparts = (int32_t)std::min<size_t>((size_t)nparts, np / 16);
    if (nparts < 2) return one(ctx, piles, out);
    for (int32_t k = 1; k < nparts; k++)
        if (!ctx->sub[k - 1])
            if (int rc = dh_ctx_create(ctx->device, nullptr, &ctx->sub[k - 1])) return rc;
    std::vector<double> cost(np);
    double total = 0;
    for (size_t p = 0; p < np; p++) {
        const double e = (double)piles->triples[p].size() / 3.0;
        cost[p] = e * e;
        total += cost[p];
    }
    // contiguous runs of pile-ups of about total / nparts each, none empty
    // cumulative shares of the parts (equal unless DH_PROCESS_SPLIT = "w0,w1,..." says otherwise: development)
    std::vector<double> wcum((size_t)nparts + 1, 0.0);
    {
        std::vector<double> wt((size_t)nparts, 1.0);
        if (const char *e = getenv("DH_PROCESS_SPLIT")) {
            const char *q = e;
            for (int32_t k = 0; k < nparts && *q; k++) {
                wt[(size_t)k] = std::max(0.01, atof(q));
                while (*q && *q != ',') q++;
                if (*q == ',') q++;
            }
        }
        double sum = 0;
        for (double x : wt) sum += x;
        for (int32_t k = 0; k < nparts; k++) wcum[(size_t)k + 1] = wcum[(size_t)k] + wt[(size_t)k] / sum;
    }
    std::vector<size_t> cut((size_t)nparts + 1, np);
    cut[0] = 0;
    {
        size_t p = 0;
        double acc = 0;
        for (int32_t k = 1; k < nparts; k++) {
            while (p < np && acc + cost[p] <= total * wcum[(size_t)k]) acc += cost[p++];
            while (p < cut[(size_t)k - 1] + 1) acc += cost[p++];
            p = std::min(p, np - (size_t)(nparts - k));
            cut[(size_t)k] = p;
        }
    }
    std::vector<dh_pileups> part((size_t)nparts);
    for (int32_t k = 0; k < nparts; k++)
        for (size_t p = cut[(size_t)k]; p < cut[(size_t)k + 1]; p++) {
            part[(size_t)k].contig_left.push_back(piles->contig_left[p]);
            part[(size_t)k].triples.push_back(piles->triples[p]);
            if (!piles->join.empty()) part[(size_t)k].join.push_back(piles->join[p]);
        }
    std::vector<dh_insertions *> res((size_t)nparts, nullptr);
    std::vector<int> rcs((size_t)nparts, DH_OK);
    std::vector<std::string> msgs((size_t)nparts);
    std::vector<ProcStats> sts((size_t)nparts);
    std::vector<std::thread> workers;
    for (int32_t k = 1; k < nparts; k++)
        workers.emplace_back([&, k] {
            try {
                rcs[(size_t)k] = one(ctx->sub[k - 1], &part[(size_t)k], &res[(size_t)k]);
                if (rcs[(size_t)k]) msgs[(size_t)k] = dh_last_error();
            } catch (const std::exception &e) {  // (an exception leaving a thread would end the process)
                rcs[(size_t)k] = DH_EINVAL;
                msgs[(size_t)k] = std::string("dh_process_pileups: a concurrent part of the batch failed: ") + e.what();
            }
            sts[(size_t)k] = g_pstats;
        });
    try {
        rcs[0] = one(ctx, &part[0], &res[0]);
    } catch (const std::exception &e) {  // (the workers must be joined whatever happens here; no exception crosses the C ABI)
        rcs[0] = dh_fail(DH_EINVAL, std::string("dh_process_pileups: the first part of the batch failed: ") + e.what());
    }
    for (std::thread &w : workers) w.join();
    // the other contexts' alignment statistics belong to this call (the streams' event times overlap: their sum
    // overstates the kernel time of the step, never understates it)
    for (int32_t k = 1; k < nparts; k++) {
        dh_cum_stats &a = ctx->cum, &b = ctx->sub[k - 1]->cum;
        a.ms_index += b.ms_index; a.ms_seed += b.ms_seed; a.ms_wave += b.ms_wave; a.ms_gather += b.ms_gather;
        a.wave_launches += b.wave_launches; a.wave_cells += b.wave_cells; a.alignments += b.alignments; a.las += b.las;
        a.aligned_bp += b.aligned_bp; a.trace_values += b.trace_values; a.hits += b.hits; a.b_bases += b.b_bases;
        b = dh_cum_stats();
    }
    for (int32_t k = 0; k < nparts; k++)
        if (rcs[(size_t)k]) {
            for (dh_insertions *r : res) dh_insertions_destroy(r);
            if (k == 0) return rcs[0];  // (its message is this thread's last error)
            return dh_fail(rcs[(size_t)k], msgs[(size_t)k].empty() ? "dh_process_pileups: a concurrent part of the batch failed" : msgs[(size_t)k]);
        }
    for (int32_t k = 1; k < nparts; k++) {
        for (int i = 0; i < 7; i++) g_pstats.ms[i] = std::max(g_pstats.ms[i], sts[(size_t)k].ms[i]);  // side by side
        for (int i = 0; i < 3; i++) g_pstats.counters[i] += sts[(size_t)k].counters[i];
        for (int i = 0; i < 4; i++) g_pstats.work[i] += sts[(size_t)k].work[i];
    }
    // later parts appended to the first
    dh_insertions *r0 = res[0];
    for (int32_t k = 1; k < nparts; k++) {
        dh_insertions *r1 = res[(size_t)k];
        const int64_t b0 = (int64_t)r0->bases.size();
        const int32_t f0 = (int32_t)r0->flank.size(), i0 = r0->ids_off.empty() ? 0 : r0->ids_off.back();
        const int64_t t0 = (int64_t)r0->flank_tr.size();
        for (dh_insertion x : r1->rec) {
            x.cons_off += b0;
            r0->rec.push_back(x);
        }
        r0->bases.insert(r0->bases.end(), r1->bases.begin(), r1->bases.end());
        for (dh_la f : r1->flank) {
            f.toff += t0;
            r0->flank.push_back(f);
        }
        r0->flank_tr.insert(r0->flank_tr.end(), r1->flank_tr.begin(), r1->flank_tr.end());
        for (int32_t v : r1->flank_of) r0->flank_of.push_back(v < 0 ? v : v + f0);
        if (!r1->ids_off.empty()) {
            if (r0->ids_off.empty()) r0->ids_off.push_back(0);
            for (size_t j = 1; j < r1->ids_off.size(); j++) r0->ids_off.push_back(r1->ids_off[j] + i0);
            r0->ids.insert(r0->ids.end(), r1->ids.begin(), r1->ids.end());
        }
        dh_insertions_destroy(r1);
    }
    *out = r0;
    return DH_OK;
}

// The pile-up stages of `dentist process` after the crop (package.d:283-374): pile-up alignment ->
// filter -> tile QV -> reference read -> consensus -> flank re-alignment -> insertion.
extern "C" int dh_process_cropped(dh_ctx *ctx, dh_db *contigs, dh_cropped *crop, const dh_process_opts *opts,
                                  dh_insertions **out)
{
    if (!ctx || !contigs || !crop || !opts || !out) return dh_fail(DH_EINVAL, "dh_process_cropped: NULL argument");
    const dh_process_opts &o = *opts;
    if (o.max_reads != 0 && (o.max_reads < 3 || o.max_reads > 250))
        return dh_fail(DH_EINVAL, "max_reads must be 0 (no cap) or in [3, 250]");
    if (o.max_partners != 0 && o.max_partners < 4) return dh_fail(DH_EINVAL, "max_partners must be 0 (every pair) or at least 4");
    // (0 is refused: it is what a caller built against the 56-byte struct of ABI 3, or one that zero-fills the struct, would
    // pass without meaning it -- "every chain at or above min_score" is 1)
    if (o.min_relative_score_ppm < 1 || o.min_relative_score_ppm > 1000000)
        return dh_fail(DH_EINVAL, "min_relative_score_ppm must be in [1, 1000000] (1000000 = the default 1.0; fill the struct with dh_default_process_opts)");
    if (o.rounds < 1 || o.rounds > 8) return dh_fail(DH_EINVAL, "rounds must be in [1, 8]");
    if (o.tspace_pile < 16 || o.tspace_pile > SEG_MAX) return dh_fail(DH_EINVAL, "tspace_pile out of range");
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    ProcStats ps;
    ps.ms[0] = crop->ms_crop;
    hipEvent_t ev[8];
    for (auto &e : ev) HIPCHK(hipEventCreate(&e));
    struct EvGuard {
        hipEvent_t *e;
        ~EvGuard()
        {
            for (int i = 0; i < 8; i++) (void)hipEventDestroy(e[i]);
        }
    } evg{ev};
    auto elapsed = [&](int a, int b, float &acc) -> int {
        float t = 0;
        HIPCHK(hipEventSynchronize(ev[b]));
        HIPCHK(hipEventElapsedTime(&t, ev[a], ev[b]));
        acc += t;
        return DH_OK;
    };
    const bool tr_on = getenv("DH_TRACE") != nullptr;
    auto now_ms = [] {
        return (double)std::chrono::duration_cast<std::chrono::microseconds>(
                   std::chrono::steady_clock::now().time_since_epoch()).count() / 1e3;
    };
    double tmark = now_ms();
    auto lap = [&](const char *what) {
        const double t = now_ms();
        if (tr_on) fprintf(stderr, "[dh_process] %-28s %.2f ms\n", what, t - tmark);
        tmark = t;
    };
    DbGuard dbg;
    SetGuard sg;
    dh_insertions *res = new dh_insertions();
    struct ResGuard {
        dh_insertions *&r;
        bool ok = false;
        ~ResGuard()
        {
            if (!ok) delete r;
        }
    } rg{res};
    const int32_t np = (int32_t)crop->rec.size();
    res->rec = crop->rec;
    res->flank_of.assign((size_t)np, -1);
    res->ids_off.assign((size_t)np + 1, 0);
    {
        std::vector<int32_t> cnt((size_t)np, 0);
        for (int32_t p : crop->pile) cnt[(size_t)p]++;
        for (int32_t p = 0; p < np; p++) res->ids_off[(size_t)p + 1] = res->ids_off[(size_t)p] + cnt[(size_t)p];
        res->ids.resize((size_t)res->ids_off.back());
        std::vector<int32_t> at(res->ids_off.begin(), res->ids_off.end() - 1);
        for (size_t i = 0; i < crop->pile.size(); i++) res->ids[(size_t)at[(size_t)crop->pile[i]]++] = crop->read_id[i];
    }
    const int32_t tsp = o.tspace_pile;
    int32_t pwidth = o.width > 0 ? o.width : 30;
    if (const char *e = getenv("DH_PILE_WIDTH")) pwidth = atoi(e);  // development override
    if (pwidth < 1 || pwidth > 62) return dh_fail(DH_EINVAL, "process: width must be in [1, 62]");
    if (o.algo != 0 && o.algo != 1) return dh_fail(DH_EINVAL, "process: algo must be 0 (DH-1) or 1 (DH-2)");
    const int32_t palgo = o.algo;
    if (palgo == 1) pwidth = o.width == 32 ? 32 : 64;  // DH-2: the band (64 rows; dh_process_opts.width = 32 asks for the narrow one)

    // ---- 1. the pile-up DB: the cropped reads of every pile-up that is large enough, grouped by
    // pile-up (group = index among the active pile-ups)
    const int32_t ncr = (int32_t)crop->pile.size();
    std::vector<int32_t> cnt_of((size_t)np, 0);
    for (int32_t i = 0; i < ncr; i++) cnt_of[(size_t)crop->pile[(size_t)i]]++;
    std::vector<int32_t> active_of((size_t)np, -1);
    std::vector<int32_t> pile_of_active;             // active index -> pile-up index
    std::vector<int32_t> first_read;                 // active index -> first read in pile-up DB
    std::vector<int32_t> read_id;                    // pile-up DB read -> read id in `reads`
    std::vector<uint8_t> rkind;                      // pile-up DB read -> 0 = it may serve as reference read (alignments on every flank of its pile-up:
                                                     // selectAllowedReferenceReadIds, package.d:461-472), else 1 / 2 = on flank 0 / 1 only
    std::vector<uint8_t> rcomp;                      // pile-up DB read -> bit f: its alignment on flank f is a complement one
    std::vector<int32_t> sgroup, keep;               // per pile-up DB read: group, index in the crop DB
    for (int32_t p = 0; p < np; p++) {
        dh_insertion &r = res->rec[(size_t)p];
        r.nreads = cnt_of[(size_t)p];
        if (r.status != DH_PILE_OK) continue;
        if (r.join == 0 && r.contig_right == 0 && r.contig_left + 1 < contigs->n) r.contig_right = r.contig_left + 1;  // records made by hand before the field existed
        if (r.contig_left < 0 || r.contig_left >= contigs->n || ((r.join & DH_JOIN_EXTENSION) ? r.contig_right != -1 : (r.contig_right <= r.contig_left || r.contig_right >= contigs->n)))
            return dh_fail(DH_EINVAL, "dh_process_cropped: gap outside the contigs DB");
        if (cnt_of[(size_t)p] < o.min_reads)
            r.status = DH_PILE_TOO_SMALL;
        else if (o.max_reads > 0 && cnt_of[(size_t)p] > o.max_reads)
            return dh_fail(DH_EINVAL, "dh_process_cropped: pile-up with more than max_reads reads");
    }
    for (int32_t i = 0; i < ncr; i++) {
        const int32_t p = crop->pile[(size_t)i];
        if (res->rec[(size_t)p].status != DH_PILE_OK) continue;
        if (active_of[(size_t)p] < 0) {
            active_of[(size_t)p] = (int32_t)pile_of_active.size();
            pile_of_active.push_back(p);
            first_read.push_back((int32_t)keep.size());
        }
        sgroup.push_back(active_of[(size_t)p]);
        keep.push_back(i);
        read_id.push_back(crop->read_id[(size_t)i]);
        {
            const uint8_t kd = crop->kind.empty() ? 0 : crop->kind[(size_t)i];
            // (an extension pile-up has one flank: every read of it has its alignment there)
            rkind.push_back((res->rec[(size_t)p].join & DH_JOIN_EXTENSION) ? ((kd & 3) == 1 ? 0 : 2) : (kd & 3));
            rcomp.push_back(kd >> 2);
        }
    }
    const int32_t na = (int32_t)pile_of_active.size();
    first_read.push_back((int32_t)keep.size());
    for (int32_t a = 0; a < na; a++) {
        const int64_t n_ = first_read[(size_t)a + 1] - first_read[(size_t)a];
        int64_t lsum = 0;
        for (int32_t x = first_read[(size_t)a]; x < first_read[(size_t)a + 1]; x++)
            lsum += crop->off[(size_t)keep[(size_t)x] + 1] - crop->off[(size_t)keep[(size_t)x]];
        ps.work[0] += 1;
        ps.work[1] += n_;
        ps.work[2] += lsum;
        ps.work[3] += n_ * lsum + 2 * lsum / std::max<int64_t>(n_, 1);  // (n^2 + 2) L with L = lsum / n
    }
    HIPCHK(hipEventRecord(ev[0], st));
    if (!crop->dev) {  // cropped reads came over the host (dh_cropped_create): upload them once
        uint8_t *d_alloc = nullptr, *d_bases = nullptr;
        if (int rc = dh_alloc_bases(st, crop->off.back(), &d_alloc, &d_bases)) return rc;
        if (int rc = dh_db_adopt(ctx, d_alloc, d_bases, crop->off, std::vector<int32_t>(), &crop->dev)) {
            dh_dev_free(d_alloc);
            return rc;
        }
        crop->ctx = ctx;
        if (crop->off.back() > 0)
            HIPCHK(hipMemcpyAsync(d_bases, crop->bases.data(), (size_t)crop->off.back(), hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
        lap("cropped reads upload");
    }
    dh_db *pile = nullptr;
    {
        std::vector<int32_t> sbeg(keep.size(), 0), slen(keep.size());
        for (size_t x = 0; x < keep.size(); x++)
            slen[x] = (int32_t)(crop->off[(size_t)keep[x] + 1] - crop->off[(size_t)keep[x]]);
        if (int rc = dh_db_from_slices(ctx, crop->dev, keep, sbeg, slen, sgroup, &pile)) return rc;
        dbg.dbs.push_back(pile);
        lap("pile DB slices");
        if (o.dust)  // DBdust pileup.db; daligner ... -mdust (package.d:476-482)
            if (int rc = dh_db_dust_impl(pile)) return rc;
    }
    HIPCHK(hipEventRecord(ev[1], st));
    if (int rc = elapsed(0, 1, ps.ms[0])) return rc;
    lap("pile DB");

    std::vector<uint8_t> active_ok((size_t)na, 1);
    dh_db *T = nullptr;
    if (na > 0) {
        // ---- 2. pile-up all-vs-all: daligner -s126 -l500 -e0.7 (commandline.d:2886-2902)
        dh_align_opts ao;
        dh_default_align_opts(&ao);
        ao.tspace = tsp;
        ao.min_len = 500;
        ao.skip_self = 2;  // every unordered pair aligned once, both records emitted (as daligner does)
        // record slots per (read, strand): a read overlaps at most every other read of its pile-up
        {
            int32_t most = std::min(crop->batch_most, 252);
            for (int32_t a = 0; a < na; a++) most = std::max(most, first_read[(size_t)a + 1] - first_read[(size_t)a]);
            if (most > 252)
                return dh_fail(DH_EOVERFLOW, "process: a pile-up with more than 252 reads (set max_reads)");
            ao.max_la = most <= 60 ? 64 : (most <= 124 ? 128 : 256);
            ao.max_cand = std::min(256, 2 * ao.max_la);
        }
        ao.width = pwidth;
        ao.algo = palgo;
        // The funnel, the tile QVs, the ranking and the first consensus round read the overlaps of the reads that may serve
        // as reference read only (selectAllowedReferenceReadIds, package.d:461-472; findReferenceReadCandidates :518-568):
        // pairs of two other reads are not aligned, and of a mixed pair only the record of the allowed read is made.  A
        // pile-up without any allowed read keeps every pair (it fails later, with the status it always had).
        // DH_PILE_ALL_PAIRS=1 aligns everything (what `daligner pile.db pile.db` itself writes; tests compare the two).
        // max_partners (dh_process_opts): the B side of the wanted records is bounded as well -- the first max_partners
        // reads of the pile-up in the order allowed reads, then the others, each in pile-up order.
        // (DH-2 only: with DH-1 every pair is aligned; oracle/process.py and oracle/pile.c set the same flags.)
        if (!getenv("DH_PILE_ALL_PAIRS") && palgo == 1) {
            std::vector<uint8_t> fl((size_t)pile->n, 3);  // bit 0: records with the read as A are wanted, bit 1: it may be their B
            bool any_cut = false;
            for (int32_t a = 0; a < na; a++) {
                const int32_t r0 = first_read[(size_t)a], r1 = first_read[(size_t)a + 1];
                int32_t nallowed = 0;
                for (int32_t r = r0; r < r1; r++) nallowed += rkind[(size_t)r] == 0 ? 1 : 0;
                if (nallowed == 0) continue;
                const bool cut_b = o.max_partners > 0 && r1 - r0 > o.max_partners;
                int32_t seen_allowed = 0, seen_other = 0;
                for (int32_t r = r0; r < r1; r++) {
                    const bool al = rkind[(size_t)r] == 0;
                    const int32_t rank = al ? seen_allowed++ : nallowed + seen_other++;  // position in (allowed, then others)
                    const bool partner = !cut_b || rank < o.max_partners;
                    fl[(size_t)r] = (uint8_t)((al ? 1 : 0) | (partner ? 2 : 0));
                    any_cut = any_cut || fl[(size_t)r] != 3;
                }
            }
            if (int rc = dh_db_set_pflags(pile, any_cut ? fl.data() : nullptr)) return rc;
        }
        dh_la_set *pset = nullptr;
        HIPCHK(hipEventRecord(ev[0], st));
        // (2: the trace values stay on the device -- the tile QVs read them there, the first consensus round fetches the
        // overlaps of the reference reads only, 1 / n of them)
        // (4: with DH-2 the records stay on the device as well -- the funnel below runs there, only the overlaps of the
        // reference reads travel; DH_HOST_FUNNEL=1 keeps the host path, which is also the fall-back)
        // (below the default --min-relative-score the chains can share LAs, which are then written once per chain: the host
        // funnel inserts them; the kernel reports a pair in which that happens and the batch comes to the host as well)
        const bool want_dev_funnel = palgo == 1 && !getenv("DH_HOST_FUNNEL") && o.min_relative_score_ppm == 1000000;
        if (int rc = dh_align_db_ex(ctx, pile, pile, &ao, 0, want_dev_funnel ? 6 : 2, &pset)) return rc;
        sg.sets.push_back(pset);
        lap("pile align call");
        const int32_t npr = pile->n;
        const int32_t maxtiles = std::max(1, (pile->max_len + tsp - 1) / tsp);
        std::vector<uint8_t> qv((size_t)npr * maxtiles, 255);
        // cov = max(#allowed reference reads, 4 if pile >= 4) == pile size here (package.d:498-503)
        // (allowed reference reads = the reads that span the gap, selectAllowedReferenceReadIds :461-472)
        std::vector<int32_t> cov_of((size_t)npr, 1);
        for (int32_t a = 0; a < na; a++) {
            const int32_t r0 = first_read[(size_t)a], r1 = first_read[(size_t)a + 1];
            int32_t cov = 0;
            for (int32_t r = r0; r < r1; r++) cov += rkind[(size_t)r] == 0 ? 1 : 0;
            if (cov < 4 && r1 - r0 >= 4) cov = 4;
            for (int32_t r = r0; r < r1; r++) cov_of[(size_t)r] = std::max(cov, 1);
        }
        // ---- 3' + 4'. the funnel and the tile QVs on the device copy of the records
        bool on_dev = pset->d_la != nullptr && pset->d_la_n > 0 && pset->d_trace != nullptr;
        std::vector<int32_t> dev_first, dev_live;  // first record / live records of every pile-up read
        DevBuf<int32_t> d_first_keep;
        if (on_dev) {
            HIPCHK(hipEventRecord(ev[2], st));
            DevBuf<int32_t> d_live, d_stat, d_cov;
            DevBuf<uint8_t> d_qv;
            HIPCHK(d_first_keep.alloc((size_t)npr + 1));
            HIPCHK(d_live.alloc((size_t)npr));
            HIPCHK(d_stat.alloc(1));
            HIPCHK(d_cov.alloc(cov_of.size()));
            HIPCHK(d_qv.alloc(qv.size()));
            HIPCHK(hipMemsetAsync(d_stat.p, 0, sizeof(int32_t), st));
            HIPCHK(hipMemcpyAsync(d_cov.p, cov_of.data(), sizeof(int32_t) * cov_of.size(), hipMemcpyHostToDevice, st));
            HIPCHK(dhk_memset(st, d_qv.p, 255, qv.size()));
            dhk_pile_funnel(st, pset->d_la, pset->d_item_off, npr, pile->d_off, o.max_align_err_ppm, tsp, d_first_keep.p, d_live.p, d_stat.p);
            dhk_tile_qv(st, pset->d_la, pset->d_trace, d_first_keep.p, pile->d_off, npr, tsp, d_cov.p, maxtiles, d_qv.p);
            HIPCHK(hipGetLastError());
            dev_first.resize((size_t)npr + 1);
            dev_live.resize((size_t)npr);
            int32_t fstat = 0;
            HIPCHK(hipMemcpyAsync(qv.data(), d_qv.p, qv.size(), hipMemcpyDeviceToHost, st));
            HIPCHK(hipMemcpyAsync(dev_first.data(), d_first_keep.p, sizeof(int32_t) * dev_first.size(), hipMemcpyDeviceToHost, st));
            HIPCHK(hipMemcpyAsync(dev_live.data(), d_live.p, sizeof(int32_t) * dev_live.size(), hipMemcpyDeviceToHost, st));
            HIPCHK(hipMemcpyAsync(&fstat, d_stat.p, sizeof(int32_t), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            HIPCHK(hipEventRecord(ev[3], st));
            if (int rc = elapsed(2, 3, ps.ms[2])) return rc;
            if (getenv("DH_FUNNEL_FALLBACK")) fstat = 1;  // (tests: the fall-back below must give the same result)
            if (fstat != 0) {
                // a read or a pair beyond the kernel's capacities: the records come to the host after all (the flags the
                // kernel has set are a subset of what the host path sets: it runs on them unchanged)
                on_dev = false;
                pset->la.resize((size_t)pset->d_la_n);
                HIPCHK(hipMemcpy(pset->la.data(), pset->d_la, sizeof(dh_la) * (size_t)pset->d_la_n, hipMemcpyDeviceToHost));
                for (dh_la &la : pset->la) la.flags &= ~FLAG_IMPROPER;
                std::fill(qv.begin(), qv.end(), (uint8_t)255);
            }
            lap("funnel + tile qv (device)");
        }
        bool grouped = true;  // the symmetric wave kernel already emits grouped by A read
        if (!on_dev) {
            std::atomic<int> out_of_order{0};
            const dh_la *lp = pset->la.data();
            dh_parallel_for((int64_t)pset->la.size(), 1 << 16, [&](int64_t lo, int64_t hi) {
                for (int64_t i = std::max<int64_t>(lo, 1); i < hi; i++)
                    if (lp[i - 1].aread > lp[i].aread) {
                        out_of_order = 1;
                        break;
                    }
            });
            grouped = out_of_order.load() == 0;
        }
        if (!grouped) {  // group by aread (counting sort, stable); traces stay where they are
            std::vector<int32_t> cnt((size_t)pile->n + 1, 0);
            for (const dh_la &la : pset->la) cnt[(size_t)la.aread + 1]++;
            for (int32_t r = 0; r < pile->n; r++) cnt[(size_t)r + 1] += cnt[(size_t)r];
            LaVec tmp(pset->la.size());
            for (const dh_la &la : pset->la) tmp[(size_t)cnt[(size_t)la.aread]++] = la;
            pset->la.swap(tmp);
        }
        HIPCHK(hipEventRecord(ev[1], st));
        if (int rc = elapsed(0, 1, ps.ms[1])) return rc;
        LaVec &pl = pset->la;
        ps.counters[0] = on_dev ? pset->d_la_n : (int64_t)pl.size();
        // reads whose overlaps did not fit the per-read slots: their pile-up is skipped with a status
        // (the reference skips a failing pile-up and carries on, package.d:319-363)
        for (int32_t r : pset->ovf_reads) {
            const int32_t a = pile->h_group[(size_t)r];
            if (active_ok[(size_t)a]) {
                active_ok[(size_t)a] = 0;
                res->rec[(size_t)pile_of_active[(size_t)a]].status = DH_PILE_ALIGN_OVERFLOW;
            }
        }
        if (!pset->ovf_reads.empty())
            for (dh_la &la : pl)
                if (!active_ok[(size_t)pile->h_group[(size_t)la.aread]]) la.flags |= DH_FLAG_DISABLED;
        lap("group by aread");
        // ---- 3. the alignment funnel of computeQVs (package.d:474-516): averageErrorRate <=
        //         maxAlignmentError -> chainLocalAlignments -> isValidPileUpAlignment with
        //         allowance = trace spacing (dazzler.d:4066-4141)
        std::vector<int32_t> la_first;
        std::vector<std::vector<ChainDup>> gdups_of_funnel;
        if (on_dev) la_first = dev_first;
        if (!on_dev) {
            // the funnel of one A read is independent of the others: host threads take read groups
            // (LAs are grouped by aread; inside a group order by bread to get (A, B) pairs)
            // la_first[r] = first LA of A read r (the LAs are grouped by aread): boundaries found in parallel
            la_first.assign((size_t)pile->n + 1, 0);
            {
                const int64_t nl = (int64_t)pl.size();
                const dh_la *lp = pl.data();
                int32_t *lf = la_first.data();
                const int32_t npr_ = pile->n;
                dh_parallel_for(nl + 1, 1 << 16, [&](int64_t lo, int64_t hi) {
                    for (int64_t i = lo; i < hi; i++) {
                        const int32_t prev = i == 0 ? -1 : lp[i - 1].aread, cur = i == nl ? npr_ : lp[i].aread;
                        for (int32_t r = prev + 1; r <= cur; r++) lf[r] = (int32_t)i;
                    }
                });
            }
            const double min_rel = (double)o.min_relative_score_ppm / 1e6;
            gdups_of_funnel.assign((size_t)pile->n, {});  // LAs that alternate chains share, per A read
            auto &gdups = gdups_of_funnel;
            dh_parallel_for(pile->n, 64, [&](int64_t glo, int64_t ghi) {
                for (int64_t g = glo; g < ghi; g++) {
                    const size_t g0 = (size_t)la_first[(size_t)g], g1 = (size_t)la_first[(size_t)g + 1];
                    if (g1 <= g0) continue;
                    for (size_t i = g0; i < g1; i++) {
                        dh_la &la = pl[i];
                        if ((int64_t)la.diffs * 1000000 > (int64_t)o.max_align_err_ppm * (la.aepos - la.abpos))
                            la.flags |= DH_FLAG_DISABLED;
                    }
                    auto by_b = [](const dh_la &x, const dh_la &y) { return x.bread < y.bread; };
                    // the device hands over one bread-ordered run per strand: merge them (stable)
                    const auto gb = pl.begin() + (long)g0, ge = pl.begin() + (long)g1;
                    const auto mid = std::is_sorted_until(gb, ge, by_b);
                    if (mid != ge) {
                        if (std::is_sorted(mid, ge, by_b))
                            std::inplace_merge(gb, mid, ge, by_b);
                        else
                            std::stable_sort(gb, ge, by_b);
                    }
                    size_t p0 = g0;
                    while (p0 < g1) {
                        size_t p1 = p0;
                        while (p1 < g1 && pl[p1].bread == pl[p0].bread) p1++;
                        chain_pair(pl, p0, p1, tsp, min_rel, gdups[(size_t)g]);
                        p0 = p1;
                    }
                    for (size_t i = g0; i < g1; i++) {
                        dh_la &la = pl[i];
                        if (la.flags & DH_FLAG_DISABLED) continue;
                        const int32_t alen = (int32_t)(pile->h_off[(size_t)la.aread + 1] - pile->h_off[(size_t)la.aread]);
                        const int32_t blen = (int32_t)(pile->h_off[(size_t)la.bread + 1] - pile->h_off[(size_t)la.bread]);
                        // improper overlaps still count for the tile QVs: DASqv runs on the chained
                        // file, filterPileUpAlignments comes after it (package.d:492-512)
                        if (!valid_pileup_alignment(la, la.aread == la.bread, alen, blen, tsp))
                            la.flags |= FLAG_IMPROPER;
                    }
                }
            });
        }
        if (!on_dev) {
            // the further occurrences of LAs that alternate chains share: behind their first occurrence (same trace).
            // NOTE (record order): the reference writes every accepted chain as ONE contiguous run (composeAlignmentChain
            // per chain, then acceptedChains.sort(): chaining.d:269-312); here a shared LA's copy sits behind its first
            // occurrence and the chain's other members stay where they were, so in record order two chains may interleave
            // (START(a), START(a'), NEXT(b) ...).  Everything downstream of the funnel works per LA (tile QVs, validity,
            // ranking, the first consensus round); code that walks a chain as "START plus the NEXT records behind it"
            // (dh_chain_view, covering_member, intersect_chain) must NOT be pointed at this output.  Only reachable below
            // min_relative_score 1.0 or with equal-score chains sharing a prefix.
            size_t ndup = 0;
            for (const auto &gd : gdups_of_funnel) ndup += gd.size();
            if (ndup) {
                LaVec out;
                out.reserve(pl.size() + ndup);
                std::vector<int32_t> nf((size_t)pile->n + 1, 0);
                for (int32_t g = 0; g < pile->n; g++) {
                    nf[(size_t)g] = (int32_t)out.size();
                    auto &gd = gdups_of_funnel[(size_t)g];
                    std::stable_sort(gd.begin(), gd.end(), [](const ChainDup &x, const ChainDup &y) { return x.i < y.i; });
                    size_t d = 0;
                    for (size_t i = (size_t)la_first[(size_t)g]; i < (size_t)la_first[(size_t)g + 1]; i++) {
                        out.push_back(pl[i]);
                        for (; d < gd.size() && gd[d].i == i; d++) {
                            dh_la c = pl[i];
                            c.flags = gd[d].flags | (pl[i].flags & FLAG_IMPROPER);
                            out.push_back(c);
                        }
                    }
                }
                nf[(size_t)pile->n] = (int32_t)out.size();
                pl.swap(out);
                la_first.swap(nf);
            }
        }
        lap("filter + chain");
        // ---- 4. tile QVs on the device (LAs are sorted by aread)
        if (!on_dev) {
        HIPCHK(hipEventRecord(ev[0], st));
        {
            DevBuf<DhLa> d_las;
            DevBuf<uint16_t> d_tr;
            DevBuf<int32_t> d_first;
            DevBuf<uint8_t> d_qv;
            HIPCHK(d_las.alloc(pl.size()));
            HIPCHK(d_first.alloc(la_first.size()));
            HIPCHK(d_qv.alloc(qv.size()));
            HIPCHK(hipMemcpyAsync(d_las.p, pl.data(), sizeof(dh_la) * pl.size(), hipMemcpyHostToDevice, st));
            // the traces of the pile-up alignment are still on the device (no alignment call since)
            const uint16_t *d_trp = pset->d_trace;
            if (!d_trp) {
                HIPCHK(d_tr.alloc(pset->trace.size()));
                HIPCHK(hipMemcpyAsync(d_tr.p, pset->trace.data(), sizeof(uint16_t) * pset->trace.size(),
                                      hipMemcpyHostToDevice, st));
                d_trp = d_tr.p;
            }
            HIPCHK(hipMemcpyAsync(d_first.p, la_first.data(), sizeof(int32_t) * la_first.size(),
                                  hipMemcpyHostToDevice, st));
            HIPCHK(hipMemsetAsync(d_qv.p, 255, qv.size(), st));
            DevBuf<int32_t> d_cov;
            HIPCHK(d_cov.alloc(cov_of.size()));
            HIPCHK(hipMemcpyAsync(d_cov.p, cov_of.data(), sizeof(int32_t) * cov_of.size(), hipMemcpyHostToDevice, st));
            dhk_tile_qv(st, d_las.p, d_trp, d_first.p, pile->d_off, npr, tsp, d_cov.p, maxtiles, d_qv.p);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(qv.data(), d_qv.p, qv.size(), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
        }
        HIPCHK(hipEventRecord(ev[1], st));
        if (int rc = elapsed(0, 1, ps.ms[2])) return rc;
        // filterPileUpAlignments (properAlignmentAllowance), dazzler.d:4043-4094: after the QVs
        dh_parallel_for((int64_t)pl.size(), 1 << 16, [&](int64_t lo, int64_t hi) {
            for (int64_t i = lo; i < hi; i++) {
                dh_la &la = pl[(size_t)i];
                if (la.flags & FLAG_IMPROPER) la.flags = (la.flags & ~FLAG_IMPROPER) | DH_FLAG_DISABLED;
            }
        });
        }
        lap("tile qv");
        // ---- 5. reference read per pile-up: findReferenceReadCandidates (package.d:518-568)
        std::vector<int32_t> ref_of((size_t)na, -1);
        const double bad_fraction = (double)o.bad_fraction_ppm / 1e6;
        dh_parallel_for(na, 8, [&](int64_t alo, int64_t ahi) {
          for (int32_t a = (int32_t)alo; a < (int32_t)ahi; a++) {  // pile-ups are independent
            const int32_t r0 = first_read[(size_t)a], r1 = first_read[(size_t)a + 1];
            bool any = false;
            if (on_dev)
                for (int32_t r = r0; r < r1 && !any; r++) any = dev_live[(size_t)r] > 0;
            else
                for (int32_t i = la_first[(size_t)r0]; i < la_first[(size_t)r1]; i++)
                    if (!(pl[(size_t)i].flags & DH_FLAG_DISABLED)) any = true;
            dh_insertion &rec = res->rec[(size_t)pile_of_active[(size_t)a]];
            if (!any) {
                if (rec.status == DH_PILE_OK) rec.status = DH_PILE_EMPTY_ALIGNMENT;
                active_ok[(size_t)a] = 0;
                continue;
            }
            int64_t hist[MAXQV] = {0};
            int64_t total = 0;
            for (int32_t r = r0; r < r1; r++) {
                if (rkind[(size_t)r] != 0) continue;  // only allowed reference reads enter the histogram and the ranking
                const int32_t len = (int32_t)(pile->h_off[(size_t)r + 1] - pile->h_off[(size_t)r]);
                const int32_t nt = (len + tsp - 1) / tsp;
                for (int32_t t = 0; t < nt; t++) {
                    const int32_t q = qv[(size_t)r * maxtiles + t];
                    if (q < MAXQV) {
                        hist[q]++;
                        total++;
                    }
                }
            }
            const int64_t bad_thres = (int64_t)(bad_fraction * (double)total);
            int32_t idx = -1;
            int64_t cum = 0;
            for (int32_t x = 0; x < MAXQV; x++) {
                cum += hist[MAXQV - 1 - x];
                if (cum >= bad_thres) {
                    idx = x;
                    break;
                }
            }
            const int32_t bad_qv = MAXQV - 1 - idx;
            int32_t best = -1;
            int64_t best_nbad = 0;
            double best_mean = 0;
            for (int32_t r = r0; r < r1; r++) {
                if (rkind[(size_t)r] != 0) continue;
                const int32_t len = (int32_t)(pile->h_off[(size_t)r + 1] - pile->h_off[(size_t)r]);
                const int32_t nt = (len + tsp - 1) / tsp;
                int64_t nb = 0, sum = 0;
                for (int32_t t = 0; t < nt; t++) {
                    const int32_t q = qv[(size_t)r * maxtiles + t];
                    if (q >= bad_qv) nb++;
                    sum += q;
                }
                const double mean = nt > 0 ? (double)sum / (double)nt : 0.0;
                if (best < 0 || nb < best_nbad || (nb == best_nbad && mean < best_mean)) {
                    best = r;
                    best_nbad = nb;
                    best_mean = mean;
                }
            }
            if (best < 0) {  // no read spans the gap: "no valid reference read found" (package.d:335-343)
                if (rec.status == DH_PILE_OK) rec.status = DH_PILE_TOO_SMALL;
                active_ok[(size_t)a] = 0;
                continue;
            }
            ref_of[(size_t)a] = best;
            rec.ref_read = best - r0;
            rec.ref_read_id = read_id[(size_t)best];
          }
        });
        lap("rank reference reads");
        // ---- 6. consensus rounds.  Templates are indexed by active pile-up (group = active idx)
        std::vector<int32_t> tidx, tbeg, tlen, tgrp;
        for (int32_t a = 0; a < na; a++) {
            const int32_t r = ref_of[(size_t)a] >= 0 ? ref_of[(size_t)a] : first_read[(size_t)a];
            tidx.push_back(r);
            tbeg.push_back(0);
            tlen.push_back((int32_t)(pile->h_off[(size_t)r + 1] - pile->h_off[(size_t)r]));
            tgrp.push_back(a);
        }
        if (int rc = dh_db_from_slices(ctx, pile, tidx, tbeg, tlen, tgrp, &T)) return rc;
        dbg.dbs.push_back(T);
        {
            std::vector<int32_t> tmpl_of(on_dev ? 0 : pl.size());
            if (!on_dev)
                dh_parallel_for((int64_t)pl.size(), 1 << 16, [&](int64_t lo, int64_t hi) {
                    for (int64_t i = lo; i < hi; i++) {
                        const int32_t a = pile->h_group[(size_t)pl[(size_t)i].aread];
                        tmpl_of[(size_t)i] = (active_ok[(size_t)a] && pl[(size_t)i].aread == ref_of[(size_t)a]) ? a : -1;
                    }
                });
            HIPCHK(hipEventRecord(ev[0], st));
            dh_db *nT = nullptr;
            int64_t nseg = 0, ncell = 0;
            if (pset->d_trace_len > 0) {
                // the overlaps of the reference reads and their trace values, gathered on the device
                std::vector<size_t> tsel;
                LaVec dev_tl;                 // device funnel: the records of the reference reads, fetched now
                std::vector<int32_t> dev_tm;  // ... and their templates
                if (on_dev) {
                    std::vector<int32_t> sel, doff{0};
                    for (int32_t a = 0; a < na; a++)
                        if (active_ok[(size_t)a] && ref_of[(size_t)a] >= 0) {
                            const int32_t r = ref_of[(size_t)a];
                            sel.push_back(r);
                            doff.push_back(doff.back() + (dev_first[(size_t)r + 1] - dev_first[(size_t)r]));
                            dev_tm.insert(dev_tm.end(), (size_t)(dev_first[(size_t)r + 1] - dev_first[(size_t)r]), a);
                        }
                    dev_tl.resize((size_t)doff.back());
                    if (!sel.empty() && doff.back() > 0) {
                        DevBuf<int32_t> d_sel, d_doff;
                        DevBuf<DhLa> d_out;
                        HIPCHK(d_sel.alloc(sel.size()));
                        HIPCHK(d_doff.alloc(doff.size()));
                        HIPCHK(d_out.alloc((size_t)doff.back()));
                        HIPCHK(hipMemcpyAsync(d_sel.p, sel.data(), sizeof(int32_t) * sel.size(), hipMemcpyHostToDevice, st));
                        HIPCHK(hipMemcpyAsync(d_doff.p, doff.data(), sizeof(int32_t) * doff.size(), hipMemcpyHostToDevice, st));
                        dhk_gather_read_records(st, pset->d_la, d_first_keep.p, d_sel.p, d_doff.p, (int32_t)sel.size(), d_out.p);
                        HIPCHK(hipGetLastError());
                        HIPCHK(hipMemcpyAsync(dev_tl.data(), d_out.p, sizeof(dh_la) * dev_tl.size(), hipMemcpyDeviceToHost, st));
                        HIPCHK(hipStreamSynchronize(st));
                    }
                    for (size_t i = 0; i < dev_tl.size(); i++) tsel.push_back(i);
                } else
                    for (size_t i = 0; i < pl.size(); i++)
                        if (tmpl_of[i] >= 0) tsel.push_back(i);
                LaVec tl(tsel.size());
                std::vector<int32_t> ttm(tsel.size());
                std::vector<int64_t, PinnedAlloc<int64_t>> desc(3 * tsel.size());
                int64_t tot = 0;
                for (size_t q = 0; q < tsel.size(); q++) {
                    tl[q] = on_dev ? dev_tl[tsel[q]] : pl[tsel[q]];
                    ttm[q] = on_dev ? dev_tm[tsel[q]] : tmpl_of[tsel[q]];
                    desc[3 * q] = tl[q].toff;
                    desc[3 * q + 1] = tot;
                    desc[3 * q + 2] = tl[q].tlen;
                    if (tl[q].toff < 0 || tl[q].toff + tl[q].tlen > pset->d_trace_len)
                        return dh_fail(DH_EINVAL, "process: trace range outside the pile-up alignment's trace");
                    tl[q].toff = tot;
                    tot += tl[q].tlen;
                }
                TraceVec ttrace((size_t)tot);
                DevBuf<int64_t> d_desc;
                DevBuf<uint16_t> d_tt;
                HIPCHK(d_desc.alloc(desc.size()));
                HIPCHK(d_tt.alloc((size_t)tot));
                if (!tsel.empty()) {
                    HIPCHK(hipMemcpyAsync(d_desc.p, desc.data(), sizeof(int64_t) * desc.size(), hipMemcpyHostToDevice, st));
                    dhk_gather_ranges16(st, pset->d_trace, d_desc.p, (int32_t)tsel.size(), d_tt.p);
                    HIPCHK(hipGetLastError());
                    HIPCHK(hipMemcpyAsync(ttrace.data(), d_tt.p, sizeof(uint16_t) * (size_t)tot, hipMemcpyDeviceToHost, st));
                }
                HIPCHK(hipStreamSynchronize(st));
                if (int rc = consensus_round(ctx, T, pile, tl, ttrace, ttm, tsp, &nT, &nseg, &ncell)) return rc;
            } else if (int rc = consensus_round(ctx, T, pile, pl, pset->trace, tmpl_of, tsp, &nT, &nseg, &ncell))
                return rc;
            dbg.dbs.push_back(nT);
            T = nT;
            ps.counters[1] += nseg;
            ps.counters[2] += ncell;
            HIPCHK(hipEventRecord(ev[1], st));
            if (int rc = elapsed(0, 1, ps.ms[3])) return rc;
        }
        for (int32_t round = 1; round < o.rounds; round++) {
            dh_align_opts ro;
            dh_default_align_opts(&ro);
            ro.tspace = tsp;
            ro.min_len = 500;
            ro.max_la = 4;
            ro.max_cand = 32;
            ro.width = pwidth;
            ro.algo = palgo;
            dh_la_set *rset = nullptr;
            HIPCHK(hipEventRecord(ev[0], st));
            if (int rc = dh_align_db_ex(ctx, T, pile, &ro, 0, 0, &rset)) return rc;
            sg.sets.push_back(rset);
            HIPCHK(hipEventRecord(ev[1], st));
            if (int rc = elapsed(0, 1, ps.ms[4])) return rc;
            std::vector<int32_t> tmpl_of(rset->la.size(), -1);
            for (size_t i = 0; i < rset->la.size(); i++) {
                dh_la &la = rset->la[i];
                const int32_t a = la.aread;
                const int32_t alen = (int32_t)(T->h_off[(size_t)a + 1] - T->h_off[(size_t)a]);
                const int32_t blen = (int32_t)(pile->h_off[(size_t)la.bread + 1] - pile->h_off[(size_t)la.bread]);
                if (!valid_pileup_alignment(la, false, alen, blen, tsp)) la.flags |= DH_FLAG_DISABLED;
                if (active_ok[(size_t)a]) tmpl_of[i] = a;
            }
            HIPCHK(hipEventRecord(ev[0], st));
            dh_db *nT = nullptr;
            int64_t nseg = 0, ncell = 0;
            if (int rc = consensus_round(ctx, T, pile, rset->la, rset->trace, tmpl_of, tsp, &nT, &nseg, &ncell))
                return rc;
            dbg.dbs.push_back(nT);
            T = nT;
            ps.counters[1] += nseg;
            ps.counters[2] += ncell;
            HIPCHK(hipEventRecord(ev[1], st));
            if (int rc = elapsed(0, 1, ps.ms[3])) return rc;
        }
        lap("consensus rounds");
        // ---- 7. flank re-alignment: daligner -A -s126 -l126 contigs consensus (commandline.d:2918-2935)
        // one slice of the flank DB per flank of a pile-up (package.d:631-667 builds the DB from the croppingPositions'
        // contigs): the contig's tail for a back-seeded flank, its head for a front-seeded one
        std::vector<int32_t> fidx, fbeg, flen, fgrp, fbase((size_t)na + 1, 0);
        for (int32_t a = 0; a < na; a++) {
            const dh_insertion &rec = res->rec[(size_t)pile_of_active[(size_t)a]];
            const int32_t nf = (rec.join & DH_JOIN_EXTENSION) ? 1 : 2;
            // flank_window <= 0: the whole contigs, as the reference hands them to daligner (commandline.d:2918-2935)
            const int32_t fw = o.flank_window > 0 ? o.flank_window : INT32_MAX;
            fbase[(size_t)a] = (int32_t)fidx.size();
            for (int32_t f = 0; f < nf; f++) {
                const int32_t g = f == 0 ? rec.contig_left : rec.contig_right;
                const bool front = f == 0 ? (rec.join & DH_JOIN_FLANK0_FRONT) != 0 : (rec.join & DH_JOIN_FLANK1_BACK) == 0;
                const int32_t cl = (int32_t)(contigs->h_off[(size_t)g + 1] - contigs->h_off[(size_t)g]);
                // (a tail window starts on the trace grid of the contig: tiles, and with them the alignment, are those of the whole contig)
                const int32_t wl = front ? 0 : std::max(0, cl - std::min(cl, fw)) / tsp * tsp;
                fidx.push_back(g);
                fbeg.push_back(wl);
                flen.push_back(front ? std::min(cl, fw) : cl - wl);
                fgrp.push_back(a);
            }
        }
        fbase[(size_t)na] = (int32_t)fidx.size();
        dh_db *F = nullptr;
        if (int rc = dh_db_from_slices(ctx, contigs, fidx, fbeg, flen, fgrp, &F, true)) return rc;
        dbg.dbs.push_back(F);
        if (o.dust)  // DBdust contigs.dam; daligner -A ... -mdust -mrep (package.d:631-667)
            if (int rc = dh_db_dust_impl(F)) return rc;
        dh_align_opts fo;
        dh_default_align_opts(&fo);
        fo.tspace = tsp;
        fo.min_len = 126;
        fo.max_la = 4;
        fo.max_cand = 32;
        fo.width = pwidth;
        fo.algo = palgo;
        dh_la_set *fset = nullptr;
        HIPCHK(hipEventRecord(ev[0], st));
        if (int rc = dh_align_db_ex(ctx, F, T, &fo, 0, 0, &fset)) return rc;
        sg.sets.push_back(fset);
        HIPCHK(hipEventRecord(ev[1], st));
        if (int rc = elapsed(0, 1, ps.ms[5])) return rc;
        lap("flank align");
        // ---- 8. consensus bases to the host, insertion per pile-up
        std::vector<uint8_t> cons((size_t)std::max<int64_t>(T->total, 1));
        if (T->total > 0) HIPCHK(hipMemcpy(cons.data(), T->d_bases, (size_t)T->total, hipMemcpyDeviceToHost));
        // the flank overlaps of a pile-up: B = its consensus; records are grouped by B read or not -- index them once
        std::vector<std::vector<int32_t>> fl_of((size_t)na);
        for (size_t i = 0; i < fset->la.size(); i++)
            if (fset->la[i].bread >= 0 && fset->la[i].bread < na) fl_of[(size_t)fset->la[i].bread].push_back((int32_t)i);
        for (int32_t a = 0; a < na; a++) {
            dh_insertion &rec = res->rec[(size_t)pile_of_active[(size_t)a]];
            if (!active_ok[(size_t)a]) continue;
            const int64_t c0 = T->h_off[(size_t)a], c1 = T->h_off[(size_t)a + 1];
            rec.cons_off = (int64_t)res->bases.size();
            rec.cons_len = (int32_t)(c1 - c0);
            res->bases.insert(res->bases.end(), cons.begin() + c0, cons.begin() + c1);
            const int32_t clen = rec.cons_len;
            const int32_t nf = (rec.join & DH_JOIN_EXTENSION) ? 1 : 2;
            const bool front[2] = {(rec.join & DH_JOIN_FLANK0_FRONT) != 0, (rec.join & DH_JOIN_FLANK1_BACK) == 0};
            // the consensus has the orientation of the reference read: an overlap whose complement flag differs from the
            // reference read's alignment on that contig is disabled (package.d:669-690); of the others exactly one per
            // flank must be a proper insertion overlap (:707-745)
            const uint8_t refc = ref_of[(size_t)a] >= 0 ? rcomp[(size_t)ref_of[(size_t)a]] : 0;
            const bool refc_known = crop->comp_known && ref_of[(size_t)a] >= 0;
            const dh_la *ov[2] = {nullptr, nullptr};
            int cnt[2] = {0, 0};
            for (int32_t i : fl_of[(size_t)a]) {
                const dh_la &la = fset->la[(size_t)i];
                const int32_t f = la.aread - fbase[(size_t)a];
                if (f < 0 || f >= nf) continue;
                if (refc_known && ((la.flags & DH_FLAG_COMP) != 0) != (((refc >> f) & 1) != 0)) continue;
                const int32_t fl_len = flen[(size_t)la.aread];
                const bool proper = front[f] ? (la.abpos <= tsp && la.bepos + tsp >= clen) : (la.aepos + tsp >= fl_len && la.bbpos <= tsp);
                if (proper) {
                    ov[f] = &la;
                    cnt[f]++;
                }
            }
            if (cnt[0] != 1 || (nf == 2 && cnt[1] != 1)) {
                rec.status = DH_PILE_FLANKS_NOT_UNIQUE;
                continue;
            }
            const dh_la *L = ov[0], *R = ov[1];
            // insertionAlignment.isParallel == referenceRead.isParallel (package.d:757-773): seeds differ <=> complements equal
            if (nf == 2 && ((L->flags & DH_FLAG_COMP) == (R->flags & DH_FLAG_COMP)) != (front[0] != front[1])) {
                rec.status = DH_PILE_ORIENTATION;
                continue;
            }
            rec.left_diffs = L->diffs;
            rec.right_diffs = R ? R->diffs : 0;
            // ensureHighQualityConsensus, output.d:388-410
            bool bad_q = false;
            for (int32_t f = 0; f < nf; f++)
                if ((int64_t)ov[f]->diffs * 1000000 > (int64_t)o.max_ins_err_ppm * (ov[f]->aepos - ov[f]->abpos)) bad_q = true;
            if (bad_q) {
                rec.status = DH_PILE_MAX_INSERTION_ERROR;
                continue;
            }
            for (int32_t f = 0; f < nf; f++) {  // kept for insertions.db (dh_insertions_write_db)
                dh_la c = *ov[f];
                const int32_t shift = fbeg[(size_t)(fbase[(size_t)a] + f)];
                c.abpos += shift;
                c.aepos += shift;
                c.toff = (int64_t)res->flank_tr.size();
                res->flank_tr.insert(res->flank_tr.end(), fset->trace.begin() + ov[f]->toff, fset->trace.begin() + ov[f]->toff + ov[f]->tlen);
                if (f == 0) res->flank_of[(size_t)pile_of_active[(size_t)a]] = (int32_t)res->flank.size();
                res->flank.push_back(c);
            }
            rec.comp = (L->flags & DH_FLAG_COMP) ? 1 : 0;
            // getCroppingPosition!"contigA" (insertions.d:110-121): front seed = begin of the overlap, back seed = its end
            const int32_t sh0 = fbeg[(size_t)fbase[(size_t)a]];
            rec.left_aepos = sh0 + (front[0] ? L->abpos : L->aepos);
            // getCroppingPosition!"contigB" (:124-146) in the frame of the flank-0 overlap
            const int32_t p0 = front[0] ? L->bbpos : L->bepos;
            if (nf == 2) {
                const int32_t sh1 = fbeg[(size_t)fbase[(size_t)a] + 1];
                rec.right_abpos = sh1 + (front[1] ? R->abpos : R->aepos);
                int32_t p1 = front[1] ? R->bbpos : R->bepos;
                if ((R->flags & DH_FLAG_COMP) != (L->flags & DH_FLAG_COMP)) p1 = clen - p1;
                // walking away from flank 0: past the end of a back-seeded overlap, before the begin of a front-seeded one
                rec.ins_begin = front[0] ? p1 : p0;
                rec.ins_end = front[0] ? p0 : p1;
            } else {
                rec.right_abpos = -1;
                rec.ins_begin = front[0] ? 0 : p0;
                rec.ins_end = front[0] ? p0 : clen;
            }
            if (rec.ins_end < rec.ins_begin) rec.status = DH_PILE_NEGATIVE_INSERTION;
        }
    }
    lap("insertions");
    ps.ms[6] = ps.ms[0] + ps.ms[1] + ps.ms[2] + ps.ms[3] + ps.ms[4] + ps.ms[5];
    g_pstats = ps;
    rg.ok = true;
    *out = res;
    return DH_OK;
}

// ------------------------------------------------------------------------------------ propagate-mask
// `dentist propagate-mask` (commands/propagateMask.d:136-305): every interval of the contig mask is cut
// to the local alignments it intersects (:214-262) and carried over to the read through the trace
// points -- begin rounded down, end rounded up (:264-293, translateTracePoint base.d:185-203) -- and
// mirrored for complement alignments (:295-300); the union per read is the result (:307-313, Region
// normalisation util/region.d:776-816: sorted, intersecting or touching intervals merged, empty ones
// dropped).  Alignments are independent of each other, so they are spread over the host threads.
// out_ptr gets nreads + 1 entries; out_iv may be NULL to size; returns the number of intervals.
extern "C" int64_t dh_propagate_mask(const dh_la *las, int64_t n, const uint16_t *trace, int32_t tspace,
                                     const int64_t *mask_ptr, const int32_t *mask_iv, int32_t ncontigs,
                                     const int64_t *read_off, int32_t nreads, int64_t *out_ptr, int32_t *out_iv,
                                     int64_t cap)
{
    if ((n > 0 && (!las || !trace)) || n < 0 || !mask_ptr || !read_off || !out_ptr || tspace < 1 || ncontigs < 0 || nreads < 0)
        return dh_fail(DH_EINVAL, "dh_propagate_mask: bad argument");
    struct Iv {
        int32_t rd, b, e;
    };
    const int64_t grain = 4096, nchunks = (n + grain - 1) / grain;
    std::vector<std::vector<Iv>> found((size_t)std::max<int64_t>(nchunks, 1));
    std::atomic<int> bad{0};
    dh_parallel_for(nchunks, 1, [&](int64_t clo, int64_t chi) {
        for (int64_t c = clo; c < chi; c++) {
            std::vector<Iv> &out = found[(size_t)c];
            const int64_t i1 = std::min(n, (c + 1) * grain);
            for (int64_t i = c * grain; i < i1; i++) {
                const dh_la &l = las[i];
                if (l.aread < 0 || l.aread >= ncontigs || l.bread < 0 || l.bread >= nreads || l.tlen < 0 || l.tlen % 2 ||
                    l.tlen / 2 != (l.aepos + tspace - 1) / tspace - l.abpos / tspace) {
                    bad = 1;
                    continue;
                }
                const int64_t m0 = mask_ptr[l.aread], m1 = mask_ptr[l.aread + 1];
                if (m1 <= m0) continue;
                // first mask interval that ends after the alignment begins
                int64_t lo = m0, hi = m1;
                while (lo < hi) {
                    const int64_t mid = (lo + hi) >> 1;
                    if (mask_iv[2 * mid + 1] <= l.abpos)
                        lo = mid + 1;
                    else
                        hi = mid;
                }
                const int32_t blen = (int32_t)(read_off[l.bread + 1] - read_off[l.bread]);
                for (int64_t j = lo; j < m1 && mask_iv[2 * j] < l.aepos; j++) {
                    const int32_t ib = std::max(mask_iv[2 * j], l.abpos), ie = std::min(mask_iv[2 * j + 1], l.aepos);
                    int32_t ta, b0, b1;
                    translate_trace_point(l, trace + l.toff, tspace, ib, 0, &ta, &b0);
                    translate_trace_point(l, trace + l.toff, tspace, ie, 1, &ta, &b1);
                    if (l.flags & DH_FLAG_COMP) {
                        const int32_t x0 = blen - b1, x1 = blen - b0;
                        b0 = x0;
                        b1 = x1;
                    }
                    if (b1 > b0) out.push_back(Iv{l.bread, b0, b1});
                }
            }
        }
    });
    if (bad) return dh_fail(DH_EINVAL, "dh_propagate_mask: id out of range or trace length does not fit the A interval");
    std::vector<Iv> all;
    for (auto &v : found) all.insert(all.end(), v.begin(), v.end());
    std::sort(all.begin(), all.end(), [](const Iv &x, const Iv &y) {
        return x.rd != y.rd ? x.rd < y.rd : (x.b != y.b ? x.b < y.b : x.e < y.e);
    });
    int64_t m = 0;
    size_t at = 0;
    for (int32_t r = 0; r < nreads; r++) {
        out_ptr[r] = m;
        while (at < all.size() && all[at].rd == r) {
            int32_t b = all[at].b, e = all[at].e;
            at++;
            while (at < all.size() && all[at].rd == r && all[at].b <= e) {  // intersecting or touching
                e = std::max(e, all[at].e);
                at++;
            }
            if (out_iv && m < cap) {
                out_iv[2 * m] = b;
                out_iv[2 * m + 1] = e;
            }
            m++;
        }
    }
    out_ptr[nreads] = m;
    return m;
}

// ------------------------------------------------------------------------------------ sharded collect + process
//
// The host work of one rank between the collectives of the sharded path (dentist_amd/parallel.py): what
// `LAmerge` + `dentist collect` + `process --batch` + `merge-insertions` do through the file system in the
// reference (snakemake/Snakefile:1173-1185, 1315-1334; commands/mergeInsertions.d:60-164).  Payloads are byte
// blobs the caller hands to RCCL as they are:
//   candidates  records of 104 bytes: int32 gap, int32 read, dh_la left, dh_la right, in (gap, read) order
//   cropped     int64 k, k x {int32 pile, entry, read, len}, then the k cropped reads' bases back to back
namespace {
#pragma pack(push, 1)
struct CandRec {
    int32_t gap, read;
    dh_la L, R;
};
struct CropHead {
    int32_t pile, entry, read, len;
};
#pragma pack(pop)
static_assert(sizeof(CandRec) == 104 && sizeof(CropHead) == 16, "blob layouts");
}  // namespace

// record arrays of destroyed plans, kept for the next plan of the process: a plan of configs[2] holds 14 MB of records, and
// giving them back to the system and faulting them in again cost a rank 2 ms per step (munmap of touched pages on destroy)
static std::mutex g_plan_las_mu;
static std::vector<dh_la_vec> g_plan_las;  // at most 8 arrays of at most 64 MB
struct dh_shard_plan {
    dh_la_vec las;                 // L0 R0 L1 R1 ... of every gathered candidate, in gather order
    dh_pileups *piles = nullptr;   // after the min / max reads cut; LA indices into `las`
    std::vector<int32_t> owner;    // rank that processes each pile-up
    dh_shard_plan()
    {
        std::lock_guard<std::mutex> lk(g_plan_las_mu);
        if (!g_plan_las.empty()) {
            las = std::move(g_plan_las.back());
            g_plan_las.pop_back();
            las.clear();
        }
    }
    ~dh_shard_plan()
    {
        delete piles;
        std::lock_guard<std::mutex> lk(g_plan_las_mu);
        if (g_plan_las.size() < 8 && las.capacity() > 0 && las.capacity() * sizeof(dh_la) <= ((size_t)64 << 20)) g_plan_las.push_back(std::move(las));
    }
};

extern "C" void dh_shard_free(void *p) { free(p); }

// this rank's candidates as a blob (malloc'd; dh_shard_free).  read_shift is added to the read ids (candidates
// collected before the alignments got their whole-DB ids)
extern "C" int dh_shard_pack_candidates(const dh_pileups *cands, const dh_la *las, int64_t n, int32_t read_shift,
                                        uint8_t **out, int64_t *nbytes)
{
    if (!cands || !out || !nbytes || (n > 0 && !las)) return dh_fail(DH_EINVAL, "dh_shard_pack_candidates: bad argument");
    if (int rc = refuse_general(cands, "dh_shard_pack_candidates")) return rc;
    int64_t tot = 0;
    for (const auto &t : cands->triples) tot += (int64_t)t.size() / 3;
    CandRec *rec = (CandRec *)malloc(std::max<size_t>((size_t)tot * sizeof(CandRec), 1));
    if (!rec) return dh_fail(DH_EINVAL, "dh_shard_pack_candidates: out of memory");
    int64_t at = 0;
    for (size_t g = 0; g < cands->contig_left.size(); g++) {
        const std::vector<int32_t> &t = cands->triples[g];
        for (size_t e = 0; e + 2 < t.size(); e += 3) {
            if (t[e + 1] < 0 || t[e + 1] >= n || t[e + 2] < 0 || t[e + 2] >= n) {
                free(rec);
                return dh_fail(DH_EINVAL, "dh_shard_pack_candidates: LA index out of range");
            }
            CandRec &r = rec[at++];
            r.gap = cands->contig_left[g];
            r.read = t[e] + read_shift;
            r.L = las[t[e + 1]];
            r.R = las[t[e + 2]];
        }
    }
    *out = (uint8_t *)rec;
    *nbytes = tot * (int64_t)sizeof(CandRec);
    return DH_OK;
}

// owners by greedy bin-packing of n^2 * (mean read span between the anchors + 1 kb), largest first (ties: lower index;
// least-loaded rank, ties: lower rank); the span is taken over the entries that span the gap
static void plan_owners(dh_shard_plan *p, int32_t world)
{
    const size_t np = p->piles->contig_left.size();
    std::vector<int64_t> cost(np);
    dh_parallel_for((int64_t)np, 16, [&](int64_t glo, int64_t ghi) {
    for (size_t g = (size_t)glo; g < (size_t)ghi; g++) {
        const std::vector<int32_t> &t = p->piles->triples[g];
        const int64_t cnt = (int64_t)t.size() / 3;
        int64_t span = 0;
        int64_t nspan = 0;
        for (size_t e = 0; e + 2 < t.size(); e += 3)
            if (t[e + 1] >= 0 && t[e + 2] >= 0) {
                span += std::max<int64_t>((int64_t)p->las[(size_t)t[e + 2]].bbpos - p->las[(size_t)t[e + 1]].bepos, 0);
                nspan++;
            }
        const double mean = (double)span / (double)std::max<int64_t>(nspan, 1) + 1000.0;
        cost[g] = (int64_t)((double)(cnt * cnt) * mean);
    }
    });
    std::vector<int32_t> order(np);
    for (size_t g = 0; g < np; g++) order[g] = (int32_t)g;
    std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return cost[(size_t)a] != cost[(size_t)b] ? cost[(size_t)a] > cost[(size_t)b] : a < b; });
    std::vector<int64_t> load((size_t)world, 0);
    p->owner.assign(np, 0);
    for (int32_t g : order) {
        int32_t best = 0;
        for (int32_t r = 1; r < world; r++)
            if (load[(size_t)r] < load[(size_t)best]) best = r;
        p->owner[(size_t)g] = best;
        load[(size_t)best] += cost[(size_t)g];
    }
}

// the sharded scaffold-graph collector: all ranks' join blobs (dh_shard_read_joins, rank order = read order) -> the
// scaffold, its gap pile-ups with the extension entries (dh_scaffold_gap_pileups) -- or, sopts->only_joins, every pile-up of
// the scaffold (dh_scaffold_all_pileups) --, the min / max reads cut, owners
extern "C" int dh_shard_graph_plan_create(const uint8_t *const *blobs, const int64_t *sizes, int32_t world, int32_t ncontigs,
                                          const int32_t *input_gaps, int32_t ngaps, const dh_scaffold_opts *sopts,
                                          const dh_process_opts *opts, dh_shard_plan **out)
{
    if (!opts || !out) return dh_fail(DH_EINVAL, "dh_shard_graph_plan_create: bad argument");
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *w) {
        if (!getenv("DH_TRACE")) return;
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[graph plan] %-20s %.2f ms\n", w, std::chrono::duration<double, std::milli>(t - t0).count());
        t0 = t;
    };
    dh_shard_plan *p = new dh_shard_plan();
    dh_scaffold *sc = nullptr;
    if (int rc = dh_scaffold_from_join_blobs(blobs, sizes, world, ncontigs, input_gaps, ngaps, sopts, p->las, &sc)) {
        delete p;
        return rc;
    }
    lap("scaffold");
    dh_pileups *all = nullptr;
    int32_t skipped = 0;
    // (only_joins: every pile-up of the scaffold -- gap joins of any two contig ends, extension joins -- as one rank's
    // `dentist process` receives them; the crop, the blobs and the process stage carry a pile-up's join with it)
    int rc = sopts && sopts->only_joins ? dh_scaffold_all_pileups(sc, p->las.data(), (int64_t)p->las.size(), sopts->only_joins & 3, &all, &skipped)
                                        : dh_scaffold_gap_pileups(sc, p->las.data(), (int64_t)p->las.size(), &all, &skipped);
    dh_scaffold_destroy(sc);
    lap("gap pile-ups");
    if (!rc) rc = dh_pileups_select(all, p->las.data(), (int64_t)p->las.size(), opts, &p->piles);
    delete all;
    if (rc) {
        delete p;
        return rc;
    }
    lap("select");
    plan_owners(p, world);
    lap("owners");
    *out = p;
    return DH_OK;
}

// every rank's candidates (rank order = read order) -> the same pile-ups on every rank: entries of a gap ordered by
// read id (stable sort by gap of the concatenation), the min / max reads cut, owners by greedy bin-packing of
// n^2 * (mean read span between the anchors + 1 kb), largest first (ties: lower index; least-loaded rank, ties: lower rank)
extern "C" int dh_shard_plan_create(const uint8_t *const *blobs, const int64_t *sizes, int32_t world,
                                    const dh_process_opts *opts, dh_shard_plan **out)
{
    if (!blobs || !sizes || !opts || !out || world < 1) return dh_fail(DH_EINVAL, "dh_shard_plan_create: bad argument");
    int64_t tot = 0;
    for (int32_t r = 0; r < world; r++) {
        if (sizes[r] < 0 || sizes[r] % (int64_t)sizeof(CandRec)) return dh_fail(DH_EINVAL, "dh_shard_plan_create: blob size");
        tot += sizes[r] / (int64_t)sizeof(CandRec);
    }
    if (2 * tot >= (1ll << 31)) return dh_fail(DH_EINVAL, "dh_shard_plan_create: too many candidates");
    dh_shard_plan *p = new dh_shard_plan();
    p->las.resize((size_t)(2 * tot));
    std::vector<std::pair<int32_t, int32_t>> key((size_t)tot);  // (gap, position in the concatenation)
    std::vector<int32_t> rd((size_t)tot);
    int64_t at = 0;
    for (int32_t r = 0; r < world; r++) {
        const CandRec *rec = (const CandRec *)blobs[r];
        for (int64_t i = 0; i < sizes[r] / (int64_t)sizeof(CandRec); i++, at++) {
            // a corrupted or short collective payload is an error, not an index
            if (rec[i].gap < 0 || rec[i].read < 0 || rec[i].L.aread != rec[i].gap || rec[i].R.aread != rec[i].gap + 1 ||
                rec[i].L.bread != rec[i].R.bread || rec[i].L.tlen < 0 || rec[i].R.tlen < 0) {
                delete p;
                return dh_fail(DH_EINVAL, "dh_shard_plan_create: candidate record with inconsistent gap / read / alignment ids");
            }
            p->las[(size_t)(2 * at)] = rec[i].L;
            p->las[(size_t)(2 * at + 1)] = rec[i].R;
            key[(size_t)at] = std::make_pair(rec[i].gap, (int32_t)at);
            rd[(size_t)at] = rec[i].read;
        }
    }
    std::sort(key.begin(), key.end());  // by gap, then gather order: the stable sort by gap
    dh_pileups all;
    for (int64_t i = 0; i < tot; i++) {
        if (all.contig_left.empty() || all.contig_left.back() != key[(size_t)i].first) {
            all.contig_left.push_back(key[(size_t)i].first);
            all.triples.emplace_back();
        }
        const int32_t x = key[(size_t)i].second;
        std::vector<int32_t> &t = all.triples.back();
        t.push_back(rd[(size_t)x]);
        t.push_back(2 * x);
        t.push_back(2 * x + 1);
    }
    if (int rc = dh_pileups_select(&all, p->las.data(), (int64_t)p->las.size(), opts, &p->piles)) {
        delete p;
        return rc;
    }
    plan_owners(p, world);
    *out = p;
    return DH_OK;
}
extern "C" void dh_shard_plan_destroy(dh_shard_plan *p) { delete p; }
extern "C" const dh_la *dh_shard_plan_las(const dh_shard_plan *p) { return p ? p->las.data() : nullptr; }
extern "C" int64_t dh_shard_plan_nlas(const dh_shard_plan *p) { return p ? (int64_t)p->las.size() : 0; }
extern "C" const dh_pileups *dh_shard_plan_pileups(const dh_shard_plan *p) { return p ? p->piles : nullptr; }
extern "C" const int32_t *dh_shard_plan_owner(const dh_shard_plan *p) { return p ? p->owner.data() : nullptr; }

// the cropped reads of this rank for the owners of their pile-ups: one blob per destination rank (malloc'd as ONE block,
// blobs[r] point into it; release blobs[0] with dh_shard_free)
extern "C" int dh_shard_pack_cropped(dh_cropped *crop, const int32_t *owner, int32_t world, uint8_t **blobs, int64_t *sizes)
{
    if (!crop || !owner || !blobs || !sizes || world < 1) return dh_fail(DH_EINVAL, "dh_shard_pack_cropped: bad argument");
    const size_t nr = crop->pile.size();
    const auto T0 = std::chrono::steady_clock::now();
    auto ms_since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - T0).count(); };
    const uint8_t *bases = nr ? dh_cropped_bases(crop) : nullptr;
    const double t_bases = ms_since();
    if (nr && !bases) return DH_EHIP;
    std::vector<int64_t> cnt((size_t)world, 0), nb((size_t)world, 0);
    // `owner` has one entry per pile-up of the crop (dh_shard_plan_owner of the plan the crop was made from)
    for (size_t i = 0; i < nr; i++) {
        if (crop->pile[i] < 0 || (size_t)crop->pile[i] >= crop->rec.size())
            return dh_fail(DH_EINVAL, "dh_shard_pack_cropped: pile-up index outside the crop's records");
        if (crop->entry[i] < 0 || crop->entry[i] >= (1 << 28) || (i < crop->kind.size() && crop->kind[i] > 15))
            return dh_fail(DH_EINVAL, "dh_shard_pack_cropped: entry index or kind does not fit the blob header");
        const int32_t d = owner[crop->pile[i]];
        if (d < 0 || d >= world) return dh_fail(DH_EINVAL, "dh_shard_pack_cropped: owner out of range");
        cnt[(size_t)d]++;
        nb[(size_t)d] += crop->off[i + 1] - crop->off[i];
    }
    int64_t total = 0;
    std::vector<int64_t> start((size_t)world);
    for (int32_t r = 0; r < world; r++) {
        start[(size_t)r] = total;
        sizes[r] = 8 + cnt[(size_t)r] * (int64_t)sizeof(CropHead) + nb[(size_t)r];
        total += sizes[r];
    }
    const double t_count = ms_since();
    uint8_t *blk = (uint8_t *)malloc((size_t)std::max<int64_t>(total, 1));
    if (!blk) return dh_fail(DH_EINVAL, "dh_shard_pack_cropped: out of memory");
    std::vector<int64_t> hat((size_t)world), bat((size_t)world);
    for (int32_t r = 0; r < world; r++) {
        blobs[r] = blk + start[(size_t)r];
        memcpy(blobs[r], &cnt[(size_t)r], 8);
        hat[(size_t)r] = 8;
        bat[(size_t)r] = 8 + cnt[(size_t)r] * (int64_t)sizeof(CropHead);
    }
    // every read's place in its destination's blob, then the copies on the host threads (20 MB per rank at N = 8: one
    // thread took 6 ms, most of it page faults of the fresh block)
    std::vector<int64_t> hpos(nr), bpos(nr);
    for (size_t i = 0; i < nr; i++) {
        const int32_t d = owner[crop->pile[i]];
        hpos[i] = hat[(size_t)d];
        bpos[i] = bat[(size_t)d];
        hat[(size_t)d] += (int64_t)sizeof(CropHead);
        bat[(size_t)d] += crop->off[i + 1] - crop->off[i];
    }
    dh_parallel_for((int64_t)nr, 256, [&](int64_t lo, int64_t hi) {
        for (int64_t ii = lo; ii < hi; ii++) {
            const size_t i = (size_t)ii;
            const int32_t d = owner[crop->pile[i]];
            const int64_t len = crop->off[i + 1] - crop->off[i];
            // the entry's kind (0 spanning, 1 / 2 extension) rides in the top bits of `entry` (entries < 2^28)
            const CropHead h{crop->pile[i], (int32_t)((uint32_t)crop->entry[i] | ((uint32_t)(i < crop->kind.size() ? crop->kind[i] : 0) << 28)), crop->read_id[i], (int32_t)len};
            memcpy(blobs[d] + hpos[i], &h, sizeof(h));
            memcpy(blobs[d] + bpos[i], bases + crop->off[i], (size_t)len);
        }
    });
    if (getenv("DH_TRACE"))
        fprintf(stderr, "[pack cropped] %zu reads, %lld bytes: bases to the host %.2f, sizes %.2f, copies %.2f ms\n", nr, (long long)total, t_bases,
                t_count - t_bases, ms_since() - t_count);
    return DH_OK;
}

// what the owners received (one blob per source rank) -> the cropped pile-ups this rank processes: its pile-ups
// renumbered 0.., their reads ordered by (pile, entry); rec = the crop records of ALL pile-ups (same on every rank)
extern "C" int dh_shard_unpack_cropped(const uint8_t *const *blobs, const int64_t *sizes, int32_t world, const dh_insertion *rec,
                                       int32_t npiles, const int32_t *owner, int32_t rank, dh_cropped **out)
{
    if (!blobs || !sizes || !out || world < 1 || npiles < 0 || (npiles > 0 && (!rec || !owner)))
        return dh_fail(DH_EINVAL, "dh_shard_unpack_cropped: bad argument");
    struct Src {
        CropHead h;
        const uint8_t *b;
        uint8_t kind;
    };
    std::vector<Src> all;
    for (int32_t r = 0; r < world; r++) {
        if (sizes[r] < 8) return dh_fail(DH_EINVAL, "dh_shard_unpack_cropped: short blob");
        int64_t k;
        memcpy(&k, blobs[r], 8);
        if (k < 0 || 8 + k * (int64_t)sizeof(CropHead) > sizes[r]) return dh_fail(DH_EINVAL, "dh_shard_unpack_cropped: corrupt blob");
        const uint8_t *hb = blobs[r] + 8, *bb = hb + k * (int64_t)sizeof(CropHead);
        for (int64_t i = 0; i < k; i++) {
            Src s;
            memcpy(&s.h, hb + i * (int64_t)sizeof(CropHead), sizeof(CropHead));
            s.kind = (uint8_t)((uint32_t)s.h.entry >> 28);
            s.h.entry &= 0x0FFFFFFF;
            s.b = bb;
            if (s.h.len < 0 || bb + s.h.len > blobs[r] + sizes[r] || s.h.pile < 0 || s.h.pile >= npiles)
                return dh_fail(DH_EINVAL, "dh_shard_unpack_cropped: corrupt blob");
            bb += s.h.len;
            all.push_back(s);
        }
    }
    std::stable_sort(all.begin(), all.end(), [](const Src &a, const Src &b) {
        return a.h.pile != b.h.pile ? a.h.pile < b.h.pile : a.h.entry < b.h.entry;
    });
    std::vector<int32_t> renum((size_t)npiles, -1);
    dh_cropped *c = new dh_cropped();
    for (int32_t p = 0; p < npiles; p++)
        if (owner[p] == rank) {
            renum[(size_t)p] = (int32_t)c->rec.size();
            c->rec.push_back(rec[p]);
        }
    int64_t nbases = 0;
    for (const Src &s : all) nbases += s.h.len;
    c->bases.resize((size_t)nbases);
    int64_t at = 0;
    for (const Src &s : all) {
        if (renum[(size_t)s.h.pile] < 0) {
            delete c;
            return dh_fail(DH_EINVAL, "dh_shard_unpack_cropped: a read of a pile-up this rank does not own");
        }
        c->pile.push_back(renum[(size_t)s.h.pile]);
        c->entry.push_back(s.h.entry);
        c->read_id.push_back(s.h.read);
        c->kind.push_back(s.kind);
        memcpy(c->bases.data() + at, s.b, (size_t)s.h.len);
        at += s.h.len;
        c->off.push_back(at);
    }
    c->host_valid = true;
    *out = c;
    return DH_OK;
}

// ------------------------------------------------------------------------------------ bubbles
// getReadAlignmentsOnContigs of `resolveBubbles` (collectPileUps/pileups.d:1316-1385): the reads of a pile-up whose
// join skips contigs are mapped again, without any mask, onto just those intermediate contigs (the reference builds
// two DB subsets and spawns damapper on them, :1337-1366); chains that do not cover their contig completely within
// `allowance` (AlignmentChain.completelyCovers!"contigA", common/alignments/base.d:562-566) are disabled, ids are
// those of the full DBs again (:1373-1380).  The graph surgery around it (BubbleResolver) stays with the caller.
extern "C" int dh_remap_skipping_reads(dh_ctx *ctx, dh_db *contigs, dh_db *reads, const int32_t *contig_ids, int32_t ncontig_ids,
                                       const int32_t *read_ids, int32_t nread_ids, const dh_align_opts *opts, int32_t allowance,
                                       dh_la_set **out)
{
    if (!ctx || !contigs || !reads || !contig_ids || !read_ids || !opts || !out || ncontig_ids < 1 || nread_ids < 1 || allowance < 0)
        return dh_fail(DH_EINVAL, "dh_remap_skipping_reads: bad argument");
    auto subset = [&](dh_db *src, const int32_t *ids, int32_t n, dh_db **sub) -> int {
        std::vector<int32_t> sidx((size_t)n), sbeg((size_t)n, 0), slen((size_t)n);
        for (int32_t i = 0; i < n; i++) {
            if (ids[i] < 0 || ids[i] >= src->n || (i > 0 && ids[i] <= ids[i - 1]))
                return dh_fail(DH_EINVAL, "dh_remap_skipping_reads: ids must be ascending, distinct and inside the DB");
            sidx[(size_t)i] = ids[i];
            slen[(size_t)i] = (int32_t)(src->h_off[(size_t)ids[i] + 1] - src->h_off[(size_t)ids[i]]);
        }
        return dh_db_from_slices(ctx, src, sidx, sbeg, slen, {}, sub);  // no mask: "align without any mask"
    };
    dh_db *sa = nullptr, *sb = nullptr;
    if (int rc = subset(contigs, contig_ids, ncontig_ids, &sa)) return rc;
    if (int rc = subset(reads, read_ids, nread_ids, &sb)) {
        dh_db_destroy(sa);
        return rc;
    }
    dh_la_set *set = nullptr;
    const int rc = dh_align_db(ctx, sa, sb, opts, 1, &set);
    if (!rc) {
        // chains in file order: START, then its NEXT records (how the reference reads them, dazzler.d:1728-1758)
        LaVec &la = set->la;
        for (size_t i = 0; i < la.size();) {
            size_t j = i + 1;
            while (j < la.size() && (la[j].flags & DH_FLAG_NEXT) && !(la[j].flags & DH_FLAG_START)) j++;
            const int32_t alen = (int32_t)(sa->h_off[(size_t)la[i].aread + 1] - sa->h_off[(size_t)la[i].aread]);
            const bool covers = la[i].abpos <= allowance && la[j - 1].aepos >= alen - allowance;
            for (size_t x = i; x < j; x++) {
                if (!covers) la[x].flags |= DH_FLAG_DISABLED;
                la[x].aread = contig_ids[la[x].aread];
                la[x].bread = read_ids[la[x].bread];
            }
            i = j;
        }
        *out = set;
    }
    dh_db_destroy(sa);
    dh_db_destroy(sb);
    return rc;
}
