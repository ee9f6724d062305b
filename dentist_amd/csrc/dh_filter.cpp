// dh_filter.cpp -- the alignment filters of `dentist collect` (host only):
// source/dentist/commands/collectPileUps/filter.d:122-356, applied in the order of
// collectPileUps/package.d:130-141 to the read->contig alignments damapper produced:
//   1 LQ              averageErrorRate > maxAlignmentError                      filter.d:122-139, base.d:695-698
//   2 Improper        !isProper(properAlignmentAllowance)                       filter.d:142-161, base.d:537-556
//   3 WeaklyAnchored  <= minAnchorLength unmasked reference bases               filter.d:325-356
//   4 Contained       inside another alignment on contig AND read, same strand  filter.d:178-211
//   5 Ambiguous       reads with two alignments that overlap on the read        filter.d:238-322
//   6 Redundant       reads with an alignment that (extended by the unaligned
//                     read ends) lies inside one contig                          filter.d:164-175, base.d:562-598
// The unit of every filter is the alignment CHAIN (base.d:306-421): records linked by the START / NEXT flags damapper
// sets (dazzler.d:1728-1758) are judged together -- first.begin .. last.end, totalDiffs over coveredBases, the union
// of the members' A intervals minus the mask -- and dropped together (dh_chain_view).  Dropped alignments get
// DH_FLAG_DISABLED (the reference removes the alignments of ambiguous reads from the array; the effect on every later
// stage is the same); the counts are chains.  Per-alignment predicates run on the host thread pool.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <numeric>
#include <vector>

#include "dh_internal.h"
#include "dh_parallel.h"

namespace {
struct Ctx {
    const int64_t *coff, *roff;
    int32_t alen(const dh_la &l) const { return (int32_t)(coff[l.aread + 1] - coff[l.aread]); }
    int32_t blen(const dh_la &l) const { return (int32_t)(roff[l.bread + 1] - roff[l.bread]); }
    // toInterval!(ReadInterval, "contigB"), common/package.d:259-288: forward read coordinates
    int32_t bfwd_begin(const dh_la &l) const { return (l.flags & DH_FLAG_COMP) ? blen(l) - l.bepos : l.bbpos; }
    int32_t bfwd_end(const dh_la &l) const { return (l.flags & DH_FLAG_COMP) ? blen(l) - l.bbpos : l.bepos; }
};
}  // namespace

void dh_chain_view_build(const dh_la *las, int64_t n, dh_chain_view &v)
{
    v = dh_chain_view();
    std::atomic<int> any{0};
    dh_parallel_for(n, 1 << 16, [&](int64_t lo, int64_t hi) {
        for (int64_t i = std::max<int64_t>(lo, 1); i < hi; i++)
            if (dh_continues_chain(las[i - 1], las[i])) {
                any = 1;
                break;
            }
    });
    if (!any) return;
    v.trivial = false;
    // chain starts per run of records (counted, then placed), then one unit per chain -- all on the host threads: as one
    // serial walk this was 1.5 of the 3 ms a mapping chunk's filters took (537 k records at configs[2])
    const int64_t grain = 1 << 13, nruns = (n + grain - 1) / grain;
    std::vector<int64_t> cnt((size_t)nruns + 1, 0);
    auto is_start = [&](int64_t i) { return i == 0 || !dh_continues_chain(las[i - 1], las[i]); };
    dh_parallel_for(nruns, 1, [&](int64_t rlo, int64_t rhi) {
        for (int64_t r = rlo; r < rhi; r++) {
            int64_t c = 0;
            const int64_t i1 = std::min(n, (r + 1) * grain);
            for (int64_t i = r * grain; i < i1; i++) c += is_start(i) ? 1 : 0;
            cnt[(size_t)r + 1] = c;
        }
    });
    for (int64_t r = 0; r < nruns; r++) cnt[(size_t)r + 1] += cnt[(size_t)r];
    const int64_t nc = cnt[(size_t)nruns];
    v.first.resize((size_t)nc + 1);
    v.first[(size_t)nc] = n;
    dh_parallel_for(nruns, 1, [&](int64_t rlo, int64_t rhi) {
        for (int64_t r = rlo; r < rhi; r++) {
            int64_t at = cnt[(size_t)r];
            const int64_t i1 = std::min(n, (r + 1) * grain);
            for (int64_t i = r * grain; i < i1; i++)
                if (is_start(i)) v.first[(size_t)at++] = i;
        }
    });
    v.unit.resize((size_t)nc);
    v.covered.resize((size_t)nc);
    dh_parallel_for(nc, 4096, [&](int64_t lo, int64_t hi) {
        for (int64_t c = lo; c < hi; c++) {
            const int64_t i = v.first[(size_t)c], e = v.first[(size_t)c + 1];
            dh_la u = las[i];
            int64_t cov = u.aepos - u.abpos;
            for (int64_t j = i + 1; j < e; j++) {
                u.aepos = las[j].aepos;
                u.bepos = las[j].bepos;
                u.diffs += las[j].diffs;
                u.flags |= las[j].flags & DH_FLAG_DISABLED;  // a chain with a disabled member is disabled
                cov += las[j].aepos - las[j].abpos;
            }
            v.unit[(size_t)c] = u;
            v.covered[(size_t)c] = cov;
        }
    });
}

static int collect_filter_units(dh_la *las, int64_t n, const int64_t *contig_off, int32_t ncontigs, const int64_t *read_off,
                                int32_t nreads, const int64_t *rep_ptr, const int32_t *rep_iv, const dh_process_opts *opts,
                                int64_t *dropped6, uint8_t *read_used, const int64_t *covered, const int64_t *unmasked);

extern "C" int dh_collect_filter(dh_la *las, int64_t n, const int64_t *contig_off, int32_t ncontigs,
                                 const int64_t *read_off, int32_t nreads, const int64_t *rep_ptr, const int32_t *rep_iv,
                                 const dh_process_opts *opts, int64_t *dropped6, uint8_t *read_used)
{
    if ((n > 0 && !las) || !contig_off || !read_off || !opts || n < 0) return dh_fail(DH_EINVAL, "dh_collect_filter: bad argument");
    auto T0_ = std::chrono::steady_clock::now();
    auto LAP_ = [&](const char *w) { if (getenv("DH_TRACE_FILTER")) { auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[filter] %-24s %.2f ms\n", w, std::chrono::duration<double, std::milli>(t - T0_).count()); T0_ = t; } };
    dh_chain_view cv;
    dh_chain_view_build(las, n, cv);
    LAP_("chain view");
    if (cv.trivial)
        return collect_filter_units(las, n, contig_off, ncontigs, read_off, nreads, rep_ptr, rep_iv, opts, dropped6, read_used,
                                    nullptr, nullptr);
    // multi-record chains: the filters see one unit per chain
    const int64_t nc = (int64_t)cv.unit.size();
    std::vector<int64_t> unmasked;
    if (rep_ptr) {
        // size of (union of the members' A intervals) - mask; members are ordered along A and may overlap by a few bases
        unmasked.assign((size_t)nc, 0);
        dh_parallel_for(nc, 4096, [&](int64_t lo, int64_t hi) {
            for (int64_t c = lo; c < hi; c++) {
                int64_t sum = 0;
                int32_t done = -1;  // A positions below `done` are accounted for
                for (int64_t i = cv.first[(size_t)c]; i < cv.first[(size_t)c + 1]; i++) {
                    const dh_la &l = las[i];
                    if (l.aread < 0 || l.aread >= ncontigs) continue;  // (reported by the unit pass)
                    const int32_t b0 = std::max(l.abpos, done), e0 = l.aepos;
                    if (e0 <= b0) continue;
                    int64_t un = e0 - b0;
                    for (int64_t j = rep_ptr[l.aread]; j < rep_ptr[l.aread + 1]; j++) {
                        const int32_t b = std::max(rep_iv[2 * j], b0), e = std::min(rep_iv[2 * j + 1], e0);
                        if (e > b) un -= e - b;
                    }
                    sum += un;
                    done = e0;
                }
                unmasked[(size_t)c] = sum;
            }
        });
    } else {
        unmasked.assign((size_t)nc, 0);
        dh_parallel_for(nc, 4096, [&](int64_t lo, int64_t hi) {
            for (int64_t c = lo; c < hi; c++) {  // union of the members' A intervals
                int32_t done = -1;
                int64_t sum = 0;
                for (int64_t i = cv.first[(size_t)c]; i < cv.first[(size_t)c + 1]; i++) {
                    const int32_t b0 = std::max(las[i].abpos, done);
                    if (las[i].aepos > b0) sum += las[i].aepos - b0;
                    done = std::max(done, las[i].aepos);
                }
                unmasked[(size_t)c] = sum;
            }
        });
    }
    LAP_("unmasked");
    if (int rc = collect_filter_units(cv.unit.data(), nc, contig_off, ncontigs, read_off, nreads, rep_ptr, rep_iv, opts, dropped6,
                                      read_used, cv.covered.data(), unmasked.data()))
        return rc;
    LAP_("units");
    dh_parallel_for(nc, 4096, [&](int64_t lo, int64_t hi) {
        for (int64_t c = lo; c < hi; c++)
            if (cv.unit[(size_t)c].flags & DH_FLAG_DISABLED)
                for (int64_t i = cv.first[(size_t)c]; i < cv.first[(size_t)c + 1]; i++) las[i].flags |= DH_FLAG_DISABLED;
    });
    return DH_OK;
}

// the six filters on units (one record each: a chain's pseudo record, or the records themselves when every chain is one
// record).  covered / unmasked (optional, per unit): A bases covered by the unit's members / of their union outside the mask
static int collect_filter_units(dh_la *las, int64_t n, const int64_t *contig_off, int32_t ncontigs, const int64_t *read_off,
                                int32_t nreads, const int64_t *rep_ptr, const int32_t *rep_iv, const dh_process_opts *opts,
                                int64_t *dropped6, uint8_t *read_used, const int64_t *covered, const int64_t *unmasked_in)
{
    {
        std::atomic<int> bad{0};
        dh_parallel_for(n, 16384, [&](int64_t lo, int64_t hi) {
            for (int64_t i = lo; i < hi; i++)
                if (las[i].aread < 0 || las[i].aread >= ncontigs || las[i].bread < 0 || las[i].bread >= nreads) bad = 1;
        });
        if (bad) return dh_fail(DH_EINVAL, "dh_collect_filter: id out of range");
    }
    const dh_process_opts &o = *opts;
    const Ctx c{contig_off, read_off};
    int64_t cnt[6] = {0, 0, 0, 0, 0, 0};
    // alignments disabled by the running stage (summed over the host threads)
    std::atomic<int64_t> newly{0};
    auto stage_done = [&](int s) { cnt[s] = newly.exchange(0); };
    // 1-3: per-alignment predicates, one pass: a record is judged by the first stage it fails (as three passes, each
    // skipping what the one before disabled, would)
    {
        std::atomic<int64_t> s0{0}, s1{0}, s2{0};
        dh_parallel_for(n, 4096, [&](int64_t lo, int64_t hi) {
            int64_t l0 = 0, l1 = 0, l2 = 0;
            for (int64_t i = lo; i < hi; i++) {
                dh_la &l = las[i];
                if (l.flags & DH_FLAG_DISABLED) continue;
                if ((int64_t)l.diffs * 1000000 > (int64_t)o.max_align_err_ppm * (covered ? covered[i] : (int64_t)(l.aepos - l.abpos))) {
                    l.flags |= DH_FLAG_DISABLED;
                    l0++;
                    continue;
                }
                const bool begins = l.abpos <= o.allowance || l.bbpos <= o.allowance;
                const bool ends = l.aepos + o.allowance >= c.alen(l) || l.bepos + o.allowance >= c.blen(l);
                if (!(begins && ends)) {
                    l.flags |= DH_FLAG_DISABLED;
                    l1++;
                    continue;
                }
                int64_t unmasked = l.aepos - l.abpos;
                if (unmasked_in)
                    unmasked = unmasked_in[i];
                else if (rep_ptr)
                    for (int64_t j = rep_ptr[l.aread]; j < rep_ptr[l.aread + 1]; j++) {
                        const int32_t b = std::max(rep_iv[2 * j], l.abpos), e = std::min(rep_iv[2 * j + 1], l.aepos);
                        if (e > b) unmasked -= e - b;
                    }
                if (unmasked <= o.min_anchor) {
                    l.flags |= DH_FLAG_DISABLED;
                    l2++;
                }
            }
            s0 += l0;
            s1 += l1;
            s2 += l2;
        });
        cnt[0] = s0.load();
        cnt[1] = s1.load();
        cnt[2] = s2.load();
    }
    // Records grouped by read (what the aligner emits, and what the chunk hook of dh_map_reads hands over): the three
    // remaining stages are decisions per read -- `contained` relates alignments of one read on one contig (in the
    // sorted order of stage 4 below the alignments of a read on a contig are neighbours, and the scan leaves a1's
    // range at the first alignment that is not inside it on the contig: alignments of other reads cannot change
    // the outcome), `ambiguous` and `redundant` are per read by definition -- so the host threads take runs of reads
    // and no global regrouping (two counting sorts, five serial passes over the records) is needed.
    std::atomic<int> ungrouped{0};
    dh_parallel_for(n, 16384, [&](int64_t lo, int64_t hi) {
        for (int64_t i = std::max<int64_t>(lo, 1); i < hi; i++)
            if (las[i].bread < las[i - 1].bread) ungrouped = 1;
    });
    auto U0_ = std::chrono::steady_clock::now();
    auto ULAP_ = [&](const char *w) { if (getenv("DH_TRACE_FILTER")) { auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[filter units] %-18s %.2f ms\n", w, std::chrono::duration<double, std::milli>(t - U0_).count()); U0_ = t; } };
    if (!ungrouped) {
        const std::vector<int64_t> rs = dh_run_starts(n, [las](int64_t i) { return las[i].bread; });
        ULAP_("run starts");
        std::vector<uint8_t> used((size_t)nreads, 1);
        ULAP_("used");
        std::atomic<int64_t> c3{0}, c4{0}, c5{0};
        dh_parallel_for((int64_t)rs.size() - 1, 1024, [&](int64_t rlo, int64_t rhi) {
            int64_t l3 = 0, l4 = 0, l5 = 0;
            std::vector<int64_t> idx;
            for (int64_t run = rlo; run < rhi; run++) {
                const int64_t i0 = rs[(size_t)run], i1 = rs[(size_t)run + 1];
                const int32_t r = las[i0].bread;
                if (i1 - i0 > 1) {  // 4: contained
                    idx.resize((size_t)(i1 - i0));
                    std::iota(idx.begin(), idx.end(), i0);
                    std::stable_sort(idx.begin(), idx.end(), [&](int64_t x, int64_t y) {
                        const dh_la &p = las[x], &q = las[y];
                        if (p.aread != q.aread) return p.aread < q.aread;
                        if (p.abpos != q.abpos) return p.abpos < q.abpos;
                        if (p.bbpos != q.bbpos) return p.bbpos < q.bbpos;
                        if (p.aepos != q.aepos) return p.aepos < q.aepos;
                        return p.bepos < q.bepos;
                    });
                    for (size_t x = 0; x < idx.size(); x++) {
                        const dh_la &a1 = las[idx[x]];
                        if (a1.flags & DH_FLAG_DISABLED) continue;
                        for (size_t y = x + 1; y < idx.size(); y++) {
                            dh_la &a2 = las[idx[y]];
                            if (a2.aread != a1.aread || !(a1.abpos <= a2.abpos && a2.aepos <= a1.aepos)) break;
                            if ((a2.flags & DH_FLAG_COMP) == (a1.flags & DH_FLAG_COMP) && c.bfwd_begin(a1) <= c.bfwd_begin(a2) &&
                                c.bfwd_end(a2) <= c.bfwd_end(a1) && !(a2.flags & DH_FLAG_DISABLED)) {
                                a2.flags |= DH_FLAG_DISABLED;
                                l3++;
                            }
                        }
                    }
                }
                bool amb = false;  // 5: ambiguous
                for (int64_t x = i0; x < i1 && !amb; x++) {
                    const dh_la &p = las[x];
                    if (p.flags & DH_FLAG_DISABLED) continue;
                    for (int64_t y = x + 1; y < i1; y++) {
                        const dh_la &q = las[y];
                        if (q.flags & DH_FLAG_DISABLED) continue;
                        if (c.bfwd_begin(p) < c.bfwd_end(q) && c.bfwd_begin(q) < c.bfwd_end(p)) {
                            amb = true;
                            break;
                        }
                    }
                }
                if (amb) {
                    used[(size_t)r] = 0;
                    for (int64_t x = i0; x < i1; x++) {
                        l4 += (las[x].flags & DH_FLAG_DISABLED) ? 0 : 1;
                        las[x].flags |= DH_FLAG_DISABLED;
                    }
                }
                bool red = false;  // 6: redundant -- isFullyContained, base.d:562-598
                for (int64_t x = i0; x < i1 && !red; x++) {
                    const dh_la &p = las[x];
                    if (p.flags & DH_FLAG_DISABLED) continue;
                    if (p.bbpos > p.abpos) continue;
                    const int64_t yy = (int64_t)p.aepos + c.blen(p) - p.bepos;
                    red = yy < c.alen(p);
                }
                if (red) {
                    used[(size_t)r] = 0;
                    for (int64_t x = i0; x < i1; x++) {
                        l5 += (las[x].flags & DH_FLAG_DISABLED) ? 0 : 1;
                        las[x].flags |= DH_FLAG_DISABLED;
                    }
                }
            }
            c3 += l3;
            c4 += l4;
            c5 += l5;
        });
        cnt[3] = c3.load();
        cnt[4] = c4.load();
        cnt[5] = c5.load();
        ULAP_("per read");
        if (dropped6) memcpy(dropped6, cnt, sizeof(cnt));
        if (read_used) memcpy(read_used, used.data(), (size_t)nreads);
        return DH_OK;
    }
    // 4: contained (AlignmentChain.opCmp order, base.d:766-777; stable).  Alignments of different
    // contigs never interact, so the contigs are sorted and scanned independently on the host threads.
    std::vector<int64_t> ord((size_t)n);
    // the two ids of every alignment as dense arrays: the counting sorts below are serial passes
    std::vector<int32_t> ka((size_t)n), kb((size_t)n);
    dh_parallel_for(n, 16384, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; i++) {
            ka[(size_t)i] = las[i].aread;
            kb[(size_t)i] = las[i].bread;
        }
    });
    {
        std::vector<int64_t> cfirst((size_t)ncontigs + 1, 0);
        for (int64_t i = 0; i < n; i++) cfirst[(size_t)ka[(size_t)i] + 1]++;
        for (int32_t a = 0; a < ncontigs; a++) cfirst[(size_t)a + 1] += cfirst[(size_t)a];
        std::vector<int64_t> cur(cfirst.begin(), cfirst.end() - 1);
        for (int64_t i = 0; i < n; i++) ord[(size_t)cur[(size_t)ka[(size_t)i]]++] = i;  // stable by input order
        dh_parallel_for(ncontigs, 8, [&](int64_t clo, int64_t chi) {
            int64_t local = 0;
            for (int64_t a = clo; a < chi; a++) {
                const auto ob = ord.begin() + cfirst[(size_t)a], oe = ord.begin() + cfirst[(size_t)a + 1];
                std::stable_sort(ob, oe, [&](int64_t x, int64_t y) {
                    const dh_la &p = las[x], &q = las[y];
                    if (p.bread != q.bread) return p.bread < q.bread;
                    if (p.abpos != q.abpos) return p.abpos < q.abpos;
                    if (p.bbpos != q.bbpos) return p.bbpos < q.bbpos;
                    if (p.aepos != q.aepos) return p.aepos < q.aepos;
                    return p.bepos < q.bepos;
                });
                for (auto x = ob; x != oe; ++x) {
                    const dh_la &a1 = las[*x];
                    if (a1.flags & DH_FLAG_DISABLED) continue;
                    for (auto y = x + 1; y != oe; ++y) {
                        dh_la &a2 = las[*y];
                        if (!(a1.abpos <= a2.abpos && a2.aepos <= a1.aepos)) break;  // sliceUntil
                        if ((a2.flags & DH_FLAG_COMP) == (a1.flags & DH_FLAG_COMP) && a2.bread == a1.bread &&
                            c.bfwd_begin(a1) <= c.bfwd_begin(a2) && c.bfwd_end(a2) <= c.bfwd_end(a1) &&
                            !(a2.flags & DH_FLAG_DISABLED)) {
                            a2.flags |= DH_FLAG_DISABLED;
                            local++;
                        }
                    }
                }
            }
            newly += local;
        });
    }
    stage_done(3);
    // group by read for 5 and 6
    std::vector<int64_t> first((size_t)nreads + 1, 0), byread((size_t)n);
    for (int64_t i = 0; i < n; i++) first[(size_t)kb[(size_t)i] + 1]++;
    for (int32_t r = 0; r < nreads; r++) first[(size_t)r + 1] += first[(size_t)r];
    {
        std::vector<int64_t> cur(first.begin(), first.end() - 1);
        for (int64_t i = 0; i < n; i++) byread[(size_t)cur[(size_t)kb[(size_t)i]]++] = i;
    }
    std::vector<uint8_t> used((size_t)nreads, 1);
    // 5: ambiguous -- two enabled alignments of a read that intersect on the read
    dh_parallel_for(nreads, 2048, [&](int64_t lo, int64_t hi) {
        int64_t local = 0;
        for (int64_t r = lo; r < hi; r++) {
            bool amb = false;
            for (int64_t x = first[(size_t)r]; x < first[(size_t)r + 1] && !amb; x++) {
                const dh_la &p = las[byread[(size_t)x]];
                if (p.flags & DH_FLAG_DISABLED) continue;
                for (int64_t y = x + 1; y < first[(size_t)r + 1]; y++) {
                    const dh_la &q = las[byread[(size_t)y]];
                    if (q.flags & DH_FLAG_DISABLED) continue;
                    if (c.bfwd_begin(p) < c.bfwd_end(q) && c.bfwd_begin(q) < c.bfwd_end(p)) {
                        amb = true;
                        break;
                    }
                }
            }
            if (amb) {
                used[(size_t)r] = 0;
                for (int64_t x = first[(size_t)r]; x < first[(size_t)r + 1]; x++) {
                    dh_la &l = las[byread[(size_t)x]];
                    local += (l.flags & DH_FLAG_DISABLED) ? 0 : 1;
                    l.flags |= DH_FLAG_DISABLED;
                }
            }
        }
        newly += local;
    });
    stage_done(4);
    // 6: redundant -- isFullyContained, base.d:562-598
    dh_parallel_for(nreads, 2048, [&](int64_t lo, int64_t hi) {
        int64_t local = 0;
        for (int64_t r = lo; r < hi; r++) {
            bool red = false;
            for (int64_t x = first[(size_t)r]; x < first[(size_t)r + 1] && !red; x++) {
                const dh_la &p = las[byread[(size_t)x]];
                if (p.flags & DH_FLAG_DISABLED) continue;
                if (p.bbpos > p.abpos) continue;
                const int64_t yy = (int64_t)p.aepos + c.blen(p) - p.bepos;
                red = yy < c.alen(p);
            }
            if (red) {
                used[(size_t)r] = 0;
                for (int64_t x = first[(size_t)r]; x < first[(size_t)r + 1]; x++) {
                    dh_la &l = las[byread[(size_t)x]];
                    local += (l.flags & DH_FLAG_DISABLED) ? 0 : 1;
                    l.flags |= DH_FLAG_DISABLED;
                }
            }
        }
        newly += local;
    });
    stage_done(5);
    if (dropped6) memcpy(dropped6, cnt, sizeof(cnt));
    if (read_used) memcpy(read_used, used.data(), (size_t)nreads);
    return DH_OK;
}

// ------------------------------------------------------------------------------------ validate-regions
// `dentist validate-regions` (commands/validateRegions.d:141-203, 274-312, 325-512): a region -- a closed
// gap on the gap-closed assembly -- extended by `region_context` on both sides (clipped to the contig)
// is valid iff (a) every sliding window of `weak_coverage_window` bases inside it is spanned by at least
// `min_coverage_reads` local alignments (:423-505) and (b) at least `min_spanning_reads` alignments span
// the whole extended region (:409-420).  `las` = the reads aligned to that assembly, grouped by aread
// (the command insists on the sort, :165-168).  An alignment spans the window [w, w + W) iff
// abpos <= w and w + W <= aepos (it opened at or before w and has not closed before the window's end),
// so the count per window start is a prefix sum over +1 at abpos, -1 at aepos - W + 1.  Regions are
// independent: host threads take one each.  weak_iv (may be NULL; cap pairs) receives the weak-coverage
// mask of all regions as (contig, begin, end) triples in region order (the command's union is left to the
// caller, e.g. dh_db_set_mask); returns the number of triples or a negative error.
extern "C" int64_t dh_validate_regions(const dh_la *las, int64_t n, const int64_t *contig_off, int32_t ncontigs,
                                       const dh_region *regions, int32_t nregions, int32_t region_context,
                                       int32_t weak_coverage_window, int32_t min_coverage_reads,
                                       int32_t min_spanning_reads, dh_region_report *reports, int32_t *weak_iv,
                                       int64_t cap)
{
    if ((n > 0 && !las) || !contig_off || (nregions > 0 && (!regions || !reports)) || n < 0 || nregions < 0 ||
        region_context < 0 || weak_coverage_window < 1)
        return dh_fail(DH_EINVAL, "dh_validate_regions: bad argument");
    for (int64_t i = 0; i < n; i++) {
        if (las[i].aread < 0 || las[i].aread >= ncontigs) return dh_fail(DH_EINVAL, "dh_validate_regions: contig id out of range");
        if (i > 0 && las[i].aread < las[i - 1].aread)
            return dh_fail(DH_EINVAL, "dh_validate_regions: reads-alignment must be sorted at least by a-read ID");
    }
    std::vector<int64_t> first((size_t)ncontigs + 1, 0);
    for (int64_t i = 0; i < n; i++) first[(size_t)las[i].aread + 1]++;
    for (int32_t c = 0; c < ncontigs; c++) first[(size_t)c + 1] += first[(size_t)c];
    for (int32_t r = 0; r < nregions; r++)
        if (regions[r].contig < 0 || regions[r].contig >= ncontigs || regions[r].begin < 0 || regions[r].end < regions[r].begin)
            return dh_fail(DH_EINVAL, "dh_validate_regions: bad region");
    std::vector<std::vector<int32_t>> weak((size_t)std::max(nregions, 1));
    const int32_t W = weak_coverage_window;
    dh_parallel_for(nregions, 1, [&](int64_t rlo, int64_t rhi) {
        std::vector<int32_t> diff;
        for (int64_t r = rlo; r < rhi; r++) {
            const dh_region &rg = regions[r];
            const int32_t clen = (int32_t)(contig_off[rg.contig + 1] - contig_off[rg.contig]);
            const int32_t cb = rg.begin > region_context ? rg.begin - region_context : 0;
            const int32_t ce = std::min(rg.end + region_context, clen);
            dh_region_report &rep = reports[r];
            rep.ctx_begin = cb;
            rep.ctx_end = ce;
            rep.num_spanning_reads = 0;
            rep.weak_bp = 0;
            const int64_t l0 = first[(size_t)rg.contig], l1 = first[(size_t)rg.contig + 1];
            // (b) alignments spanning the extended region
            for (int64_t i = l0; i < l1; i++)
                if (las[i].abpos < cb && ce < las[i].aepos) rep.num_spanning_reads++;
            // (a) windows [w, w + W) for w = cb .. ce - W (window clipped to the extended region)
            const int32_t wlen = std::min(W, ce - cb), nw = (ce - cb) - wlen + 1;
            std::vector<int32_t> &wk = weak[(size_t)r];
            bool any = false;
            for (int64_t i = l0; i < l1 && !any; i++) any = las[i].abpos < ce && cb < las[i].aepos;
            if (any && nw > 0 && wlen > 0) {  // no alignment bound in sight: the command's loop does not run (:471)
                diff.assign((size_t)nw + 1, 0);
                for (int64_t i = l0; i < l1; i++) {
                    if (!(las[i].abpos < ce && cb < las[i].aepos)) continue;
                    const int32_t w0 = std::max(las[i].abpos, cb) - cb, w1 = las[i].aepos - wlen - cb;  // last spanned start
                    if (w1 < w0 || w0 >= nw) continue;
                    diff[(size_t)w0]++;
                    diff[(size_t)std::min(w1 + 1, nw)]--;
                }
                int32_t cov = 0;
                for (int32_t w = 0; w < nw; w++) {
                    cov += diff[(size_t)w];
                    if (cov >= min_coverage_reads) continue;
                    const int32_t b = cb + w, e = cb + w + wlen;
                    if (wk.empty() || wk[wk.size() - 1] < b) {
                        wk.push_back(b);
                        wk.push_back(e);
                    } else
                        wk[wk.size() - 1] = e;
                }
                for (size_t x = 0; x + 1 < wk.size(); x += 2) rep.weak_bp += wk[x + 1] - wk[x];
            }
            rep.is_valid = rep.num_spanning_reads >= min_spanning_reads && rep.weak_bp == 0;
        }
    });
    int64_t m = 0;
    for (int32_t r = 0; r < nregions; r++)
        for (size_t x = 0; x + 1 < weak[(size_t)r].size(); x += 2) {
            if (weak_iv && m < cap) {
                weak_iv[3 * m] = regions[r].contig;
                weak_iv[3 * m + 1] = weak[(size_t)r][x];
                weak_iv[3 * m + 2] = weak[(size_t)r][x + 1];
            }
            m++;
        }
    return m;
}
