/*
 * pile.c -- oracle restatement of `dentist collect` (spanning reads only) and of the per-pile-up
 * call sequence of `dentist process`, in C with an OpenMP loop over pile-ups (the reference runs
 * `foreach (pileUp; parallel(pileUps))`, source/dentist/commands/processPileUps/package.d:153-154).
 *
 * TEST INFRASTRUCTURE ONLY (see dh_oracle.h): the checker of the parity tests at sizes where the
 * Python driver (oracle/process.py, same algorithm, kept as an independent cross-check) is too slow,
 * and the timed CPU baseline of bench.py ("port": a CPU restatement, NOT the reference binaries).
 *
 * Follows processPileUps/package.d:283-374 (crop -> pile-up alignment -> filter -> QV -> reference
 * read -> consensus -> flank re-alignment -> insertion), the cropping arithmetic of
 * processPileUps/cropper.d:446-550 (+ support patches :224-262, :363-378), chainLocalAlignments
 * (common/alignments/chaining.d:122-334, predicates :434-475), isValidPileUpAlignment
 * (dazzler.d:4108-4141), the reference-read ranking (package.d:518-568), the splice coordinates
 * of common/insertions.d:110-146 and the quality gate of commands/output.d:388-410.
 * The aligner and consensus arithmetic it drives (align.c, consensus.c) is PARITY UNPINNED.
 */
#include "dh_oracle.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>

static int g_inner_threads = 1; /* threads of a pile-up's all-vs-all (oz_process_piles sets it) */
#endif

void oz_default_process_opts(oz_process_opts *o)
{
    memset(o, 0, sizeof(*o));
    o->ts_map = 100;
    o->allowance = 100;
    o->min_anchor = 500;
    o->min_reads = 3;
    o->max_reads = 60;
    o->ts_pile = 126;
    o->rounds = 3;
    o->flank_window = 20000;
    o->max_align_err_ppm = 300000;
    o->max_ins_err_ppm = 100000;
    o->bad_fraction_ppm = 80000;
    o->width = 30;
    o->dust = 1;
    o->min_relative_score_ppm = 1000000;
}

/* ------------------------------------------------------------------ collect ------------- */

typedef struct {
    int64_t key;
    int32_t e;
} key_ent;
static int key_cmp(const void *x, const void *y)
{
    const key_ent *p = (const key_ent *)x, *q = (const key_ent *)y;
    if (p->key != q->key) return p->key < q->key ? -1 : 1;
    return p->e < q->e ? -1 : (p->e > q->e ? 1 : 0);
}
static int i32_cmp_(const void *x, const void *y)
{
    const int32_t p = *(const int32_t *)x, q = *(const int32_t *)y;
    return p < q ? -1 : (p > q ? 1 : 0);
}

/* Every read enters the pile-up of a gap once, with its qualifying LA pair of the longest anchors;
 * fewer than min_reads -> no pile-up; more than max_reads -> the max_reads entries with the
 * lowest error rate of their anchoring LAs (ties: lower read id), in read order. */
int oz_collect_spanning(const oz_la *las, int64_t n, const oz_db *contigs, const oz_process_opts *o,
                        int32_t **gap_out, int32_t **count_out, int32_t **triples_out, int32_t *npiles)
{
    const int32_t nc = contigs->n;
    int32_t nreads = 0;
    for (int64_t i = 0; i < n; i++)
        if (las[i].bread + 1 > nreads) nreads = las[i].bread + 1;
    int64_t *first = (int64_t *)calloc((size_t)nreads + 2, sizeof(int64_t));
    int64_t *order = (int64_t *)malloc((size_t)(n ? n : 1) * sizeof(int64_t));
    for (int64_t i = 0; i < n; i++) first[las[i].bread + 1]++;
    for (int32_t r = 0; r < nreads; r++) first[r + 1] += first[r];
    {
        int64_t *cur = (int64_t *)malloc(((size_t)nreads + 1) * sizeof(int64_t));
        memcpy(cur, first, ((size_t)nreads + 1) * sizeof(int64_t));
        for (int64_t i = 0; i < n; i++) order[cur[las[i].bread]++] = i;
        free(cur);
    }
    /* per gap: growing list of (read, iL, iR) */
    int32_t **lst = (int32_t **)calloc((size_t)nc + 1, sizeof(int32_t *));
    int32_t *cnt = (int32_t *)calloc((size_t)nc + 1, sizeof(int32_t));
    int32_t *cap = (int32_t *)calloc((size_t)nc + 1, sizeof(int32_t));
    for (int32_t rd = 0; rd < nreads; rd++) {
        const int64_t *idx = order + first[rd], c = first[rd + 1] - first[rd];
        if (c < 2) continue;
        struct {
            int32_t g;
            int64_t anchors, iL, iR;
        } rb[64];
        int nrb = 0;
        for (int64_t x = 0; x < c; x++) {
            const oz_la *L = &las[idx[x]];
            const int32_t g = L->aread;
            if ((L->flags & OZ_FLAG_DISABLED) || g < 0 || g + 1 >= nc) continue;
            const int64_t cl = contigs->off[g + 1] - contigs->off[g];
            if (L->aepos + o->allowance < cl || L->aepos - L->abpos < o->min_anchor) continue;
            for (int64_t y = 0; y < c; y++) {
                const oz_la *R = &las[idx[y]];
                if ((R->flags & OZ_FLAG_DISABLED) || R->aread != g + 1 || (R->flags & 1u) != (L->flags & 1u)) continue;
                if (R->abpos > o->allowance || R->aepos - R->abpos < o->min_anchor) continue;
                if (R->bbpos + o->allowance < L->bepos - o->allowance) continue; /* right part follows the left part */
                const int64_t anchors = (int64_t)(L->aepos - L->abpos) + (R->aepos - R->abpos);
                int k = 0;
                while (k < nrb && rb[k].g != g) k++;
                if (k == nrb) {
                    if (nrb == 64) continue;
                    rb[nrb].g = g;
                    rb[nrb].anchors = -1;
                    nrb++;
                }
                if (anchors > rb[k].anchors) {
                    rb[k].anchors = anchors;
                    rb[k].iL = idx[x];
                    rb[k].iR = idx[y];
                }
            }
        }
        for (int k = 0; k < nrb; k++) {
            const int32_t g = rb[k].g;
            if (cnt[g] == cap[g]) {
                cap[g] = cap[g] ? cap[g] * 2 : 16;
                lst[g] = (int32_t *)realloc(lst[g], (size_t)cap[g] * 3 * sizeof(int32_t));
            }
            lst[g][3 * cnt[g]] = rd;
            lst[g][3 * cnt[g] + 1] = (int32_t)rb[k].iL;
            lst[g][3 * cnt[g] + 2] = (int32_t)rb[k].iR;
            cnt[g]++;
        }
    }
    int32_t np = 0;
    int64_t ntri = 0;
    for (int32_t g = 0; g < nc; g++)
        if (cnt[g] >= o->min_reads) {
            np++;
            ntri += (o->max_reads <= 0 || cnt[g] < o->max_reads) ? cnt[g] : o->max_reads; /* max_reads 0: no cap */
        }
    int32_t *gap = (int32_t *)malloc((size_t)(np ? np : 1) * sizeof(int32_t));
    int32_t *count = (int32_t *)malloc((size_t)(np ? np : 1) * sizeof(int32_t));
    int32_t *tri = (int32_t *)malloc((size_t)(ntri ? ntri : 1) * 3 * sizeof(int32_t));
    int32_t p = 0;
    int64_t at = 0;
    for (int32_t g = 0; g < nc; g++) {
        if (cnt[g] < o->min_reads) continue;
        gap[p] = g;
        if (o->max_reads <= 0 || cnt[g] <= o->max_reads) {
            memcpy(tri + 3 * at, lst[g], (size_t)cnt[g] * 3 * sizeof(int32_t));
            count[p] = cnt[g];
        } else {
            key_ent *ke = (key_ent *)malloc((size_t)cnt[g] * sizeof(key_ent));
            for (int32_t e = 0; e < cnt[g]; e++) {
                const oz_la *L = &las[lst[g][3 * e + 1]], *R = &las[lst[g][3 * e + 2]];
                int64_t len = (int64_t)(L->aepos - L->abpos) + (R->aepos - R->abpos);
                if (len < 1) len = 1;
                ke[e].key = ((int64_t)L->diffs + R->diffs) * 1000000 / len;
                ke[e].e = e;
            }
            qsort(ke, (size_t)cnt[g], sizeof(key_ent), key_cmp);
            int32_t *keep = (int32_t *)malloc((size_t)o->max_reads * sizeof(int32_t));
            for (int32_t x = 0; x < o->max_reads; x++) keep[x] = ke[x].e;
            qsort(keep, (size_t)o->max_reads, sizeof(int32_t), i32_cmp_);
            for (int32_t x = 0; x < o->max_reads; x++) memcpy(tri + 3 * (at + x), lst[g] + 3 * keep[x], 3 * sizeof(int32_t));
            count[p] = o->max_reads;
            free(keep);
            free(ke);
        }
        at += count[p];
        p++;
    }
    for (int32_t g = 0; g <= nc; g++) free(lst[g]);
    free(lst);
    free(cnt);
    free(cap);
    free(first);
    free(order);
    *gap_out = gap;
    *count_out = count;
    *triples_out = tri;
    *npiles = np;
    return 0;
}

/* ------------------------------------------------------------------ crop ----------------- */

static int32_t ceil_to(int32_t x, int32_t m) { return (x + m - 1) / m * m; }

/* getCommonTracePoint, cropper.d:446-500, with an empty repeat mask */
static int32_t common_trace_point(int32_t lo, int32_t hi, int32_t contig_len, int32_t ts, int seed_front)
{
    if (lo >= hi) return -1;
    const int32_t tp_min = ceil_to(lo, ts), tp_sup = ceil_to(hi, ts);
    const int32_t ncand = (tp_sup - tp_min) / ts + (tp_sup > contig_len ? 1 : 0);
    for (int32_t x = 0; x < ncand; x++) {
        const int32_t y = seed_front ? ncand - 1 - x : x;
        const int32_t c = (y < (tp_sup - tp_min) / ts) ? tp_min + y * ts : contig_len;
        if ((lo <= c && c < hi) || c == hi) return c;
    }
    return -1;
}

typedef struct {
    uint8_t *bases;
    int64_t *off;
    int32_t n;
} seq_list;

static void seq_list_free(seq_list *s)
{
    free(s->bases);
    free(s->off);
    s->bases = NULL;
    s->off = NULL;
}

/* ------------------------------------------------------------------ chain ---------------- */

static int32_t la_score(const oz_la *x) { return ((x->aepos - x->abpos) + (x->bepos - x->bbpos)) / 2; }
static int chainable(const oz_la *x, const oz_la *y)
{
    const int32_t max_indel = 1000, max_gap = 10000;
    if ((x->flags & 1u) != (y->flags & 1u)) return 0;
    const int32_t ga = y->abpos - x->aepos, gb = y->bbpos - x->bepos;
    if (!(x->abpos < y->abpos && x->bbpos < y->bbpos)) return 0;
    const int32_t ind = ga - gb < 0 ? gb - ga : ga - gb;
    const int32_t aga = ga < 0 ? -ga : ga, agb = gb < 0 ? -gb : gb;
    if (ind > max_indel || (aga > agb ? aga : agb) > max_gap) return 0;
    const int32_t la1 = x->aepos - x->abpos, la2 = y->aepos - y->abpos;
    const int32_t lb1 = x->bepos - x->bbpos, lb2 = y->bepos - y->bbpos;
    const int32_t mla = la1 < la2 ? la1 : la2, mlb = lb1 < lb2 ? lb1 : lb2;
    return (ga < 0 ? -ga : 0) <= 0.3 * mla && (gb < 0 ? -gb : 0) <= 0.3 * mlb;
}
static int32_t chain_score(const oz_la *x, const oz_la *y)
{
    const int32_t ga = y->abpos - x->aepos, gb = y->bbpos - x->bepos;
    const int32_t ind = ga - gb < 0 ? gb - ga : ga - gb;
    const int32_t aga = ga < 0 ? -ga : ga, agb = gb < 0 ? -gb : gb;
    return ind + (aga > agb ? aga : agb) / 10 - la_score(y);
}

typedef struct {
    int32_t abpos, bbpos;
    int64_t i;
} ord_ent;
static int ord_cmp(const void *x, const void *y)
{
    const ord_ent *p = (const ord_ent *)x, *q = (const ord_ent *)y;
    if (p->abpos != q->abpos) return p->abpos < q->abpos ? -1 : 1;
    if (p->bbpos != q->bbpos) return p->bbpos < q->bbpos ? -1 : 1;
    return p->i < q->i ? -1 : (p->i > q->i ? 1 : 0);
}
typedef struct {
    int32_t dist, v;
} end_ent;
static int end_cmp(const void *x, const void *y)
{
    const end_ent *p = (const end_ent *)x, *q = (const end_ent *)y;
    if (p->dist != q->dist) return p->dist < q->dist ? -1 : 1;
    return p->v < q->v ? -1 : (p->v > q->v ? 1 : 0);
}

/* a second (third ...) occurrence of record i: an LA that an ALTERNATE chain shares with a better chain of the pair is
 * written once per chain (the chains are written one after the other, dazzler.d:2050-2085) */
typedef struct {
    int64_t i;
    uint32_t flags;
} dup_ent;
typedef struct {
    dup_ent *d;
    int64_t n, cap;
} dup_list;
static void dup_push(dup_list *l, int64_t i, uint32_t flags)
{
    if (l->n == l->cap) {
        l->cap = l->cap ? 2 * l->cap : 16;
        l->d = (dup_ent *)realloc(l->d, (size_t)l->cap * sizeof(dup_ent));
    }
    l->d[l->n].i = i;
    l->d[l->n].flags = flags;
    l->n++;
}
static int dup_cmp(const void *x, const void *y)
{
    const dup_ent *p = (const dup_ent *)x, *q = (const dup_ent *)y;
    return p->i < q->i ? -1 : (p->i > q->i ? 1 : 0);
}
typedef struct {
    int32_t end, alt, score;
} sel_ent;

/* buildAlignmentChains (common/alignments/chaining.d:151-312) for the enabled LAs of one (A, B) pair [first, last):
 *  - the LAs are split into the connected components of the undirected chainability relation (:182);
 *  - per component a shortest-path problem rates the chains (:227-233; the relaxations run over the LAs ordered by
 *    (abpos, bbpos, index) -- a topological order; no edge joins two components, so one pass serves all of them);
 *  - per component the end nodes within effectiveMinScore of the component's best chain are taken from best to worst
 *    (:236-266): a node that already lies on a taken chain is no end node; a chain that runs into nodes of a better
 *    chain is an ALTERNATE chain (it shares that chain's prefix) -- it is composed of its WHOLE path (:269-285), so the
 *    shared LAs are written twice;
 *  - the chains within effectiveMinScore = max(minScore, minRelativeScore * best) of the pair's best chain are accepted
 *    (:305-312).
 * First LA of a chain: START (+ BEST unless the chain is an alternate chain, dazzler.d:2063-2068), the others NEXT; LAs on
 * no accepted chain: DISABLED; further occurrences of an LA go to `dups`.
 * Ties (equal distances of two end nodes, of two predecessors): the lower position in the (abpos, bbpos, index) order
 * first -- the reference sorts the end nodes with an unstable sort (:240-245) and relaxes in the order of a depth-first
 * topological sort (util/graphalgo.d:926-960), neither of which is a property of the data. */
static void chain_pair(oz_la *la, int64_t first, int64_t last, int32_t min_score, double min_rel, dup_list *dups)
{
    const uint32_t cmask = OZ_FLAG_START | OZ_FLAG_NEXT | OZ_FLAG_BEST;
    int32_t n = 0;
    for (int64_t i = first; i < last; i++)
        if (!(la[i].flags & OZ_FLAG_DISABLED)) n++;
    if (n == 0) return;
    ord_ent *ord = (ord_ent *)malloc((size_t)n * sizeof(ord_ent));
    n = 0;
    for (int64_t i = first; i < last; i++)
        if (!(la[i].flags & OZ_FLAG_DISABLED)) {
            ord[n].abpos = la[i].abpos;
            ord[n].bbpos = la[i].bbpos;
            ord[n].i = i;
            n++;
        }
    qsort(ord, (size_t)n, sizeof(ord_ent), ord_cmp);
    int32_t *dist = (int32_t *)malloc((size_t)n * sizeof(int32_t)), *pred = (int32_t *)malloc((size_t)n * sizeof(int32_t));
    int32_t *comp = (int32_t *)malloc((size_t)n * sizeof(int32_t));  /* label = a member of the component */
    for (int32_t v = 0; v < n; v++) {
        dist[v] = -la_score(&la[ord[v].i]);
        pred[v] = -1;
        comp[v] = v;
    }
    for (int32_t u = 0; u < n; u++)
        for (int32_t v = u + 1; v < n; v++)
            if (chainable(&la[ord[u].i], &la[ord[v].i])) {  /* (v before u is impossible: abpos ascends) */
                const int32_t d = dist[u] + chain_score(&la[ord[u].i], &la[ord[v].i]);
                if (dist[v] > d) {
                    dist[v] = d;
                    pred[v] = u;
                }
                const int32_t cu = comp[u], cv = comp[v];
                if (cu != cv)
                    for (int32_t w = 0; w < n; w++)
                        if (comp[w] == cv) comp[w] = cu;
            }
    /* components in the order of their smallest record index (util/graphalgo.d:43-66) */
    int64_t *cmin = (int64_t *)malloc((size_t)n * sizeof(int64_t));
    int32_t *cord = (int32_t *)malloc((size_t)n * sizeof(int32_t)), nc = 0;
    for (int32_t v = 0; v < n; v++) cmin[v] = -1;
    for (int32_t v = 0; v < n; v++)
        if (cmin[comp[v]] < 0 || ord[v].i < cmin[comp[v]]) cmin[comp[v]] = ord[v].i;
    for (int32_t v = 0; v < n; v++)
        if (cmin[v] >= 0) cord[nc++] = v;
    for (int32_t x = 1; x < nc; x++) {
        const int32_t c = cord[x];
        int32_t y = x - 1;
        while (y >= 0 && cmin[cord[y]] > cmin[c]) {
            cord[y + 1] = cord[y];
            y--;
        }
        cord[y + 1] = c;
    }
    end_ent *ends = (end_ent *)malloc((size_t)n * sizeof(end_ent));
    uint8_t *forbidden = (uint8_t *)calloc((size_t)n, 1);
    sel_ent *sel = (sel_ent *)malloc((size_t)n * sizeof(sel_ent));
    int32_t nsel = 0;
    for (int32_t x = 0; x < nc; x++) {
        int32_t ne = 0;
        for (int32_t v = 0; v < n; v++)
            if (comp[v] == cord[x]) {
                ends[ne].dist = dist[v];
                ends[ne].v = v;
                ne++;
            }
        qsort(ends, (size_t)ne, sizeof(end_ent), end_cmp);
        const int32_t cbest = -ends[0].dist;
        const double cthr_d = (double)min_score > min_rel * cbest ? (double)min_score : min_rel * cbest;
        const int32_t cthr = (int32_t)cthr_d;
        for (int32_t y = 0; y < ne; y++) {
            const int32_t e = ends[y].v;
            if (forbidden[e] || -dist[e] < cthr) continue;
            int32_t alt = 0;
            for (int32_t v = e; v >= 0; v = pred[v]) {
                if (forbidden[v]) alt = 1;
                forbidden[v] = 1;
            }
            sel[nsel].end = e;
            sel[nsel].alt = alt;
            sel[nsel].score = -dist[e];
            nsel++;
        }
    }
    int32_t best = 0;
    for (int32_t x = 0; x < nsel; x++)
        if (x == 0 || sel[x].score > best) best = sel[x].score;
    const double thr_d = (double)min_score > min_rel * best ? (double)min_score : min_rel * best;
    const int32_t thr = (int32_t)thr_d;
    uint8_t *occ = (uint8_t *)calloc((size_t)n, 1);
    int32_t *path = (int32_t *)malloc((size_t)n * sizeof(int32_t));
    for (int32_t x = 0; x < nsel; x++) {
        if (sel[x].score < thr) continue;
        int32_t np = 0;
        for (int32_t v = sel[x].end; v >= 0; v = pred[v]) path[np++] = v;
        for (int32_t k = 0; k < np; k++) {
            const int32_t v = path[np - 1 - k];
            oz_la *l = &la[ord[v].i];
            const uint32_t f = k == 0 ? (OZ_FLAG_START | (sel[x].alt ? 0u : OZ_FLAG_BEST)) : OZ_FLAG_NEXT;
            if (!occ[v]) {
                l->flags = (l->flags & ~cmask) | f;
                occ[v] = 1;
            } else
                dup_push(dups, ord[v].i, (l->flags & ~cmask) | f);
        }
    }
    for (int32_t v = 0; v < n; v++)
        if (!occ[v]) la[ord[v].i].flags |= OZ_FLAG_DISABLED;
    free(path);
    free(occ);
    free(sel);
    free(forbidden);
    free(ends);
    free(cord);
    free(cmin);
    free(comp);
    free(dist);
    free(pred);
    free(ord);
}

/* chainLocalAlignments over a set sorted by (aread, bread): the pairs one after the other; the further occurrences of
 * LAs that alternate chains share are inserted behind their first occurrence (same trace) */
void oz_chain_set(oz_la_set *s, int32_t min_score, int32_t min_rel_ppm)
{
    dup_list dups = {NULL, 0, 0};
    const double rel = (double)min_rel_ppm / 1e6;
    for (int64_t p0 = 0; p0 < s->n;) {
        int64_t p1 = p0;
        while (p1 < s->n && s->la[p1].aread == s->la[p0].aread && s->la[p1].bread == s->la[p0].bread) p1++;
        chain_pair(s->la, p0, p1, min_score, rel, &dups);
        p0 = p1;
    }
    if (dups.n) {
        qsort(dups.d, (size_t)dups.n, sizeof(dup_ent), dup_cmp);  /* (pairs come in order: stable enough -- equal i keep any order) */
        oz_la *out = (oz_la *)malloc((size_t)(s->n + dups.n) * sizeof(oz_la));
        int64_t w = 0, d = 0;
        for (int64_t i = 0; i < s->n; i++) {
            out[w++] = s->la[i];
            while (d < dups.n && dups.d[d].i == i) {
                out[w] = s->la[i];
                out[w].flags = dups.d[d].flags;
                w++;
                d++;
            }
        }
        free(s->la);
        s->la = out;
        s->n = w;
        s->cap = w;
    }
    free(dups.d);
}

/* the same on an array (tests): returns the number of records of *out (malloc'd, oz_free) */
int64_t oz_chain_las(const oz_la *las, int64_t n, int32_t min_score, int32_t min_rel_ppm, oz_la **out)
{
    oz_la_set s;
    oz_la_set_init(&s);
    s.la = (oz_la *)malloc((size_t)(n > 0 ? n : 1) * sizeof(oz_la));
    memcpy(s.la, las, (size_t)n * sizeof(oz_la));
    s.n = s.cap = n;
    oz_chain_set(&s, min_score, min_rel_ppm);
    *out = s.la;
    return s.n;
}

/* ------------------------------------------------------------------ one pile-up ---------- */

static void set_opts(oz_opts *a, int32_t tspace, int32_t min_len, int32_t skip_self, int32_t max_la,
                     int32_t max_cand, int32_t width, int32_t algo)
{
    oz_default_opts(a);
    a->tspace = tspace;
    a->min_len = min_len;
    a->skip_self = skip_self;
    a->max_la = max_la;
    a->max_cand = max_cand;
    a->width = algo == 1 ? (width == 32 ? 32 : 64) : width; /* DH-2: the band, 64 rows unless 32 are asked for */
    a->algo = algo;
}

static void process_one(const oz_db *contigs, const oz_db *reads, const oz_la *las, const uint16_t *trace,
                        int32_t g, const int32_t *tri, int32_t ne, const oz_process_opts *o, oz_insertion *r,
                        uint8_t **cons_out)
{
    memset(r, 0, sizeof(*r));
    r->gap = g;
    r->ref_idx = r->ref_read_id = -1;
    r->crop_left = r->crop_right = -1;
    *cons_out = NULL;
    const int32_t tsm = o->ts_map, tsp = o->ts_pile;
    int32_t llo = 0, lhi = INT32_MAX, rlo = 0, rhi = INT32_MAX;
    for (int32_t e = 0; e < ne; e++) {
        const oz_la *L = &las[tri[3 * e + 1]], *R = &las[tri[3 * e + 2]];
        if (L->abpos > llo) llo = L->abpos;
        if (L->aepos < lhi) lhi = L->aepos;
        if (R->abpos > rlo) rlo = R->abpos;
        if (R->aepos < rhi) rhi = R->aepos;
    }
    const uint8_t *cl = contigs->bases + contigs->off[g], *cr = contigs->bases + contigs->off[g + 1];
    const int32_t cll = (int32_t)(contigs->off[g + 1] - contigs->off[g]);
    const int32_t clr = (int32_t)(contigs->off[g + 2] - contigs->off[g + 1]);
    const int32_t cropL = common_trace_point(llo, lhi, cll, tsm, 0);
    const int32_t cropR = common_trace_point(rlo, rhi, clr, tsm, 1);
    r->crop_left = cropL;
    r->crop_right = cropR;
    if (cropL < 0 || cropR < 0) {
        r->status = 1;
        return;
    }
    /* fetchSupportPatches, cropper.d:224-262 */
    int32_t lp0 = 0, lp1 = 0, rp0 = 0, rp1 = 0;
    if (cll - cropL < o->min_anchor) {
        lp0 = cll - o->min_anchor > 0 ? cll - o->min_anchor : 0;
        lp1 = cropL;
    }
    if (cropR < o->min_anchor) {
        rp0 = cropR;
        rp1 = clr < o->min_anchor ? clr : o->min_anchor;
    }
    seq_list pile = {NULL, NULL, 0};
    pile.off = (int64_t *)calloc((size_t)ne + 1, sizeof(int64_t));
    int32_t *ids = (int32_t *)malloc((size_t)ne * sizeof(int32_t));
    int64_t capb = 0;
    for (int32_t e = 0; e < ne; e++) capb += reads->off[tri[3 * e] + 1] - reads->off[tri[3 * e]] + 2 * o->min_anchor;
    pile.bases = (uint8_t *)malloc((size_t)(capb ? capb : 1));
    for (int32_t e = 0; e < ne; e++) {
        const int32_t rd = tri[3 * e];
        const oz_la *L = &las[tri[3 * e + 1]], *R = &las[tri[3 * e + 2]];
        int32_t ta, bL, bR;
        oz_translate_trace_point_a(L->abpos, L->aepos, L->bbpos, tsm, trace + L->toff, L->tlen / 2, cropL, OZ_FLOOR, &ta, &bL);
        oz_translate_trace_point_a(R->abpos, R->aepos, R->bbpos, tsm, trace + R->toff, R->tlen / 2, cropR, OZ_FLOOR, &ta, &bR);
        const int32_t rl = (int32_t)(reads->off[rd + 1] - reads->off[rd]);
        const int comp = (L->flags & 1u) != 0;
        int32_t b0 = bL, b1 = bR;
        if (comp) { /* getCroppingSlice, cropper.d:533-538 */
            b0 = rl - bR;
            b1 = rl - bL;
        }
        if (b1 - b0 < 14) continue;
        uint8_t *dst = pile.bases + pile.off[pile.n];
        int64_t at = 0;
        /* getSingleReadPatch, cropper.d:363-378 */
        if (!comp) {
            memcpy(dst + at, cl + lp0, (size_t)(lp1 - lp0));
            at += lp1 - lp0;
        } else {
            oz_revcomp(cr + rp0, rp1 - rp0, dst + at);
            at += rp1 - rp0;
        }
        memcpy(dst + at, reads->bases + reads->off[rd] + b0, (size_t)(b1 - b0));
        at += b1 - b0;
        if (!comp) {
            memcpy(dst + at, cr + rp0, (size_t)(rp1 - rp0));
            at += rp1 - rp0;
        } else {
            oz_revcomp(cl + lp0, lp1 - lp0, dst + at);
            at += lp1 - lp0;
        }
        ids[pile.n] = rd;
        pile.off[pile.n + 1] = pile.off[pile.n] + at;
        pile.n++;
    }
    r->nreads = pile.n;
    if (pile.n < o->min_reads) {
        r->status = 2;
        goto done_pile;
    }
    {
        oz_db pdb = {pile.n, pile.off, pile.bases, NULL, NULL, NULL, NULL};
        /* DBdust pileup.db; daligner ... -mdust (package.d:476-482) */
        int64_t *dptr = NULL;
        int32_t *div = NULL;
        if (o->dust) {
            dptr = (int64_t *)malloc(((size_t)pile.n + 1) * sizeof(int64_t));
            oz_dust(&pdb, dptr, &div);
            pdb.mask_ptr = dptr;
            pdb.mask_iv = div;
        }
        int32_t *rlen = (int32_t *)malloc((size_t)pile.n * sizeof(int32_t));
        int32_t maxlen = 0;
        for (int32_t i = 0; i < pile.n; i++) {
            rlen[i] = (int32_t)(pile.off[i + 1] - pile.off[i]);
            if (rlen[i] > maxlen) maxlen = rlen[i];
        }
        oz_opts po;
        const int32_t pla = pile.n <= 60 ? 64 : (pile.n <= 124 ? 128 : 256);
        set_opts(&po, tsp, 500, 2, pla, pla * 2 < 256 ? pla * 2 : 256, o->width, o->algo);
        /* dh_process_opts.max_partners: a read is aligned with the first max_partners reads only (every read of these
         * pile-ups spans the gap, i.e. may serve as reference read: bit 0 everywhere) */
        uint8_t *pfl = NULL;
        if (o->algo == 1 && o->max_partners > 0 && pile.n > o->max_partners) {
            pfl = (uint8_t *)malloc((size_t)pile.n);
            for (int32_t i = 0; i < pile.n; i++) pfl[i] = (uint8_t)(1 | (i < o->max_partners ? 2 : 0));
            pdb.pflags = pfl;
        }
        oz_la_set ps;
        oz_la_set_init(&ps);
        int64_t st[4];
        oz_align_db(&pdb, &pdb, &po, g_inner_threads, &ps, st);  /* (threads inside a pile-up when there are fewer pile-ups than threads) */
        pdb.pflags = NULL;
        free(pfl);
        oz_la_set_sort(&ps);
        /* computeQVs' funnel (package.d:474-516) */
        for (int64_t i = 0; i < ps.n; i++)
            if ((int64_t)ps.la[i].diffs * 1000000 > (int64_t)o->max_align_err_ppm * (ps.la[i].aepos - ps.la[i].abpos))
                ps.la[i].flags |= OZ_FLAG_DISABLED;
        oz_chain_set(&ps, tsp, o->min_relative_score_ppm);
        /* DAScover + DASqv on the chained file, then filterPileUpAlignments (package.d:492-512) */
        const int32_t maxtiles = (maxlen + tsp - 1) / tsp > 0 ? (maxlen + tsp - 1) / tsp : 1;
        uint8_t *qv = (uint8_t *)malloc((size_t)pile.n * maxtiles);
        memset(qv, 255, (size_t)pile.n * maxtiles);
        oz_tile_qv(&ps, pile.n, rlen, tsp, pile.n, qv, maxtiles);
        int any = 0;
        for (int64_t i = 0; i < ps.n; i++) {
            oz_la *la = &ps.la[i];
            if (la->flags & OZ_FLAG_DISABLED) continue;
            if (!oz_valid_pileup_alignment(la, rlen[la->aread], rlen[la->bread], tsp))
                la->flags |= OZ_FLAG_DISABLED;
            else
                any = 1;
        }
        if (!any) {
            r->status = 3;
            free(qv);
            oz_la_set_free(&ps);
            free(rlen);
            free(dptr);
            free(div);
            goto done_pile;
        }
        int32_t *order = (int32_t *)malloc((size_t)pile.n * sizeof(int32_t)), norder = 0;
        oz_rank_reference_reads(qv, pile.n, rlen, tsp, maxtiles, NULL, (double)o->bad_fraction_ppm / 1e6, order, &norder);
        const int32_t ref = order[0];
        r->ref_idx = ref;
        r->ref_read_id = ids[ref];
        free(order);
        free(qv);
        /* consensus rounds */
        int32_t clen = rlen[ref];
        uint8_t *cons = (uint8_t *)malloc((size_t)clen * (1 + OZ_MAXINS) + 8);
        {
            uint8_t *tmp = (uint8_t *)malloc((size_t)clen * (1 + OZ_MAXINS) + 8);
            clen = oz_consensus(pile.bases + pile.off[ref], rlen[ref], &pdb, &ps, ref, tsp, tmp, NULL);
            free(cons);
            cons = tmp;
        }
        oz_la_set_free(&ps);
        for (int32_t round = 1; round < o->rounds; round++) {
            int64_t toff[2] = {0, clen};
            oz_db tdb = {1, toff, cons, NULL, NULL, NULL, NULL};
            oz_opts ro;
            set_opts(&ro, tsp, 500, 0, 4, 32, o->width, o->algo);
            oz_la_set rs;
            oz_la_set_init(&rs);
            oz_align_db(&tdb, &pdb, &ro, 1, &rs, st);
            oz_la_set_sort(&rs);
            for (int64_t i = 0; i < rs.n; i++) {
                oz_la t = rs.la[i];
                t.aread = -1;
                if (!oz_valid_pileup_alignment(&t, clen, rlen[t.bread], tsp)) rs.la[i].flags |= OZ_FLAG_DISABLED;
            }
            uint8_t *tmp = (uint8_t *)malloc((size_t)clen * (1 + OZ_MAXINS) + 8);
            const int32_t nl = oz_consensus(cons, clen, &pdb, &rs, 0, tsp, tmp, NULL);
            oz_la_set_free(&rs);
            free(cons);
            cons = tmp;
            clen = nl;
        }
        free(rlen);
        r->cons_len = clen;
        *cons_out = cons;
        /* flank re-alignment: daligner -A -s126 -l126 contigs consensus (commandline.d:2918-2935) */
        const int32_t fw = o->flank_window > 0 ? o->flank_window : INT32_MAX; /* 0 = the whole contigs */
        const int32_t wl = (cll > fw ? cll - fw : 0) / o->ts_pile * o->ts_pile; /* on the contig's trace grid */
        const int32_t fl_len = cll - wl, fr_len = clr < fw ? clr : fw;
        uint8_t *fb = (uint8_t *)malloc((size_t)fl_len + fr_len + 1);
        memcpy(fb, cl + wl, (size_t)fl_len);
        memcpy(fb + fl_len, cr, (size_t)fr_len);
        int64_t foff[3] = {0, fl_len, (int64_t)fl_len + fr_len};
        oz_db fdb = {2, foff, fb, NULL, NULL, NULL, NULL};
        /* DBdust contigs.dam; daligner -A ... -mdust -mrep (package.d:631-667; no repeat mask here) */
        int64_t *fdptr = NULL;
        int32_t *fdiv = NULL;
        if (o->dust) {
            fdptr = (int64_t *)malloc(3 * sizeof(int64_t));
            oz_dust(&fdb, fdptr, &fdiv);
            fdb.mask_ptr = fdptr;
            fdb.mask_iv = fdiv;
        }
        int64_t coff[2] = {0, clen};
        oz_db cdb = {1, coff, cons, NULL, NULL, NULL, NULL};
        oz_opts fo;
        set_opts(&fo, tsp, 126, 0, 4, 32, o->width, o->algo);
        oz_la_set fs;
        oz_la_set_init(&fs);
        oz_align_db(&fdb, &cdb, &fo, 1, &fs, st);
        oz_la_set_sort(&fs);
        const oz_la *L = NULL, *R = NULL;
        int nL = 0, nR = 0;
        for (int64_t i = 0; i < fs.n; i++) {
            const oz_la *la = &fs.la[i];
            if (la->aread == 0 && la->aepos + tsp >= fl_len && la->bbpos <= tsp) {
                L = la;
                nL++;
            }
            if (la->aread == 1 && la->abpos <= tsp && la->bepos + tsp >= clen) {
                R = la;
                nR++;
            }
        }
        if (nL != 1 || nR != 1)
            r->status = 4;
        else if ((L->flags & 1u) != (R->flags & 1u))
            r->status = 5;
        else {
            r->left_diffs = L->diffs;
            r->right_diffs = R->diffs;
            if ((int64_t)L->diffs * 1000000 > (int64_t)o->max_ins_err_ppm * (L->aepos - L->abpos) ||
                (int64_t)R->diffs * 1000000 > (int64_t)o->max_ins_err_ppm * (R->aepos - R->abpos))
                r->status = 6; /* ensureHighQualityConsensus, output.d:388-410 */
            else {
                r->comp = (L->flags & 1u) ? 1 : 0;
                r->left_aepos = wl + L->aepos;
                r->right_abpos = R->abpos;
                r->ins_begin = L->bepos;
                r->ins_end = R->bbpos;
                if (r->ins_end < r->ins_begin) r->status = 7;
            }
        }
        oz_la_set_free(&fs);
        free(fb);
        free(fdptr);
        free(fdiv);
        free(dptr);
        free(div);
    }
done_pile:
    free(ids);
    seq_list_free(&pile);
}

/* every pile-up through the `process` sequence; OpenMP over pile-ups, and -- when there are fewer pile-ups than threads --
 * over the reads of a pile-up's all-vs-all (nested; the result does not depend on the thread counts).  out[npiles];
 * *bases_out = malloc'd concatenation of the consensus sequences (cons_off / cons_len locate them). */
int oz_process_piles(const oz_db *contigs, const oz_db *reads, const oz_la *las, int64_t n, const uint16_t *trace,
                     const int32_t *gap, const int32_t *count, const int32_t *triples, int32_t npiles,
                     const oz_process_opts *o, int nthreads, oz_insertion *out, uint8_t **bases_out, int64_t *nbases)
{
    (void)n;
    if (nthreads < 1) nthreads = 1;
    g_inner_threads = npiles > 0 && nthreads > npiles ? nthreads / npiles : 1;
    if (g_inner_threads > 1) omp_set_max_active_levels(2);
    if (nthreads > npiles && npiles > 0) nthreads = npiles;
    int64_t *first = (int64_t *)calloc((size_t)npiles + 1, sizeof(int64_t));
    for (int32_t p = 0; p < npiles; p++) first[p + 1] = first[p] + count[p];
    uint8_t **cons = (uint8_t **)calloc((size_t)(npiles ? npiles : 1), sizeof(uint8_t *));
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (int32_t p = 0; p < npiles; p++)
        process_one(contigs, reads, las, trace, gap[p], triples + 3 * first[p], count[p], o, &out[p], &cons[p]);
    int64_t tot = 0;
    for (int32_t p = 0; p < npiles; p++) {
        out[p].cons_off = tot;
        tot += cons[p] ? out[p].cons_len : 0;
        if (!cons[p]) out[p].cons_len = 0;
    }
    uint8_t *all = (uint8_t *)malloc((size_t)(tot ? tot : 1));
    for (int32_t p = 0; p < npiles; p++)
        if (cons[p]) {
            memcpy(all + out[p].cons_off, cons[p], (size_t)out[p].cons_len);
            free(cons[p]);
        }
    free(cons);
    free(first);
    *bases_out = all;
    *nbases = tot;
    return 0;
}

void oz_free(void *p) { free(p); }
