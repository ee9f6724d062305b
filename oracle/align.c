/*
 * align.c -- oracle restatement of the local-alignment pass (daligner / damapper role).
 *
 * TEST INFRASTRUCTURE ONLY (see dh_oracle.h).  PARITY UNPINNED for this file: the reference
 * delegates this arithmetic to external tools (call sites source/dentist/dazzler.d:6121-6140
 * `dalign`, :6158-6170 `damapper`; argv source/dentist/commandline.d:2886-2902, 2918-2955),
 * whose sources are not under /root/reference.  The algorithm below restates the published
 * scheme (Myers, "Efficient local alignment discovery amongst noisy long reads", WABI 2014):
 *   1. all k-mers of A are indexed, k-mers of B (and of its reverse complement) are looked up;
 *   2. hits are binned by diagonal into bands of 2^w; a pair of adjacent bands whose
 *      hit-covered bases reach h triggers an alignment through a seed hit;
 *   3. from the seed an O(ND) furthest-reaching wave runs forward and backward, trimmed to the
 *      points within `xdrop` of the best score, with at most `width` live diagonals, recording a
 *      trace point (diffs, b-bases) every `tspace` A-bases;
 *   4. LAs shorter than `min_len` or noisier than 1-e are dropped.
 * What the consumer pins (and this file honours): the record layout and the trace invariants
 * of source/dentist/common/alignments/base.d:434-458 (sum(bbases) == bepos-bbpos,
 * sum(diffs) == diffs, #trace points == ceil(aepos/s) - floor(abpos/s)).
 */
#include "dh_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

void oz_default_opts(oz_opts *o)
{
    memset(o, 0, sizeof(*o));
    o->k = 14;
    o->hmin = 35;
    o->band_shift = 6;
    o->tspace = 100;
    o->min_len = 500;
    o->pen = 6; /* floor(2 / (1 - 0.7)) */
    o->xdrop = 120;
    o->max_err_ppm = 300000;
    o->max_cand = 32;
    o->max_la = 4;
    o->tcap = 64;
    o->strands = 3;
    o->skip_self = 0;
    o->dmax = 60000;
    o->width = 64;
    o->kmer_mod = 1;
}

/* ------------------------------------------------------------------ LA set ---------- */

void oz_la_set_init(oz_la_set *s) { memset(s, 0, sizeof(*s)); }

void oz_la_set_free(oz_la_set *s)
{
    free(s->la);
    free(s->trace);
    memset(s, 0, sizeof(*s));
}

static void la_set_push(oz_la_set *s, const oz_la *la, const uint16_t *trace)
{
    if (s->n == s->cap) {
        s->cap = s->cap ? s->cap * 2 : 64;
        s->la = (oz_la *)realloc(s->la, (size_t)s->cap * sizeof(oz_la));
    }
    if (s->tn + la->tlen > s->tcap) {
        while (s->tn + la->tlen > s->tcap) s->tcap = s->tcap ? s->tcap * 2 : 4096;
        s->trace = (uint16_t *)realloc(s->trace, (size_t)s->tcap * sizeof(uint16_t));
    }
    s->la[s->n] = *la;
    s->la[s->n].toff = s->tn;
    memcpy(s->trace + s->tn, trace, (size_t)la->tlen * sizeof(uint16_t));
    s->tn += la->tlen;
    s->n++;
}

/* LAsort order: (a, b, comp, abpos, aepos, bbpos, bepos, diffs)
 * source/dentist/common/alignments/base.d:1787-1809 */
static int la_cmp(const void *x, const void *y)
{
    const oz_la *p = (const oz_la *)x, *q = (const oz_la *)y;
#define CMPF(f)                                                                                  \
    if (p->f != q->f) return p->f < q->f ? -1 : 1;
    CMPF(aread)
    CMPF(bread)
    if ((p->flags & OZ_FLAG_COMP) != (q->flags & OZ_FLAG_COMP))
        return (p->flags & OZ_FLAG_COMP) < (q->flags & OZ_FLAG_COMP) ? -1 : 1;
    CMPF(abpos)
    CMPF(aepos)
    CMPF(bbpos)
    CMPF(bepos)
    CMPF(diffs)
#undef CMPF
    return 0;
}

void oz_la_set_sort(oz_la_set *s)
{
    if (s->n < 2) return;
    oz_la *copy = (oz_la *)malloc((size_t)s->n * sizeof(oz_la));
    memcpy(copy, s->la, (size_t)s->n * sizeof(oz_la));
    qsort(copy, (size_t)s->n, sizeof(oz_la), la_cmp);
    uint16_t *nt = (uint16_t *)malloc((size_t)(s->tn ? s->tn : 1) * sizeof(uint16_t));
    int64_t t = 0;
    for (int64_t i = 0; i < s->n; i++) {
        memcpy(nt + t, s->trace + copy[i].toff, (size_t)copy[i].tlen * sizeof(uint16_t));
        copy[i].toff = t;
        t += copy[i].tlen;
    }
    free(s->la);
    free(s->trace);
    s->la = copy;
    s->cap = s->n;
    s->trace = nt;
    s->tcap = s->tn ? s->tn : 1;
}

/* ------------------------------------------------------------------ k-mer index ------ */

/* does the k-mer [p, p+k) of sequence s touch a masked interval?  *cur walks the sorted intervals
 * of the sequence (positions are visited in increasing order) */
static inline int kmer_masked(const oz_db *db, int32_t s, int64_t p, int k, int64_t *cur)
{
    if (!db->mask_ptr) return 0;
    const int64_t end = db->mask_ptr[s + 1];
    while (*cur < end && db->mask_iv[2 * *cur + 1] <= p) (*cur)++;
    return *cur < end && db->mask_iv[2 * *cur] < p + k;
}


/* modimer sampling: the same k-mers are kept on the A and on the B side.  The decision is taken on
 * the CANONICAL k-mer (the smaller of the k-mer and its reverse complement), so a k-mer and its
 * reverse complement are sampled together: one pass over a read then serves both strands. */
static inline uint64_t kmer_rc(uint64_t km, int k)
{
    uint64_t r = 0;
    for (int i = 0; i < k; i++) {
        r = (r << 2) | (3 - (km & 3));
        km >>= 2;
    }
    return r;
}
static inline int kmer_sampled_k(uint64_t km, int32_t mod, int k)
{
    if (mod <= 1) return 1;
    const uint64_t rc = kmer_rc(km, k), c = km < rc ? km : rc;
    return (uint32_t)((c * 0x9E3779B97F4A7C15ull) >> 32) % (uint32_t)mod == 0;
}


typedef struct {
    uint64_t key; /* group * 4^k + kmer */
    int32_t aseq;
    int32_t apos;
} ix_ent;

struct oz_index {
    int64_t n;
    ix_ent *e;
    int64_t *goff; /* virtual global offset of every A sequence: sum(len + sepv) */
    int32_t na;
    int32_t sepv;
    int32_t k;
    int32_t shift;
    int64_t nb;
    int64_t *dir; /* bucket directory over key >> shift */
};

static int ent_cmp(const void *x, const void *y)
{
    const ix_ent *p = (const ix_ent *)x, *q = (const ix_ent *)y;
    if (p->key != q->key) return p->key < q->key ? -1 : 1;
    if (p->aseq != q->aseq) return p->aseq < q->aseq ? -1 : 1;
    if (p->apos != q->apos) return p->apos < q->apos ? -1 : 1;
    return 0;
}

/* sepv separates the sequences of A on the virtual diagonal axis so that no band of 2^w
 * diagonals is shared by two A sequences; it has to be >= the longest B sequence + 2^w. */
/* both are multiples of 4096 (>= the widest band), and so is every goff: the band of a hit then
 * depends only on the pair (A sequence, B read), not on the rest of the DB */
static int32_t sepv_for(int32_t max_blen) { return (max_blen + 64 + 4095) & ~4095; }

static oz_index *index_build(const oz_db *A, const oz_opts *o, int32_t max_blen)
{
    oz_index *ix = (oz_index *)calloc(1, sizeof(*ix));
    const int k = o->k;
    ix->k = k;
    ix->na = A->n;
    ix->sepv = sepv_for(max_blen);
    ix->goff = (int64_t *)malloc(((size_t)A->n + 1) * sizeof(int64_t));
    int64_t g = 0, total = 0;
    for (int32_t s = 0; s < A->n; s++) {
        ix->goff[s] = g;
        int64_t len = A->off[s + 1] - A->off[s];
        g += (len + ix->sepv + 4095) & ~4095ll;
        if (len >= k) total += len - k + 1;
    }
    ix->goff[A->n] = g;
    ix->e = (ix_ent *)malloc((size_t)(total ? total : 1) * sizeof(ix_ent));
    const uint64_t mask = (k == 32) ? ~0ull : ((1ull << (2 * k)) - 1);
    int64_t n = 0;
    for (int32_t s = 0; s < A->n; s++) {
        const uint8_t *a = A->bases + A->off[s];
        int64_t len = A->off[s + 1] - A->off[s];
        uint64_t grp = A->group ? (uint64_t)A->group[s] : 0;
        uint64_t km = 0;
        int valid = 0;
        int64_t mcur = A->mask_ptr ? A->mask_ptr[s] : 0;
        for (int64_t p = 0; p < len; p++) {
            if (a[p] < 4) {
                km = ((km << 2) | a[p]) & mask;
                valid++;
            } else {
                valid = 0;
                km = 0;
            }
            if (valid >= k && kmer_sampled_k(km, o->kmer_mod, k) && !kmer_masked(A, s, p - k + 1, k, &mcur)) {
                ix->e[n].key = (grp << (2 * k)) | km;
                ix->e[n].aseq = s;
                ix->e[n].apos = (int32_t)(p - k + 1);
                n++;
            }
        }
    }
    ix->n = n;
    /* order by (key, aseq, apos): counting sort on the top bits of the key, then each bucket is
     * sorted on its own -- same final order as one big sort, but O(n) instead of O(n log n) */
    int ngrp = 1;
    if (A->group)
        for (int32_t s = 0; s < A->n; s++)
            if (A->group[s] + 1 > ngrp) ngrp = A->group[s] + 1;
    int keybits = 2 * k, gb = 0;
    while ((1 << gb) < ngrp) gb++;
    keybits += gb;
    int pbits = 10;
    while (pbits < 24 && (1ll << pbits) < n) pbits++;
    if (pbits > keybits) pbits = keybits;
    ix->shift = keybits - pbits;
    ix->nb = (int64_t)((((uint64_t)ngrp << (2 * k)) - 1) >> ix->shift) + 1;
    ix->dir = (int64_t *)calloc((size_t)ix->nb + 1, sizeof(int64_t));
    for (int64_t i = 0; i < n; i++) ix->dir[(ix->e[i].key >> ix->shift) + 1]++;
    for (int64_t b = 0; b < ix->nb; b++) ix->dir[b + 1] += ix->dir[b];
    ix_ent *tmp = (ix_ent *)malloc((size_t)(n ? n : 1) * sizeof(ix_ent));
    int64_t *cur = (int64_t *)malloc((size_t)ix->nb * sizeof(int64_t));
    memcpy(cur, ix->dir, (size_t)ix->nb * sizeof(int64_t));
    for (int64_t i = 0; i < n; i++) tmp[cur[ix->e[i].key >> ix->shift]++] = ix->e[i];
    free(cur);
    free(ix->e);
    ix->e = tmp;
    for (int64_t b = 0; b < ix->nb; b++)
        if (ix->dir[b + 1] - ix->dir[b] > 1)
            qsort(ix->e + ix->dir[b], (size_t)(ix->dir[b + 1] - ix->dir[b]), sizeof(ix_ent), ent_cmp);
    return ix;
}

oz_index *oz_index_build(const oz_db *A, const oz_opts *o) { return index_build(A, o, 1 << 20); }

void oz_index_free(oz_index *ix)
{
    if (!ix) return;
    free(ix->e);
    free(ix->goff);
    free(ix->dir);
    free(ix);
}

int64_t oz_index_size(const oz_index *ix) { return ix->n; }

/* first entry with key >= key: directory bucket, then a short scan */
static int64_t ix_lower(const oz_index *ix, uint64_t key)
{
    const int64_t b = (int64_t)(key >> ix->shift);
    if (b >= ix->nb) return ix->n;
    int64_t lo = ix->dir[b];
    const int64_t hi = ix->dir[b + 1];
    while (lo < hi && ix->e[lo].key < key) lo++;
    return lo;
}

/* ------------------------------------------------------------------ seeds ------------- */

static int u64_cmp(const void *x, const void *y)
{
    uint64_t p = *(const uint64_t *)x, q = *(const uint64_t *)y;
    return p < q ? -1 : (p > q ? 1 : 0);
}

typedef struct {
    int64_t band;
    int32_t s, e; /* hit range */
    int32_t cov;
} band_ent;

static int cand_cmp(const void *x, const void *y)
{
    const oz_cand *p = (const oz_cand *)x, *q = (const oz_cand *)y;
    if (p->score != q->score) return p->score > q->score ? -1 : 1;
    if (p->band != q->band) return p->band < q->band ? -1 : 1;
    return 0;
}

static int cand_cmp_aseq(const void *x, const void *y)
{
    const oz_cand *p = (const oz_cand *)x, *q = (const oz_cand *)y;
    if (p->aseq != q->aseq) return p->aseq < q->aseq ? -1 : 1;
    return cand_cmp(x, y);
}

#define HIT_QBITS 24
#define HIT_QMASK ((1u << HIT_QBITS) - 1)

/* bmask/nbmask: sorted masked intervals of this B sequence in the orientation of `b` (or NULL) */
static int seed_candidates(const oz_index *ix, const uint8_t *b, int32_t blen, int32_t bgroup,
                           int32_t bself, const int32_t *bmask, int64_t nbmask, const oz_opts *o,
                           oz_cand *out, int32_t *nhits_out, const uint8_t *pflags);

/* oz_db.pflags: is the record with A read a and B read b wanted / does the pair {a, b} yield hits at all */
static int rec_wanted(const uint8_t *f, int32_t a, int32_t b) { return !f || ((f[a] & 1) && (f[b] & 2)); }
static int pair_seeded(const uint8_t *f, int32_t a, int32_t b) { return !f || rec_wanted(f, a, b) || rec_wanted(f, b, a); }

int oz_seed_candidates(const oz_index *ix, const oz_db *A, const uint8_t *b, int32_t blen,
                       int32_t bgroup, int32_t bself, int32_t sepv_unused, const oz_opts *o,
                       oz_cand *out, int32_t *nhits_out)
{
    (void)sepv_unused;
    return seed_candidates(ix, b, blen, bgroup, bself, NULL, 0, o, out, nhits_out, o->skip_self == 2 && o->algo == 1 ? A->pflags : NULL);
}

static int seed_candidates(const oz_index *ix, const uint8_t *b, int32_t blen, int32_t bgroup,
                           int32_t bself, const int32_t *bmask, int64_t nbmask, const oz_opts *o,
                           oz_cand *out, int32_t *nhits_out, const uint8_t *pflags)
{
    const int k = o->k;
    const int32_t sepv = ix->sepv;
    int64_t bcur = 0;
    const uint64_t mask = (1ull << (2 * k)) - 1;
    int64_t cap = 1024, n = 0;
    uint64_t *hits = (uint64_t *)malloc((size_t)cap * sizeof(uint64_t));
    uint64_t km = 0;
    int valid = 0;
    for (int32_t p = 0; p < blen; p++) {
        if (b[p] < 4) {
            km = ((km << 2) | b[p]) & mask;
            valid++;
        } else {
            valid = 0;
            km = 0;
        }
        if (valid < k || !kmer_sampled_k(km, o->kmer_mod, k)) continue;
        const int32_t q = p - k + 1;
        if (bmask) {
            while (bcur < nbmask && bmask[2 * bcur + 1] <= q) bcur++;
            if (bcur < nbmask && bmask[2 * bcur] < q + k) continue;
        }
        const uint64_t key = ((uint64_t)bgroup << (2 * k)) | km;
        int64_t s = ix_lower(ix, key), e = s;
        while (e < ix->n && ix->e[e].key == key) e++;
        if (e == s || e - s > o->tcap) continue;
        for (int64_t t = s; t < e; t++) {
            if (o->skip_self == 1 && ix->e[t].aseq == bself) continue;
            /* tandem (datander): a read against itself, below the main diagonal only (a > b) */
            if (o->skip_self == 3 && (ix->e[t].aseq != bself || ix->e[t].apos - q < 1)) continue;
            /* symmetric: each unordered pair once; which read plays B alternates with the parity of
             * a + b so that every read is B for about half of its partners (balanced work) */
            if (o->skip_self == 2) {
                const int32_t pa = ix->e[t].aseq;
                if (pa == bself || ((pa < bself) != (((pa + bself) & 1) == 0))) continue;
                if (!pair_seeded(pflags, pa, bself)) continue; /* neither record of the pair is wanted */
            }
            int64_t D = ix->goff[ix->e[t].aseq] + ix->e[t].apos + sepv - q;
            if (n == cap) {
                cap *= 2;
                hits = (uint64_t *)realloc(hits, (size_t)cap * sizeof(uint64_t));
            }
            hits[n++] = ((uint64_t)D << HIT_QBITS) | (uint32_t)q;
        }
    }
    if (nhits_out) *nhits_out = (int32_t)n;
    if (n == 0) {
        free(hits);
        return 0;
    }
    qsort(hits, (size_t)n, sizeof(uint64_t), u64_cmp);

    /* per-hit covered-base contribution and band runs */
    int32_t *c = (int32_t *)malloc((size_t)n * sizeof(int32_t));
    band_ent *bd = (band_ent *)malloc((size_t)n * sizeof(band_ent));
    int64_t m = 0;
    for (int64_t i = 0; i < n; i++) {
        int64_t D = (int64_t)(hits[i] >> HIT_QBITS);
        int32_t q = (int32_t)(hits[i] & HIT_QMASK);
        if (i > 0 && (int64_t)(hits[i - 1] >> HIT_QBITS) == D) {
            int32_t dq = q - (int32_t)(hits[i - 1] & HIT_QMASK);
            c[i] = dq < k ? dq : k;
        } else
            c[i] = k;
        int64_t band = D >> o->band_shift;
        if (m == 0 || bd[m - 1].band != band) {
            bd[m].band = band;
            bd[m].s = (int32_t)i;
            bd[m].e = (int32_t)i;
            bd[m].cov = 0;
            m++;
        }
        bd[m - 1].e = (int32_t)i + 1;
        bd[m - 1].cov += c[i];
    }

    int64_t ccap = 64, nc = 0;
    oz_cand *cands = (oz_cand *)malloc((size_t)ccap * sizeof(oz_cand));
    for (int64_t j = 0; j < m; j++) {
        const int64_t bj = bd[j].band;
        /* coverage of band x looked up among the neighbouring entries */
#define COV(x)                                                                                   \
    ((j > 0 && bd[j - 1].band == (x))                                                            \
         ? bd[j - 1].cov                                                                         \
         : (bd[j].band == (x)                                                                    \
                ? bd[j].cov                                                                      \
                : ((j + 1 < m && bd[j + 1].band == (x))                                          \
                       ? bd[j + 1].cov                                                           \
                       : ((j + 2 < m && bd[j + 2].band == (x)) ? bd[j + 2].cov : 0))))
        const int32_t P = COV(bj) + COV(bj + 1);
        const int32_t Pm1 = COV(bj - 1) + COV(bj);
        const int32_t Pp1 = COV(bj + 1) + COV(bj + 2);
#undef COV
        if (P < o->hmin || P < Pm1 || P <= Pp1) continue;
        int32_t rs = bd[j].s, re = bd[j].e;
        if (j + 1 < m && bd[j + 1].band == bj + 1) re = bd[j + 1].e;
        /* seed = first hit of the run (same diagonal, steps <= k) covering most bases */
        int32_t best_first = rs, best_cov = -1, run_first = rs;
        for (int32_t i = rs; i < re; i++) {
            int linked = 0;
            if (i > rs && (hits[i] >> HIT_QBITS) == (hits[i - 1] >> HIT_QBITS)) {
                int32_t dq = (int32_t)(hits[i] & HIT_QMASK) - (int32_t)(hits[i - 1] & HIT_QMASK);
                linked = dq <= k;
            }
            if (!linked) run_first = i;
            int32_t cov = k + (int32_t)(hits[i] & HIT_QMASK) - (int32_t)(hits[run_first] & HIT_QMASK);
            if (cov > best_cov) {
                best_cov = cov;
                best_first = run_first;
            }
        }
        int64_t D = (int64_t)(hits[best_first] >> HIT_QBITS);
        int32_t q = (int32_t)(hits[best_first] & HIT_QMASK);
        int64_t gv = D - sepv + q;
        /* aseq = last sequence with goff <= gv */
        int32_t lo = 0, hi = ix->na;
        while (hi - lo > 1) {
            int32_t mid = (lo + hi) >> 1;
            if (ix->goff[mid] <= gv)
                lo = mid;
            else
                hi = mid;
        }
        if (nc == ccap) {
            ccap *= 2;
            cands = (oz_cand *)realloc(cands, (size_t)ccap * sizeof(oz_cand));
        }
        cands[nc].score = P;
        cands[nc].aseq = lo;
        cands[nc].apos = (int32_t)(gv - ix->goff[lo]);
        cands[nc].bpos = q;
        cands[nc].band = bj;
        nc++;
    }
    qsort(cands, (size_t)nc, sizeof(oz_cand), cand_cmp);
    if (nc > o->max_cand) nc = o->max_cand;
    /* symmetric all-vs-all: the kept candidates are grouped by A read (rank order inside a group),
     * so that the groups -- the only candidates that depend on each other -- are separate work
     * units for the device */
    if (o->skip_self == 2) qsort(cands, (size_t)nc, sizeof(oz_cand), cand_cmp_aseq);
    memcpy(out, cands, (size_t)nc * sizeof(oz_cand));
    free(cands);
    free(bd);
    free(c);
    free(hits);
    return (int)nc;
}

/* ------------------------------------------------------------------ wave -------------- */

typedef struct {
    int32_t parent, d, j;
} tp_node;

typedef struct {
    tp_node *v;
    int32_t n, cap;
} tp_pool;

static int32_t pool_push(tp_pool *p, int32_t parent, int32_t d, int32_t j)
{
    if (p->n == p->cap) {
        p->cap = p->cap ? p->cap * 2 : 256;
        p->v = (tp_node *)realloc(p->v, (size_t)p->cap * sizeof(tp_node));
    }
    p->v[p->n].parent = parent;
    p->v[p->n].d = d;
    p->v[p->n].j = j;
    return p->n++;
}

#define WMASK 127

/* number of trace boundaries (first at tp_first, then every ts) that are <= x */
static inline int32_t nbound(int32_t x, int32_t tp_first, int32_t ts)
{
    return x >= tp_first ? (x - tp_first) / ts + 1 : 0;
}

/*
 * One-directional greedy extension.  Sequence element i of A' is ap[i*astep], of B' bp[j*bstep].
 * Diagonal k = i - j.  R[k] = furthest i reached on k with the current number of diffs.
 * Score of a point = i + j - pen*d.  Returns best point and the boundary crossings
 * (cd[m], cj[m]) = (diffs, j) when the best path first reached A'-offset tp_first + m*ts.
 * If tpb_first > 0 the crossings of the B'-offsets tpb_first + m*ts are recorded as well:
 * (cdb[m], cib[m]) = (diffs, i) when the path first reached that j (symmetric mode: the same
 * path also yields the trace of the transposed record); *nbb receives their number.
 */
static int extend(const uint8_t *ap, int astep, int32_t an, const uint8_t *bp, int bstep,
                  int32_t bn, int32_t tp_first, int32_t tpb_first, const oz_opts *o, int32_t *bi,
                  int32_t *bj, int32_t *bd_, int32_t *cd, int32_t *cj, int32_t *cdb, int32_t *cib,
                  int32_t *nbb, int32_t *dlo, int32_t *dhi, int64_t *cells)
{
    const int32_t ts = o->tspace, pen = o->pen, xdrop = o->xdrop;
    int32_t R[2][WMASK + 1], H[2][WMASK + 1], HB[2][WMASK + 1];
    uint8_t alive[2][WMASK + 1];
    tp_pool pool = {0, 0, 0};
    memset(alive, 0, sizeof(alive));

    int32_t i = 0;
    while (i < an && i < bn && ap[(int64_t)i * astep] == bp[(int64_t)i * bstep]) i++;
    int32_t head = -1, headb = -1;
    for (int32_t m = 0, nb = nbound(i, tp_first, ts); m < nb; m++)
        head = pool_push(&pool, head, 0, tp_first + m * ts);
    if (tpb_first > 0)
        for (int32_t m = 0, nb = nbound(i, tpb_first, ts); m < nb; m++)
            headb = pool_push(&pool, headb, 0, tpb_first + m * ts);
    int cur = 0;
    R[cur][0] = i;
    H[cur][0] = head;
    HB[cur][0] = headb;
    alive[cur][0] = 1;
    int32_t L = 0, U = 0;
    int32_t best_score = 2 * i, best_i = i, best_k = 0, best_d = 0, best_head = head, best_headb = headb;
    int64_t ncell = 1;

    for (int32_t d = 1; d <= o->dmax; d++) {
        const int prv = cur;
        cur ^= 1;
        const int32_t nL = L - 1, nU = U + 1;
        int32_t step_best = INT32_MIN, step_k = 0;
        for (int32_t k = nL; k <= nU; k++) {
            int32_t ni = -1, src = 0;
            /* substitution on k, deletion from k-1 (consumes A), insertion from k+1 (consumes B) */
            if (k >= L && k <= U && alive[prv][k & WMASK]) {
                int32_t c = R[prv][k & WMASK] + 1;
                if (c <= an && c - k <= bn && c - k >= 0) {
                    ni = c;
                    src = k;
                }
            }
            if (k - 1 >= L && k - 1 <= U && alive[prv][(k - 1) & WMASK]) {
                int32_t c = R[prv][(k - 1) & WMASK] + 1;
                if (c <= an && c - k <= bn && c - k >= 0 && c > ni) {
                    ni = c;
                    src = k - 1;
                }
            }
            if (k + 1 >= L && k + 1 <= U && alive[prv][(k + 1) & WMASK]) {
                int32_t c = R[prv][(k + 1) & WMASK];
                if (c <= an && c - k <= bn && c - k >= 0 && c > ni) {
                    ni = c;
                    src = k + 1;
                }
            }
            if (ni < 0) {
                alive[cur][k & WMASK] = 0;
                continue;
            }
            const int32_t prev_i = R[prv][src & WMASK];
            int32_t hd = H[prv][src & WMASK], hb = HB[prv][src & WMASK];
            int32_t j = ni - k;
            while (ni < an && j < bn && ap[(int64_t)ni * astep] == bp[(int64_t)j * bstep]) {
                ni++;
                j++;
            }
            ncell++;
            for (int32_t m = nbound(prev_i, tp_first, ts), nb = nbound(ni, tp_first, ts); m < nb;
                 m++)
                hd = pool_push(&pool, hd, d, tp_first + m * ts - k);
            if (tpb_first > 0)
                for (int32_t m = nbound(prev_i - src, tpb_first, ts), nb = nbound(j, tpb_first, ts);
                     m < nb; m++)
                    hb = pool_push(&pool, hb, d, tpb_first + m * ts + k);
            R[cur][k & WMASK] = ni;
            H[cur][k & WMASK] = hd;
            HB[cur][k & WMASK] = hb;
            alive[cur][k & WMASK] = 1;
            const int32_t sc = 2 * ni - k - pen * d;
            if (sc > step_best) {
                step_best = sc;
                step_k = k;
            }
        }
        if (step_best == INT32_MIN) break;
        if (step_best > best_score) {
            best_score = step_best;
            best_k = step_k;
            best_i = R[cur][step_k & WMASK];
            best_d = d;
            best_head = H[cur][step_k & WMASK];
            best_headb = HB[cur][step_k & WMASK];
        }
        /* trim to points within xdrop of the best */
        int32_t l2 = INT32_MAX, u2 = INT32_MIN;
        for (int32_t k = nL; k <= nU; k++) {
            if (!alive[cur][k & WMASK]) continue;
            const int32_t sc = 2 * R[cur][k & WMASK] - k - pen * d;
            if (sc < best_score - xdrop) {
                alive[cur][k & WMASK] = 0;
                continue;
            }
            if (k < l2) l2 = k;
            if (k > u2) u2 = k;
        }
        if (l2 > u2) break;
        /* at most `width` live diagonals: drop the lower-scoring edge (ties: the low edge) */
        while (u2 - l2 + 1 > o->width) {
            const int32_t sl = 2 * R[cur][l2 & WMASK] - l2, su = 2 * R[cur][u2 & WMASK] - u2;
            if (sl <= su) {
                alive[cur][l2 & WMASK] = 0;
                do l2++;
                while (!alive[cur][l2 & WMASK]);
            } else {
                alive[cur][u2 & WMASK] = 0;
                do u2--;
                while (!alive[cur][u2 & WMASK]);
            }
        }
        /* clear stale slots just outside the new window so they cannot alias */
        for (int32_t k = nL; k < l2; k++) alive[cur][k & WMASK] = 0;
        for (int32_t k = u2 + 1; k <= nU; k++) alive[cur][k & WMASK] = 0;
        L = l2;
        U = u2;
    }

    *bi = best_i;
    *bj = best_i - best_k;
    *bd_ = best_d;
    int32_t nb = nbound(best_i, tp_first, ts);
    int32_t h = best_head;
    int32_t lo = 0, hi = 0; /* diagonal excursion of the path relative to the seed */
    if (best_k < lo) lo = best_k;
    if (best_k > hi) hi = best_k;
    for (int32_t m = nb - 1; m >= 0; m--) {
        if (h < 0) {
            fprintf(stderr, "oracle: trace chain too short\n");
            abort();
        }
        cd[m] = pool.v[h].d;
        cj[m] = pool.v[h].j;
        int32_t kk = (tp_first + m * ts) - cj[m];
        if (kk < lo) lo = kk;
        if (kk > hi) hi = kk;
        h = pool.v[h].parent;
    }
    if (h >= 0) {
        fprintf(stderr, "oracle: trace chain too long\n");
        abort();
    }
    if (tpb_first > 0) {
        const int32_t nbB = nbound(best_i - best_k, tpb_first, ts);
        h = best_headb;
        for (int32_t m = nbB - 1; m >= 0; m--) {
            if (h < 0) {
                fprintf(stderr, "oracle: B trace chain too short\n");
                abort();
            }
            cdb[m] = pool.v[h].d;
            cib[m] = pool.v[h].j;
            h = pool.v[h].parent;
        }
        if (h >= 0) {
            fprintf(stderr, "oracle: B trace chain too long\n");
            abort();
        }
        *nbb = nbB;
    }
    *dlo = lo;
    *dhi = hi;
    if (cells) *cells += ncell;
    free(pool.v);
    return nb;
}

/*
 * DH-2: tile-by-tile banded extension (selected by oz_opts.algo == 1; band = o->width rows, 32 or 64).
 *
 * The extension advances through A' one trace tile at a time (the first tile ends at tp_first, the
 * others are `tspace` columns; the last one ends with A').  A tile that starts at (a0, b0) is a banded
 * edit-distance DP over its columns c = 0..cols and the rows j (B' bases consumed since b0) with
 *   c - W/2 <= j <= c + W/2 - 1                         (the band slides down one row per column)
 *   D[0][j] = |j|
 *   D[c][j] = min(D[c-1][j-1] + !eq(c, j), left + 1, up + 1)
 * where up = D[c][j-1] is missing for the top row of the band, and left = D[c-1][j] is, for the
 * bottom row of the band, the value V[c-1] of a virtual row beneath the band that never matches:
 *   V[0] = D[0][W/2 - 1] + 1,  V[c] = min(D[c][bottom] + 1, V[c-1] + 1)
 * (this is exactly what Hyyro's diagonal-band bit-vector recurrence computes; the device kernel is
 * that recurrence, one alignment per lane).  eq(c, j): rows j <= 0 (before the tile) never match; rows
 * j > bnr = bn - b0 (past the end of B') match everything, so that D[c][bnr + t] = D[c - t][bnr]:
 * the cells beneath the end of B' in the last column are the history of B's last row.
 * At the last column the row with the smallest D wins (ties: closest to the tile's diagonal, then
 * the lower row).  A row past the end of B' stands for the end point (a0 + cols - t, bn) and ends
 * the extension, so does the end of A'.  Otherwise the tile's (diffs, b-bases) is a trace point, the
 * next tile starts at the chosen cell with the band centred on it, and the running score
 * a + b - pen * diffs decides as in DH-1: the best boundary is remembered, a boundary scoring less
 * than best - xdrop stops the extension, which then ends at the best boundary.  An end point inside
 * the last tile replaces the best boundary only if it scores higher.
 * Same outputs as extend(): best point, crossings (cd[m], cj[m]) = (diffs, j) at A'-offset
 * tp_first + m*ts for m < nbound(best i), diagonal excursion, cells (= band cells computed).
 */
#define T2_WMAX 64
/* tan (tandem mode, oz_opts.skip_self == 3: A' and B' are the same read): cells in which B's base is not BEFORE A's on
 * the read never match -- the alignment of a read with itself stays below the main diagonal (datander reports the
 * non-trivial self alignments; on the diagonal everything matches).  sd = seed diagonal as - bs > 0; forward (tan 1) a
 * tile at (a0, b0) lies on the diagonal dd = sd + a0 - b0 and its rows i >= dd + W/2 are barred; backward (tan 2) the
 * axes are mirrored: dd = sd - (a0 - b0), rows i <= W/2 - dd.  Rows past the end of B' stay wild (they are history). */
static int extend_tiled(const uint8_t *ap, int astep, int32_t an, const uint8_t *bp, int bstep,
                        int32_t bn, int32_t tp_first, const oz_opts *o, int32_t *bi, int32_t *bj,
                        int32_t *bd_, int32_t *cd, int32_t *cj, int32_t *dlo, int32_t *dhi,
                        int64_t *cells, int tan, int32_t sd)
{
    const int32_t W = o->width, lo = -(W / 2), ts = o->tspace, pen = o->pen, xdrop = o->xdrop;
    int32_t a0 = 0, b0 = 0, dsum = 0, ntp = 0;
    int32_t best_s = 0, best_a = 0, best_b = 0, best_d = 0;
    int64_t ncell = 0;
    int32_t D[2][T2_WMAX];
    if (W != 32 && W != 64) {
        fprintf(stderr, "oracle: DH-2 needs width 32 or 64\n");
        abort();
    }
    while (a0 < an && b0 < bn) {
        const int32_t T = ntp == 0 ? tp_first : ts;
        const int32_t anr = an - a0, bnr = bn - b0;
        const int32_t cols = T < anr ? T : anr;
        int cur = 0;
        int32_t V = W / 2;
        for (int32_t i = 0; i < W; i++) D[0][i] = lo + i < 0 ? -(lo + i) : lo + i;
        for (int32_t c = 1; c <= cols; c++) {
            const int prv = cur;
            cur ^= 1;
            const uint8_t ach = ap[(int64_t)(a0 + c - 1) * astep];
            for (int32_t i = 0; i < W; i++) {
                const int32_t j = c + lo + i;
                int eq;
                if (j <= 0)
                    eq = 0;
                else if (j > bnr)
                    eq = 1;
                else if (tan == 1 && i >= sd + (a0 - b0) + W / 2)
                    eq = 0;
                else if (tan == 2 && i <= W / 2 - (sd - (a0 - b0)))
                    eq = 0;
                else
                    eq = ach == bp[(int64_t)(b0 + j - 1) * bstep];
                int32_t v = D[prv][i] + !eq;
                const int32_t left = (i + 1 < W ? D[prv][i + 1] : V) + 1;
                if (left < v) v = left;
                if (i > 0 && D[cur][i - 1] + 1 < v) v = D[cur][i - 1] + 1;
                D[cur][i] = v;
            }
            V = (D[cur][W - 1] < V ? D[cur][W - 1] : V) + 1;
            ncell += W;
        }
        /* the row of the last column to go on from (or to end at) */
        uint32_t key = UINT32_MAX;
        for (int32_t i = 0; i < W; i++) {
            const int32_t j = cols + lo + i;
            if (j < 0 || j - bnr > cols) continue;
            const int32_t off = lo + i < 0 ? -(lo + i) : lo + i;
            const uint32_t kk = ((uint32_t)D[cur][i] << 16) | ((uint32_t)off << 8) | (uint32_t)(W - 1 - i);
            if (kk < key) key = kk;
        }
        if (key == UINT32_MAX) break; /* (cannot happen: row max(0, ..) of the band is always eligible) */
        const int32_t ci = W - 1 - (int32_t)(key & 255), dt = (int32_t)(key >> 16);
        const int32_t j = cols + lo + ci, t = j > bnr ? j - bnr : 0;
        if (t > 0 || cols < T) {
            /* the end of B' or of A': an end point inside (or at the end of) this tile */
            const int32_t ea = a0 + cols - t, eb = b0 + j - t, ed = dsum + dt;
            const int32_t sc = ea + eb - pen * ed;
            if (sc > best_s) {
                best_s = sc;
                best_a = ea;
                best_b = eb;
                best_d = ed;
            }
            break;
        }
        a0 += T;
        b0 += j;
        dsum += dt;
        cd[ntp] = dsum;
        cj[ntp] = b0;
        ntp++;
        const int32_t sc = a0 + b0 - pen * dsum;
        if (sc > best_s) {
            best_s = sc;
            best_a = a0;
            best_b = b0;
            best_d = dsum;
        } else if (sc < best_s - xdrop)
            break;
    }
    *bi = best_a;
    *bj = best_b;
    *bd_ = best_d;
    const int32_t nb = nbound(best_a, tp_first, ts);
    int32_t klo = 0, khi = 0;
    const int32_t bk = best_a - best_b;
    if (bk < klo) klo = bk;
    if (bk > khi) khi = bk;
    for (int32_t m = 0; m < nb; m++) {
        const int32_t kk = tp_first + m * ts - cj[m];
        if (kk < klo) klo = kk;
        if (kk > khi) khi = kk;
    }
    *dlo = klo;
    *dhi = khi;
    if (cells) *cells += ncell;
    return nb;
}

/* trace pairs (delta diffs, delta other) between consecutive grid boundaries, in increasing grid
 * coordinate: grid = the coordinate the trace spacing refers to, other = the opposite sequence.
 * gs/os = seed on the grid / other axis; res = residue of the boundaries (grid = res mod ts). */
static int32_t assemble_trace(int32_t ts, int32_t res, int32_t gs, int32_t os, int32_t gbeg,
                              int32_t gend, int32_t obeg, int32_t oend, int32_t rdv, int32_t fdv,
                              int32_t rev_first, int32_t nr, const int32_t *rd, const int32_t *ro,
                              int32_t fwd_first, int32_t nf, const int32_t *fd, const int32_t *fo,
                              uint16_t *trace)
{
    int32_t n = 0, po = obeg, pD = -rdv;
    for (int32_t m = nr - 1; m >= 0; m--) { /* reverse crossings, outermost first */
        const int32_t g = gs - (rev_first + m * ts);
        if (g <= gbeg) continue;
        const int32_t ov = os - ro[m], D = -rd[m];
        trace[n++] = (uint16_t)(D - pD);
        trace[n++] = (uint16_t)(ov - po);
        po = ov;
        pD = D;
    }
    if (((gs - res) % ts + ts) % ts == 0 && gs > gbeg && gs < gend) { /* the seed is a boundary */
        trace[n++] = (uint16_t)(0 - pD);
        trace[n++] = (uint16_t)(os - po);
        po = os;
        pD = 0;
    }
    for (int32_t m = 0; m < nf; m++) {
        const int32_t g = gs + fwd_first + m * ts;
        if (g >= gend) break;
        const int32_t ov = os + fo[m], D = fd[m];
        trace[n++] = (uint16_t)(D - pD);
        trace[n++] = (uint16_t)(ov - po);
        po = ov;
        pD = D;
    }
    if (gend > gbeg) {
        trace[n++] = (uint16_t)(fdv - pD);
        trace[n++] = (uint16_t)(oend - po);
    }
    return n;
}

/* la2/trace2 (optional): the transposed record (A and B swapped) of the same alignment with its
 * trace on the grid of B; comp2 = 1 when B is the reverse complement of its read (then the grid
 * is the forward strand of that read and the record is mirrored accordingly). */
static int local_align2(const uint8_t *a, int32_t alen, const uint8_t *b, int32_t blen, int32_t as,
                        int32_t bs, const oz_opts *o, oz_la *la, uint16_t *trace, oz_la *la2,
                        uint16_t *trace2, int comp2, int32_t *dlo, int32_t *dhi, int64_t *cells)
{
    const int32_t ts = o->tspace;
    /* forward: boundaries at real a = m*ts > as ; reverse: at real a = m*ts < as */
    const int32_t fwd_first = ts - (as % ts);             /* in (0, ts] */
    const int32_t rev_first = (as % ts) ? (as % ts) : ts; /* in (0, ts] */
    const int32_t resb = comp2 ? blen % ts : 0;           /* B grid: b = resb (mod ts) */
    const int32_t bm = ((bs - resb) % ts + ts) % ts;
    const int32_t fwdb_first = la2 ? ts - bm : 0, revb_first = la2 ? (bm ? bm : ts) : 0;
    const int32_t maxb = (alen > blen ? alen : blen) / ts + 3;
    int32_t *buf = (int32_t *)malloc((size_t)maxb * 8 * sizeof(int32_t));
    int32_t *fd = buf, *fj = fd + maxb, *rd = fj + maxb, *rj = rd + maxb;
    int32_t *fdb = rj + maxb, *fib = fdb + maxb, *rdb = fib + maxb, *rib = rdb + maxb;
    int32_t fi, fjv, fdv, ri, rjv, rdv, flo, fhi, rlo, rhi, nfb = 0, nrb = 0;
    int32_t nf, nr;
    if (o->algo == 1) {
        if (la2) {
            fprintf(stderr, "oracle: DH-2 has no symmetric mode\n");
            abort();
        }
        const int tan = o->skip_self == 3;
        nf = extend_tiled(a + as, 1, alen - as, b + bs, 1, blen - bs, fwd_first, o, &fi, &fjv, &fdv, fd, fj,
                          &flo, &fhi, cells, tan ? 1 : 0, as - bs);
        nr = extend_tiled(a + as - 1, -1, as, b + bs - 1, -1, bs, rev_first, o, &ri, &rjv, &rdv, rd, rj, &rlo,
                          &rhi, cells, tan ? 2 : 0, as - bs);
    } else {
        nf = extend(a + as, 1, alen - as, b + bs, 1, blen - bs, fwd_first, fwdb_first, o, &fi, &fjv, &fdv, fd,
                    fj, fdb, fib, &nfb, &flo, &fhi, cells);
        nr = extend(a + as - 1, -1, as, b + bs - 1, -1, bs, rev_first, revb_first, o, &ri, &rjv, &rdv, rd, rj,
                    rdb, rib, &nrb, &rlo, &rhi, cells);
    }
    la->abpos = as - ri;
    la->bbpos = bs - rjv;
    la->aepos = as + fi;
    la->bepos = bs + fjv;
    la->diffs = fdv + rdv;
    /* diagonal range (a - b) touched by the path; reverse extension mirrors the sign */
    const int32_t sd = as - bs;
    int32_t lo = sd + flo, hi = sd + fhi;
    if (sd - rhi < lo) lo = sd - rhi;
    if (sd - rlo > hi) hi = sd - rlo;
    *dlo = lo;
    *dhi = hi;
    la->tlen = assemble_trace(ts, 0, as, bs, la->abpos, la->aepos, la->bbpos, la->bepos, rdv, fdv,
                              rev_first, nr, rd, rj, fwd_first, nf, fd, fj, trace);
    if (la2) {
        const int32_t n2 = assemble_trace(ts, resb, bs, as, la->bbpos, la->bepos, la->abpos,
                                          la->aepos, rdv, fdv, revb_first, nrb, rdb, rib, fwdb_first,
                                          nfb, fdb, fib, trace2);
        *la2 = *la;
        la2->tlen = n2;
        if (!comp2) {
            la2->abpos = la->bbpos;
            la2->aepos = la->bepos;
            la2->bbpos = la->abpos;
            la2->bepos = la->aepos;
        } else {
            /* grid = forward strand of the read behind B: mirror both axes, reverse the pairs */
            la2->abpos = blen - la->bepos;
            la2->aepos = blen - la->bbpos;
            la2->bbpos = alen - la->aepos;
            la2->bepos = alen - la->abpos;
            for (int32_t x = 0, y = n2 / 2 - 1; x < y; x++, y--) {
                uint16_t t0 = trace2[2 * x], t1 = trace2[2 * x + 1];
                trace2[2 * x] = trace2[2 * y];
                trace2[2 * x + 1] = trace2[2 * y + 1];
                trace2[2 * y] = t0;
                trace2[2 * y + 1] = t1;
            }
        }
    }
    free(buf);
    return la->aepos > la->abpos;
}

int oz_local_align(const uint8_t *a, int32_t alen, const uint8_t *b, int32_t blen, int32_t as,
                   int32_t bs, const oz_opts *o, oz_la *la, uint16_t *trace, int32_t *dlo,
                   int32_t *dhi, int64_t *cells)
{
    return local_align2(a, alen, b, blen, as, bs, o, la, trace, NULL, NULL, 0, dlo, dhi, cells);
}

/* ------------------------------------------------------------------ whole pass -------- */

typedef struct {
    int32_t aseq, abpos, aepos, bbpos, bepos, dlo, dhi;
} region;

static int la_accept(const oz_la *la, const oz_opts *o)
{
    const int64_t al = la->aepos - la->abpos, bl = la->bepos - la->bbpos;
    if (al < o->min_len) return 0;
    return (int64_t)2 * la->diffs * 1000000 <= (int64_t)o->max_err_ppm * (al + bl);
}

/* out2 (optional, DH-2 and A != B only): the records of the transposed pairs -- `damapper -C`'s second file */
static void align_read(const oz_index *ix, const oz_db *A, const oz_db *B, int32_t r,
                       const oz_opts *o, oz_la_set *out, oz_la_set *out2, int64_t *stats, oz_cand *cands,
                       uint8_t *rc, uint16_t *trace, uint16_t *trace2)
{
    const uint8_t *bf = B->bases + B->off[r];
    const int32_t blen = (int32_t)(B->off[r + 1] - B->off[r]);
    const int32_t bgroup = B->group ? B->group[r] : 0;
    for (int strand = 0; strand < 2; strand++) {
        if (!(o->strands & (1 << strand))) continue;
        const uint8_t *b = bf;
        if (strand) {
            oz_revcomp(bf, blen, rc);
            b = rc;
        }
        int32_t nh = 0;
        /* mask of this B read in the orientation of the strand (mirrored for the complement) */
        const int32_t *bm = NULL;
        int64_t nbm = 0;
        int32_t *tmpm = NULL;
        if (B->mask_ptr) {
            nbm = B->mask_ptr[r + 1] - B->mask_ptr[r];
            bm = B->mask_iv + 2 * B->mask_ptr[r];
            if (strand && nbm > 0) {
                tmpm = (int32_t *)malloc((size_t)nbm * 2 * sizeof(int32_t));
                for (int64_t x = 0; x < nbm; x++) {
                    tmpm[2 * x] = blen - bm[2 * (nbm - 1 - x) + 1];
                    tmpm[2 * x + 1] = blen - bm[2 * (nbm - 1 - x)];
                }
                bm = tmpm;
            }
        }
        const uint8_t *pf = o->skip_self == 2 && o->algo == 1 ? A->pflags : NULL;
        int nc = seed_candidates(ix, b, blen, bgroup, r, bm, nbm, o, cands, &nh, pf);
        free(tmpm);
        stats[0] += nh;
        stats[1] += nc;
        region done[64];
        int nd = 0, nacc = 0;
        /* symmetric mode: max_la is only a capacity (a record may also arrive from its partner) */
        for (int c = 0; c < nc && (o->skip_self == 2 || nacc < o->max_la) && nd < 64; c++) {
            const oz_cand *cd = &cands[c];
            const int32_t sd = cd->apos - cd->bpos;
            int covered = 0;
            for (int t = 0; t < nd; t++)
                if (done[t].aseq == cd->aseq && cd->apos >= done[t].abpos &&
                    cd->apos < done[t].aepos && cd->bpos >= done[t].bbpos &&
                    cd->bpos < done[t].bepos && sd >= done[t].dlo - 64 && sd <= done[t].dhi + 64)
                    covered = 1;
            if (covered) continue;
            const uint8_t *a = A->bases + A->off[cd->aseq];
            const int32_t alen = (int32_t)(A->off[cd->aseq + 1] - A->off[cd->aseq]);
            oz_la la, la2;
            memset(&la, 0, sizeof(la));
            memset(&la2, 0, sizeof(la2));
            int32_t dlo, dhi;
            const int sym = o->skip_self == 2, tiled = o->algo == 1;
            /* DH-1 derives the transposed record from the same path (second trace grid); DH-2 aligns the
             * transposed pair on its own, below, once the first record is in */
            local_align2(a, alen, b, blen, cd->apos, cd->bpos, o, &la, trace, sym && !tiled ? &la2 : NULL,
                         sym && !tiled ? trace2 : NULL, strand, &dlo, &dhi, &stats[3]);
            stats[2]++;
            done[nd].aseq = cd->aseq;
            done[nd].abpos = la.abpos;
            done[nd].aepos = la.aepos;
            done[nd].bbpos = la.bbpos;
            done[nd].bepos = la.bepos;
            done[nd].dlo = dlo;
            done[nd].dhi = dhi;
            nd++;
            if (!la_accept(&la, o)) continue;
            la.aread = cd->aseq;
            la.bread = r;
            la.flags = strand ? OZ_FLAG_COMP : 0;
            if (!sym || rec_wanted(pf, cd->aseq, r)) la_set_push(out, &la, trace);
            if ((sym || out2) && tiled && (!sym || rec_wanted(pf, r, cd->aseq))) {
                /* DH-2, symmetric (or the transposed file of a mapping): the record (b, a) is the tiled alignment of the transposed pair through the
                 * same seed -- A'' = the read behind B on its forward strand (the trace grid of that record),
                 * B'' = the A read, complemented when B is (then both axes are mirrored: the seed point
                 * (as, bs) becomes (blen - bs, alen - as)).  It is accepted on its own length and error. */
                uint8_t *a2 = (uint8_t *)malloc((size_t)blen + 1), *b2 = (uint8_t *)malloc((size_t)alen + 1);
                if (strand) {
                    oz_revcomp(b, blen, a2);
                    oz_revcomp(a, alen, b2);
                } else {
                    memcpy(a2, b, (size_t)blen);
                    memcpy(b2, a, (size_t)alen);
                }
                int32_t d2lo, d2hi;
                local_align2(a2, blen, b2, alen, strand ? blen - cd->bpos : cd->bpos, strand ? alen - cd->apos : cd->apos,
                             o, &la2, trace2, NULL, NULL, 0, &d2lo, &d2hi, &stats[3]);
                free(a2);
                free(b2);
                if (la_accept(&la2, o)) {
                    la2.aread = r;
                    la2.bread = cd->aseq;
                    la2.flags = la.flags;
                    la_set_push(sym ? out : out2, &la2, trace2);
                }
            } else if (sym) { /* the transposed record of the same alignment: A and B swapped */
                la2.aread = r;
                la2.bread = cd->aseq;
                la2.flags = la.flags;
                la_set_push(out, &la2, trace2);
            }
            nacc++;
        }
    }
}

int oz_align_db(const oz_db *A, const oz_db *B, const oz_opts *o, int nthreads, oz_la_set *out,
                int64_t *stats)
{
    return oz_align_db2(A, B, o, nthreads, out, NULL, stats);
}

int oz_align_db2(const oz_db *A, const oz_db *B, const oz_opts *o, int nthreads, oz_la_set *out, oz_la_set *out2,
                 int64_t *stats)
{
    if (out2 && (o->algo != 1 || o->skip_self == 2)) return -1; /* the transposed file is defined for DH-2 mappings */
    if (o->skip_self == 3 && (o->algo != 1 || o->strands != 1)) return -1; /* tandem: DH-2, forward strand (A and B: the same DB) */
    int32_t max_blen = 0, max_alen = 0;
    for (int32_t r = 0; r < B->n; r++) {
        int32_t l = (int32_t)(B->off[r + 1] - B->off[r]);
        if (l > max_blen) max_blen = l;
    }
    for (int32_t s = 0; s < A->n; s++) {
        int32_t l = (int32_t)(A->off[s + 1] - A->off[s]);
        if (l > max_alen) max_alen = l;
    }
    oz_index *ix = index_build(A, o, max_blen);
    int64_t st[4] = {0, 0, 0, 0};
    if (nthreads < 1) nthreads = 1;
    oz_la_set *parts = (oz_la_set *)calloc((size_t)nthreads, sizeof(oz_la_set));
    oz_la_set *parts2 = (oz_la_set *)calloc((size_t)nthreads, sizeof(oz_la_set));
    int64_t(*pst)[4] = (int64_t(*)[4])calloc((size_t)nthreads, sizeof(int64_t[4]));
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
    {
#ifdef _OPENMP
        const int tid = omp_get_thread_num(), nt = omp_get_num_threads();
#else
        const int tid = 0, nt = 1;
#endif
        oz_cand *cands = (oz_cand *)malloc((size_t)(o->max_cand + 1) * sizeof(oz_cand));
        uint8_t *rc = (uint8_t *)malloc((size_t)max_blen + 1);
        const int32_t mlen = max_alen > max_blen ? max_alen : max_blen;
        uint16_t *trace = (uint16_t *)malloc((size_t)(2 * (mlen / o->tspace + 4)) * sizeof(uint16_t));
        uint16_t *trace2 = (uint16_t *)malloc((size_t)(2 * (mlen / o->tspace + 4)) * sizeof(uint16_t));
        /* contiguous read ranges per thread keep the merged output in read order */
        const int64_t lo = (int64_t)B->n * tid / nt, hi = (int64_t)B->n * (tid + 1) / nt;
        for (int64_t r = lo; r < hi; r++)
            align_read(ix, A, B, (int32_t)r, o, &parts[tid], out2 ? &parts2[tid] : NULL, pst[tid], cands, rc, trace, trace2);
        free(cands);
        free(rc);
        free(trace);
        free(trace2);
    }
    for (int t = 0; t < nthreads; t++) {
        for (int64_t i = 0; i < parts[t].n; i++)
            la_set_push(out, &parts[t].la[i], parts[t].trace + parts[t].la[i].toff);
        for (int64_t i = 0; out2 && i < parts2[t].n; i++)
            la_set_push(out2, &parts2[t].la[i], parts2[t].trace + parts2[t].la[i].toff);
        for (int q = 0; q < 4; q++) st[q] += pst[t][q];
        oz_la_set_free(&parts[t]);
        oz_la_set_free(&parts2[t]);
    }
    free(parts);
    free(parts2);
    free(pst);
    if (stats)
        for (int q = 0; q < 4; q++) stats[q] = st[q];
    oz_index_free(ix);
    return 0;
}

/*
 * damapper's chain flags per B read (consumer semantics: source/dentist/dazzler.d:1728-1758, flags :1991-1998).
 * The LAs of a read on one contig and strand, ordered as LAsort orders them, are linked into chains: an LA continues
 * its predecessor's chain when it follows it on both sequences (at most 100 bases of overlap, both ends further on),
 * the gaps are at most 10 000 on either sequence and differ by at most 6 000 (a long indel splits a mapping into
 * collinear LAs).  Chain score = sum(alen - 2 diffs).  A chain is BEST unless a higher-scoring chain of the same strand
 * (ties: the one whose first LA sorts later) covers more than half of its B span.  START on the first LA, NEXT on the
 * others, BEST on all LAs of a best chain; near_best_ppm > 0 (damapper -n): an alternate chain below that fraction of
 * the chain that beats it is DISABLED.  Expects LAs grouped by bread (any order inside the group).
 */
static int32_t oz_near_best_ppm = 0;
void oz_set_near_best(int32_t ppm) { oz_near_best_ppm = ppm < 0 ? 0 : ppm; }

typedef struct {
    int64_t score;
    int32_t bb, be, comp, n;
    int64_t first, last;
    int64_t mem0; /* index of the first member in the members array */
} oz_chain;

static const oz_la *cmp_base_;
static int idx_cmp_(const void *x, const void *y)
{
    return la_cmp(&cmp_base_[*(const int64_t *)x], &cmp_base_[*(const int64_t *)y]);
}

void oz_select_best(oz_la_set *s)
{
    int64_t g0 = 0;
    while (g0 < s->n) {
        int64_t g1 = g0;
        while (g1 < s->n && s->la[g1].bread == s->la[g0].bread) g1++;
        const int64_t m = g1 - g0;
        int64_t *ord = (int64_t *)malloc((size_t)m * sizeof(int64_t));
        int64_t *mem = (int64_t *)malloc((size_t)m * sizeof(int64_t));
        oz_chain *ch = (oz_chain *)malloc((size_t)m * sizeof(oz_chain));
        for (int64_t x = 0; x < m; x++) ord[x] = g0 + x;
        cmp_base_ = s->la;
        qsort(ord, (size_t)m, sizeof(int64_t), idx_cmp_);
        int64_t nc = 0, nm = 0;
        for (int64_t k = 0; k < m; k++) {
            const oz_la *q = &s->la[ord[k]];
            int linked = 0;
            if (nc > 0) {
                oz_chain *c = &ch[nc - 1];
                const oz_la *p = &s->la[c->last];
                const int64_t ga = (int64_t)q->abpos - p->aepos, gb = (int64_t)q->bbpos - p->bepos;
                const int64_t dg = ga > gb ? ga - gb : gb - ga;
                linked = p->aread == q->aread && (p->flags & OZ_FLAG_COMP) == (q->flags & OZ_FLAG_COMP) && ga >= -100 &&
                         gb >= -100 && ga <= 10000 && gb <= 10000 && dg <= 6000 && q->aepos > p->aepos && q->bepos > p->bepos;
                if (linked) {
                    mem[nm++] = ord[k];
                    c->n++;
                    c->last = ord[k];
                    c->score += (int64_t)(q->aepos - q->abpos) - 2 * (int64_t)q->diffs;
                    c->be = q->bepos;
                }
            }
            if (!linked) {
                oz_chain *c = &ch[nc++];
                c->score = (int64_t)(q->aepos - q->abpos) - 2 * (int64_t)q->diffs;
                c->bb = q->bbpos;
                c->be = q->bepos;
                c->comp = (int32_t)(q->flags & OZ_FLAG_COMP);
                c->n = 1;
                c->first = c->last = ord[k];
                c->mem0 = nm;
                mem[nm++] = ord[k];
            }
        }
        for (int64_t x = 0; x < nc; x++) {
            const oz_chain *p = &ch[x];
            int best = 1, drop = 0;
            for (int64_t y = 0; y < nc; y++) {
                if (x == y) continue;
                const oz_chain *q = &ch[y];
                if (q->score < p->score || (q->score == p->score && la_cmp(&s->la[q->first], &s->la[p->first]) < 0)) continue;
                if (q->comp != p->comp) continue;
                const int32_t lo = p->bb > q->bb ? p->bb : q->bb, hi = p->be < q->be ? p->be : q->be;
                if (hi - lo > (p->be - p->bb) / 2) {
                    best = 0;
                    if (oz_near_best_ppm > 0 && p->score * 1000000ll < (int64_t)oz_near_best_ppm * q->score) drop = 1;
                }
            }
            for (int32_t k = 0; k < p->n; k++) {
                oz_la *l = &s->la[mem[p->mem0 + k]];
                l->flags &= ~(OZ_FLAG_START | OZ_FLAG_NEXT | OZ_FLAG_BEST);
                l->flags |= (k == 0 ? OZ_FLAG_START : OZ_FLAG_NEXT) | (best ? OZ_FLAG_BEST : 0u) | (drop ? OZ_FLAG_DISABLED : 0u);
            }
        }
        {
            /* the records of the read in LAsort order: chain members become neighbours (START, then its NEXT records) */
            oz_la *tmp = (oz_la *)malloc((size_t)m * sizeof(oz_la));
            for (int64_t x = 0; x < m; x++) tmp[x] = s->la[ord[x]];
            memcpy(s->la + g0, tmp, (size_t)m * sizeof(oz_la));
            free(tmp);
        }
        free(ord);
        free(mem);
        free(ch);
        g0 = g1;
    }
}
