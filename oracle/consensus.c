/*
 * consensus.c -- oracle: tile QVs, reference-read ranking and pile-up consensus.
 * TEST INFRASTRUCTURE ONLY (see dh_oracle.h).
 *
 * Pinned by the reference's own D code:
 *   - pile-up alignment validity          source/dentist/dazzler.d:4108-4141
 *   - reference-read ranking (badQV etc.) source/dentist/commands/processPileUps/package.d:518-568
 * PARITY UNPINNED (tools absent from /root/reference, restated from their published idea):
 *   - tile QV   = DASqv role   (call site dazzler.d:6142-6156): per trace tile of an A read, the
 *     mean of the lowest `cov` per-overlap error percentages, capped at 50;
 *   - consensus = daccord role (call site dazzler.d:6185-6231, getConsensus :4213-4255): every
 *     overlap of the reference read is re-aligned tile by tile with the Needleman-Wunsch of
 *     util/string.d (oz_nw), each column of the reference read is decided by majority over
 *     {base, deletion}, insertion slots by majority over the covering reads; uncovered
 *     stretches keep the reference read (daccord -f).
 */
#include "dh_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* OZ_MAXQV = DbRecord.maxQV, source/dentist/dazzler.d:2873 */

/* isValidPileUpAlignment for flat LAs, source/dentist/dazzler.d:4126-4141 */
int oz_valid_pileup_alignment(const oz_la *la, int32_t alen, int32_t blen, int32_t allowance)
{
    const int ab = la->abpos <= allowance, bb = la->bbpos <= allowance;
    const int ae = la->aepos + allowance >= alen, be = la->bepos + allowance >= blen;
    const int left_anchored = ab && bb, left_proper = ab || bb;
    const int right_anchored = ae && be, right_proper = ae || be;
    return la->aread != la->bread &&
           ((left_anchored && right_proper) || (right_anchored && left_proper));
}

static int i32_cmp(const void *x, const void *y)
{
    int32_t p = *(const int32_t *)x, q = *(const int32_t *)y;
    return p < q ? -1 : (p > q ? 1 : 0);
}

/*
 * qv[r * maxtiles + t] for every read r (length rlen[r]) and tile t < ceil(rlen/ts).
 * Only LAs with the DISABLED flag clear are used.  A tile counts for an LA when the LA covers
 * the whole tile; its value is floor(200 * diffs / (tile_len + bbases)).
 */
void oz_tile_qv(const oz_la_set *s, int32_t nreads, const int32_t *rlen, int32_t tspace,
                int32_t cov, uint8_t *qv, int32_t maxtiles)
{
    int32_t *vals = (int32_t *)malloc((size_t)(s->n ? s->n : 1) * sizeof(int32_t));
    for (int32_t r = 0; r < nreads; r++) {
        const int32_t nt = (rlen[r] + tspace - 1) / tspace;
        for (int32_t t = 0; t < nt && t < maxtiles; t++) {
            const int32_t t0 = t * tspace, t1 = (t0 + tspace < rlen[r]) ? t0 + tspace : rlen[r];
            int32_t m = 0;
            for (int64_t i = 0; i < s->n; i++) {
                const oz_la *la = &s->la[i];
                if (la->aread != r || (la->flags & OZ_FLAG_DISABLED)) continue;
                if (la->abpos > t0 || la->aepos < t1) continue;
                const int32_t e = t - la->abpos / tspace; /* trace segment of this tile */
                const uint16_t *tr = s->trace + la->toff;
                /* the segment is the whole tile only if the LA does not start/end inside it */
                const int32_t seg0 = e == 0 ? la->abpos : t0;
                const int32_t seg1 = (e == la->tlen / 2 - 1) ? la->aepos : t1;
                if (seg0 != t0 || seg1 != t1) continue;
                vals[m++] = 200 * (int32_t)tr[2 * e] / ((t1 - t0) + (int32_t)tr[2 * e + 1]);
            }
            int32_t q = OZ_MAXQV;
            if (m > 0) {
                qsort(vals, (size_t)m, sizeof(int32_t), i32_cmp);
                const int32_t use = m < cov ? m : cov;
                int64_t sum = 0;
                for (int32_t x = 0; x < use; x++) sum += vals[x];
                q = (int32_t)(sum / use);
                if (q > OZ_MAXQV) q = OZ_MAXQV;
            }
            qv[(size_t)r * maxtiles + t] = (uint8_t)q;
        }
    }
    free(vals);
}

/*
 * findReferenceReadCandidates, processPileUps/package.d:518-568: histogram of QVs < maxQV over
 * the allowed reads, badQV at `bad_fraction`, reads ordered by (numBadQVs, meanQV, id).
 * order[] receives read indices best first; returns badQV.
 */
typedef struct {
    int64_t nbad;
    double mean;
    int32_t id;
} rank_ent;

static int rank_cmp(const void *x, const void *y)
{
    const rank_ent *p = (const rank_ent *)x, *q = (const rank_ent *)y;
    if (p->nbad != q->nbad) return p->nbad < q->nbad ? -1 : 1;
    if (p->mean != q->mean) return p->mean < q->mean ? -1 : 1;
    return p->id < q->id ? -1 : (p->id > q->id ? 1 : 0);
}

int32_t oz_rank_reference_reads(const uint8_t *qv, int32_t nreads, const int32_t *rlen,
                                int32_t tspace, int32_t maxtiles, const uint8_t *allowed,
                                double bad_fraction, int32_t *order, int32_t *norder)
{
    int64_t hist[OZ_MAXQV];
    int64_t total = 0;
    memset(hist, 0, sizeof(hist));
    for (int32_t r = 0; r < nreads; r++) {
        if (allowed && !allowed[r]) continue;
        const int32_t nt = (rlen[r] + tspace - 1) / tspace;
        for (int32_t t = 0; t < nt; t++) {
            const int32_t q = qv[(size_t)r * maxtiles + t];
            if (q < OZ_MAXQV) {
                hist[q]++;
                total++;
            }
        }
    }
    const int64_t bad_thres = (int64_t)(bad_fraction * (double)total);
    int32_t idx = -1;
    int64_t cum = 0;
    for (int32_t x = 0; x < OZ_MAXQV; x++) { /* cumulativeFold over hist.retro, countUntil >= */
        cum += hist[OZ_MAXQV - 1 - x];
        if (cum >= bad_thres) {
            idx = x;
            break;
        }
    }
    /* countUntil returns -1 when nothing matches: badQV = maxQV - 1 - (-1) = maxQV */
    const int32_t bad_qv = OZ_MAXQV - 1 - idx;
    rank_ent *e = (rank_ent *)malloc((size_t)(nreads ? nreads : 1) * sizeof(rank_ent));
    int32_t n = 0;
    for (int32_t r = 0; r < nreads; r++) {
        if (allowed && !allowed[r]) continue;
        const int32_t nt = (rlen[r] + tspace - 1) / tspace;
        int64_t nb = 0, sum = 0;
        for (int32_t t = 0; t < nt; t++) {
            const int32_t q = qv[(size_t)r * maxtiles + t];
            if (q >= bad_qv) nb++;
            sum += q;
        }
        e[n].nbad = nb;
        e[n].mean = nt > 0 ? (double)sum / (double)nt : 0.0;
        e[n].id = r;
        n++;
    }
    qsort(e, (size_t)n, sizeof(rank_ent), rank_cmp);
    for (int32_t i = 0; i < n; i++) order[i] = e[i].id;
    *norder = n;
    free(e);
    return bad_qv;
}

/*
 * Consensus of reference sequence `ref` from the LAs of `s` whose aread == aidx and that are
 * not DISABLED.  B sequences come from `reads` (reverse-complemented for COMP LAs).
 * out must hold rlen * (1 + OZ_MAXINS) bases.  votes (optional, may be NULL) receives the raw
 * vote table: per column 4 base votes, 1 deletion vote, 1 cover count, OZ_MAXINS*4 insertion
 * votes = 22 uint32.
 */
int32_t oz_consensus(const uint8_t *ref, int32_t rlen, const oz_db *reads, const oz_la_set *s,
                     int32_t aidx, int32_t tspace, uint8_t *out, uint32_t *votes_out)
{
    uint32_t *v = (uint32_t *)calloc((size_t)(rlen + 1) * OZ_VOTE_STRIDE, sizeof(uint32_t));
    int32_t maxb = 0;
    for (int32_t r = 0; r < reads->n; r++) {
        const int32_t l = (int32_t)(reads->off[r + 1] - reads->off[r]);
        if (l > maxb) maxb = l;
    }
    uint8_t *rc = (uint8_t *)malloc((size_t)maxb + 1);
    uint8_t *ops = (uint8_t *)malloc((size_t)(tspace + 70000));
    uint8_t *colst = (uint8_t *)malloc((size_t)tspace + 1);
    int32_t *icnt = (int32_t *)malloc((size_t)(tspace + 1) * sizeof(int32_t));
    uint8_t *ibase = (uint8_t *)malloc((size_t)(tspace + 1) * OZ_MAXINS);
    for (int64_t i = 0; i < s->n; i++) {
        const oz_la *la = &s->la[i];
        if (la->aread != aidx || (la->flags & OZ_FLAG_DISABLED)) continue;
        const uint8_t *b = reads->bases + reads->off[la->bread];
        const int32_t blen = (int32_t)(reads->off[la->bread + 1] - reads->off[la->bread]);
        if (la->flags & OZ_FLAG_COMP) {
            oz_revcomp(b, blen, rc);
            b = rc;
        }
        const uint16_t *tr = s->trace + la->toff;
        /* an overlap with a tile spanning more than OZ_SEG_MAX B bases (a > 100 % local indel rate)
         * takes no part in the vote */
        int too_long = 0;
        for (int32_t e = 0; e < la->tlen / 2; e++)
            if (tr[2 * e + 1] > OZ_SEG_MAX) too_long = 1;
        if (too_long) continue;
        int32_t a0 = la->abpos, b0 = la->bbpos;
        for (int32_t e = 0; e < la->tlen / 2; e++) {
            int32_t a1 = (a0 / tspace + 1) * tspace;
            if (a1 > la->aepos) a1 = la->aepos;
            const int32_t b1 = b0 + tr[2 * e + 1];
            int32_t nops = 0;
            oz_nw(ref + a0, a1 - a0, b + b0, b1 - b0, 1, 0, ops, &nops);
            /* per-column view of the segment: col[x-a0] = aligned B base (0..4) or 5 = deleted;
             * ins[slot][0..n) = bases inserted before column slot (slot a1-a0 = after the last) */
            const int32_t w = a1 - a0;
            memset(icnt, 0, (size_t)(w + 1) * sizeof(int32_t));
            {
                int32_t x = 0, y = b0;
                for (int32_t t = 0; t < nops; t++) {
                    if (ops[t] == OZ_OP_SUB) {
                        colst[x++] = b[y++];
                    } else if (ops[t] == OZ_OP_DEL) {
                        colst[x++] = 5;
                    } else {
                        if (icnt[x] < OZ_MAXINS) ibase[(size_t)x * OZ_MAXINS + icnt[x]] = b[y];
                        icnt[x]++;
                        y++;
                    }
                }
            }
            /* canonical (leftmost) placement of indels inside homopolymer runs of the reference:
             * a deleted column moves left across columns that are exact matches of the same
             * base; a run of inserted copies of base c moves left across exact matches of c.
             * Equivalent alignments then vote in the same column. */
            for (int32_t x = 0; x < w; x++) {
                if (colst[x] != 5) continue;
                const uint8_t c = ref[a0 + x];
                int32_t st = x;
                while (st > 0 && colst[st - 1] == c && ref[a0 + st - 1] == c && icnt[st] == 0) st--;
                if (st < x) {
                    colst[st] = 5;
                    colst[x] = c;
                }
            }
            for (int32_t x = 1; x <= w; x++) {
                const int32_t n = icnt[x];
                if (n == 0 || n > OZ_MAXINS) continue;
                const uint8_t c = ibase[(size_t)x * OZ_MAXINS];
                int same = c < 4;
                for (int32_t t = 1; t < n; t++) same = same && ibase[(size_t)x * OZ_MAXINS + t] == c;
                if (!same) continue;
                int32_t st = x;
                while (st > 0 && colst[st - 1] == c && ref[a0 + st - 1] == c && icnt[st - 1] == 0) st--;
                if (st < x) {
                    for (int32_t t = 0; t < n; t++) ibase[(size_t)st * OZ_MAXINS + t] = c;
                    icnt[st] = n;
                    icnt[x] = 0;
                }
            }
            for (int32_t x = 0; x <= w; x++) {
                uint32_t *col = v + (size_t)(a0 + x) * OZ_VOTE_STRIDE;
                const int32_t n = icnt[x] < OZ_MAXINS ? icnt[x] : OZ_MAXINS;
                for (int32_t t = 0; t < n; t++) {
                    const uint8_t c = ibase[(size_t)x * OZ_MAXINS + t];
                    if (c < 4) col[6 + 4 * t + c]++;
                }
                if (x == w) break;
                if (colst[x] == 5)
                    col[4]++;
                else if (colst[x] < 4)
                    col[colst[x]]++;
                col[5]++;
            }
            a0 = a1;
            b0 = b1;
        }
    }
    /* Emission, one homopolymer run [rs, re) of the reference at a time.  With indels in
     * canonical (leftmost) position, all reads that see a shorter/longer run vote inside the
     * run; the run length changes by round(net / (cover + 1)) where net = deletion votes minus
     * votes for inserted copies of the run's base.  Columns whose winning base differs from the
     * run's base are emitted in place; foreign-base insertion slots use a plain majority. */
    int32_t n = 0;
    for (int32_t rs = 0; rs < rlen;) {
        int32_t re = rs + 1;
        while (re < rlen && ref[re] == ref[rs]) re++;
        const uint8_t c = ref[rs];
        const int64_t den = (int64_t)v[(size_t)rs * OZ_VOTE_STRIDE + 5] + 1;
        int64_t net = 0;
        int32_t ncols = 0;
        for (int32_t x = rs; x < re; x++) {
            const uint32_t *col = v + (size_t)x * OZ_VOTE_STRIDE;
            net += col[4];
            if (c < 4)
                for (int t = 0; t < OZ_MAXINS; t++) net -= col[6 + 4 * t + c];
        }
        if (c < 4 && re < rlen) /* slot after the run: inserted copies of c that could not move */
            for (int t = 0; t < OZ_MAXINS; t++) net -= v[(size_t)re * OZ_VOTE_STRIDE + 6 + 4 * t + c];
        int64_t adj = net >= 0 ? (2 * net + den) / (2 * den) : -((2 * (-net) + den) / (2 * den));
        /* winning base per column (deletions are handled at run level) */
        for (int32_t x = rs; x < re; x++) {
            const uint32_t *col = v + (size_t)x * OZ_VOTE_STRIDE;
            int best = c < 4 ? c : 0;
            uint32_t bv[4];
            for (int k = 0; k < 4; k++) bv[k] = col[k] + ((c == k) ? 1u : 0u);
            for (int k = 0; k < 4; k++)
                if (bv[k] > bv[best]) best = k;
            if (best == c) ncols++;
        }
        int64_t target = (int64_t)ncols - adj;
        if (target < 0) target = 0;
        if (target > ncols + OZ_MAXINS) target = ncols + OZ_MAXINS;
        int64_t extra = target > ncols ? target - ncols : 0, keep = target < ncols ? target : ncols;
        for (int32_t x = rs; x < re; x++) {
            const uint32_t *col = v + (size_t)x * OZ_VOTE_STRIDE;
            const uint32_t cover = col[5];
            /* foreign-base insertions before column x (copies of c are part of the run vote;
             * at the first column of the run copies of the previous run's base were counted
             * there) */
            const uint8_t pc = (x == rs && rs > 0) ? ref[rs - 1] : 255;
            for (int t = 0; t < OZ_MAXINS; t++) {
                const uint32_t *iv = col + 6 + 4 * t;
                uint32_t tot = 0;
                int best = -1;
                for (int k = 0; k < 4; k++) {
                    if (k == c || k == pc) continue;
                    tot += iv[k];
                    if (best < 0 || iv[k] > iv[best]) best = k;
                }
                if (best < 0 || 2 * tot <= cover + 1) break;
                out[n++] = (uint8_t)best;
            }
            int best = c < 4 ? c : 0;
            uint32_t bv[4];
            for (int k = 0; k < 4; k++) bv[k] = col[k] + ((c == k) ? 1u : 0u);
            for (int k = 0; k < 4; k++)
                if (bv[k] > bv[best]) best = k;
            if (best != c) {
                if (2 * col[4] <= cover + 1) out[n++] = (uint8_t)best;
                continue;
            }
            if (x == rs)
                for (int64_t t = 0; t < extra; t++) out[n++] = c;
            if (keep > 0) {
                out[n++] = c;
                keep--;
            }
        }
        rs = re;
    }
    if (votes_out) memcpy(votes_out, v, (size_t)rlen * OZ_VOTE_STRIDE * sizeof(uint32_t));
    free(ibase);
    free(icnt);
    free(colst);
    free(ops);
    free(rc);
    free(v);
    return n;
}
