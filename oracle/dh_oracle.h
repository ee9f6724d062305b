/*
 * dh_oracle.h -- CPU restatement ("oracle") of DENTIST's alignment + consensus hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked, imported or executed by the
 * product path (dentist_amd/, libdentist_hip.so).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may use it -- as the checker / the timed CPU baseline.
 *
 * Parity status (see DESIGN.md "Oracle"):
 *   PINNED against reference-owned vectors:
 *     - .las codec                 source/dentist/dazzler.d:1665-1834, 1913-1960, 1988-2032, 2130-2170
 *                                  goldens dazzler.d:962-1166 (testLasDump)
 *     - trace-point translation    source/dentist/common/alignments/base.d:169-299, goldens :883-944, 993-1129
 *     - Needleman-Wunsch           source/dentist/util/string.d:478-520, 775-831, goldens :523-751
 *     - gap closing fixture        tests/test-commands.sh:17-44, 62-65 (md5 of gap-closed.fasta)
 *     - 3-read consensus           source/dentist/dazzler.d:4257-4299
 *   PARITY UNPINNED (restated from the public literature, the arithmetic lives in third-party
 *   tools that are absent from /root/reference -- DALIGNER c2b47da, DAMAPPER b2c9d7f,
 *   daccord 0.0.18 / libmaus2 2.0.724, DASCRUBBER a53dbe8, DAZZ_DB d22ae58):
 *     - k-mer seeding + diagonal band filter, O(ND) wave local alignment with trace points
 *       (call sites dazzler.d:6121-6170), tile QV (dazzler.d:6142-6156), pile-up consensus
 *       (dazzler.d:6172-6231).
 */
#ifndef DH_ORACLE_H
#define DH_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------- sequences (DAZZ_DB code order a,c,g,t = 0..3; anything else = 4) ---------- */

typedef struct {
    int32_t n;            /* number of sequences                                   */
    const int64_t *off;   /* off[n+1] offsets into bases                           */
    const uint8_t *bases; /* codes 0..4                                            */
    const int32_t *group; /* optional group id per sequence (NULL = all group 0)   */
    /* optional soft mask (daligner -m tracks): intervals mask_iv[2*j], mask_iv[2*j+1] for
     * j in [mask_ptr[s], mask_ptr[s+1]) of sequence s, sorted, disjoint; k-mers touching a
     * masked interval are neither indexed nor looked up */
    const int64_t *mask_ptr;
    const int32_t *mask_iv;
    /* optional, symmetric all-vs-all (skip_self == 2, algo 1) only: pflags[s] bit 0 = records with sequence s as A read
     * are wanted, bit 1 = s may be the B read of a wanted record.  Record (a, b) is wanted iff (pflags[a] & 1) &&
     * (pflags[b] & 2); a pair yields hits iff one of its two records is wanted, an unwanted record is not emitted.  The
     * process drivers set it for the pile-up all-vs-all (allowed reference reads as A, dh_process_opts.max_partners for
     * B); NULL = every record. */
    const uint8_t *pflags;
} oz_db;

void oz_encode(const char *ascii, int64_t n, uint8_t *codes);
void oz_decode(const uint8_t *codes, int64_t n, char *ascii);
void oz_revcomp(const uint8_t *src, int64_t n, uint8_t *dst);

/* ---------- alignment options (all integer so CPU and GPU agree bit for bit) ---------- */

typedef struct {
    int32_t k;           /* k-mer length (daligner -k, default 14)                         */
    int32_t hmin;        /* min covered bases in a band pair (daligner -h, default 35)     */
    int32_t band_shift;  /* log2 band width (daligner -w, default 6)                       */
    int32_t tspace;      /* trace point spacing (daligner -s: 100 mapping, 126 pile-ups)   */
    int32_t min_len;     /* min A-length of a reported LA (daligner -l)                    */
    int32_t pen;         /* wave score penalty per difference: floor(2 / (1 - e))          */
    int32_t xdrop;       /* wave is trimmed to points within xdrop of the best score       */
    int32_t max_err_ppm; /* 2*diffs*1e6 <= max_err_ppm * (alen + blen)  (1 - e)            */
    int32_t max_cand;    /* seed candidates kept per (B read, strand)                      */
    int32_t max_la;      /* LAs reported per (B read, strand)                              */
    int32_t tcap;        /* k-mers occurring more than tcap times in A are ignored (-t)    */
    int32_t strands;     /* bit0: forward B, bit1: reverse-complement B                    */
    int32_t skip_self;   /* A and B are the same DB: 1 = skip aread == bread (no -I);
                          * 2 = symmetric: every unordered pair is aligned once (the smaller id
                          *     is A when a + b is even, else B) and both records (a,b), (b,a)
                          *     are emitted from that alignment;
                          * 3 = tandem (datander): a read is aligned with ITSELF only, below the main diagonal (seeds
                          *     with a > b; cells with b >= a never match).  DH-2 (algo 1), forward strand (strands 1) */
    int32_t dmax;        /* hard cap on differences per extension                          */
    int32_t width;       /* max live diagonals of the wave (64 = one wavefront)            */
    int32_t kmer_mod;    /* modimer sampling (daligner -%): only k-mers with hash % kmer_mod == 0, 1 = all */
    int32_t algo;        /* extension: 0 = DH-1 (O(ND) wave), 1 = DH-2 (tiled banded DP, band = width in {32, 64}) */
} oz_opts;

void oz_default_opts(oz_opts *o);

/* ---------- local alignments (the .las record, source/dentist/dazzler.d:1988-2032) ---------- */

#define OZ_FLAG_COMP 0x1u
#define OZ_FLAG_START 0x4u
#define OZ_FLAG_NEXT 0x8u
#define OZ_FLAG_BEST 0x10u
#define OZ_FLAG_DISABLED 0x20u

typedef struct {
    int32_t tlen, diffs, abpos, bbpos, aepos, bepos;
    uint32_t flags;
    int32_t aread, bread; /* 0-based, as on disk */
    int32_t pad;
    int64_t toff;         /* offset (in u16 units) of this LA's trace in the trace array */
} oz_la;

typedef struct {
    int64_t n, cap;
    oz_la *la;
    int64_t tn, tcap;
    uint16_t *trace; /* (diffs, bbases) pairs */
} oz_la_set;

void oz_la_set_init(oz_la_set *s);
void oz_la_set_free(oz_la_set *s);
void oz_la_set_sort(oz_la_set *s); /* LAsort order, base.d:1787-1809 */

/* seed candidate produced by the k-mer band filter */
typedef struct {
    int32_t score;  /* covered bases of the band pair */
    int32_t aseq;
    int32_t apos;
    int32_t bpos;
    int64_t band;
} oz_cand;

typedef struct oz_index oz_index;
oz_index *oz_index_build(const oz_db *A, const oz_opts *o);
void oz_index_free(oz_index *ix);
int64_t oz_index_size(const oz_index *ix);

/* k-mer hits + band filter for one B sequence (already strand-oriented). Returns #cands. */
int oz_seed_candidates(const oz_index *ix, const oz_db *A, const uint8_t *b, int32_t blen,
                       int32_t bgroup, int32_t bself, int32_t sepv, const oz_opts *o,
                       oz_cand *out, int32_t *nhits_out);

/* one local alignment through seed (as, bs); returns 1 and fills la/trace (trace capacity
 * >= 2*(alen/tspace+3)) when an alignment was produced (not yet filtered for length/error). */
int oz_local_align(const uint8_t *a, int32_t alen, const uint8_t *b, int32_t blen, int32_t as,
                   int32_t bs, const oz_opts *o, oz_la *la, uint16_t *trace, int32_t *dlo,
                   int32_t *dhi, int64_t *cells);

/* whole pass: every B read against the index of A (daligner / damapper role) */
int oz_align_db(const oz_db *A, const oz_db *B, const oz_opts *o, int nthreads, oz_la_set *out,
                int64_t *stats /* [0]=hits [1]=cands [2]=alignments [3]=wave cells */);
/* + out2 (may be NULL): for DH-2 mappings (algo 1, A != B) the records of the transposed pairs, (aread = read,
 * bread = contig): the tiled alignment of A'' = the read on its forward strand, B'' = the contig (complemented for a
 * reverse-strand mapping) through the same seed, accepted on its own -- the second file of `damapper -C`
 * (source/dentist/dazzler.d:6158-6170, getLasFile :4339-4354) */
int oz_align_db2(const oz_db *A, const oz_db *B, const oz_opts *o, int nthreads, oz_la_set *out, oz_la_set *out2,
                 int64_t *stats);

/* damapper-style per-read selection: sets START/BEST flags (dazzler.d:1728-1758 consumer) */
void oz_select_best(oz_la_set *s);
void oz_set_near_best(int32_t ppm); /* damapper -n as parts per million, 0 = keep every chain */

/* ---------- .las codec ---------- */
int oz_las_write(const char *path, const oz_la_set *s, int32_t tspace);
int oz_las_read(const char *path, oz_la_set *s, int32_t *tspace);

/* ---------- trace semantics (base.d:169-299) ---------- */
#define OZ_FLOOR 0
#define OZ_CEIL 1
int32_t oz_trace_points_up_to_a(int32_t abpos, int32_t aepos, int32_t tspace, int32_t ntp,
                                int32_t apos, int mode);
int32_t oz_trace_points_up_to_b(int32_t bbpos, int32_t bepos, const uint16_t *trace, int32_t ntp,
                                int32_t bpos, int mode);
void oz_translate_trace_point_a(int32_t abpos, int32_t aepos, int32_t bbpos, int32_t tspace,
                                const uint16_t *trace, int32_t ntp, int32_t apos, int mode,
                                int32_t *outa, int32_t *outb);

/* ---------- Needleman-Wunsch (util/string.d:478-520 + traceback :775-831) ---------- */
#define OZ_OP_SUB 0
#define OZ_OP_DEL 1
#define OZ_OP_INS 2
/* returns score; ops (capacity rlen+qlen) receives the edit path, *nops its length */
uint32_t oz_nw(const uint8_t *ref, int32_t rlen, const uint8_t *qry, int32_t qlen,
               uint32_t indel, int free_shift, uint8_t *ops, int32_t *nops);

/* ---------- pile-up consensus path (consensus.c) ---------- */
#define OZ_MAXQV 50
#define OZ_MAXINS 4
#define OZ_VOTE_STRIDE (6 + 4 * OZ_MAXINS)
#define OZ_SEG_MAX 250 /* longest B stretch of one trace tile the consensus vote accepts */
int oz_valid_pileup_alignment(const oz_la *la, int32_t alen, int32_t blen, int32_t allowance);
void oz_tile_qv(const oz_la_set *s, int32_t nreads, const int32_t *rlen, int32_t tspace,
                int32_t cov, uint8_t *qv, int32_t maxtiles);
int32_t oz_rank_reference_reads(const uint8_t *qv, int32_t nreads, const int32_t *rlen,
                                int32_t tspace, int32_t maxtiles, const uint8_t *allowed,
                                double bad_fraction, int32_t *order, int32_t *norder);
int32_t oz_consensus(const uint8_t *ref, int32_t rlen, const oz_db *reads, const oz_la_set *s,
                     int32_t aidx, int32_t tspace, uint8_t *out, uint32_t *votes_out);

/* ---------- `dentist collect` (spanning reads) + `dentist process` per pile-up (pile.c) ---------- */
typedef struct {
    int32_t ts_map, allowance, min_anchor, min_reads, max_reads, ts_pile, rounds, flank_window,
        max_align_err_ppm, max_ins_err_ppm, bad_fraction_ppm, width, dust,
        algo, /* alignments of the process stages: 0 = DH-1 (wave, `width` live diagonals), 1 = DH-2 (tiled band of 64) */
        max_partners, /* 0 = every pair of a pile-up is aligned; n > 0 (algo 1): the first n reads are the partners (dh_process_opts) */
        min_relative_score_ppm; /* --min-relative-score of the pile-up chaining (commandline.d:2141-2153), default 1 000 000 */
} oz_process_opts;
void oz_default_process_opts(oz_process_opts *o);
/* chainLocalAlignments (common/alignments/chaining.d:122-334) over a set sorted by (aread, bread); min_score = the trace
 * spacing (commandline.d:2165-2173), min_rel_ppm = minRelativeScore.  Flags in place; LAs shared by alternate chains are
 * duplicated behind their first occurrence. */
void oz_chain_set(oz_la_set *s, int32_t min_score, int32_t min_rel_ppm);
int64_t oz_chain_las(const oz_la *las, int64_t n, int32_t min_score, int32_t min_rel_ppm, oz_la **out);
typedef struct {
    int32_t gap, status /* 0 ok, 1..7 = DH_PILE_* */, nreads, ref_idx, ref_read_id, crop_left, crop_right,
        left_aepos, right_abpos, ins_begin, ins_end, comp, cons_len, left_diffs, right_diffs, pad;
    int64_t cons_off;
} oz_insertion;
/* malloc'd outputs (free with oz_free): gap[npiles], count[npiles], triples[3 * sum(count)] */
int oz_collect_spanning(const oz_la *las, int64_t n, const oz_db *contigs, const oz_process_opts *o,
                        int32_t **gap_out, int32_t **count_out, int32_t **triples_out, int32_t *npiles);
int oz_process_piles(const oz_db *contigs, const oz_db *reads, const oz_la *las, int64_t n, const uint16_t *trace,
                     const int32_t *gap, const int32_t *count, const int32_t *triples, int32_t npiles,
                     const oz_process_opts *o, int nthreads, oz_insertion *out, uint8_t **bases_out, int64_t *nbases);
void oz_free(void *p);

/* ---------- low-complexity mask, DBdust's role (dust.c) ---------- */
int64_t oz_dust(const oz_db *db, int64_t *ptr, int32_t **iv_out);

#ifdef __cplusplus
}
#endif
#endif
