/*
 * dentist_hip.h -- C ABI of libdentist_hip.so, the MI355X (gfx950) implementation of DENTIST's
 * alignment + consensus hot path.
 *
 * The reference reaches this path through process spawns, not FFI (source/dentist/dazzler.d:
 * 6121-6231 wrappers, 6519-6594 executeCommand).  Each entry point below states the reference
 * interface it replaces; INTEGRATION.md shows the D `extern(C)` module a maintainer would add.
 * Conventions: plain pointers and sizes, POD structs, no exceptions across the boundary, every
 * function returns 0 on success or a negative DH_E* code (message via dh_last_error()).
 * Sequences are base codes a,c,g,t = 0..3 (DAZZ_DB order), anything else = 4.
 * The library never falls back to the CPU: without a usable HIP device every compute entry
 * point fails with DH_ENODEV.
 */
#ifndef DENTIST_HIP_H
#define DENTIST_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DH_OK 0
#define DH_EINVAL (-1)
#define DH_ENODEV (-2)
#define DH_EHIP (-3)
#define DH_EOVERFLOW (-4) /* a device-side capacity planned from the input (trace pool, hit slab) was exceeded */
#define DH_EIO (-5)
#define DH_ENOMEM (-6)

const char *dh_last_error(void);
/* library / ABI version, bumped on any struct change (4: dh_process_opts is 64 bytes -- max_partners,
 * min_relative_score_ppm; callers check it before passing structs) */
int32_t dh_abi_version(void);

/* ---- context: one per process == one per GPU (torch.distributed launches one rank per GPU) */
typedef struct dh_ctx dh_ctx;
/* stream: a hipStream_t created by the caller (e.g. torch's current stream) or NULL for the
 * library's own stream. */
int dh_ctx_create(int32_t device, void *stream, dh_ctx **out);
void dh_ctx_destroy(dh_ctx *ctx);
int dh_ctx_sync(dh_ctx *ctx);

/* ---- alignment options: the flag subset DENTIST derives for daligner/damapper
 *      (source/dentist/commandline.d:2886-2902, 2918-2935, 2943-2955; enums dazzler.d:5745-6019) */
typedef struct {
    int32_t k;           /* -k  k-mer length, default 14                                     */
    int32_t hmin;        /* -h  covered bases in a band pair needed to trigger, default 35   */
    int32_t band_shift;  /* -w  log2 band width, default 6                                   */
    int32_t tspace;      /* -s  trace spacing: 100 (damapper), 126 (pile-up daligner)        */
    int32_t min_len;     /* -l  minimum A-length of a reported local alignment               */
    int32_t pen;         /* derived from -e: floor(2 / (1 - e)); e = 0.7 -> 6                */
    int32_t xdrop;       /* wave trimmed to points within xdrop of the best score            */
    int32_t max_err_ppm; /* (1 - e) * 1e6: 2*diffs*1e6 <= max_err_ppm * (alen + blen)        */
    int32_t max_cand;    /* seed candidates kept per (B read, strand)                        */
    int32_t max_la;      /* local alignments reported per (B read, strand)                   */
    int32_t tcap;        /* -t  k-mers occurring more often in A are ignored                 */
    int32_t strands;     /* bit0 forward B, bit1 reverse-complement B                        */
    int32_t skip_self;   /* A is B: 1 = skip aread == bread (absence of -I); 2 = symmetric: every
                          * unordered pair is aligned once and both records are emitted;
                          * 3 = tandem (`datander`, DAMASKER; DENTIST's call commandline.d:2866-2876): every read is
                          *     aligned with ITSELF only, below the main diagonal -- seeds with A position > B position,
                          *     cells in which B's base does not come before A's never match; records have aread ==
                          *     bread and abpos > bbpos.  DH-2 (algo 1), strands = 1 */
    int32_t dmax;        /* cap on differences per extension                                 */
    int32_t width;       /* live diagonals of the wave, <= 62 (one 64-lane wavefront)        */
    int32_t kmer_mod;    /* -%  modimer sampling: only k-mers with hash % kmer_mod == 0; 1 = all     */
    int32_t algo;        /* extension algorithm: 0 = DH-1, the O(ND) furthest-reaching wave (k_wave2);
                          * 1 = DH-2, tile-by-tile banded bit-parallel DP, one alignment per lane (k_tile);
                          *     band = width, which must be 64; with skip_self = 2 the second record of a pair is the
                          *     tiled alignment of the transposed pair through the same seed */
} dh_align_opts;
void dh_default_align_opts(dh_align_opts *o);

/* ---- device-resident sequence DB: replaces the DAZZ_DB .db/.dam the tools open
 *      (DB stub + .idx/.bps; created by dazzler.d:6233-6330 fasta2DB/DAM). */
typedef struct dh_db dh_db;
/* bases: concatenated codes, off[n+1] offsets, group: optional per-sequence group id (pile-up
 * index when many pile-ups are batched in one DB; alignments never cross groups) or NULL.     */
int dh_db_create(dh_ctx *ctx, const uint8_t *bases, const int64_t *off, int32_t n,
                 const int32_t *group, dh_db **out);
void dh_db_destroy(dh_db *db);
/* soft mask = union of the daligner/damapper -m tracks (commandline.d:2886-2955 passes -mdust,
 * -mdentist-self, -mtan, -mrep): per sequence sorted disjoint intervals iv[2j], iv[2j+1] for j in
 * [ptr[s], ptr[s+1]).  k-mers touching a masked interval are neither indexed (A side) nor looked
 * up (B side); alignments still extend through masked sequence.  The call REPLACES the tracks of an
 * earlier dh_db_set_mask; the bits the library derived itself (dh_db_dust, dh_db_mask_coverage) are a
 * layer of their own and stay -- the effective mask is the OR of the two layers.  ptr == NULL clears
 * the whole mask, both layers. */
int dh_db_set_mask(dh_db *db, const int64_t *ptr, const int32_t *iv);
/* DBdust (symmetric DUST, -w64 -t2.0 -m10; DENTIST runs it on every DB it aligns with -mdust,
 * processPileUps/package.d:476-482, 655-667): low-complexity windows are found on the device and
 * ORed into the DB's soft mask.  A window of L = 16, 32 or 64 bases is masked when its triplets
 * repeat with a DUST score above 2.0: sum_t c_t (c_t - 1) / 2 > 2 (L - 3).
 * dh_db_get_mask returns the current soft mask as intervals (what DBdust writes into the `dust`
 * track): ptr gets n + 1 entries, iv may be NULL to size; returns the interval count or < 0. */
int dh_db_dust(dh_db *db);
int64_t dh_db_get_mask(dh_db *db, int64_t *ptr, int32_t *iv, int64_t iv_cap);
/* drop cached derived data (k-mer index, reverse complement): the next dh_align_db rebuilds it */
int dh_db_drop_cache(dh_db *db);
int32_t dh_db_nreads(const dh_db *db);
int64_t dh_db_total_bases(const dh_db *db);

/* ---- local alignments: the .las record (struct Overlap/Path of dalign.h as mirrored at
 *      source/dentist/dazzler.d:1988-2032; bytes [8,48) of it are what goes to disk)          */
#define DH_FLAG_COMP 0x1u
#define DH_FLAG_START 0x4u
#define DH_FLAG_NEXT 0x8u
#define DH_FLAG_BEST 0x10u
#define DH_FLAG_DISABLED 0x20u
typedef struct {
    int32_t tlen, diffs, abpos, bbpos, aepos, bepos;
    uint32_t flags;
    int32_t aread, bread; /* 0-based as on disk; DENTIST adds 1 in memory (dazzler.d:1731-1734) */
    int32_t pad;
    int64_t toff; /* offset of this LA's (diffs, bbases) pairs in the u16 trace array */
} dh_la;

typedef struct dh_la_set dh_la_set; /* host-side result set owned by the library */
void dh_la_set_destroy(dh_la_set *s);
int64_t dh_la_set_count(const dh_la_set *s);
int64_t dh_la_set_trace_len(const dh_la_set *s);
const dh_la *dh_la_set_records(const dh_la_set *s);
const uint16_t *dh_la_set_trace(const dh_la_set *s);
int32_t dh_la_set_tspace(const dh_la_set *s);

/* statistics of the last dh_align_db call on this context */
typedef struct {
    int64_t hits, cands, alignments, wave_cells, las;
    int64_t b_bases;       /* bases of B processed (both strands counted once)               */
    float ms_index, ms_seed, ms_wave, ms_gather, ms_total; /* HIP-event times on ctx stream   */
    int32_t wave_launches;
    int32_t overflow_items; /* (read, strand) items that exceeded a per-item capacity: more than 256
                             * candidate band pairs (the item yields no alignments) or, in symmetric
                             * mode, more than max_la overlaps (the excess records are dropped).  The
                             * call still succeeds; dh_process_pileups skips the pile-ups concerned
                             * with DH_PILE_ALIGN_OVERFLOW, as the reference skips a failing pile-up
                             * (processPileUps/package.d:319-363) */
    int64_t big_items;     /* (read, strand) items whose hits were staged in HBM instead of LDS  */
} dh_align_stats;
int dh_get_align_stats(dh_ctx *ctx, dh_align_stats *out);

/* cumulative over every dh_align_db executed on the context since the last reset (the pile-up
 * path calls it several times per batch): HIP-event kernel times and work counters */
typedef struct {
    double ms_index, ms_seed, ms_wave, ms_gather;
    int64_t wave_launches, wave_cells, alignments, las, aligned_bp, trace_values, hits, b_bases;
} dh_cum_stats;
int dh_get_cum_stats(dh_ctx *ctx, dh_cum_stats *out, int32_t reset);
/* Releases the context's device scratch (grow-only buffers kept between calls so that no call pays for GB-sized
 * allocations: the wave / tile slots, the derived copies of a read chunk, the partitioned join's entry and hit pools --
 * about 100 GB after an unsampled mapping of configs[2]).  For a long-lived host that moves on to another workload; the
 * next call allocates what it needs again.  (The reference's tools are processes: their memory goes with them,
 * dazzler.d:6519-6594.) */
int dh_ctx_release_scratch(dh_ctx *ctx);
/* Chunks of reads of the mapping calls on this context whose seeds came from the radix-partitioned k-mer join
 * (csrc/dh_mjoin.h: the damapper role, dazzler.d:6158-6170, without a random directory line per k-mer) -- out2[0] --
 * and chunks that exceeded one of its capacities and were redone by the directory lookups -- out2[1].  Both paths give
 * the same alignments; the counts say which one ran (tests, traces). */
int dh_get_mjoin_counts(dh_ctx *ctx, int64_t *out2, int32_t reset);

/*
 * dh_align_db -- every sequence of B against all of A: k-mer seeds, diagonal band filter, wave
 * local alignment with trace points.  Replaces the spawns
 *   `damapper -C -T<t> -e0.7 ... <ref> <reads>`   dazzler.d:6158-6170 (getDamapping :3855-3866,
 *                                                 workflow call snakemake/Snakefile:1143-1170)
 *   `daligner -T<a> -B -s126 -l500 -e0.7 db db`   dazzler.d:6121-6140 (getDalignment :3829-3844)
 *   `daligner -A ... contigs consensus`           processPileUps/package.d:655-667
 * Output: LAs in LAsort order (base.d:1787-1809); each record's toff locates its trace pairs in
 * the trace array (the trace array itself is not reordered).  select_best != 0 additionally sets the
 * chain flags damapper emits (START/BEST, consumer dazzler.d:1728-1758).
 */
int dh_align_db(dh_ctx *ctx, dh_db *A, dh_db *B, const dh_align_opts *opts, int32_t select_best,
                dh_la_set **out);
/* select_best: damapper's chains.  The LAs of a read on one contig and strand that follow each other (gaps <= 10 kb,
 * gap difference <= 6 kb: a long indel splits a mapping into collinear LAs) form a chain: START (0x4) on its first
 * LA, NEXT (0x8) on the others, BEST (0x10) on the LAs of the chain that no higher-scoring chain of the strand covers
 * by more than half (consumer: dazzler.d:1728-1758).  dh_set_near_best(ppm) is damapper's -n: alternate chains scoring
 * less than that fraction of the chain that beats them are DISABLED; 0 (default) keeps every chain.  dh_set_near_best sets
 * the process-wide default, dh_ctx_set_near_best the value of one context (-1: back to the process default) -- a library
 * user who never asked for -n is not affected by what another context set.  (damapper's own default is -n1.00,
 * commandline.d:2943-2955 always passes -n.7.) */
void dh_set_near_best(int32_t ppm);
int dh_ctx_set_near_best(dh_ctx *ctx, int32_t ppm);

/* One read block against the whole reference, the unit the workflow shards the mapping by
 * (`damapper <ref> <reads>.<block>`, snakemake/Snakefile:1143-1170; blocks = DBsplit ranges of one
 * DB): reads [first, first + count) of B; bread in the records are ids of the whole DB.  The k-mer
 * index of A is built once and stays with A between calls.  dh_la_set_merge is LAmerge
 * (Snakefile:1173-1185) on the in-memory results: one set in LAsort order. */
int dh_align_db_block(dh_ctx *ctx, dh_db *A, dh_db *B, int32_t first, int32_t count,
                      const dh_align_opts *opts, int32_t select_best, dh_la_set **out);
/* `damapper -C`: the mapping of B onto A and, in one pass, the set of its transposed records (aread = B read, bread = A
 * sequence; source/dentist/dazzler.d:6158-6170 writes them as <B>.<A>.las, getLasFile :4339-4354).  DH-2 only
 * (opts->algo = 1, A != B): for every accepted local alignment the transposed pair -- A'' = the read on its forward
 * strand, B'' = the contig, complemented for a reverse-strand mapping -- is aligned through the same seed and accepted
 * on its own (min_len, max_err_ppm); its trace lies on the read's grid.  want_best: chain flags on both sets (the
 * transposed set with the roles of the sequences exchanged).  Both sets in LAsort order. */
int dh_align_db_transposed(dh_ctx *ctx, dh_db *A, dh_db *B, const dh_align_opts *opts, int32_t want_best,
                           dh_la_set **out, dh_la_set **out_transposed);
int dh_la_set_merge(const dh_la_set *const *sets, int32_t nsets, dh_la_set **out);

/* ---- .las files: replaces the reader/writer pair of source/dentist/dazzler.d:1665-1834
 *      (LocalAlignmentReader) and :1913-1960, 2130-2170 (writeAlignments/writeDazzlerOverlap). */
int dh_las_write(const char *path, const dh_la *las, int64_t n, const uint16_t *trace,
                 int32_t tspace);
int dh_las_read(const char *path, dh_la_set **out);
/* LAmerge (snakemake/Snakefile:1173-1185): the .las files of read blocks (one per GPU) merged into one
 * file in LAsort order; all inputs must share the trace spacing.  Host only. */
int dh_las_merge(const char *const *paths, int32_t npaths, const char *out_path);


/* ---- trace-point arithmetic of the alignment model (Trace.translateTracePoint!"contigA",
 *      source/dentist/common/alignments/base.d:185-244; the cropper is built on it, cropper.d:503-550):
 *      apos is assigned to a trace point of the LA (mode 0 = RoundingMode.floor, 1 = ceil); out_a /
 *      out_b = that trace point on A and on B.  trace = the u16 array the record's toff indexes.
 *      DH_EINVAL when apos lies outside [abpos, aepos] (the reference asserts). Host only. */
int dh_translate_trace_point(const dh_la *la, const uint16_t *trace, int32_t tspace, int32_t apos,
                             int32_t mode, int32_t *out_a, int32_t *out_b);

/* getCommonTracePoint (commands/processPileUps/cropper.d:446-500) of one flank of a pile-up as an entry of its own:
 * first[count] names the first record of every alignment chain of the flank (same contig, same seed; chain members
 * follow their first record, dazzler.d:1728-1758); the common region is the intersection of the chains' A regions
 * (union of the members' A intervals, common/package.d:228-241).  Candidates are the trace points of the region plus
 * the contig end, taken from the inner side for seed_front != 0; the region minus the repeat mask (mask_iv = nmask
 * sorted disjoint (begin, end) pairs of this contig) is tried first, then the region itself.  *out = -1: none. Host only. */
int dh_common_trace_point(const dh_la *las, int64_t n, const int32_t *first, int32_t count, int32_t contig_len,
                          int32_t tspace, int32_t seed_front, const int32_t *mask_iv, int64_t nmask, int32_t *out);

/* ---- pile-ups: which reads span which gap.  Host-side stand-in for the part of `dentist collect`
 *      the consensus path needs (spanning reads only; the scaffold-graph builder of
 *      source/dentist/commands/collectPileUps/pileups.d is outside this library). */
typedef struct {
    int32_t tspace_map;        /* trace spacing of the read->contig LAs (100)                       */
    int32_t allowance;         /* proper-alignment-allowance of the mapping LAs (= tspace_map)      */
    int32_t min_anchor;        /* --min-anchor-length, commandline.d:2036 (500)                     */
    int32_t min_reads;         /* --min-reads-per-pile-up, commandline.d:2125-2187 (3)              */
    int32_t max_reads;         /* reads kept per pile-up (default 60); 0 = every read, as the reference
                                * (commandline.d:2125-2187 knows minimum counts only)                */
    int32_t tspace_pile;       /* -s126 of the pile-up daligner call, commandline.d:2886-2902       */
    int32_t rounds;            /* consensus rounds (1 = reference read + its overlaps only)         */
    int32_t flank_window;      /* bases of each flanking contig given to the flank re-alignment (default 20 000);
                                * 0 = the whole contigs, as the reference does (commandline.d:2918-2935) */
    int32_t max_align_err_ppm; /* --max-alignment-error 0.30, commandline.d:1808                    */
    int32_t max_ins_err_ppm;   /* --max-insertion-error 0.10, commandline.d:1997                    */
    int32_t bad_fraction_ppm;  /* --bad-fraction 0.08, commandline.d:1101                           */
    int32_t width;             /* live diagonals of the wave in the pile-up stages (dh_align_opts.width), 0 = 30 */
    int32_t dust;              /* 1: DBdust + -mdust on the pile-up DB and the flank DB (package.d:476-482,
                                * 655-667); the flank DB also inherits the contigs' soft mask (-mrep)       */
    int32_t algo;              /* alignments of the process stages (pile-up all-vs-all, re-alignment to the template,
                                * flanks): 0 = DH-1 (wave, `width` live diagonals), 1 = DH-2 (tiled band of 64, k_tile) */
    int32_t max_partners;      /* 0 (default) = the pile-up all-vs-all aligns every read with every other, as
                                * `daligner pile.db pile.db` does (package.d:474-485).  n > 0 (DH-2 only): a pile-up of more
                                * than n reads takes n PARTNER reads -- the first n in the order: reads that may serve as
                                * reference read (:461-472), then the others, each in pile-up order -- and aligns a read with
                                * the partners only: the tile QVs that rank the reference-read candidates (:498-568) and the
                                * first consensus round see n overlaps per read instead of all, the later consensus rounds
                                * still re-align EVERY read of the pile-up to the template.  The n^2 stage becomes n x
                                * max_partners; every read keeps its vote.  oracle/process.py, oracle/pile.c: same rule. */
    int32_t min_relative_score_ppm; /* --min-relative-score of the pile-up chaining (commandline.d:2141-2153; 1 000 000 =
                                * the default 1.0: only chains with the pair's best score; valid range [1, 1 000 000], 0 is
                                * refused as the mark of a zero-filled or too short struct).  Below it the chains of a pair
                                * within that fraction of its best chain are kept, per connected component of the
                                * chainability relation, ALTERNATE chains (sharing a prefix with a better chain) included:
                                * the LAs they share then count once per chain, as in the reference's chained .las
                                * (chaining.d:166-312, dazzler.d:2050-2085).  The funnel then runs on the host. */
} dh_process_opts;
void dh_default_process_opts(dh_process_opts *o);

typedef struct dh_pileups dh_pileups;
/* las/trace: read->contig LAs (A = contig, B = read) as returned by dh_align_db.  A read spans
 * the gap between contig c and c+1 when it has an LA reaching the end of c and an LA starting at
 * the begin of c+1, same orientation, in read order, both anchors >= min_anchor.  Every read
 * enters a pile-up once (its pair of LAs with the longest anchors); pile-ups with fewer than
 * min_reads reads are dropped (commandline.d:2125-2187); of more than max_reads the max_reads
 * reads whose anchoring LAs have the lowest error rate are kept (ties: lower read id).
 *   dh_collect_spanning   = dh_collect_candidates + dh_pileups_select
 *   dh_collect_candidates   every spanning read of every gap, no cut (the multi-GPU path exchanges
 *                           candidates between ranks first)
 *   dh_pileups_create       pile-ups from explicit arrays: npiles gaps ordered by contig_left,
 *                           count[i] (read, left LA index, right LA index) triples each
 *   dh_pileups_select       the min_reads / max_reads cut on a candidate set */
int dh_collect_spanning(const dh_la *las, int64_t n, const int64_t *contig_off, int32_t ncontigs,
                        const dh_process_opts *opts, dh_pileups **out);
int dh_collect_candidates(const dh_la *las, int64_t n, const int64_t *contig_off, int32_t ncontigs,
                          const dh_process_opts *opts, dh_pileups **out);
int dh_pileups_create(const int32_t *contig_left, const int32_t *count, int32_t npiles,
                      const int32_t *triples, dh_pileups **out);
int dh_pileups_select(const dh_pileups *cands, const dh_la *las, int64_t n, const dh_process_opts *opts,
                      dh_pileups **out);
void dh_pileups_destroy(dh_pileups *p);
int32_t dh_pileups_count(const dh_pileups *p);
/* pile-up i: left contig id (gap lies between it and the next contig), number of reads, and the
 * (read, left LA index, right LA index) triples */
int32_t dh_pileups_get(const dh_pileups *p, int32_t i, int32_t *contig_left, const int32_t **triples);
/* Pile-ups of ANY join of the scaffold graph (ReadAlignment types of common/alignments/base.d:2160-2330; PileUp types
 * :2725-2805; cropper.d:113-175 crops them all the same way): nodes4[4 * i ..] = (contig0, seed0, contig1, seed1) of
 * pile-up i -- flank 0 = contig0 cropped at its seed0 side (DH_SEED_FRONT: the reads hang over the contig's begin,
 * DH_SEED_BACK: over its end), flank 1 likewise; contig1 = -1: an extension pile-up (one flank).  The gap between
 * contig c and c + 1 of dh_pileups_create is (c, DH_SEED_BACK, c + 1, DH_SEED_FRONT); (c, BACK, d, BACK) and
 * (c, FRONT, d, FRONT) are anti-parallel joins, d > c + 1 skips contigs.  contig0 < contig1 (a contig joined with
 * itself is refused), pile-ups ordered by their nodes, triples = (read, LA on flank 0 or -1, LA on flank 1 or -1).
 * dh_pileups_get_join returns the four values of pile-up i (also for pile-ups made by the other creators). */
#define DH_SEED_FRONT 0
#define DH_SEED_BACK 1
int dh_pileups_create_joins(const int32_t *nodes4, const int32_t *count, int32_t npiles, const int32_t *triples,
                            dh_pileups **out);
int dh_pileups_get_join(const dh_pileups *p, int32_t i, int32_t *nodes4);

/* maskRepetitiveRegions (commands/maskRepetitiveRegions.d:129-176 assessRepeatStructure, :238-430
 * BadAlignmentCoverageAssessor): ORs into the soft mask of `db` every region whose coverage by the
 * alignment intervals [abpos, aepos) of `las` is < lower or > upper; improper_only != 0 counts only
 * alignments that are not proper within `allowance` (base.d:537-557; needs read_off).  The command's mask
 * is the union of one call over all alignments with --max-coverage-reads and one improper-only call with
 * --max-improper-coverage-reads; read the result with dh_db_get_mask, write it with dh_dazz_write_mask.
 * n == 0 masks nothing (:347-348). */
int dh_db_mask_coverage(dh_db *db, const dh_la *las, int64_t n, const int64_t *read_off, int32_t nreads,
                        int32_t lower, int32_t upper, int32_t improper_only, int32_t allowance);
/* the bounds DENTIST derives from --read-coverage (commandline.d:1876-1889, 1957-1970) */
int32_t dh_max_coverage_reads(double read_coverage);
int32_t dh_max_improper_coverage_reads(double read_coverage);

/* `dentist propagate-mask` (commands/propagateMask.d:136-305): the contig mask (mask_ptr[ncontigs + 1],
 * mask_iv = sorted disjoint (begin, end) pairs) carried over to the reads through the trace points of the
 * read->contig LAs -- begin rounded down, end rounded up, mirrored for complement alignments -- and merged
 * per read.  out_ptr gets nreads + 1 entries, out_iv (begin, end) pairs on the forward read; out_iv may be
 * NULL to size (cap = pairs it can hold); returns the number of intervals or a negative error.  Host only. */
int64_t dh_propagate_mask(const dh_la *las, int64_t n, const uint16_t *trace, int32_t tspace, const int64_t *mask_ptr,
                          const int32_t *mask_iv, int32_t ncontigs, const int64_t *read_off, int32_t nreads,
                          int64_t *out_ptr, int32_t *out_iv, int64_t cap);

/* `dentist validate-regions` (commands/validateRegions.d:141-203 region context, :325-512 RegionValidator):
 * every region (closed gap on the gap-closed assembly) extended by region_context (default 1000,
 * commandline.d:2411) is valid iff every window of weak_coverage_window bases (default 500, :2497) inside
 * it is spanned by >= min_coverage_reads alignments (from --read-coverage: 0.5 * x / ploidy, :2080-2085)
 * and >= min_spanning_reads alignments span the whole extended region.  las: reads aligned to that
 * assembly, grouped by aread.  reports[nregions]; weak_iv (may be NULL, cap triples) = weakly covered
 * (contig, begin, end) per region in region order (--weak-coverage-mask); returns their number.  Host only. */
typedef struct dh_region { int32_t contig, begin, end; } dh_region;
typedef struct dh_region_report { int32_t num_spanning_reads, weak_bp, is_valid, ctx_begin, ctx_end; } dh_region_report;
int64_t dh_validate_regions(const dh_la *las, int64_t n, const int64_t *contig_off, int32_t ncontigs,
                            const dh_region *regions, int32_t nregions, int32_t region_context,
                            int32_t weak_coverage_window, int32_t min_coverage_reads, int32_t min_spanning_reads,
                            dh_region_report *reports, int32_t *weak_iv, int64_t cap);

/* The mapping pass with the six filters of `dentist collect` applied on the way: `damapper` per read block
 * (snakemake/Snakefile:1143-1170) + collectPileUps/filter.d:122-356.  Every filter decides per read, so
 * the records of a finished chunk of reads get their chain flags and are filtered on a host thread while
 * the device maps the next chunk.  want_sorted != 0: result = dh_align_db_block(want_best = 1) followed by
 * dh_collect_filter (same records, flags and dropped6 counts, LAsort order).  want_sorted == 0: the same
 * records in mapping order (by read, strand); then `cands` (may be NULL) receives the spanning-read
 * candidates of dh_collect_candidates, collected chunk by chunk as well (indices into the result).
 * want_sorted is a bit set: 1 = LAsort order; 8 = the trace values stay on the device in a buffer the result owns
 * (dh_la_set_trace downloads them when asked; dh_process_pileups_set gathers what the cropper needs there). */
int dh_map_reads(dh_ctx *ctx, dh_db *contigs, dh_db *reads, int32_t first, int32_t count, const dh_align_opts *opts,
                 const dh_process_opts *popts, const int64_t *rep_ptr, const int32_t *rep_iv, int32_t want_sorted,
                 int64_t *dropped6, dh_la_set **out, dh_pileups **cands);

/* ---- the scaffold-graph pile-up builder of `dentist collect` (collectPileUps/pileups.d:173-208 build;
 * collectPileUps/package.d:174-184 is the call site).  Nodes are (contig, part) with part 0 = pre,
 * 1 = begin, 2 = end, 3 = post (scaffold.d:75-90); a read alignment is one seeded LA (an extension over
 * a contig end) or two (a read spanning a gap), seed 0 = front, 1 = back (base.d:1938-1944).  Steps:
 * collectReadAlignments per read (pileups.d:821-888), joins merged per edge (pileups.d:626-636), forks
 * resolved by read support with a bonus for joins of the input assembly (pileups.d:1592-1657,
 * 1754-1804), min_spanning_reads (pileups.d:1807-1838), extensions merged into their gap
 * (scaffold.d:789-816), pile-ups in edge order (pileups.d:435-444).  resolveBubbles (pileups.d:1124-1590): see
 * dh_scaffold_pileups_resolved below.  DISABLED LAs are ignored.  input_gaps: ngaps pairs (begin contig, end contig) of the
 * input assembly's scaffolding (pileups.d:796-811).  Host only. */
typedef struct dh_scaffold_opts {
    int32_t min_spanning_reads;  /* --min-spanning-reads, default 3 */
    int32_t merge_extensions;    /* 0 = --no-merge-extension */
    double best_pile_up_margin;  /* --best-pile-up-margin, default 3.0 */
    double existing_gap_bonus;   /* --existing-gap-bonus, default 6.0 */
    int32_t only_joins;          /* the SHARDED plan only (dh_shard_graph_plan_create, dh_shard_run): 0 (default) = the gap
                                  * pile-ups between neighbouring contigs with the extension entries merged into them
                                  * (dh_scaffold_gap_pileups; --only spanning --join-policy scaffoldGaps); otherwise every
                                  * pile-up of the scaffold `dentist process` is handed, as dh_scaffold_all_pileups(only):
                                  * & 1 the gap joins of any two contig ends (anti-parallel, contig-skipping), & 2 the
                                  * extension joins -- what `process --batch` jobs take (snakemake/Snakefile:1315-1334,
                                  * processPileUps/package.d:146-159).  Ignored by dh_scaffold_pileups*. */
    int32_t pad_;
} dh_scaffold_opts;
void dh_default_scaffold_opts(dh_scaffold_opts *o);
typedef struct dh_join {        /* one pile-up = the payload of one edge of the scaffold graph */
    int32_t contig0, part0, contig1, part1; /* (contig0, part0) <= (contig1, part1) */
    int32_t type;                /* ReadAlignmentType (base.d:2052-2070): 0 front extension, 1 gap, 2 back extension */
    int32_t count;               /* read alignments of the pile-up: entries[first .. first + count) */
    int64_t first;
} dh_join;
typedef struct dh_read_alignment {
    int32_t read, la0, la1;      /* LA indices into the input; la1 = -1 for an extension */
    uint8_t seed0, seed1, n, pad; /* n = 1 | 2 seeded alignments */
} dh_read_alignment;
typedef struct dh_scaffold dh_scaffold;
int dh_scaffold_pileups(const dh_la *las, int64_t n, const int64_t *contig_off, int32_t ncontigs,
                        const int64_t *read_off, int32_t nreads, const int32_t *input_gaps, int32_t ngaps,
                        const dh_scaffold_opts *opts, dh_scaffold **out);
/* The same builder WITH resolveBubbles (pileups.d:1100-1315, 1387-1590; position in build(): pileups.d:186): cycles of
 * at most max_bubble_size nodes (0 = the reference's 8, commandline.d:1826-1833) with exactly two nodes of degree >= 3
 * joined by an edge that carries a pile-up -- its reads skip the contigs on the other side of the cycle, whose
 * alignments a mask or a filter removed -- are linearised: the skipping reads are mapped onto the intermediate contigs
 * again without any mask (getReadAlignmentsOnContigs :1316-1385), their read alignments are collected from old and new
 * alignments together, kept iff they walk the skipped path in order (collectFixedSimpleBubbles :1414-1491), and replace
 * the skipping pile-up; at most max_iterations sweeps (0 = 4, :1835-1837).  The alignments added on the way come back in
 * *extra: LA index n + i in the result's read alignments = record i of *extra.  *resolved (optional) = bubbles handled.
 *   dh_scaffold_pileups_resolved  the device maps (dh_remap_skipping_reads with map_opts; chains must cover their contig
 *                                 within `allowance`); contig / read offsets are those of the DBs; *extra carries the traces
 *   dh_scaffold_pileups_cb        host only: `remap` supplies the alignments (las_out malloc'd by the callback, freed here;
 *                                 ids of the full DBs, DISABLED unless the chain covers its contig) */
typedef int (*dh_remap_fn)(void *user, const int32_t *contig_ids, int32_t ncontig_ids, const int32_t *read_ids, int32_t nread_ids,
                           dh_la **las_out, int64_t *n_out);
int dh_scaffold_pileups_cb(const dh_la *las, int64_t n, const int64_t *contig_off, int32_t ncontigs, const int64_t *read_off,
                           int32_t nreads, const int32_t *input_gaps, int32_t ngaps, const dh_scaffold_opts *opts,
                           int32_t max_bubble_size, int32_t max_iterations, dh_remap_fn remap, void *user, dh_scaffold **out,
                           dh_la_set **extra, int32_t *resolved);
int dh_scaffold_pileups_resolved(dh_ctx *ctx, dh_db *contigs, dh_db *reads, const dh_la *las, int64_t n, const int32_t *input_gaps,
                                 int32_t ngaps, const dh_scaffold_opts *opts, const dh_align_opts *map_opts, int32_t allowance,
                                 int32_t max_bubble_size, int32_t max_iterations, dh_scaffold **out, dh_la_set **extra,
                                 int32_t *resolved);
int32_t dh_scaffold_npiles(const dh_scaffold *s);
int64_t dh_scaffold_nentries(const dh_scaffold *s);
const dh_join *dh_scaffold_joins(const dh_scaffold *s);
const dh_read_alignment *dh_scaffold_entries(const dh_scaffold *s);
void dh_scaffold_destroy(dh_scaffold *s);
/* the pile-ups dh_process_pileups handles -- gap joins (c, end)--(c + 1, begin) -- with their spanning
 * read alignments as (read, left LA, right LA) triples; *skipped = pile-ups of any other kind */
/* The re-mapping call of `resolveBubbles` (getReadAlignmentsOnContigs, collectPileUps/pileups.d:1316-1385): the reads
 * read_ids[] (those of a pile-up whose join skips contigs) are mapped, without any mask, onto the intermediate contigs
 * contig_ids[] alone -- the reference builds two DB subsets and spawns damapper on them (:1337-1366); here the subsets
 * are gathered on the device.  Chains that do not cover their contig completely within `allowance`
 * (completelyCovers!"contigA", common/alignments/base.d:562-566) come back DISABLED; ids are those of the full DBs
 * (:1373-1380).  Ids 0-based, ascending, distinct.  The graph surgery of BubbleResolver stays with the caller. */
int dh_remap_skipping_reads(dh_ctx *ctx, dh_db *contigs, dh_db *reads, const int32_t *contig_ids, int32_t ncontig_ids,
                            const int32_t *read_ids, int32_t nread_ids, const dh_align_opts *opts, int32_t allowance,
                            dh_la_set **out);
int dh_scaffold_spanning(const dh_scaffold *s, const dh_la *las, int64_t n, dh_pileups **out, int32_t *skipped);
/* the same pile-ups with EVERY read alignment the builder put into them, as `dentist process` gets them
 * (pile-ups.db): besides the spanning reads the extension-type read alignments that mergeExtensionsWithGaps
 * (scaffold.d:789-816) moved into the gap -- (read, LA, -1): back extension of the left contig, (read, -1, LA):
 * front extension of the right one.  The cropper cuts them from their crop point to the read's end
 * (cropper.d:339-361, 503-550); they are members of the pile-up but never its reference read
 * (processPileUps/package.d:461-472). */
int dh_scaffold_gap_pileups(const dh_scaffold *s, const dh_la *las, int64_t n, dh_pileups **out, int32_t *skipped);
/* Every pile-up of the scaffold `dentist process` is handed (collectPileUps/package.d:88-96 writes them all; which
 * insertions are used is `dentist output`'s --only, commandline.d:2230-2250): only & 1 = the gap joins of any two
 * contig ends (same orientation, anti-parallel, contig-skipping), only & 2 = the extension joins.  The pile-ups carry
 * their nodes (dh_pileups_create_joins / dh_pileups_get_join); entries as in dh_scaffold_gap_pileups with flank 0 / 1
 * in place of left / right.  *skipped = joins without entries. */
int dh_scaffold_all_pileups(const dh_scaffold *s, const dh_la *las, int64_t n, int32_t only, dh_pileups **out,
                            int32_t *skipped);

/* the six alignment filters of `dentist collect` (collectPileUps/filter.d:122-356, order of
 * collectPileUps/package.d:130-141) on the read->contig LAs: LQ (averageErrorRate > max_align_err),
 * Improper (allowance), WeaklyAnchored (<= min_anchor bases outside the repeat mask rep_ptr / rep_iv of
 * the contigs, may be NULL), Contained, Ambiguous (reads with alignments overlapping on the read),
 * Redundant (reads that fit inside one contig).  Dropped LAs get DH_FLAG_DISABLED in place;
 * dropped6[stage] = LAs dropped per stage, read_used[r] = 0 for reads discarded by 5 / 6 (either may be
 * NULL).  Host only. */
int dh_collect_filter(dh_la *las, int64_t n, const int64_t *contig_off, int32_t ncontigs, const int64_t *read_off,
                      int32_t nreads, const int64_t *rep_ptr, const int32_t *rep_iv, const dh_process_opts *opts,
                      int64_t *dropped6, uint8_t *read_used);

/* all pile-ups at once: contig_left[npiles], count[npiles], triples[3 * total] (any may be NULL);
 * returns the total number of triples */
int64_t dh_pileups_flat(const dh_pileups *p, int32_t *contig_left, int32_t *count, int32_t *triples);

/* ---- `dentist process` for a batch of pile-ups: crop -> pile-up alignment -> filter -> tile QV
 *      -> reference read -> consensus -> flank re-alignment -> insertion
 *      (source/dentist/commands/processPileUps/package.d:283-374; cropper.d:446-550;
 *      common/insertions.d:110-146).  Replaces the ~15 tool spawns per pile-up of
 *      package.d:474-697 (daligner, DAScover/DASqv, computeintrinsicqv, daccord, daligner -A). */
#define DH_PILE_OK 0
#define DH_PILE_NO_COMMON_TRACE_POINT 1
#define DH_PILE_TOO_SMALL 2
#define DH_PILE_EMPTY_ALIGNMENT 3
#define DH_PILE_FLANKS_NOT_UNIQUE 4
#define DH_PILE_ORIENTATION 5
#define DH_PILE_MAX_INSERTION_ERROR 6
#define DH_PILE_NEGATIVE_INSERTION 7
#define DH_PILE_ALIGN_OVERFLOW 8 /* a read of the pile-up exceeded a per-read capacity of the aligner */
#define DH_PILE_UNSUPPORTED_JOIN 9 /* a contig joined with itself */
/* dh_insertion.join: 0 = the gap between contig_left and contig_left + 1 in the same orientation
 * ((contig_left, end) -> (contig_right, begin)); otherwise bits */
#define DH_JOIN_FLANK0_FRONT 1 /* flank 0 is the BEGIN of contig_left (seed front) instead of its end     */
#define DH_JOIN_FLANK1_BACK 2  /* flank 1 is the END of contig_right (seed back) instead of its begin     */
#define DH_JOIN_EXTENSION 4    /* no flank 1: the consensus extends contig_left over the flank-0 side     */
typedef struct {
    int32_t contig_left;   /* flank 0 (the `left` fields below); the contig with the smaller id                */
    int32_t status;        /* DH_PILE_*                                                             */
    int32_t nreads;        /* reads in the cropped pile-up                                          */
    int32_t ref_read;      /* index of the reference read inside the pile-up, -1 if none            */
    int32_t ref_read_id;   /* its read id in the reads DB                                           */
    int32_t crop_left;     /* common trace point on the flank-0 contig                              */
    int32_t crop_right;    /* common trace point on the flank-1 contig (-1 for an extension)        */
    int32_t left_aepos;    /* getCroppingPosition!"contigA" of the flank-0 overlap (insertions.d:110-121): the
                            * contig is kept up to here for seed back, from here for seed front       */
    int32_t right_abpos;   /* the same for flank 1                                                  */
    int32_t ins_begin;     /* insertion = oriented consensus [ins_begin, ins_end) (the slice of        */
    int32_t ins_end;       /* getInfoForNewSequenceInsertion, insertions.d:230-284, in the frame of `comp`) */
    int32_t comp;          /* 1: the flank-0 overlap is a complement one -- walking from flank 0 to flank 1 (a
                            * front-seeded flank 0: from flank 1 to flank 0) the consensus is reverse-complemented */
    int32_t cons_len;
    int32_t left_diffs, right_diffs; /* of the two flank overlaps                                   */
    int32_t join;          /* DH_JOIN_* bits, 0 for the plain gap                                   */
    int64_t cons_off;      /* consensus bases (read orientation) in the result's sequence buffer    */
    int32_t contig_right;  /* flank 1: its contig (contig_left + 1 for the plain gap, -1 for an extension) */
    int32_t pad;
} dh_insertion;

typedef struct dh_insertions dh_insertions;
int dh_process_pileups(dh_ctx *ctx, dh_db *contigs, dh_db *reads, const dh_la *las, int64_t n,
                       const uint16_t *trace, const dh_pileups *piles, const dh_process_opts *opts,
                       dh_insertions **out);
void dh_insertions_destroy(dh_insertions *r);
int32_t dh_insertions_count(const dh_insertions *r);
const dh_insertion *dh_insertions_records(const dh_insertions *r);
const uint8_t *dh_insertions_bases(const dh_insertions *r);
int64_t dh_insertions_bases_len(const dh_insertions *r);
/* read ids (0-based) of every record's pile-up, the Insertion.readIds of makeInsertion (processPileUps/
 * package.d:789-798): ids[off[i] .. off[i + 1]), off has count + 1 entries; NULL when the result carries none */
const int32_t *dh_insertions_read_ids(const dh_insertions *r);
const int32_t *dh_insertions_read_ids_off(const dh_insertions *r);
/* The two halves of dh_process_pileups as entry points of their own (dh_process_pileups runs them
 * back to back with the cropped reads staying on the device):
 *   dh_crop_pileups     cropPileUp (cropper.d:113-175, 446-550) for a batch: the common trace point of
 *                       each flank from ALL entries of a pile-up, then [support patch] + read slice +
 *                       [support patch] for the entries whose read is in `reads`.  Read ids in the
 *                       triples are ids of the whole reads DB; `reads` holds [read_first, read_first +
 *                       nreads(reads)) of it -- one rank's share when the mapping is sharded (SURVEY
 *                       8(e)); LAs of reads held elsewhere only need their A intervals.
 *   dh_cropped_create   cropped pile-ups assembled from parts received from other ranks: rec = the
 *                       per-pile-up records of dh_crop_pileups (crop points, status), reads ordered by
 *                       (pile, entry)
 *   dh_process_cropped  pile-up alignment -> filter -> tile QV -> reference read -> consensus -> flank
 *                       re-alignment -> insertion on cropped pile-ups (package.d:283-374 after crop()) */
typedef struct dh_cropped dh_cropped;
int dh_crop_pileups(dh_ctx *ctx, dh_db *contigs, dh_db *reads, int32_t read_first, const dh_la *las, int64_t n,
                    const uint16_t *trace, const dh_pileups *piles, const dh_process_opts *opts,
                    dh_cropped **out);
/* the same with the repeat mask of the contigs (`dentist process --mask`): rep_ptr[ncontigs + 1] / rep_iv = sorted
 * disjoint (begin, end) pairs per contig, as dh_map_reads takes them; the common trace point of a flank is taken
 * outside the mask when one exists there (cropper.d:446-500).  NULL = no mask. */
int dh_crop_pileups_masked(dh_ctx *ctx, dh_db *contigs, dh_db *reads, int32_t read_first, const dh_la *las, int64_t n,
                           const uint16_t *trace, const dh_pileups *piles, const int64_t *rep_ptr, const int32_t *rep_iv,
                           const dh_process_opts *opts, dh_cropped **out);
int dh_process_pileups_masked(dh_ctx *ctx, dh_db *contigs, dh_db *reads, const dh_la *las, int64_t n,
                              const uint16_t *trace, const dh_pileups *piles, const int64_t *rep_ptr, const int32_t *rep_iv,
                              const dh_process_opts *opts, dh_insertions **out);
/* dh_process_pileups_masked on the result set of a mapping itself.  For a set whose trace values stayed on the device
 * (dh_map_reads with want_sorted & 8) only the trace of the pile-up reads' records is brought to the host -- the cropper
 * (cropper.d:446-550) reads nothing else; otherwise the same as passing dh_la_set_records / dh_la_set_trace. */
int dh_process_pileups_set(dh_ctx *ctx, dh_db *contigs, dh_db *reads, dh_la_set *set, const dh_pileups *piles,
                           const int64_t *rep_ptr, const int32_t *rep_iv, const dh_process_opts *opts, dh_insertions **out);
/* 1: the set's trace values are on the device only (dh_la_set_trace would fetch them now) */
int32_t dh_la_set_trace_on_device(const dh_la_set *s);
int dh_cropped_create(const dh_insertion *rec, int32_t npiles, int32_t nreads, const int32_t *pile,
                      const int32_t *entry, const int32_t *read_id, const int64_t *off, const uint8_t *bases,
                      dh_cropped **out);
/* the same with the kind of every read (dh_cropped_kind); kind == NULL: every read spans its gap */
int dh_cropped_create2(const dh_insertion *rec, int32_t npiles, int32_t nreads, const int32_t *pile,
                       const int32_t *entry, const int32_t *read_id, const uint8_t *kind, const int64_t *off,
                       const uint8_t *bases, dh_cropped **out);
void dh_cropped_destroy(dh_cropped *c);
int32_t dh_cropped_npiles(const dh_cropped *c);
/* per cropped read, bits 0-1: 0 = it has alignments on both flanks (spans the gap), 1 = on flank 0 only (plain gap: back
 * extension of the left contig), 2 = on flank 1 only (front extension of the right contig) -- entries with one
 * alignment, see dh_scaffold_gap_pileups; bit 2 / 3: its alignment on flank 0 / 1 is a complement one (the flank
 * overlaps of the consensus are checked against the reference read's, package.d:669-690) */
const uint8_t *dh_cropped_kind(const dh_cropped *c);

/* ---- the host work of one rank between the collectives of the sharded path (one process per GPU): what LAmerge +
 *      `dentist collect` + `process --batch` do through the file system in the reference (snakemake/Snakefile:
 *      1173-1185, 1315-1334).  Blobs are what the caller hands to the collective as they are:
 *        candidates  records of 104 bytes {int32 gap, int32 read, dh_la left, dh_la right} in (gap, read) order
 *        cropped     int64 k, k x {int32 pile, entry, read, len}, then the k reads' bases back to back
 *      dh_shard_pack_candidates  this rank's candidates (dh_collect_candidates / dh_map_reads) -> blob
 *      dh_shard_plan_create      all ranks' blobs (rank order) -> the same pile-ups on every rank (entries of a gap by
 *                                read id, min / max reads cut) and their owners (greedy bin-packing of n^2 L)
 *      dh_shard_read_joins       (scaffold-graph collector) the raw joins of this rank's reads -- collectReadAlignments,
 *                                pileups.d:821-888, a per-read computation -- as a blob of 120-byte records {4 x int32
 *                                edge (contig, part) x 2, int32 read, u8 seed0, seed1, n, pad, dh_la, dh_la}; las[].bread
 *                                are global read ids, read_off the offsets of the rank's reads [read_first, +nreads)
 *      dh_shard_graph_plan_create  all ranks' join blobs (rank order) -> the scaffold every rank derives (forks, min
 *                                spanning reads, extensions merged into gaps), its gap pile-ups incl. extension entries,
 *                                the min / max reads cut, owners -- the pile-ups are those of dh_scaffold_pileups +
 *                                dh_scaffold_gap_pileups + dh_pileups_select on the merged alignments
 *      dh_shard_pack_cropped     the reads this rank cropped -> one blob per owner (one malloc, free blobs[0])
 *      dh_shard_unpack_cropped   the blobs an owner received -> its cropped pile-ups for dh_process_cropped */
typedef struct dh_shard_plan dh_shard_plan;
void dh_shard_free(void *p);
int dh_shard_pack_candidates(const dh_pileups *cands, const dh_la *las, int64_t n, int32_t read_shift, uint8_t **out,
                             int64_t *nbytes);
int dh_shard_plan_create(const uint8_t *const *blobs, const int64_t *sizes, int32_t world, const dh_process_opts *opts,
                         dh_shard_plan **out);
int dh_shard_read_joins(const dh_la *las, int64_t n, const int64_t *contig_off, int32_t ncontigs, const int64_t *read_off,
                        int32_t read_first, int32_t nreads, uint8_t **blob, int64_t *nbytes);
int dh_shard_graph_plan_create(const uint8_t *const *blobs, const int64_t *sizes, int32_t world, int32_t ncontigs,
                               const int32_t *input_gaps, int32_t ngaps, const dh_scaffold_opts *sopts,
                               const dh_process_opts *opts, dh_shard_plan **out);
void dh_shard_plan_destroy(dh_shard_plan *p);
const dh_la *dh_shard_plan_las(const dh_shard_plan *p);
int64_t dh_shard_plan_nlas(const dh_shard_plan *p);
const dh_pileups *dh_shard_plan_pileups(const dh_shard_plan *p);   /* owned by the plan */
const int32_t *dh_shard_plan_owner(const dh_shard_plan *p);         /* one per pile-up of dh_shard_plan_pileups */
int dh_shard_pack_cropped(dh_cropped *crop, const int32_t *owner, int32_t world, uint8_t **blobs, int64_t *sizes);
int dh_shard_unpack_cropped(const uint8_t *const *blobs, const int64_t *sizes, int32_t world, const dh_insertion *rec,
                            int32_t npiles, const int32_t *owner, int32_t rank, dh_cropped **out);
const dh_insertion *dh_cropped_records(const dh_cropped *c);
/* ---- the multi-GPU entry (dh_comm.cpp): one process per GPU, the exchanges over RCCL / xGMI.  Replaces the file system
 *      between the workflow's jobs: LAmerge of the per-block mappings (snakemake/Snakefile:1173-1185), the
 *      `dentist process --batch` jobs (:1315-1358) and `dentist merge-insertions` (commands/mergeInsertions.d:60-164).
 *      Rank 0 calls dh_comm_unique_id and hands the 128 bytes to the other processes (file, socket, MPI ...); every
 *      process calls dh_comm_create(id, rank, world, its context) once and dh_shard_run per batch.  RCCL is loaded on
 *      first use (dlopen); without it these calls fail with DH_ENODEV, everything else works.
 *      dh_comm_create_local: `world` communicators inside ONE process that exchange through memory -- one per host
 *      thread and context (tests, the N-rank emulation on one GPU); same calls, same dh_shard_run.
 *      dh_comm_all_gather / dh_comm_all_to_all: the two collectives on host blobs (staged through page-locked memory and
 *      the context's device scratch); *out = one malloc'd block with the blobs in rank order, release with dh_shard_free.
 *      dh_shard_run: `collect` + `process` of this rank's share of the reads (reads = the reads [read_first, read_first +
 *      nreads(reads)) of the whole DB; las / trace its mapping, bread = ids of the whole DB).  cands != NULL: the
 *      spanning-read collector on the candidates dh_map_reads listed; cands == NULL: the scaffold-graph collector of
 *      `dentist collect` (read_off = offsets of this rank's reads, input_gaps / ngaps / sopts as dh_scaffold_pileups; sopts
 *      NULL = defaults with min_spanning_reads = opts->min_reads).  *out = the closed-gap records of ALL ranks ordered by
 *      gap, identical on every rank and bit-identical to the single-GPU result; info4 (optional) = pile-ups, pile-ups
 *      owned by this rank, entries, bytes of cropped reads sent. */
typedef struct dh_comm dh_comm;
int dh_comm_unique_id(uint8_t *id128);
int dh_comm_create(const uint8_t *id128, int32_t rank, int32_t world, dh_ctx *ctx, dh_comm **out);
int dh_comm_create_local(int32_t world, dh_ctx *const *ctxs, dh_comm **out);
void dh_comm_destroy(dh_comm *c);
int32_t dh_comm_rank(const dh_comm *c);
int32_t dh_comm_world(const dh_comm *c);
int dh_comm_all_gather(dh_comm *c, const uint8_t *payload, int64_t nbytes, uint8_t **out, int64_t *sizes);
int dh_comm_all_to_all(dh_comm *c, const uint8_t *const *per_dest, const int64_t *send_sizes, uint8_t **out,
                       int64_t *recv_sizes);
int dh_shard_run(dh_comm *c, dh_db *contigs, dh_db *reads, int32_t read_first, const int64_t *contig_off, int32_t ncontigs,
                 const dh_la *las, int64_t n, const uint16_t *trace, const dh_process_opts *opts, const dh_pileups *cands,
                 const int64_t *read_off, const int32_t *input_gaps, int32_t ngaps, const struct dh_scaffold_opts *sopts,
                 dh_insertions **out, int64_t *info4);

int32_t dh_cropped_nreads(const dh_cropped *c);
const int32_t *dh_cropped_pile(const dh_cropped *c);     /* pile-up of every cropped read               */
const int32_t *dh_cropped_entry(const dh_cropped *c);    /* its position in the pile-up's read list      */
const int32_t *dh_cropped_read_id(const dh_cropped *c);
const int64_t *dh_cropped_offsets(const dh_cropped *c);  /* nreads + 1                                   */
const uint8_t *dh_cropped_bases(dh_cropped *c);          /* copies device -> host on first use           */
int dh_process_cropped(dh_ctx *ctx, dh_db *contigs, dh_cropped *crop, const dh_process_opts *opts,
                       dh_insertions **out);
/* per-stage HIP-event times (ms) of the last dh_process_pileups call on this context:
 * [0] crop+gather [1] pile-up alignment [2] tile QV [3] consensus vote+emit (all rounds)
 * [4] read->consensus re-alignment (rounds > 1) [5] flank re-alignment [6] total;
 * counters: [0] pile LAs [1] tiles aligned (NW) [2] NW cells */
int dh_get_process_stats(dh_ctx *ctx, float *ms7, int64_t *counters3);
/* the work of the same call: [0] pile-ups processed [1] their entries (cropped reads) [2] cropped bases [3] algorithmic
 * bytes, sum over pile-ups of (n^2 + 2) L for n entries of mean cropped length L (the process stage's roofline figure:
 * every read streamed once per partner, once for the consensus, the flanks and the output) */
int dh_get_process_work(dh_ctx *ctx, int64_t *work4);


/* ---- stage-level entry points (the fused dh_process_pileups runs the same code):
 *  dh_tile_qv    DAScover + DASqv -c<cov> (dazzler.d:3782-3792, 6142-6156): qv[r * maxtiles + t] in
 *                [0, 50] for tile t of read r of the pile-up DB, 255 behind the last tile; las must be
 *                grouped by aread (ascending), traces at tspace.
 *  dh_consensus  computeintrinsicqv + daccord -f -I<i>,<i> (dazzler.d:4213-4255, 6172-6231): consensus
 *                of read ref_read from the overlaps with aread == ref_read; rounds > 1 re-aligns the
 *                reads of the DB to the consensus and votes again.  out: caller's buffer of cap bases. */
int dh_tile_qv(dh_ctx *ctx, dh_db *db, const dh_la *las, int64_t n, const uint16_t *trace, int32_t tspace,
               int32_t cov, uint8_t *qv, int32_t maxtiles);
int dh_consensus(dh_ctx *ctx, dh_db *db, const dh_la *las, int64_t n, const uint16_t *trace, int32_t tspace,
                 int32_t ref_read, int32_t rounds, uint8_t *out, int64_t cap, int64_t *out_len);

/* ---- gap-closed assembly writer (host only): the linear-scaffold subset of `dentist output`
 *      (source/dentist/commands/output.d:743-925): header "<id>\tscaffold-<first contig id>", contig
 *      slices lower case, insertions upper case (highlight != 0), unclosed gaps as 'n' runs, lines
 *      wrapped at line_width (commandline.d:1699, default 50); optional closed-gaps BED
 *      (output.d:879-891).  scaffold_of[c]: input scaffold of contig c (contigs of a scaffold are
 *      consecutive); headers[s]: its FASTA header without '>'; gap_len[c]: gap after contig c.
 *      Splice coordinates are dh_insertion.left_aepos / right_abpos / ins_begin / ins_end
 *      (common/insertions.d:110-146).  Pinned by the md5 of tests/test-commands.sh:62-65. */
int dh_output_fasta(const char *fasta_path, const char *bed_path, const uint8_t *contig_bases,
                    const int64_t *contig_off, int32_t ncontigs, const int32_t *scaffold_of,
                    const char *const *headers, const int32_t *gap_len, const dh_insertion *ins,
                    int32_t nins, const uint8_t *ins_bases, int32_t line_width, int32_t highlight);
/* `dentist output` with its graph step and all three writers (output.d:305-348 buildAssemblyGraph with
 * enforceJoinPolicy common/scaffold.d:642-723, normalizeUnkownJoins :373-451, fixCropping :931-1003; scaffoldStarts +
 * linearWalk scaffold.d:1021-1295; FASTA :782-925; AGP :454-573; BED :879-891 with every read id of the pile-up) for
 * insertions of any join (dh_insertion.join: gaps between any two contig ends, extensions).  join_policy: 0
 * scaffoldGaps (gap joins that do not sit on a gap of an input scaffold are dropped, *dropped counts them), 1 scaffolds
 * (they come back where both contig ends are still free), 2 contigs (all stay).  agp_dazzler: component ids are contig numbers / "reads-<ids>"; otherwise scaffold header ids and
 * read_names[id - 1]; agp_skip_read_ids: "<n> reads".  read_ids / read_ids_off[nins + 1]: 0-based read ids of
 * every insertion's pile-up, offsets int32 exactly as dh_insertions_read_ids_off returns them (NULL: the reference read
 * alone); nreads = length of read_names, every id is checked against it (-1: unknown, only without a name table). */
typedef struct {
    int32_t line_width, highlight, join_policy, agp_dazzler, agp_skip_read_ids;
    int32_t only;                 /* --only (commandline.d:2230-2250): 1 spanning (default; 0 means the same), 2 extending, 3 both */
    const char *agp_version, *tool, *input_assembly;
    int32_t min_extension_length; /* --min-extension-length (commandline.d:2098-2100, default 100): shorter extensions are skipped */
    int32_t pad;
} dh_output_opts;
void dh_default_output_opts(dh_output_opts *o);
/* Test surface of the writer's graph code (normalizeUnkownJoins common/scaffold.d:373-451, linearWalk :1021-1170,
 * scaffoldStarts :1209-1295 -- the reference's unit vectors run against it): the default edges of `ncontigs` contigs plus
 * joins4 = (contig0, part0, contig1, part1) per join, parts 0 pre, 1 begin, 2 end, 3 post, contigs 0-based; normalize != 0
 * runs normalizeUnkownJoins.  Out (each optional, `cap` entries each): the edges, the scaffold starts as (contig, part), and
 * the linear walk from walk_start2 (through the join walk_first4 when not NULL) as node pairs in walking direction. */
int dh_scaffold_graph_probe(int32_t ncontigs, const int32_t *joins4, int32_t njoins, int32_t normalize, int32_t *edges4,
                            int32_t *nedges, int32_t *starts2, int32_t *nstarts, const int32_t *walk_start2,
                            const int32_t *walk_first4, int32_t *walk4, int32_t *walk_len, int32_t *cyclic, int32_t cap);
int dh_output_assembly(const char *fasta_path, const char *bed_path, const char *agp_path,
                       const uint8_t *contig_bases, const int64_t *contig_off, int32_t ncontigs,
                       const int32_t *scaffold_of, const char *const *headers, const int32_t *gap_len,
                       const dh_insertion *ins, int32_t nins, const uint8_t *ins_bases, const int32_t *read_ids,
                       const int32_t *read_ids_off, int32_t nreads, const char *const *read_names,
                       const dh_output_opts *opts, int32_t *dropped);

/* ---- DAZZ_DB files on disk (.db / .dam stub + hidden .idx / .bps / .hdr), host only.
 *      Replaces what DENTIST obtains by spawning fasta2DB / fasta2DAM / DBsplit
 *      (source/dentist/dazzler.d:6233-6330) and what the aligners open themselves.  The .idx, .bps
 *      and .hdr images are byte-identical to DAZZ_DB's (pinned by tests/test-commands.sh:54-61). */
typedef struct dh_dazz dh_dazz;
int dh_dazz_create_dam(const char *path, const char *fasta_text, int64_t n); /* fasta2DAM -i */
int dh_dazz_create_db(const char *path, const char *fasta_text, int64_t n);  /* fasta2DB -i  */
int dh_dazz_split(const char *path, int32_t cutoff, int32_t all, int64_t size_mb); /* DBsplit -x -a -s */
/* path may name a block ("reads.3"); the trimmed view is returned (ids are trimmed ids) */
int dh_dazz_open(const char *path, dh_dazz **out);
void dh_dazz_close(dh_dazz *db);
int32_t dh_dazz_nreads(const dh_dazz *db);
int32_t dh_dazz_first_id(const dh_dazz *db);         /* trimmed id of the first read of the block */
const uint8_t *dh_dazz_bases(const dh_dazz *db);      /* base codes 0..3, concatenated              */
const int64_t *dh_dazz_offsets(const dh_dazz *db);    /* nreads + 1                                 */
const int32_t *dh_dazz_origin(const dh_dazz *db);     /* well (DB) / contig number in scaffold (DAM)*/
const int32_t *dh_dazz_fpulse(const dh_dazz *db);     /* first pulse (DB) / contig start (DAM)      */
const int32_t *dh_dazz_flags(const dh_dazz *db);      /* DAZZ_READ.flags: low 10 bits = RQ * 1000 (DB) */
const char *dh_dazz_header(const dh_dazz *db, int32_t i); /* DAM: scaffold header of contig i; DB: prolog */
/* mask tracks `<dir>/.<db>.<name>.anno/.data` (source/dentist/dazzler.d:4870-5170): .anno = int32
 * nreads, int32 size (0), int64 byte offsets[nreads + 1]; .data = int32 (begin, end) pairs.
 * read: intervals of the opened (trimmed) view, ptr has nreads + 1 entries; returns the number of
 * intervals or a negative error; iv may be NULL to size.  write: for the whole trimmed DB. */
int64_t dh_dazz_read_mask(const dh_dazz *db, const char *db_path, const char *name, int64_t *ptr, int32_t *iv,
                          int64_t iv_cap);
int dh_dazz_write_mask(const char *db_path, const char *name, int32_t nreads, const int64_t *ptr,
                       const int32_t *iv);

/* ---- the binary containers between DENTIST's commands (host only):
 *      pile-ups.db (collect -> process, source/dentist/common/binio/pileupdb.d:400-897) and
 *      insertions.db (process -> output, binio/insertiondb.d:738-1031), byte-compatible with the D
 *      structs on x86-64.  Flat description used on both sides of the ABI:
 *        dh_seeded    one SeededAlignment: alignment chain (id, contig A = reference contig, contig B =
 *                     read, DENTIST flag bits 1 complement 2 disabled 4 alternateChain 8 chainContinuation
 *                     16 unchained), trace spacing, seed (0 front, 1 back), nla local alignments
 *        dh_chain_la  one LocalAlignment with ntp trace points
 *        tp           (numDiffs, numBasePairs) u16 pairs of all local alignments, in order
 *      pile-ups.db: npiles pile-ups of nra_of_pile[p] read alignments of nsa_of_ra[r] (1 or 2) seeded
 *      alignments.  insertions.db: nins insertions, each with its sequence (codes a,c,g,t = 0..3,
 *      stored 4 per byte as a=0 c=1 t=2 g=3), noverlaps seeded alignments and nread_ids read ids. */
typedef struct {
    int64_t id;
    uint32_t contig_a_id, contig_a_len, contig_b_id, contig_b_len;
    uint8_t flags, seed;
    uint16_t tspace;
    int32_t nla;
} dh_seeded;
typedef struct {
    uint32_t a_begin, a_end, b_begin, b_end, diffs;
    int32_t ntp;
} dh_chain_la;
typedef struct {
    int64_t start_contig, end_contig; /* ContigNode.contigId (1-based contig ids)                         */
    uint8_t start_part, end_part;     /* ContigPart: 0 pre, 1 begin, 2 end, 3 post (scaffold.d:77-90)     */
    uint8_t pad[6];
    int64_t seq_len, contig_len;
    int32_t noverlaps, nread_ids;
} dh_insertion_rec;
typedef struct dh_chaindb dh_chaindb; /* a parsed container, owned by the library */
int dh_pileupdb_write(const char *path, int32_t npiles, const int32_t *nra_of_pile, const int32_t *nsa_of_ra,
                      const dh_seeded *sa, const dh_chain_la *la, const uint16_t *tp);
int dh_pileupdb_read(const char *path, dh_chaindb **out);
int dh_insertiondb_write(const char *path, int32_t nins, const dh_insertion_rec *ins, const uint8_t *bases,
                         const uint32_t *read_ids, const dh_seeded *sa, const dh_chain_la *la, const uint16_t *tp);
int dh_insertiondb_read(const char *path, dh_chaindb **out);
/* `dentist merge-insertions` (commands/mergeInsertions.d:42-164): the insertions.db files of the process batches
 * (snakemake/Snakefile:1315-1334) merged into one file, ordered by (start contig, start part, end contig, end
 * part) like Insertion.opCmp (util/math.d:527-545); an unsorted input is sorted first (:66-72), equal keys keep
 * the order of the inputs (:120-138).  *ntotal (may be NULL) = insertions written. */
int dh_insertiondb_merge(const char *const *paths, int32_t npaths, const char *out_path, int64_t *ntotal);
/* pile-ups.db of a collect result (collectPileUps/package.d:88-96): every read of a pile-up becomes a
 * ReadAlignment of two SeededAlignments (left contig seeded at the back, right contig at the front) */
int dh_pileups_write_db(const dh_pileups *p, const dh_la *las, int64_t n, const uint16_t *trace,
                        const int64_t *contig_off, int32_t ncontigs, const int64_t *read_off, int32_t nreads,
                        int32_t tspace, const char *path);
/* insertions.db of a process result (processPileUps/package.d:156-158, 789-805): one insertion per
 * closed gap = (left contig, end) -> (right contig, begin), the whole consensus, its two flank overlaps
 * with trace points (tspace = dh_process_opts.tspace_pile) and the sorted 1-based read ids */
int dh_insertions_write_db(const dh_insertions *r, const int64_t *contig_off, int32_t ncontigs, int32_t tspace,
                           const char *path);
void dh_chaindb_destroy(dh_chaindb *d);
int32_t dh_chaindb_npiles(const dh_chaindb *d);
const int32_t *dh_chaindb_pile_counts(const dh_chaindb *d);
int32_t dh_chaindb_nread_alignments(const dh_chaindb *d);
const int32_t *dh_chaindb_read_alignment_counts(const dh_chaindb *d);
int64_t dh_chaindb_nseeded(const dh_chaindb *d);
const dh_seeded *dh_chaindb_seeded(const dh_chaindb *d);
int64_t dh_chaindb_nlas(const dh_chaindb *d);
const dh_chain_la *dh_chaindb_las(const dh_chaindb *d);
int64_t dh_chaindb_ntrace(const dh_chaindb *d); /* number of u16 values = 2 x trace points */
const uint16_t *dh_chaindb_trace(const dh_chaindb *d);
int32_t dh_chaindb_ninsertions(const dh_chaindb *d);
const dh_insertion_rec *dh_chaindb_insertions(const dh_chaindb *d);
const uint8_t *dh_chaindb_bases(const dh_chaindb *d);
const uint32_t *dh_chaindb_read_ids(const dh_chaindb *d);

/* byte tracks `<dir>/.<db>.<name>.anno/.data` (the `qual` / `inqual` intrinsic-QV tracks DASqv and
 * computeintrinsicqv write and `DBdump -i` shows, dazzler.d:2877-2897, 6142-6183): .anno = int32 nreads,
 * int32 8, int64 byte offsets[nreads + 1]; .data = the bytes (one QV per trace tile).  read: bytes of the
 * opened (trimmed) view, ptr has nreads + 1 entries, data may be NULL to size; returns the byte count. */
int dh_dazz_write_track(const char *db_path, const char *name, int32_t nreads, const int64_t *ptr,
                        const uint8_t *data);
int64_t dh_dazz_read_track(const dh_dazz *db, const char *db_path, const char *name, int64_t *ptr, uint8_t *data,
                           int64_t cap);
/* DBrm (dazzler.d:216, 6115-6119): removes the stub and every hidden file of the DB */
int dh_dazz_remove(const char *db_path);

#ifdef __cplusplus
}
#endif
#endif
