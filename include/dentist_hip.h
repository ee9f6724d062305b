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
#define DH_EOVERFLOW (-4) /* a device-side capacity (hit buffer, trace pool) was exceeded */
#define DH_EIO (-5)
#define DH_ENOMEM (-6)

const char *dh_last_error(void);
/* library / ABI version, bumped on any struct change */
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
    int32_t skip_self;   /* 1: A is B, skip aread == bread (absence of -I)                   */
    int32_t dmax;        /* cap on differences per extension                                 */
    int32_t width;       /* live diagonals of the wave, <= 62 (one 64-lane wavefront)        */
    int32_t reserved;
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
    int32_t wave_launches, pad;
} dh_align_stats;
int dh_get_align_stats(dh_ctx *ctx, dh_align_stats *out);

/*
 * dh_align_db -- every sequence of B against all of A: k-mer seeds, diagonal band filter, wave
 * local alignment with trace points.  Replaces the spawns
 *   `damapper -C -T<t> -e0.7 ... <ref> <reads>`   dazzler.d:6158-6170 (getDamapping :3855-3866,
 *                                                 workflow call snakemake/Snakefile:1143-1170)
 *   `daligner -T<a> -B -s126 -l500 -e0.7 db db`   dazzler.d:6121-6140 (getDalignment :3829-3844)
 *   `daligner -A ... contigs consensus`           processPileUps/package.d:655-667
 * Output: LAs in LAsort order (base.d:1787-1809).  select_best != 0 additionally sets the
 * chain flags damapper emits (START/BEST, consumer dazzler.d:1728-1758).
 */
int dh_align_db(dh_ctx *ctx, dh_db *A, dh_db *B, const dh_align_opts *opts, int32_t select_best,
                dh_la_set **out);

/* ---- .las files: replaces the reader/writer pair of source/dentist/dazzler.d:1665-1834
 *      (LocalAlignmentReader) and :1913-1960, 2130-2170 (writeAlignments/writeDazzlerOverlap). */
int dh_las_write(const char *path, const dh_la *las, int64_t n, const uint16_t *trace,
                 int32_t tspace);
int dh_las_read(const char *path, dh_la_set **out);

#ifdef __cplusplus
}
#endif
#endif
