// dh_device.h -- POD views shared by the kernels (dh_kernels.hip) and the host side (dh_api.cpp).
#ifndef DH_DEVICE_H
#define DH_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#define KM_TILE 4096 /* A positions per block of the k-mer passes (256 threads x 16) */

#define DH_ST_HIT_OVERFLOW 0x1
#define DH_ST_CAND_OVERFLOW 0x2
#define DH_ST_POOL_OVERFLOW 0x4

struct DhOpts {  // == dh_align_opts
    int32_t k, hmin, band_shift, tspace, min_len, pen, xdrop, max_err_ppm, max_cand, max_la, tcap,
        strands, skip_self, dmax, width, kmer_mod, algo;
};

struct DbView {
    const uint8_t *bases;  // concatenated base codes
    const int64_t *off;    // n + 1 offsets
    const int32_t *group;  // optional
    int32_t n;
    // optional soft mask (daligner -m tracks, DBdust): one bit per base of `bases` (bit g = base g is
    // masked); k-mers touching a masked base are neither indexed nor looked up
    const uint8_t *mask_bits;
    // optional, symmetric all-vs-all only (skip_self == 2): pflags[s] bit 0 = records with sequence s as A read are
    // wanted, bit 1 = s may be the B read of a wanted record.  The record (a, b) is wanted iff (pflags[a] & 1) and
    // (pflags[b] & 2); a pair yields hits iff one of its two records is wanted, an unwanted record is not written and the
    // transposed pair is aligned only for a wanted one (dh_process: the funnel, the tile QVs and the consensus read the
    // overlaps of the reads that may serve as reference read only, processPileUps/package.d:461-472, 518-568; with
    // dh_process_opts.max_partners against a bounded set of partners).  NULL = every record.  oracle/align.c has the
    // same rule (oz_db.pflags).
    const uint8_t *pflags;
};
#if defined(__HIPCC__)
#define DH_HDI __host__ __device__ inline
#else
#define DH_HDI inline
#endif
DH_HDI bool dh_rec_wanted(const uint8_t *f, int32_t a, int32_t b) { return (f[a] & 1) && (f[b] & 2); }
DH_HDI bool dh_pair_seeded(const uint8_t *f, int32_t a, int32_t b) { return dh_rec_wanted(f, a, b) || dh_rec_wanted(f, b, a); }

// fat directory word of a bucket (16 bytes, one load per looked-up k-mer):
//   x == ~0            empty bucket
//   x >> 62 == 1       several entries: y = first entry | count << 32
//   else               the bucket's only entry itself, (x, y) as in `ent` -- no second line to fetch
#define DH_FAT_EMPTY (~0ull)
struct IndexView {
    const ulonglong2 *fat;  // one word per bucket
    const ulonglong2 *ent;  // x = group * 4^k + canonical k-mer, bit 63: the k-mer of A is its reverse complement; y = aseq << 40 | virtual position
    const int64_t *goff;   // virtual offset of every A sequence
    const int32_t *page_seq;  // A sequence of every 4096-base page of the virtual axis (sequences start on page boundaries)
    int64_t n;
    int32_t na, sepv, shift, pbits;
};

struct DhCand {
    int32_t score, aseq, apos, bpos;
};

struct DhLa {  // == dh_la
    int32_t tlen, diffs, abpos, bbpos, aepos, bepos;
    uint32_t flags;
    int32_t aread, bread, pad;
    int64_t toff;
};

struct DhNode {
    int32_t parent, d, j;
};

struct WaveScratch {
    DhNode *pool;     // nslots * poolcap trace-tree nodes
    int32_t *cdj;     // nslots * 4 * nbmax boundary records
    uint32_t *queue;  // work-item counter
    const int4 *units;        // optional work units (item - item0, first candidate, end candidate, 0)
    const uint32_t *nunits;   // their number (device side)
    int32_t poolcap, nbmax;
    int32_t *item_ovf;        // per item (absolute index): set when a record of the item was dropped (SYM: > max_la)
};

#ifdef __cplusplus
extern "C" {
#endif
void dhk_revcomp(hipStream_t st, const uint8_t *src, uint8_t *dst, const int64_t *off, int32_t n,
                 int32_t max_len);
void dhk_kmer_pass(hipStream_t st, int fill, DbView A, const int2 *tiles, int32_t ntiles, int32_t k,
                   int32_t kmer_mod, int32_t shift, uint32_t *dir, ulonglong2 *ent, const int64_t *goff);
/* the two passes for a grouped DB whose groups own contiguous bucket ranges: LDS counters, no global atomics */
#define DH_GI_SLICE 32768
void dhk_group_index(hipStream_t st, int fill, DbView A, const int2 *tiles, const int32_t *gtile, int32_t ngroups,
                     int32_t slices_per_group, int32_t slice, int32_t k, int32_t kmer_mod, int32_t shift, uint32_t *dir,
                     ulonglong2 *ent, const int64_t *goff);
void dhk_scan(hipStream_t st, uint32_t *v, int64_t n, uint32_t *sums);
void dhk_scan_total(hipStream_t st, uint32_t *v, int64_t n, uint32_t *sums, unsigned long long *total64);
void dhk_seed_summary(hipStream_t st, const int32_t *ncand, const int32_t *nhits, int32_t n, unsigned long long *out4);
// dir[b] = end of bucket b (dir[-1] == 0) -> the fat directory
void dhk_fat_dir(hipStream_t st, const uint32_t *dir, const ulonglong2 *ent, int64_t nb, ulonglong2 *fat);
void dhk_seed(hipStream_t st, int cap, DbView B, IndexView ix, DhOpts o,
              int32_t item0, int32_t nitems, DhCand *cand, int32_t *ncand, int32_t *nhits,
              int32_t *status, uint32_t *queue, int32_t ncu, uint64_t *fscr);
// scratch of the 8192-entry seed variant: DH_SEED_FSCR_WORDS 8-byte words per resident block
// (prefix sums u32[8192] + band heads u16[8192]), DH_SEED_FSCR_BLOCKS_PER_CU blocks per CU at most
#define DH_SEED_FSCR_WORDS 6144
#define DH_SEED_FSCR_WORDS16 24576 /* the 16384-entry variant fed from segments: u64 sums + u32 band heads */
#define DH_SEED_FSCR_BLOCKS_PER_CU 4
void dhk_seed_big(hipStream_t st, DbView B, IndexView ix, DhOpts o,
                  const int32_t *read_list, int32_t nreads, uint64_t *gbuf, int32_t gcap, DhCand *cand,
                  int32_t *ncand, int32_t *nhits, int32_t *status, uint32_t *queue, int32_t ncu);
void dhk_wave(hipStream_t st, int32_t nslots, DbView A, DbView B, const uint8_t *brc, const uint8_t *apk,
              const uint8_t *bpk, const uint8_t *brcpk, DhOpts o,
              int32_t item0, int32_t nitems, const DhCand *cand, const int32_t *ncand,
              WaveScratch ws, DhLa *out_la, uint16_t *out_trace, int32_t trmax, int32_t *out_nla,
              int32_t *out_ntr, unsigned long long *counters, int32_t *status);
void dhk_wave2(hipStream_t st, int32_t nslots, DbView A, DbView B, const uint8_t *arc, const uint8_t *brc,
               const uint8_t *apk, const uint8_t *arcpk, const uint8_t *bpk, const uint8_t *brcpk, DhOpts o,
               int32_t item0, int32_t nitems, const DhCand *cand, const int32_t *ncand, WaveScratch ws,
               DhLa *out_la, uint16_t *out_trace, int32_t trmax, int32_t *out_nla, int32_t *out_ntr,
               unsigned long long *counters, int32_t *status);
void dhk_units(hipStream_t st, const DhCand *cand, const int32_t *ncand, int32_t item0, int32_t nitems,
               int32_t max_cand, void *units, uint32_t *nunits);
void dhk_pack2(hipStream_t st, const uint8_t *src, int64_t total, uint8_t *dst, int32_t *flag);
void dhk_pack2_planes(hipStream_t st, const uint8_t *src, int64_t total, uint8_t *dst, int32_t *flag);
void dhk_pack2_rc_planes(hipStream_t st, const uint8_t *src, const int64_t *off, int32_t n, int32_t max_len, int64_t a0,
                         uint8_t *dst);  // zeroes the shared words itself
void dhk_planes_rc(hipStream_t st, const uint8_t *fwd_planes, const int64_t *off, int32_t n, int32_t max_len, int64_t a0,
                   uint8_t *dst);  // plane-packed reverse complements from the plane-packed forward copy; zeroes the shared words itself
void dhk_or_words(hipStream_t st, uint32_t *dst, const uint32_t *a, const uint32_t *b, int64_t n);
void dhk_dust(hipStream_t st, const uint8_t *bases, const int64_t *off, const int2 *tiles, int32_t ntiles,
              int32_t chunk, uint32_t *bits);
void dhk_cov_events(hipStream_t st, const DhLa *las, int64_t n, const int64_t *off, const int64_t *roff,
                    int32_t improper_only, int32_t allowance, uint32_t *diff);
void dhk_cov_mask(hipStream_t st, const uint32_t *cov, const int64_t *off, int32_t nseq, int32_t max_len, int32_t lower,
                  int32_t upper, uint32_t *bits);
void dhk_pack2_rc(hipStream_t st, const uint8_t *src, const int64_t *off, int32_t n, int32_t max_len, int64_t a0,
                  uint8_t *dst);
void dhk_pack2_rc_bounds(hipStream_t st, const int64_t *off, int32_t n, int64_t a0, uint8_t *dst);
void dhk_mask_slices(hipStream_t st, const uint32_t *src_bits, const int64_t *src_off, const int32_t *sidx,
                     const int32_t *sbeg, const int64_t *dst_off, int32_t n, int32_t max_len, uint32_t *dst_bits);
void dhk_compact(hipStream_t st, const DhLa *la_slots, const uint16_t *tr_slots, int32_t trmax,
                 int32_t max_la, int32_t ordered, int32_t nitems, const uint32_t *la_off,
                 const uint32_t *tr_off, int64_t tr_base, DhLa *la_out, uint16_t *tr_out);
#ifdef __cplusplus
}
#endif
#endif
