// dh_mjoin.h -- the seeds of a MAPPING pass (A = contigs, B = a chunk of reads; `damapper ref reads.block`,
// source/dentist/dazzler.d:6158-6170, one call per read block: snakemake/Snakefile:1143-1170) as a radix-partitioned
// k-mer join instead of one random directory line per looked-up k-mer.
//
// The generic seed filter (k_seed, dh_kernels.hip) asks the fat directory of A for every sampled k-mer of a read: a
// random 64-byte line each, of which 16 bytes are used and 94 % find nothing (13 % error: 0.87^20 intact 20-mers).  The
// part's random-line rate (55 G lines/s, scripts/rand_access_probe.cpp) bounds it: 2 x 28.6 ms per step at 1/8
// sampling, 458 ms without sampling.  Here the k-mers of the reads are first binned by the top MJ_PBITS bits of their
// canonical k-mer -- a contiguous slice of the directory and of a presence bitmap per bin --, then every bin is joined
// with its slice held on chip:
//
//   k_mj_bitmap    (per index) one presence bit per bucket of a fine directory of A's keys (2^nbbits buckets, about 16 per
//                  indexed k-mer): the filter a partition's block keeps in LDS (2^(nbbits - 10) bits <= 128 KB).
//   k_mj_part      one pass over the chunk's bases: a block takes a TILE of `tb` consecutive bases, rolls the canonical
//                  k-mers of its positions (every lane starts from k - 1 bases packed with a few bit operations instead
//                  of k - 1 rolling steps), and turns every sampled k-mer inside one read into an 8-byte entry
//                      partition << 53 | key remainder << 19 | palindrome << 18 | orientation << 17 | position in tile
//                  staged in LDS, counting-sorted by partition there and written to the tile's own 64 KB region with one
//                  coalesced pass (no global atomics) + the 1024 segment offsets of the tile.
//   k_mj_transpose segment offsets tile-major -> partition-major (start << 16 | count), so that a partition's block
//                  reads them coalesced.
//   k_mj_probe     a block per CU keeps the bitmap slice of ONE partition in LDS; blocks of one XCD work on the same
//                  partition (its 2 MB directory slice then lives in that XCD's L2: 255 G lookups/s measured against
//                  54 G/s from HBM, scripts/join_probe.cpp).  A lane takes the segment (tile, partition) -- 8 entries on
//                  average --, tests every entry against the bitmap (LDS), looks the survivors (~7 %) up in the fat
//                  directory with exactly the rules of k_seed (-t cap per orientation class, strands) and appends the
//                  hits  strand << 63 | virtual A position << 23 | position in the tile group  to the wavefront's page of
//                  the hit pool, tile group by tile group; hseg[group][partition] = first hit << 24 | count.
//   k_mj_regroup   a block per tile group: gathers the group's hits from the 1024 partitions' lists, finds the read of
//                  every hit, finishes it (diagonal, position on the oriented read: the hit encoding of k_seed), sorts the
//                  hits by read in LDS and writes one contiguous range per read; segtab[read][group - first group of the
//                  read] = first hit << 24 | count.
//   k_seed<.., JOIN>  the seed filter's back end unchanged: it gathers a read's hits from its segments.
//
// The multiset of hits a read gets is the one the directory lookups produce (same index entries, same rules), so the
// candidates and everything behind them are bit-identical; DH_NO_MJOIN=1 forces the directory path (tests compare the
// two).  The directory path stays for A == B, grouped DBs, k > 22, DBs with codes outside a/c/g/t, small calls, and as the
// fall-back of a chunk in which any capacity below was exceeded (repeat-rich reads: the entries of a segment can yield
// more hits than a page holds, more than 2 048 reads can begin inside one tile group).
#ifndef DH_MJOIN_H
#define DH_MJOIN_H
#include <stdint.h>

#include "dh_device.h"
#include "dh_join.h"

#define MJ_PBITS 10
#define MJ_P (1 << MJ_PBITS)      /* partitions */
#define MJ_CAP 8192               /* entries a tile can hold (its LDS staging buffer and its region of the entry array) */
#define MJ_THREADS 512            /* k_mj_part, k_mj_regroup */
#define MJ_RS 1024                /* read starts inside one tile */
#define MJ_POSBITS 17             /* position inside a tile */
#define MJ_MAXK 22                /* 2k - 10 bits of key remainder + 19 low bits + 10 partition bits <= 63 */
#define MJ_MINK 10
#define MJ_GROUP 16               /* tiles per group (the unit the hits are regrouped by read in) */
#define MJ_BATCH 64               /* tiles per wavefront batch of k_mj_probe = 4 groups */
#define MJ_PAGE 16384             /* hits per page of the hit pool (a wavefront owns its current page) */
#define MJ_PROBE_THREADS 1024
#define MJ_SLICES 32              /* work items per partition of k_mj_probe */
#define MJ_RG_READS 2048          /* reads that begin inside one tile group */
#define MJ_MAXBITS 30             /* bits of the presence bitmap: 2^20 per partition = 128 KB of LDS */

#define DH_ST_MJ_OVERFLOW 0x20    /* a capacity of the partitioned join was exceeded: the chunk is redone by the directory path */
#define DH_ST_MJ_POOL 0x40        /* the hit pool was too small: ctr[12] counts what found no room, the chunk is run again */

struct MjView {
    int64_t c0, c1;        // base range of the chunk in B.bases
    int32_t r0, r1;        // its reads [r0, r1)
    int32_t tb;            // bases per tile
    int32_t ntiles, ntiles_pad, ngroups;  // ntiles_pad: multiple of MJ_BATCH; ngroups = ntiles_pad / MJ_GROUP
    int32_t k, kmer_mod, nbbits, nseg;    // nseg: segments (tile groups) a read can span
    uint64_t *ent;         // ntiles * MJ_CAP
    uint16_t *segoff;      // ntiles * MJ_P: first entry of segment (tile, partition) inside the tile
    uint32_t *tile_n;      // entries of tile t
    int32_t *tile_r;       // first read that starts behind the first base of tile t
    uint32_t *seg;         // MJ_P * ntiles_pad: start << 16 | count
    const uint32_t *bitmap;
    unsigned long long *hseg;  // ngroups * MJ_P: first hit << 24 | count
    uint64_t *hits;        // npages * MJ_PAGE
    uint64_t *rhits;       // hits grouped by read (capacity rcap)
    unsigned long long *segtab;  // (r1 - r0) * nseg
    uint32_t *ctr;         // [0..7] work queues of k_mj_probe (one per XCD), [8] page cursor, [10..11] 64-bit cursor of rhits, [12..13] survivors that found no page
    int64_t rcap;
    int32_t npages;
    int32_t *status;
    int32_t dbg;           // development switches (DH_MJ_DBG)
};

#ifdef __cplusplus
extern "C" {
#endif
/* presence bitmap of the index entries: bit (key >> (2k - nbbits)); bm must hold 2^(nbbits - 5) zeroed words */
void dhk_mj_bitmap(hipStream_t st, const ulonglong2 *ent, int64_t n, int32_t k, int32_t nbbits, uint32_t *bm);
/* first read behind the first base of every tile (m.tile_r): launched ahead of the chunk's other kernels */
void dhk_mj_tile_reads(hipStream_t st, DbView B, MjView m);
/* partition, transpose, filter, hits of one chunk (ctr zeroed by the callee; m.tile_r filled by dhk_mj_tile_reads) */
void dhk_mj_run(hipStream_t st, DbView B, IndexView ix, DhOpts o, MjView m, int32_t ncu);
#ifdef __cplusplus
}
#endif
#endif
