// dh_join.h -- the per-pile-up k-mer join that seeds the symmetric all-vs-all of a GROUPED DB (A == B, group =
// pile-up; `daligner -s126 -l500 pileup.db pileup.db`, source/dentist/commands/processPileUps/package.d:474-485).
//
// The generic seed filter (k_seed) looks every k-mer of a read up in a hash directory of the whole DB: one random
// 64-byte line per k-mer, and for a pile-up -- whose intact k-mers all share a bucket with their ~coverage copies --
// a walk of dependent loads per bucket; the directory itself (k_group_index) is written to HBM and read back.
// Reads of one pile-up only ever meet reads of the same pile-up, so the join is done per pile-up and never leaves
// the CU:
//
//   k_join_part   one pass over the DB: every sampled, unmasked k-mer becomes an 8-byte entry
//                     canon << 32 | orientation << 31 | palindrome << 30 | read-in-group << 21 | position
//                 (k <= 16) and is binned by a hash of its canonical k-mer into one of NS_g slices of its group.
//                 A block covers JP_POS consecutive k-mer chunks of one group, groups its entries by slice in
//                 LDS and writes them to ITS OWN region of the entry array (region = block index * JP_POS:
//                 no global atomics, no capacity to overflow) plus a row (start, count) per slice.
//   k_join        a block per (group, slice): gathers the slice's entries from the group's part blocks (coalesced),
//                 chains them in an LDS hash table keyed by the canonical k-mer, and for every entry taken as the
//                 B side walks its chain: equal k-mers of the same orientation are forward-strand hits, of the
//                 opposite orientation reverse-strand hits (palindromes both) -- exactly the hits the directory
//                 lookups produce (same -t cap per orientation class, same self / symmetric-pair rules,
//                 dh_kernels.hip seed_item::emit).  Hits are counted per B read, a contiguous range of the hit
//                 buffer is reserved with ONE device-scope atomic per block, and the hits are written grouped by
//                 B read; segtab[segrow[read] + slice] = (first hit, count).
//   k_seed<.., JOIN>  the seed filter's back end unchanged (sort by (strand, diagonal, position), band coverage,
//                 candidates): it gathers the read's hits from its NS_g segments instead of looking k-mers up.
//
// Results are bit-identical to the directory path by construction (the same multiset of hits per read); the
// directory path stays for everything else (A != B, k > 16, ungrouped DBs) and as the fallback when a slice
// overflows its LDS table.
#ifndef DH_JOIN_H
#define DH_JOIN_H
#include <stdint.h>

#include "dh_device.h"

#define JP_THREADS 512
#define JP_PER 16                       /* k-mer start positions per thread of k_join_part */
#define JP_POS (JP_THREADS * JP_PER)    /* entries a part block can produce */
#define JOIN_MAX_READS 512              /* reads per group (9 bits of an entry) */
#define JOIN_MAX_LEN (1 << 21)          /* bases per read (21 bits of an entry) */
#define JOIN_MAX_SLICES 512
#define JOIN_CAP 4096                   /* entries of a (group, slice) the LDS table of k_join holds */
#define JOIN_FILL 2300                  /* planned entries per slice (56 % of the table) */
#define JOIN_THREADS 512

#define DH_ST_JOIN_OVERFLOW 0x8         /* a (group, slice) exceeded JOIN_CAP: the call falls back to the directory path */
#define DH_ST_JOIN_HITCAP 0x10          /* the hit buffer was too small: cursor holds the size needed, k_join is rerun */

struct JoinView {
    // plan (host-built tables, one upload per call)
    const int32_t *gfirst;    // [ngroups + 1] first read of group g
    const int32_t *gns;       // [ngroups] slices of group g
    const int32_t *pfirst;    // [ngroups + 1] part blocks of group g
    const int2 *pblk;         // [npart] (group, first chunk of the block inside the group's chunk space)
    const int64_t *psubrow;   // [npart] row of the block in psub
    const int2 *jblk;         // [njoin] (group, slice)
    const int64_t *segrow;    // [nreads] row of the read in segtab
    // device-produced
    uint32_t *psub;           // per part block and slice: start << 16 | count  (start, count <= JP_POS = 8192 < 2^16)
    uint64_t *entries;        // npart * JP_POS
    uint64_t *segtab;         // per read and slice: first hit << 24 | count
    uint64_t *hits;           // hit buffer (same encoding as the seed kernel's: strand << 63 | D << 24 | q)
    unsigned long long *cursor;  // hits reserved so far
    int64_t hits_cap;
    int32_t *status;
    int32_t npart, njoin;
    // the partitioned join of a mapping pass (dh_mjoin.h) feeds the same back end: no groups, every read has ns_fixed
    // segments, row (read - read0) * ns_fixed of segtab (gns == NULL selects this form)
    int32_t ns_fixed, read0;
};

#ifdef __cplusplus
extern "C" {
#endif
void dhk_join_part(hipStream_t st, JoinView jv, DbView B, int32_t k, int32_t kmer_mod);
void dhk_join(hipStream_t st, JoinView jv, DbView B, DhOpts o, const int64_t *goff, int32_t sepv);
/* reads with more than 2048 / 4096 / 8192 hits and the largest hit count of a read */
void dhk_join_hist(hipStream_t st, JoinView jv, const int32_t *group, int32_t nreads, unsigned int *out4);
/* the seed filter's back end fed from the join's hit segments (cap: LDS hit capacity as in dhk_seed) */
void dhk_seed_join(hipStream_t st, int cap, DbView B, IndexView ix, DhOpts o, JoinView jv, int32_t item0, int32_t nitems,
                   DhCand *cand, int32_t *ncand, int32_t *nhits, int32_t *status, uint32_t *queue, int32_t ncu,
                   uint64_t *fscr, const int32_t *read_list, int32_t nlist);
void dhk_seed_big_join(hipStream_t st, DbView B, IndexView ix, DhOpts o, JoinView jv, const int32_t *read_list,
                       int32_t nreads, uint64_t *gbuf, int32_t gcap, DhCand *cand, int32_t *ncand, int32_t *nhits,
                       int32_t *status, uint32_t *queue, int32_t ncu);
#ifdef __cplusplus
}
#endif
#endif
