// dh_internal.h -- declarations shared by the host translation units of libdentist_hip.so.
#ifndef DH_INTERNAL_H
#define DH_INTERNAL_H

#include "../../include/dentist_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>
#include <new>
#include <utility>
#include <vector>

#include "dh_device.h"

#define DB_PAD 64
#define PK_PAD 128 /* bytes of padding on both sides of a packed copy (k_tile reads 64 bytes past a window origin) */

// Caching device allocator: hipMalloc / hipFree synchronise the device and cost 0.1-1 ms each;
// the same buffer sizes recur on every call, so freed blocks are kept in size-class free lists and
// handed out again (all work runs on one stream, so stream order protects reuse).  The cache is
// released when the last context is destroyed.
hipError_t dh_dev_alloc(void **p, size_t bytes);
void dh_dev_free(void *p);
void dh_dev_trim();
template <typename T>
inline hipError_t dh_dev_alloc(T **p, size_t bytes)
{
    return dh_dev_alloc((void **)p, bytes);
}

// memset that fills the chip for large buffers (dh_kernels.hip: k_fill16); small ones go to the runtime
extern "C" hipError_t dhk_memset(hipStream_t st, void *ptr, int value, size_t nbytes);

int dh_fail(int code, const std::string &msg);
#include <memory>
#include <utility>
#include <vector>
struct dh_scaffold;
// a vector whose resize() leaves trivially constructible elements unwritten: the gathered records of a plan are written
// once by the host threads (a value-initialising resize was 2.2 of the plan's 6.3 ms: 14 MB of first-touch page faults on
// one thread)
template <class T>
struct dh_noinit_alloc : std::allocator<T> {
    template <class U>
    struct rebind {
        using other = dh_noinit_alloc<U>;
    };
    template <class U, class... A>
    void construct(U *p, A &&...a)
    {
        if constexpr (sizeof...(A) == 0)
            ::new ((void *)p) U;
        else
            ::new ((void *)p) U(std::forward<A>(a)...);
    }
};
typedef std::vector<dh_la, dh_noinit_alloc<dh_la>> dh_la_vec;
// dh_scaffold.cpp: the scaffold of the gathered join blobs of the sharded collector (glas: two LA records per join)
int dh_scaffold_from_join_blobs(const uint8_t *const *blobs, const int64_t *sizes, int32_t world, int32_t ncontigs,
                                const int32_t *input_gaps, int32_t ngaps, const struct dh_scaffold_opts *opts,
                                dh_la_vec &glas, dh_scaffold **out);
// allocate total + 2 * DB_PAD bytes filled with code 4; *base = alloc + DB_PAD
int dh_alloc_bases(hipStream_t st, int64_t total, uint8_t **alloc, uint8_t **base, bool pads_only = false);

#define HIPCHK(expr)                                                                             \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return dh_fail(DH_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));          \
    } while (0)

struct dh_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int ncu = 0;
    hipEvent_t ev[8] = {};
    // second stream for the device-to-host copies of a chunk's records (they overlap the next chunk's
    // kernels); cev[0..1]: compaction done (main stream), cev[2..3]: copy done (copy stream), by chunk parity
    hipStream_t cstream = nullptr;
    hipEvent_t cev[4] = {};
    dh_align_stats stats = {};
    dh_cum_stats cum = {};
    double mj_hit_frac = 0.2;  // hits per sampled k-mer the hit pool of the partitioned join is sized for (raised when a pool overflowed)
    int seed_wave_tier = 1;  // the wavefront-per-read first tier of the segment-fed seed back end (0: most reads overflowed it)
    int64_t mj_chunks = 0, mj_fallbacks = 0;  // chunks seeded by the partitioned join / redone by the directory (dh_get_mjoin_counts)
    int32_t near_best_ppm = -1;  // damapper -n of this context (dh_ctx_set_near_best); -1 = the process default
    // second context of the same device (own streams and scratch), created on first use: the process stage runs the
    // two halves of a batch of pile-ups concurrently, one on each (dh_process_pileups)
    dh_ctx *sub[3] = {nullptr, nullptr, nullptr};  // contexts of the concurrent parts of dh_process_pileups
    // grow-only device scratch buffers reused across calls (hipMalloc/hipFree of GB-sized
    // buffers per call costs milliseconds and synchronises the device)
    struct Arena {
        void *p = nullptr;
        size_t cap = 0;
    } arena[96];
};
// slot `id` of the context's scratch arena, at least `bytes` large
int dh_scratch(dh_ctx *ctx, int id, size_t bytes, void **out);
// per-sequence flags of a symmetric all-vs-all (DbView::pflags); NULL clears them
int dh_db_set_pflags(struct dh_db *db, const uint8_t *flags);

struct dh_index {
    // d_dir (build only, released afterwards) points one word into its allocation: d_dir[-1] == 0, so that
    // bucket b is [d_dir[b - 1], d_dir[b]) for every b; the seed kernel reads d_fat (dh_device.h)
    uint32_t *d_dir = nullptr, *d_dir_alloc = nullptr;
    ulonglong2 *d_fat = nullptr;
    ulonglong2 *d_ent = nullptr;
    int64_t *d_goff = nullptr;
    int32_t *d_page_seq = nullptr;  // virtual page (4096 bases) -> sequence
    int64_t n = 0;
    int32_t k = 0, sepv = 0, shift = 0, pbits = 0, na = 0, kmer_mod = 1;
    // presence bitmap of the entries' keys for the partitioned join of a mapping pass (dh_mjoin.h): bit key >> (2k - nbbits),
    // built on first use
    uint32_t *d_bitmap = nullptr;
    int32_t nbbits = 0;
    bool light = false;  // virtual axis only (no directory): the hits come from the k-mer join (dh_join.hip)
    void release()
    {
        dh_dev_free(d_dir_alloc);
        d_dir_alloc = nullptr;
        dh_dev_free(d_ent);
        dh_dev_free(d_goff);
        dh_dev_free(d_fat);
        d_fat = nullptr;
        dh_dev_free(d_page_seq);
        d_page_seq = nullptr;
        dh_dev_free(d_bitmap);
        d_bitmap = nullptr;
        nbbits = 0;
        d_dir = nullptr;
        d_ent = nullptr;
        d_goff = nullptr;
    }
};

struct dh_db {
    dh_ctx *ctx = nullptr;
    int32_t n = 0, max_len = 0, ngroups = 1;
    int64_t total = 0;
    // bases/rc point DB_PAD bytes into their allocations: kernels read 8 bases at a time on both
    // sides of a position, the padding (code 4) keeps those loads inside the buffer
    uint8_t *d_bases = nullptr, *d_rc = nullptr, *d_bases_alloc = nullptr, *d_rc_alloc = nullptr;
    // 2-bit packed copies for the wave kernel (built on demand; 16 bytes of padding on both sides);
    // has_n: -1 unknown, 0 only codes 0..3 (packed path usable), 1 other codes present
    uint8_t *d_pk = nullptr, *d_rcpk = nullptr, *d_pk_alloc = nullptr, *d_rcpk_alloc = nullptr;
    int32_t has_n = -1;
    int64_t *d_off = nullptr;
    int32_t *d_group = nullptr;
    std::vector<int64_t> h_off;
    std::vector<int32_t> h_group;
    dh_index ix;
    bool has_ix = false;
    // soft mask: one bit per base of d_bases (bit g of the buffer = base g), 16 bytes of slack at the
    // end for the kernels' unaligned 8-byte reads; nullptr = nothing masked
    uint8_t *d_mask_bits = nullptr;
    // the two layers behind it: the caller's tracks (dh_db_set_mask replaces this layer only; slices
    // inherit into it) and what the library derived (DBdust, alignment coverage).  d_mask_bits is the
    // layer itself while only one exists, their OR in a buffer of its own when both do.
    uint8_t *d_mask_user = nullptr, *d_mask_derived = nullptr;
    uint8_t *d_pflags = nullptr;  // see DbView::pflags (dh_db_set_pflags)
    DbView view() const { return DbView{d_bases, d_off, d_group, n, d_mask_bits, d_pflags}; }
};

// Result buffers live in pooled page-locked host memory: device-to-host copies into them run at
// PCIe speed without the runtime's staging pass, and resize() does not zero-fill.  Without a
// device (CPU-only use of the .las codec) the pool falls back to malloc.
void *dh_pinned_alloc(size_t bytes);
void dh_pinned_free(void *p, size_t bytes);
void dh_pinned_trim();
template <typename T>
struct PinnedAlloc {
    using value_type = T;
    PinnedAlloc() = default;
    template <class U>
    PinnedAlloc(const PinnedAlloc<U> &) {}
    T *allocate(size_t n)
    {
        void *p = dh_pinned_alloc(n * sizeof(T));
        if (!p) throw std::bad_alloc();
        return (T *)p;
    }
    void deallocate(T *p, size_t n) { dh_pinned_free(p, n * sizeof(T)); }
    template <class U>
    void construct(U *p) { ::new ((void *)p) U; }  // default-init: no zero fill on resize()
    template <class U, class... Args>
    void construct(U *p, Args &&...a) { ::new ((void *)p) U(std::forward<Args>(a)...); }
    bool operator==(const PinnedAlloc &) const { return true; }
    bool operator!=(const PinnedAlloc &) const { return false; }
};
using LaVec = std::vector<dh_la, PinnedAlloc<dh_la>>;
using TraceVec = std::vector<uint16_t, PinnedAlloc<uint16_t>>;

struct dh_la_set {
    LaVec la;
    TraceVec trace;
    int32_t tspace = 0;
    // device copy of `trace` left behind by dh_align_db_ex (scratch arena): valid until the next
    // alignment call on the same context, nullptr when the result came in several chunks
    const uint16_t *d_trace = nullptr;
    int64_t d_trace_len = 0;  // > 0: `trace` was left on the device on request (dh_align_db_ex want_sorted & 2), the host vector is empty
    // want_sorted & 4 (with & 2, symmetric DH-2 call in one chunk): the RECORDS stay on the device as well -- d_la_n of them,
    // grouped by A-read item, d_item_off[item] = first record of item (read * 2 + strand), nitems + 1 entries; `la` is empty.
    // Valid until the next alignment call on the context.
    DhLa *d_la = nullptr;
    int64_t d_la_n = 0;
    const uint32_t *d_item_off = nullptr;
    // B reads whose items overflowed a per-item capacity (dropped records / no candidates), see
    // dh_align_stats.overflow_items; the pile-up path skips the pile-ups of such reads
    std::vector<int32_t> ovf_reads;
    // dh_map_reads(want_sorted & 8): the trace values of ALL chunks stay on the device in a buffer this set owns (toff
    // indexes it); `trace` is empty until somebody asks for it (dh_la_set_trace downloads it then).  The cropper of the
    // process stage reads the trace of one record in ten: dh_process_pileups_set gathers those on the device.
    uint16_t *d_trace_own = nullptr;
    int64_t d_trace_own_len = 0, d_trace_own_cap = 0;
    int device = 0;
    ~dh_la_set();
};
// the host copy of a set's trace values, downloaded on first use when they were left on the device
int dh_la_set_ensure_host_trace(dh_la_set *s);

// result of the process stage (dh_process.cpp; dh_comm.cpp assembles the gathered result of all ranks)
struct dh_insertions {
    std::vector<dh_insertion> rec;
    std::vector<uint8_t> bases;
    // what insertions.db stores besides the sequence (insertiondb.d:987-1031): the two flank overlaps
    // of every closed gap with their trace points (A coordinates on the whole contig) and the read
    // ids of the pile-up
    std::vector<dh_la> flank;        // 2 per closed gap: left, right; toff into flank_tr
    std::vector<uint16_t> flank_tr;
    std::vector<int32_t> flank_of;   // per record: index of its left overlap in `flank`, -1 if none
    std::vector<int32_t> ids_off, ids;  // per record [ids_off[i], ids_off[i+1]): read ids of the pile-up
};

// Alignment chains as units (base.d:306-421; chain flags dazzler.d:1728-1758, 1991-1998): a record with NEXT (and not
// START) continues the chain of the record before it (same contig, read and strand).  The view holds one pseudo record
// per chain -- the first member's fields with aepos / bepos of the last member and diffs summed, i.e. what
// first.contigX.begin / last.contigX.end / totalDiffs read -- plus the A bases its members cover (coveredBases!"contigA")
// and where the chain's members are.  `trivial`: every chain is one record (the view is not filled).
struct dh_chain_view {
    bool trivial = true;
    std::vector<dh_la> unit;       // one per chain
    std::vector<int64_t> first;    // chain -> index of its first record; first[nchains] = n
    std::vector<int64_t> covered;  // chain -> sum of (aepos - abpos) over its members
};
void dh_chain_view_build(const dh_la *las, int64_t n, dh_chain_view &v);
inline bool dh_continues_chain(const dh_la &prev, const dh_la &cur)
{
    return (cur.flags & DH_FLAG_NEXT) && !(cur.flags & DH_FLAG_START) && cur.aread == prev.aread && cur.bread == prev.bread &&
           (cur.flags & DH_FLAG_COMP) == (prev.flags & DH_FLAG_COMP);
}

void dh_pileups_shift(dh_pileups *p, int32_t by);
int dh_pileups_concat(dh_pileups *const *parts, int32_t nparts, dh_pileups **out);

template <typename T>
struct DevBuf {
    T *p = nullptr;
    ~DevBuf() { dh_dev_free(p); }
    hipError_t alloc(size_t n) { return dh_dev_alloc(&p, sizeof(T) * std::max<size_t>(n, 1)); }
};

// build a DB whose bases are slices [beg, beg+len) of sequences of `src` (device-to-device)
int dh_db_from_slices(dh_ctx *ctx, const dh_db *src, const std::vector<int32_t> &sidx,
                      const std::vector<int32_t> &sbeg, const std::vector<int32_t> &slen,
                      const std::vector<int32_t> &group, dh_db **out, bool inherit_mask = false);
// adopt device bases allocated with dh_alloc_bases (ownership moves to the DB)
int dh_db_adopt(dh_ctx *ctx, uint8_t *d_alloc, uint8_t *d_bases, const std::vector<int64_t> &off,
                const std::vector<int32_t> &group, dh_db **out);
int dh_ensure_rc(dh_db *db);
// a layer of the DB's mask bitmap (derived != 0: the library's own bits, else the caller's / inherited
// tracks), allocated and zeroed on first use; write into it, then call dh_mask_recompose
int dh_ensure_mask_layer(dh_db *db, int derived, uint8_t **out);
int dh_mask_recompose(dh_db *db);
void dh_mask_free(dh_db *db);
// DBdust: ORs the low-complexity mask (k_dust) into the DB's mask bitmap; drops the cached index
int dh_db_dust_impl(dh_db *db);
int dh_ensure_packed(dh_db *db, bool with_rc);
// dh_align_db with the final LAsort made optional (internal callers regroup on their own): want_sorted bit 0 = LAsort,
// bit 1 = leave the trace values on the device (dh_la_set.d_trace / d_trace_len) when the call is one chunk
int dh_align_db_ex(dh_ctx *ctx, dh_db *A, dh_db *B, const dh_align_opts *opts, int32_t want_best,
                   int32_t want_sorted, dh_la_set **out);

#endif
