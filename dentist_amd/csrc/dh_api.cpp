// dh_api.cpp -- host side of libdentist_hip.so: the C ABI of include/dentist_hip.h.
//
// Owns device memory (hipMalloc, sized for 288 GB HBM3E: whole DBs stay resident, the k-mer
// index is built in HBM, per-slot wave scratch is preallocated), sequences launches on the
// context's stream and times the stages with HIP events on that stream.  No CPU fallback: every
// compute entry point needs a working HIP device.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>

#include "dh_internal.h"
#include "dh_join.h"
#include "dh_mjoin.h"
#include "dh_tile.h"
#include "dh_parallel.h"

static_assert(sizeof(dh_align_opts) == sizeof(DhOpts), "opts layout");
static_assert(sizeof(dh_la) == sizeof(DhLa), "la layout");
static_assert(sizeof(dh_la) == 48, "la size");

static thread_local std::string g_err;

int dh_fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}
#define fail dh_fail

#ifdef DH_SEED_PROF
extern "C" void dhk_seed_prof_dump();
extern "C" void dhk_join_prof_dump();
extern "C" void dhk_tile_prof_dump();
#endif
// ------------------------------------------------------------------------------------ allocator
#include <map>
#include <mutex>
#include <unordered_map>
namespace {
std::mutex g_alloc_mu;
// free blocks per (device, size class): a block is only handed back to the device it lives on
std::map<std::pair<int, size_t>, std::vector<void *>> g_free_lists;
std::unordered_map<void *, std::pair<int, size_t>> g_block_size;
int current_device()
{
    int d = 0;
    (void)hipGetDevice(&d);
    return d;
}
size_t size_class(size_t bytes)
{
    if (bytes < 4096) return 4096;
    size_t p = 4096;
    while (p < bytes) p <<= 1;  // next power of two, then steps of p/8 below it
    const size_t step = p >> 4;
    return (bytes + step - 1) / step * step;
}
}  // namespace

hipError_t dh_dev_alloc(void **p, size_t bytes)
{
    const size_t cls = size_class(bytes);
    const int dev = current_device();
    {
        std::lock_guard<std::mutex> lk(g_alloc_mu);
        auto it = g_free_lists.find(std::make_pair(dev, cls));
        if (it != g_free_lists.end() && !it->second.empty()) {
            *p = it->second.back();
            it->second.pop_back();
            return hipSuccess;
        }
    }
    hipError_t e = hipMalloc(p, cls);
    if (e != hipSuccess) {  // out of memory: drop the cache and retry once
        (void)hipGetLastError();  // (the failure is sticky: a later hipGetLastError() after a launch would report it)
        dh_dev_trim();
        e = hipMalloc(p, cls);
        if (e != hipSuccess) (void)hipGetLastError();
    }
    if (e == hipSuccess) {
        std::lock_guard<std::mutex> lk(g_alloc_mu);
        g_block_size[*p] = std::make_pair(dev, cls);
    }
    return e;
}

void dh_dev_free(void *p)
{
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_alloc_mu);
    auto it = g_block_size.find(p);
    if (it == g_block_size.end()) {
        (void)hipFree(p);
        return;
    }
    g_free_lists[it->second].push_back(p);
}

void dh_dev_trim()
{
    std::lock_guard<std::mutex> lk(g_alloc_mu);
    for (auto &kv : g_free_lists)
        for (void *p : kv.second) {
            g_block_size.erase(p);
            (void)hipFree(p);
        }
    g_free_lists.clear();
}

// pooled page-locked host memory (power-of-two classes from 64 KiB); smaller requests and the
// no-device case use malloc
namespace {
std::mutex g_pin_mu;
std::map<size_t, std::vector<void *>> g_pin_free;
std::unordered_map<void *, size_t> g_pin_size;  // pinned blocks (in use or pooled) -> class
size_t g_pin_pooled = 0;
constexpr size_t PIN_MIN = 1u << 16, PIN_POOL_MAX = 4ull << 30;
size_t pin_class(size_t bytes)
{
    size_t p = PIN_MIN;
    while (p < bytes) p <<= 1;
    return p;
}
}  // namespace

void *dh_pinned_alloc(size_t bytes)
{
    if (bytes < PIN_MIN) return malloc(std::max<size_t>(bytes, 1));
    const size_t cls = pin_class(bytes);
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        auto it = g_pin_free.find(cls);
        if (it != g_pin_free.end() && !it->second.empty()) {
            void *p = it->second.back();
            it->second.pop_back();
            g_pin_pooled -= cls;
            return p;
        }
    }
    void *p = nullptr;
    if (hipHostMalloc(&p, cls, hipHostMallocDefault) == hipSuccess && p) {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        g_pin_size[p] = cls;
        return p;
    }
    (void)hipGetLastError();
    return malloc(bytes);
}

void dh_pinned_free(void *p, size_t bytes)
{
    if (!p) return;
    if (bytes >= PIN_MIN) {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        auto it = g_pin_size.find(p);
        if (it != g_pin_size.end()) {
            if (g_pin_pooled + it->second <= PIN_POOL_MAX) {
                g_pin_free[it->second].push_back(p);
                g_pin_pooled += it->second;
            } else {
                g_pin_size.erase(it);
                (void)hipHostFree(p);
            }
            return;
        }
    }
    free(p);
}

void dh_pinned_trim()
{
    std::lock_guard<std::mutex> lk(g_pin_mu);
    for (auto &kv : g_pin_free)
        for (void *p : kv.second) {
            g_pin_size.erase(p);
            (void)hipHostFree(p);
        }
    g_pin_free.clear();
    g_pin_pooled = 0;
}

extern "C" const char *dh_last_error(void) { return g_err.c_str(); }
extern "C" int32_t dh_abi_version(void) { return 4; }  // 4: dh_process_opts.max_partners, .min_relative_score_ppm (64 bytes); 3: dh_align_opts.algo

// ------------------------------------------------------------------------------------ context


extern "C" int dh_ctx_create(int32_t device, void *stream, dh_ctx **out)
{
    if (!out) return fail(DH_EINVAL, "dh_ctx_create: out is NULL");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(DH_ENODEV, "no HIP device available (libdentist_hip has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(DH_EINVAL, "dh_ctx_create: bad device index");
    HIPCHK(hipSetDevice(device));
    dh_ctx *c = new dh_ctx();
    struct CtxCreateGuard {
        dh_ctx *&c;
        bool ok = false;
        ~CtxCreateGuard()
        {
            if (!ok && c) {
                for (auto &e : c->ev)
                    if (e) (void)hipEventDestroy(e);
                for (auto &e : c->cev)
                    if (e) (void)hipEventDestroy(e);
                if (c->cstream) (void)hipStreamDestroy(c->cstream);
                if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
                delete c;
            }
        }
    } cguard{c};
    c->device = device;
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    c->ncu = prop.multiProcessorCount;
    if (stream) {
        c->stream = (hipStream_t)stream;
    } else {
        HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    for (auto &e : c->ev) HIPCHK(hipEventCreate(&e));
    HIPCHK(hipStreamCreateWithFlags(&c->cstream, hipStreamNonBlocking));
    for (auto &e : c->cev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    cguard.ok = true;
    *out = c;
    return DH_OK;
}

extern "C" void dh_ctx_destroy(dh_ctx *c)
{
    if (!c) return;
    for (dh_ctx *s : c->sub)
        if (s) dh_ctx_destroy(s);
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->cstream) (void)hipStreamSynchronize(c->cstream);
    for (auto &e : c->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto &e : c->cev)
        if (e) (void)hipEventDestroy(e);
    if (c->cstream) (void)hipStreamDestroy(c->cstream);
    for (auto &a : c->arena)
        if (a.p) dh_dev_free(a.p);
    dh_dev_trim();
    dh_pinned_trim();
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int dh_scratch(dh_ctx *ctx, int id, size_t bytes, void **out)
{
    dh_ctx::Arena &a = ctx->arena[id];
    if (bytes > a.cap) {
        if (a.p) {
            HIPCHK(hipStreamSynchronize(ctx->stream));
            dh_dev_free(a.p);
            a.p = nullptr;
            a.cap = 0;
        }
        const size_t want = bytes + bytes / 8 + 256;
        hipError_t e = dh_dev_alloc(&a.p, want);
        if (e != hipSuccess) {
            // (the other slots cannot be released from here: the caller holds pointers into the ones it asked for earlier
            // in the same call.  Between calls the host can: dh_ctx_release_scratch)
            a.p = nullptr;
            a.cap = 0;
            return fail(DH_EHIP, std::string("device scratch of ") + std::to_string(want >> 20) + " MB: " +
                                     hipGetErrorString(e) + " (dh_ctx_release_scratch frees what earlier calls keep)");
        }
        a.cap = want;
    }
    *out = a.p;
    return DH_OK;
}

extern "C" int dh_ctx_release_scratch(dh_ctx *c)
{
    if (!c) return fail(DH_EINVAL, "ctx is NULL");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->cstream) HIPCHK(hipStreamSynchronize(c->cstream));
    for (dh_ctx *s2 : c->sub)
        if (s2) (void)dh_ctx_release_scratch(s2);
    for (auto &a : c->arena)
        if (a.p) {
            dh_dev_free(a.p);
            a.p = nullptr;
            a.cap = 0;
        }
    dh_dev_trim();
    return DH_OK;
}

extern "C" int dh_ctx_sync(dh_ctx *c)
{
    if (!c) return fail(DH_EINVAL, "ctx is NULL");
    HIPCHK(hipStreamSynchronize(c->stream));
    return DH_OK;
}

int dh_db_set_pflags(dh_db *db, const uint8_t *flags)
{
    if (!db) return fail(DH_EINVAL, "dh_db_set_pflags: NULL");
    if (!flags) {
        dh_dev_free(db->d_pflags);
        db->d_pflags = nullptr;
        return DH_OK;
    }
    if (!db->d_pflags) HIPCHK(dh_dev_alloc(&db->d_pflags, (size_t)std::max(db->n, 1)));
    HIPCHK(hipMemcpyAsync(db->d_pflags, flags, (size_t)db->n, hipMemcpyHostToDevice, db->ctx->stream));
    HIPCHK(hipStreamSynchronize(db->ctx->stream));
    return DH_OK;
}

extern "C" int dh_get_mjoin_counts(dh_ctx *c, int64_t *out2, int32_t reset)
{
    if (!c || !out2) return fail(DH_EINVAL, "dh_get_mjoin_counts: NULL");
    out2[0] = c->mj_chunks;
    out2[1] = c->mj_fallbacks;
    if (reset) c->mj_chunks = c->mj_fallbacks = 0;
    return DH_OK;
}

extern "C" int dh_get_align_stats(dh_ctx *c, dh_align_stats *out)
{
    if (!c || !out) return fail(DH_EINVAL, "dh_get_align_stats: NULL");
    *out = c->stats;
    return DH_OK;
}

extern "C" int dh_get_cum_stats(dh_ctx *c, dh_cum_stats *out, int32_t reset)
{
    if (!c || !out) return fail(DH_EINVAL, "dh_get_cum_stats: NULL");
    *out = c->cum;
    if (reset) c->cum = dh_cum_stats();
    return DH_OK;
}

extern "C" void dh_default_align_opts(dh_align_opts *o)
{
    memset(o, 0, sizeof(*o));
    o->k = 14;
    o->hmin = 35;
    o->band_shift = 6;
    o->tspace = 100;
    o->min_len = 500;
    o->pen = 6;
    o->xdrop = 120;
    o->max_err_ppm = 300000;
    o->max_cand = 32;
    o->max_la = 4;
    o->tcap = 64;
    o->strands = 3;
    o->skip_self = 0;
    o->dmax = 60000;
    o->width = 30;  /* two alignments per wavefront (k_wave2); up to 62 selects one per wavefront */
    o->kmer_mod = 1;
}

// ------------------------------------------------------------------------------------ DB

// pads_only: the caller writes every base itself (dh_db_create: one copy of the whole array) -- only the DB_PAD bytes
// on both sides get the code 4.  Filling all of a reads DB first wrote 15.7 GB for configs[2] that the copy then overwrote.
int dh_alloc_bases(hipStream_t st, int64_t total, uint8_t **alloc, uint8_t **base, bool pads_only)
{
    const size_t nb = (size_t)std::max<int64_t>(total, 0) + 2 * DB_PAD;
    HIPCHK(dh_dev_alloc(alloc, nb));
    if (pads_only) {
        HIPCHK(hipMemsetAsync(*alloc, 4, DB_PAD, st));
        HIPCHK(hipMemsetAsync(*alloc + nb - DB_PAD, 4, DB_PAD, st));
    } else
        HIPCHK(dhk_memset(st, *alloc, 4, nb));
    *base = *alloc + DB_PAD;
    return DH_OK;
}



extern "C" int dh_db_create(dh_ctx *ctx, const uint8_t *bases, const int64_t *off, int32_t n,
                            const int32_t *group, dh_db **out)
{
    if (!ctx || !off || !out || n < 0) return fail(DH_EINVAL, "dh_db_create: bad argument");
    if (n > 0 && !bases) return fail(DH_EINVAL, "dh_db_create: bases is NULL");
    HIPCHK(hipSetDevice(ctx->device));
    dh_db *db = new dh_db();
    struct DbCreateGuard {  // releases the half-built DB on any early return
        dh_db *&d;
        bool ok = false;
        ~DbCreateGuard()
        {
            if (!ok && d) {
                dh_dev_free(d->d_bases_alloc);
                dh_dev_free(d->d_off);
                dh_dev_free(d->d_group);
                delete d;
            }
        }
    } guard{db};
    db->ctx = ctx;
    db->n = n;
    db->h_off.assign(off, off + n + 1);
    db->total = off[n] - off[0];
    if (off[0] != 0) return fail(DH_EINVAL, "dh_db_create: off[0] must be 0");
    for (int32_t i = 0; i < n; i++) {
        const int64_t l = off[i + 1] - off[i];
        if (l < 0 || l >= (1 << 24)) return fail(DH_EINVAL, "dh_db_create: sequence length must be in [0, 2^24)");
        db->max_len = std::max<int32_t>(db->max_len, (int32_t)l);
    }
    if (group) {
        db->h_group.assign(group, group + n);
        for (int32_t g : db->h_group) {
            if (g < 0) return fail(DH_EINVAL, "dh_db_create: negative group id");
            db->ngroups = std::max(db->ngroups, g + 1);
        }
    }
    if (int rc = dh_alloc_bases(ctx->stream, db->total, &db->d_bases_alloc, &db->d_bases, true)) return rc;
    HIPCHK(dh_dev_alloc(&db->d_off, sizeof(int64_t) * (size_t)(n + 1)));
    if (db->total > 0)
        HIPCHK(hipMemcpyAsync(db->d_bases, bases, (size_t)db->total, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(db->d_off, off, sizeof(int64_t) * (size_t)(n + 1), hipMemcpyHostToDevice,
                          ctx->stream));
    if (group && n > 0) {
        HIPCHK(dh_dev_alloc(&db->d_group, sizeof(int32_t) * (size_t)n));
        HIPCHK(hipMemcpyAsync(db->d_group, group, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice,
                              ctx->stream));
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    guard.ok = true;
    *out = db;
    return DH_OK;
}

extern "C" void dh_db_destroy(dh_db *db)
{
    if (!db) return;
    (void)hipSetDevice(db->ctx->device);
    (void)hipStreamSynchronize(db->ctx->stream);
    dh_dev_free(db->d_bases_alloc);
    dh_dev_free(db->d_rc_alloc);
    dh_dev_free(db->d_pk_alloc);
    dh_dev_free(db->d_rcpk_alloc);
    dh_dev_free(db->d_off);
    dh_dev_free(db->d_group);
    dh_mask_free(db);
    dh_dev_free(db->d_pflags);
    if (db->has_ix) db->ix.release();
    delete db;
}

static size_t mask_bytes(const dh_db *db) { return (size_t)((db->total + 31) / 32) * 4 + 16; }

void dh_mask_free(dh_db *db)
{
    if (db->d_mask_bits != db->d_mask_user && db->d_mask_bits != db->d_mask_derived) dh_dev_free(db->d_mask_bits);
    dh_dev_free(db->d_mask_user);
    dh_dev_free(db->d_mask_derived);
    db->d_mask_bits = db->d_mask_user = db->d_mask_derived = nullptr;
}

int dh_ensure_mask_layer(dh_db *db, int derived, uint8_t **out)
{
    uint8_t *&layer = derived ? db->d_mask_derived : db->d_mask_user;
    if (!layer) {
        HIPCHK(dh_dev_alloc(&layer, mask_bytes(db)));
        HIPCHK(dhk_memset(db->ctx->stream, layer, 0, mask_bytes(db)));
    }
    *out = layer;
    return DH_OK;
}

// d_mask_bits = the only layer there is, or the OR of the two in a buffer of its own
int dh_mask_recompose(dh_db *db)
{
    uint8_t *u = db->d_mask_user, *d = db->d_mask_derived;
    const bool own = db->d_mask_bits && db->d_mask_bits != u && db->d_mask_bits != d;
    if (u && d) {
        if (!own) {
            db->d_mask_bits = nullptr;
            HIPCHK(dh_dev_alloc(&db->d_mask_bits, mask_bytes(db)));
        }
        dhk_or_words(db->ctx->stream, (uint32_t *)db->d_mask_bits, (const uint32_t *)u, (const uint32_t *)d,
                     (int64_t)(mask_bytes(db) / 4));
        HIPCHK(hipGetLastError());
        return DH_OK;
    }
    if (own) {
        HIPCHK(hipStreamSynchronize(db->ctx->stream));
        dh_dev_free(db->d_mask_bits);
    }
    db->d_mask_bits = u ? u : d;
    return DH_OK;
}

// soft mask of the DB (union of the daligner -m tracks): per sequence sorted, disjoint intervals.
// SET semantics: the call replaces the tracks of an earlier call; what the library derived itself
// (dh_db_dust, dh_db_mask_coverage) is a layer of its own and stays -- the effective mask is the OR
// of the two.  Passing ptr == NULL clears the whole mask, both layers.  The cached k-mer index is dropped.
extern "C" int dh_db_set_mask(dh_db *db, const int64_t *ptr, const int32_t *iv)
{
    if (!db) return fail(DH_EINVAL, "db is NULL");
    HIPCHK(hipSetDevice(db->ctx->device));
    HIPCHK(hipStreamSynchronize(db->ctx->stream));
    if (db->has_ix) db->ix.release();
    db->has_ix = false;
    if (!ptr) {
        dh_mask_free(db);
        return DH_OK;
    }
    for (int32_t s = 0; s < db->n; s++) {
        if (ptr[s] > ptr[s + 1]) return fail(DH_EINVAL, "dh_db_set_mask: pointers must be non-decreasing");
        const int64_t len = db->h_off[(size_t)s + 1] - db->h_off[(size_t)s];
        for (int64_t j = ptr[s]; j < ptr[s + 1]; j++)
            if (iv[2 * j] < 0 || iv[2 * j] > iv[2 * j + 1] || iv[2 * j + 1] > len ||
                (j > ptr[s] && iv[2 * j] < iv[2 * j - 1]))
                return fail(DH_EINVAL, "dh_db_set_mask: intervals must be sorted, disjoint and inside the sequence");
    }
    std::vector<uint8_t> bits(mask_bytes(db), 0);
    for (int32_t s = 0; s < db->n; s++)
        for (int64_t j = ptr[s]; j < ptr[s + 1]; j++)
            for (int64_t g = db->h_off[(size_t)s] + iv[2 * j]; g < db->h_off[(size_t)s] + iv[2 * j + 1]; g++)
                bits[(size_t)(g >> 3)] |= (uint8_t)(1u << (g & 7));
    uint8_t *layer;
    if (int rc = dh_ensure_mask_layer(db, 0, &layer)) return rc;
    HIPCHK(hipMemcpyAsync(layer, bits.data(), bits.size(), hipMemcpyHostToDevice, db->ctx->stream));
    if (int rc = dh_mask_recompose(db)) return rc;
    HIPCHK(hipStreamSynchronize(db->ctx->stream));
    return DH_OK;
}

int dh_db_dust_impl(dh_db *db)
{
    dh_ctx *ctx = db->ctx;
    if (db->has_ix) db->ix.release();
    db->has_ix = false;
    uint8_t *layer;
    if (int rc = dh_ensure_mask_layer(db, 1, &layer)) return rc;
    const int32_t chunk = db->max_len < 16384 ? 64 : 512;
    const int64_t tile = 256ll * chunk;
    std::vector<int2> tiles;
    for (int32_t s = 0; s < db->n; s++) {
        const int64_t len = db->h_off[(size_t)s + 1] - db->h_off[(size_t)s];
        for (int64_t a = 0; a < len - 15; a += tile) tiles.push_back(int2{s, (int32_t)a});
    }
    if (tiles.empty()) return dh_mask_recompose(db);
    DevBuf<int2> d_tiles;
    HIPCHK(d_tiles.alloc(tiles.size()));
    HIPCHK(hipMemcpyAsync(d_tiles.p, tiles.data(), sizeof(int2) * tiles.size(), hipMemcpyHostToDevice, ctx->stream));
    dhk_dust(ctx->stream, db->d_bases, db->d_off, d_tiles.p, (int32_t)tiles.size(), chunk, (uint32_t *)layer);
    HIPCHK(hipGetLastError());
    if (int rc = dh_mask_recompose(db)) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return DH_OK;
}

// DBdust (symmetric DUST, -w64 -t2.0 -m10; the reference runs it on every DB it aligns with -mdust,
// processPileUps/package.d:476, 655): the low-complexity mask is computed on the device and ORed
// into the DB's soft mask
extern "C" int dh_db_dust(dh_db *db)
{
    if (!db) return fail(DH_EINVAL, "db is NULL");
    HIPCHK(hipSetDevice(db->ctx->device));
    return dh_db_dust_impl(db);
}

// maskRepetitiveRegions (commands/maskRepetitiveRegions.d:129-176, 238-430): sequence regions whose
// alignment coverage lies outside [lower, upper] are ORed into the DB's soft mask; improper_only
// restricts the coverage to alignments that are not proper within `allowance` (the second assessor of
// the reads case, :157-176).  No alignments, no mask (:347-348).  The coverage is computed on the device:
// +1 / -1 events, one scan, one classification pass.
extern "C" int dh_db_mask_coverage(dh_db *db, const dh_la *las, int64_t n, const int64_t *read_off, int32_t nreads,
                                   int32_t lower, int32_t upper, int32_t improper_only, int32_t allowance)
{
    if (!db || (n > 0 && !las) || n < 0 || (improper_only && !read_off))
        return fail(DH_EINVAL, "dh_db_mask_coverage: bad argument");
    for (int64_t i = 0; i < n; i++) {
        const dh_la &l = las[i];
        if (l.aread < 0 || l.aread >= db->n || (improper_only && (l.bread < 0 || l.bread >= nreads)))
            return fail(DH_EINVAL, "dh_db_mask_coverage: id out of range");
        const int64_t alen = db->h_off[(size_t)l.aread + 1] - db->h_off[(size_t)l.aread];
        if (l.abpos < 0 || l.aepos > alen || l.abpos > l.aepos)
            return fail(DH_EINVAL, "dh_db_mask_coverage: alignment outside its contig");
    }
    if (n == 0) return DH_OK;
    dh_ctx *ctx = db->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    if (db->has_ix) db->ix.release();
    db->has_ix = false;
    uint8_t *layer;
    if (int rc = dh_ensure_mask_layer(db, 1, &layer)) return rc;
    const int64_t nslots = db->total + db->n + 2;
    DevBuf<uint32_t> d_cov, d_sums;
    DevBuf<dh_la> d_las;
    DevBuf<int64_t> d_roff;
    HIPCHK(d_cov.alloc((size_t)nslots));
    HIPCHK(d_sums.alloc((size_t)nslots / 2048 + 4));
    HIPCHK(d_las.alloc((size_t)n));
    HIPCHK(dhk_memset(st, d_cov.p, 0, sizeof(uint32_t) * (size_t)nslots));
    HIPCHK(hipMemcpyAsync(d_las.p, las, sizeof(dh_la) * (size_t)n, hipMemcpyHostToDevice, st));
    if (improper_only) {
        HIPCHK(d_roff.alloc((size_t)nreads + 1));
        HIPCHK(hipMemcpyAsync(d_roff.p, read_off, sizeof(int64_t) * ((size_t)nreads + 1), hipMemcpyHostToDevice, st));
    }
    dhk_cov_events(st, (const DhLa *)d_las.p, n, db->d_off, d_roff.p, improper_only ? 1 : 0, allowance, d_cov.p);
    HIPCHK(hipGetLastError());
    dhk_scan(st, d_cov.p, nslots, d_sums.p);
    HIPCHK(hipGetLastError());
    dhk_cov_mask(st, d_cov.p, db->d_off, db->n, db->max_len, lower, upper, (uint32_t *)layer);
    HIPCHK(hipGetLastError());
    if (int rc = dh_mask_recompose(db)) return rc;
    HIPCHK(hipStreamSynchronize(st));
    return DH_OK;
}

// --max-coverage-reads / --max-improper-coverage-reads from --read-coverage (commandline.d:1876-1889,
// 1957-1970)
extern "C" int32_t dh_max_coverage_reads(double x)
{
    return (int32_t)(x / std::log(std::log(std::log(0.1650612 * x + 5.9354533) / std::log(1.65))));
}
extern "C" int32_t dh_max_improper_coverage_reads(double x) { return (int32_t)(0.5 * x + std::exp(0.1875 * (8.0 - x))); }

// the mask as intervals (what `DBdust` writes into the `dust` track, dazzler.d:4943-5170): ptr gets
// n + 1 entries; iv may be NULL to size; returns the number of intervals or a negative error
extern "C" int64_t dh_db_get_mask(dh_db *db, int64_t *ptr, int32_t *iv, int64_t iv_cap)
{
    if (!db || !ptr) return fail(DH_EINVAL, "dh_db_get_mask: NULL argument");
    std::vector<uint8_t> bits(mask_bytes(db), 0);
    if (db->d_mask_bits) {
        if (hipSetDevice(db->ctx->device) != hipSuccess || hipStreamSynchronize(db->ctx->stream) != hipSuccess ||
            hipMemcpy(bits.data(), db->d_mask_bits, bits.size(), hipMemcpyDeviceToHost) != hipSuccess)
            return fail(DH_EHIP, "dh_db_get_mask: device to host copy failed");
    }
    int64_t m = 0;
    for (int32_t s = 0; s < db->n; s++) {
        ptr[s] = m;
        const int64_t o = db->h_off[(size_t)s], e = db->h_off[(size_t)s + 1];
        int64_t g = o;
        while (g < e) {
            if (!(bits[(size_t)(g >> 3)] >> (g & 7) & 1)) {
                g++;
                continue;
            }
            int64_t h = g;
            while (h < e && (bits[(size_t)(h >> 3)] >> (h & 7) & 1)) h++;
            if (iv && m < iv_cap) {
                iv[2 * m] = (int32_t)(g - o);
                iv[2 * m + 1] = (int32_t)(h - o);
            }
            m++;
            g = h;
        }
    }
    ptr[db->n] = m;
    return m;
}

extern "C" int32_t dh_db_nreads(const dh_db *db) { return db ? db->n : 0; }
extern "C" int64_t dh_db_total_bases(const dh_db *db) { return db ? db->total : 0; }

// drop cached derived data (k-mer index, reverse complement) so the next call rebuilds it
extern "C" int dh_db_drop_cache(dh_db *db)
{
    if (!db) return fail(DH_EINVAL, "db is NULL");
    (void)hipSetDevice(db->ctx->device);
    (void)hipStreamSynchronize(db->ctx->stream);
    if (db->has_ix) db->ix.release();
    db->has_ix = false;
    dh_dev_free(db->d_rc_alloc);
    db->d_rc = db->d_rc_alloc = nullptr;
    dh_dev_free(db->d_pk_alloc);
    dh_dev_free(db->d_rcpk_alloc);
    db->d_pk = db->d_pk_alloc = db->d_rcpk = db->d_rcpk_alloc = nullptr;
    db->has_n = -1;
    return DH_OK;
}

int dh_ensure_rc(dh_db *db)
{
    if (db->d_rc) return DH_OK;
    if (int rc = dh_alloc_bases(db->ctx->stream, db->total, &db->d_rc_alloc, &db->d_rc)) return rc;
    dhk_revcomp(db->ctx->stream, db->d_bases, db->d_rc, db->d_off, db->n, db->max_len);
    HIPCHK(hipGetLastError());
    return DH_OK;
}

// 2-bit packed copies for the wave kernel; leaves has_n = 1 (and no packed copy) when the DB
// holds codes outside 0..3
int dh_ensure_packed(dh_db *db, bool with_rc)
{
    if (db->has_n == 1) return DH_OK;
    hipStream_t st = db->ctx->stream;
    const size_t bytes = (size_t)((db->total + 31) / 32) * 8 + 2 * PK_PAD;
    if (!db->d_pk) {
        int32_t *d_flag;
        if (int rc = dh_scratch(db->ctx, 3, 4 * sizeof(int32_t), (void **)&d_flag)) return rc;
        HIPCHK(dh_dev_alloc((void **)&db->d_pk_alloc, bytes));
        db->d_pk = db->d_pk_alloc + PK_PAD;
        HIPCHK(hipMemsetAsync(d_flag + 1, 0, sizeof(int32_t), st));
        dhk_pack2(st, db->d_bases, db->total, db->d_pk, d_flag + 1);
        HIPCHK(hipGetLastError());
        int32_t flag = 0;
        HIPCHK(hipMemcpyAsync(&flag, d_flag + 1, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        db->has_n = flag ? 1 : 0;
        if (flag) {
            dh_dev_free(db->d_pk_alloc);
            db->d_pk = db->d_pk_alloc = nullptr;
            return DH_OK;
        }
    }
    if (with_rc && !db->d_rcpk) {
        if (int rc = dh_ensure_rc(db)) return rc;
        int32_t *d_flag;
        if (int rc = dh_scratch(db->ctx, 3, 4 * sizeof(int32_t), (void **)&d_flag)) return rc;
        HIPCHK(dh_dev_alloc((void **)&db->d_rcpk_alloc, bytes));
        db->d_rcpk = db->d_rcpk_alloc + PK_PAD;
        dhk_pack2(st, db->d_rc, db->total, db->d_rcpk, d_flag + 2);
        HIPCHK(hipGetLastError());
    }
    return DH_OK;
}

static int32_t ceil_log2(uint64_t x)
{
    int32_t b = 0;
    while ((1ull << b) < x) b++;
    return b;
}

// light: only the virtual axis (goff, page table) -- what the seed filter's back end needs when the hits come from the
// per-pile-up k-mer join (dh_join.hip) instead of directory lookups
static int build_index(dh_db *A, int32_t k, int32_t sepv, int32_t kmer_mod, bool light = false)
{
    dh_ctx *ctx = A->ctx;
    if (A->has_ix && A->ix.k == k && A->ix.sepv == sepv && A->ix.kmer_mod == kmer_mod && (light || !A->ix.light)) return DH_OK;
    if (A->has_ix) A->ix.release();
    A->has_ix = false;
    dh_index &ix = A->ix;
    ix = dh_index();
    ix.k = k;
    ix.sepv = sepv;
    ix.kmer_mod = kmer_mod;
    ix.na = A->n;
    ix.light = light;
    // virtual offsets and tile table
    std::vector<int64_t> goff((size_t)A->n + 1);
    std::vector<int2> tiles;
    int64_t g = 0, nk = 0;
    for (int32_t s = 0; s < A->n; s++) {
        goff[(size_t)s] = g;
        const int64_t len = A->h_off[(size_t)s + 1] - A->h_off[(size_t)s];
        g += (len + sepv + 4095) & ~4095ll;  // 4096-aligned starts: see sepv in align_range
        if (len >= k && !light) {
            nk += len - k + 1;
            for (int64_t st = 0; st < len - k + 1; st += KM_TILE) tiles.push_back(int2{s, (int32_t)st});
        }
    }
    goff[(size_t)A->n] = g;
    if (g >= (1ll << 39))
        return fail(DH_EINVAL, "index: virtual coordinate space exceeds 2^39 (every sequence takes its length + the longest "
                               "B read + 64, rounded up to 4096)");
    if (A->n >= (1 << 24)) return fail(DH_EINVAL, "index: more than 2^24 sequences");
    if (light) {
        HIPCHK(dh_dev_alloc(&ix.d_goff, sizeof(int64_t) * (size_t)(A->n + 1)));
        HIPCHK(hipMemcpyAsync(ix.d_goff, goff.data(), sizeof(int64_t) * goff.size(), hipMemcpyHostToDevice, ctx->stream));
        std::vector<int32_t> page_seq((size_t)(g >> 12) + 1, A->n > 0 ? A->n - 1 : 0);
        for (int32_t s2 = 0; s2 < A->n; s2++)
            for (int64_t pg = goff[(size_t)s2] >> 12; pg < (goff[(size_t)s2 + 1] >> 12); pg++) page_seq[(size_t)pg] = s2;
        HIPCHK(dh_dev_alloc(&ix.d_page_seq, sizeof(int32_t) * page_seq.size()));
        HIPCHK(hipMemcpyAsync(ix.d_page_seq, page_seq.data(), sizeof(int32_t) * page_seq.size(), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));  // the vectors go out of scope
        A->has_ix = true;
        return DH_OK;
    }
    const int32_t keybits = 2 * k + ceil_log2((uint64_t)A->ngroups);
    if (keybits > 62) return fail(DH_EINVAL, "index: k-mer key does not fit 62 bits");
    int32_t pbits = ceil_log2((uint64_t)std::max<int64_t>(nk, 1));
    int32_t pmax = 27;
    // more indexed k-mers than 2^27 buckets can keep apart (a 3 Gb assembly at kmer_mod 4: 750 M): about one bucket per
    // entry, up to 2^30 -- at 5.6 entries per bucket every lookup walked a chain of dependent loads (configs[4]: seeds
    // 631 -> 223 ms per 25 Gbp of reads, index build 81 -> 128 ms; 17 GB of directory, the part has 288)
    const int64_t expect = nk / std::max(1, kmer_mod);
    if (expect > (1ll << 27)) pmax = std::min(30, ceil_log2((uint64_t)expect) + 1);
    if (const char *e = getenv("DH_INDEX_PBITS")) pmax = std::max(10, std::min(30, atoi(e)));  // development
    pbits = std::max(10, std::min(pbits, std::min(keybits, pmax)));
    ix.pbits = pbits;
    ix.shift = keybits - pbits;
    // the largest key is ngroups * 4^k - 1, so buckets up to (that >> shift) are addressable
    const int64_t nb = (int64_t)((((uint64_t)A->ngroups << (2 * k)) - 1) >> ix.shift) + 1;
    // bucket offsets are 32 bits wide: the k-mers actually indexed (about nk / kmer_mod of the positions: the
    // modimer hash samples evenly) have to stay below 2^32, with 1/16 of headroom for the sampling's spread
    if (nk / std::max(1, kmer_mod) >= (1ll << 32) - (1ll << 28))
        return fail(DH_EINVAL, "index: more than 2^32 indexed k-mers (32-bit bucket offsets); raise kmer_mod");
    HIPCHK(dh_dev_alloc(&ix.d_dir_alloc, sizeof(uint32_t) * (size_t)(nb + 2)));
    ix.d_dir = ix.d_dir_alloc + 1;
    HIPCHK(dh_dev_alloc(&ix.d_goff, sizeof(int64_t) * (size_t)(A->n + 1)));
    int2 *d_tiles = nullptr;
    uint32_t *d_sums = nullptr;
    const int64_t nsum = (nb + 1 + 2047) / 2048 + 1;
    HIPCHK(dh_dev_alloc(&d_tiles, sizeof(int2) * std::max<size_t>(tiles.size(), 1)));
    HIPCHK(dh_dev_alloc(&d_sums, sizeof(uint32_t) * (size_t)nsum));
    HIPCHK(hipMemcpyAsync(ix.d_goff, goff.data(), sizeof(int64_t) * goff.size(), hipMemcpyHostToDevice,
                          ctx->stream));
    // sequence of every page of the virtual axis: one load instead of a binary search over goff per candidate
    std::vector<int32_t> page_seq((size_t)(g >> 12) + 1, A->n > 0 ? A->n - 1 : 0);
    for (int32_t s2 = 0; s2 < A->n; s2++)
        for (int64_t pg = goff[(size_t)s2] >> 12; pg < (goff[(size_t)s2 + 1] >> 12); pg++) page_seq[(size_t)pg] = s2;
    HIPCHK(dh_dev_alloc(&ix.d_page_seq, sizeof(int32_t) * page_seq.size()));
    HIPCHK(hipMemcpyAsync(ix.d_page_seq, page_seq.data(), sizeof(int32_t) * page_seq.size(), hipMemcpyHostToDevice, ctx->stream));
    if (!tiles.empty())
        HIPCHK(hipMemcpyAsync(d_tiles, tiles.data(), sizeof(int2) * tiles.size(), hipMemcpyHostToDevice,
                              ctx->stream));
    HIPCHK(dhk_memset(ctx->stream, ix.d_dir_alloc, 0, sizeof(uint32_t) * (size_t)(nb + 2)));
    const DbView av = A->view();
    // grouped DB (pile-ups): a group's keys share their top bits, i.e. its buckets are one contiguous range; when the
    // sequences come group by group and a group is cut into few slices, the passes count in LDS (k_group_index)
    // instead of 2 x nk device-scope atomics on random counters (configs[2]: see LABNOTES 8)
    int32_t *d_gtile = nullptr;
    int32_t gi_slices = 0, gi_slice = 0;
    struct GtGuard {
        int32_t *&p;
        ~GtGuard() { dh_dev_free(p); }
    } gtg{d_gtile};
    // (a small grouped DB -- the templates of a consensus round: 500 sequences -- takes the generic passes: a block per
    // group and slice that zeroes and writes back 128 KB of LDS counters cost 8.6 ms per step at configs[2] for 1.3 M
    // k-mers; DH_INDEX_LDS_MIN overrides the threshold, tests run both paths)
    int64_t gi_min = 1 << 24;
    if (const char *e = getenv("DH_INDEX_LDS_MIN")) gi_min = atoll(e);
    if (A->d_group && A->ngroups > 1 && ix.shift <= 2 * k && nk >= gi_min && !getenv("DH_INDEX_ATOMICS")) {
        const int64_t nbg = 1ll << (2 * k - ix.shift);
        gi_slice = (int32_t)std::min<int64_t>(nbg, DH_GI_SLICE);
        gi_slices = (int32_t)(nbg / gi_slice);
        bool ordered = true;
        for (int32_t s2 = 1; s2 < A->n && ordered; s2++) ordered = A->h_group[(size_t)s2] >= A->h_group[(size_t)s2 - 1];
        if (!ordered || gi_slices > 16 || (int64_t)A->ngroups * gi_slices > (1ll << 30)) gi_slices = 0;
    }
    std::vector<int32_t> gtile;  // tiles of group g: [gtile[g], gtile[g + 1]); alive until the stream is synchronised below
    if (gi_slices > 0) {
        gtile.assign((size_t)A->ngroups + 1, 0);
        for (const int2 &t : tiles) gtile[(size_t)A->h_group[(size_t)t.x] + 1]++;
        for (int32_t g2 = 0; g2 < A->ngroups; g2++) gtile[(size_t)g2 + 1] += gtile[(size_t)g2];
        HIPCHK(dh_dev_alloc(&d_gtile, sizeof(int32_t) * gtile.size()));
        HIPCHK(hipMemcpyAsync(d_gtile, gtile.data(), sizeof(int32_t) * gtile.size(), hipMemcpyHostToDevice, ctx->stream));
        dhk_group_index(ctx->stream, 0, av, d_tiles, d_gtile, A->ngroups, gi_slices, gi_slice, k, kmer_mod, ix.shift,
                        ix.d_dir, ix.d_ent, ix.d_goff);
    } else
        dhk_kmer_pass(ctx->stream, 0, av, d_tiles, (int32_t)tiles.size(), k, kmer_mod, ix.shift, ix.d_dir, ix.d_ent,
                      ix.d_goff);
    // (the k-mers actually indexed are known only now -- modimer sampling is not even on repetitive sequence --: the scan
    // also sums them in 64 bits, a total that does not fit the 32-bit bucket offsets is an error, never a wrapped directory)
    unsigned long long *d_total = nullptr;
    struct TotGuard {
        unsigned long long *&p;
        ~TotGuard() { dh_dev_free(p); }
    } totg{d_total};
    HIPCHK(dh_dev_alloc(&d_total, sizeof(unsigned long long)));
    HIPCHK(hipMemsetAsync(d_total, 0, sizeof(unsigned long long), ctx->stream));
    dhk_scan_total(ctx->stream, ix.d_dir, nb + 1, d_sums, d_total);
    // the entry array is sized by the k-mers that were actually indexed (sampled, unmasked): the
    // exclusive scan leaves their number in dir[nb]
    uint32_t nent = 0;
    unsigned long long total = 0;
    HIPCHK(hipMemcpyAsync(&nent, ix.d_dir + nb, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(&total, d_total, sizeof(total), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (total != (unsigned long long)nent || total >= (1ull << 32) - 16)
        return fail(DH_EOVERFLOW, "index: more than 2^32 indexed k-mers (32-bit bucket offsets); raise kmer_mod");
    ix.n = (int64_t)nent;
    HIPCHK(dh_dev_alloc(&ix.d_ent, sizeof(ulonglong2) * (size_t)std::max<int64_t>(ix.n, 1)));
    if (gi_slices > 0)
        dhk_group_index(ctx->stream, 1, av, d_tiles, d_gtile, A->ngroups, gi_slices, gi_slice, k, kmer_mod, ix.shift,
                        ix.d_dir, ix.d_ent, ix.d_goff);
    else
        dhk_kmer_pass(ctx->stream, 1, av, d_tiles, (int32_t)tiles.size(), k, kmer_mod, ix.shift, ix.d_dir, ix.d_ent,
                      ix.d_goff);
    HIPCHK(hipGetLastError());
    // the directory the seed kernel reads: 16 bytes per bucket that hold the bucket's only entry itself, so that a
    // looked-up k-mer costs one random line unless its bucket holds several entries
    HIPCHK(dh_dev_alloc(&ix.d_fat, sizeof(ulonglong2) * (size_t)nb));
    dhk_fat_dir(ctx->stream, ix.d_dir, ix.d_ent, nb, ix.d_fat);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));  // tiles vector goes out of scope
    dh_dev_free(d_tiles);
    dh_dev_free(d_sums);
    dh_dev_free(ix.d_dir_alloc);
    ix.d_dir_alloc = ix.d_dir = nullptr;
    A->has_ix = true;
    return DH_OK;
}

// ------------------------------------------------------------------------------------ LA sets


extern "C" void dh_la_set_destroy(dh_la_set *s) { delete s; }
extern "C" int64_t dh_la_set_count(const dh_la_set *s) { return s ? (int64_t)s->la.size() : 0; }
dh_la_set::~dh_la_set()
{
    if (d_trace_own) {
        (void)hipSetDevice(device);
        dh_dev_free(d_trace_own);
    }
}
int dh_la_set_ensure_host_trace(dh_la_set *s)
{
    if (!s || !s->trace.empty() || s->d_trace_own_len <= 0) return DH_OK;
    HIPCHK(hipSetDevice(s->device));
    s->trace.resize((size_t)s->d_trace_own_len);
    HIPCHK(hipMemcpy(s->trace.data(), s->d_trace_own, sizeof(uint16_t) * (size_t)s->d_trace_own_len, hipMemcpyDeviceToHost));
    return DH_OK;
}
extern "C" int32_t dh_la_set_trace_on_device(const dh_la_set *s) { return s && s->d_trace_own_len > 0 && s->trace.empty() ? 1 : 0; }
extern "C" int64_t dh_la_set_trace_len(const dh_la_set *s)
{
    return s ? (s->trace.empty() && s->d_trace_own_len > 0 ? s->d_trace_own_len : (int64_t)s->trace.size()) : 0;
}
extern "C" const dh_la *dh_la_set_records(const dh_la_set *s) { return s ? s->la.data() : nullptr; }
extern "C" const uint16_t *dh_la_set_trace(const dh_la_set *s)
{
    if (!s) return nullptr;
    if (dh_la_set_ensure_host_trace(const_cast<dh_la_set *>(s)) != DH_OK) return nullptr;  // (left on the device: fetched now)
    return s->trace.data();
}
extern "C" int32_t dh_la_set_tspace(const dh_la_set *s) { return s ? s->tspace : 0; }

// LAsort order (a, b, comp, abpos, aepos, bbpos, bepos, diffs): base.d:1787-1809
static bool la_less(const dh_la &p, const dh_la &q)
{
    if (p.aread != q.aread) return p.aread < q.aread;
    if (p.bread != q.bread) return p.bread < q.bread;
    const uint32_t pc = p.flags & DH_FLAG_COMP, qc = q.flags & DH_FLAG_COMP;
    if (pc != qc) return pc < qc;
    if (p.abpos != q.abpos) return p.abpos < q.abpos;
    if (p.aepos != q.aepos) return p.aepos < q.aepos;
    if (p.bbpos != q.bbpos) return p.bbpos < q.bbpos;
    if (p.bepos != q.bepos) return p.bepos < q.bepos;
    return p.diffs < q.diffs;
}

// damapper-style chain flags per B read: every LA is a chain of its own (START); it is BEST
// unless a higher-scoring LA of the same read and orientation covers more than half of it on B
// (consumer: dazzler.d:1728-1758 reads START without BEST as alternateChain).
// `la` must be grouped by bread (the kernels emit it that way).
// damapper's chain flags (consumer: source/dentist/dazzler.d:1728-1758, 1991-1998).  Per read:
//   1. the local alignments on one contig and strand, ordered by their A interval, are linked into chains: an LA
//      continues the chain of its predecessor when it lies after it on both sequences (up to CHAIN_OVERLAP bases of
//      overlap), the gaps are at most CHAIN_GAP on either sequence and differ by at most CHAIN_INDEL (a read that
//      carries a long indel maps as two collinear LAs -- SURVEY section 7, K6);
//   2. score of a chain = sum of (A length - 2 * diffs) of its LAs;
//   3. a chain is the BEST one of its stretch of the read unless a higher-scoring chain of the same strand (ties:
//      the one whose first LA sorts later in LAsort order) covers more than half of its B span;
//   4. flags: START on the first LA of a chain, NEXT on the others, BEST on every LA of a best chain; START without
//      BEST reads as `alternateChain`.  near_best_ppm > 0 (damapper -n): an alternate chain scoring less than that
//      fraction of the chain that beats it is DISABLED (damapper does not report it).
#define CHAIN_GAP 10000
#define CHAIN_INDEL 6000
#define CHAIN_OVERLAP 100
// process-wide default (dh_set_near_best) and the per-context override (dh_ctx_set_near_best, -1 = use the default):
// a library user who never asked for -n is not affected by another context's setting
static std::atomic<int32_t> g_near_best_ppm{0};
extern "C" void dh_set_near_best(int32_t ppm) { g_near_best_ppm.store(ppm < 0 ? 0 : ppm); }
extern "C" int dh_ctx_set_near_best(dh_ctx *ctx, int32_t ppm)
{
    if (!ctx) return fail(DH_EINVAL, "dh_ctx_set_near_best: NULL context");
    ctx->near_best_ppm = ppm < 0 ? -1 : ppm;
    return DH_OK;
}

static void select_best_range(dh_la *la, size_t nla, int32_t near_ppm)
{
    // groups of equal bread are independent: host threads take runs of groups
    const std::vector<int64_t> gstart = dh_run_starts((int64_t)nla, [la](int64_t i) { return la[i].bread; });
    dh_parallel_for((int64_t)gstart.size() - 1, 2048, [&](int64_t glo, int64_t ghi) {
        struct Chain {  // members = ord[k0 .. k1): a chain only ever continues the chain before it (no vector per chain:
            int64_t score;  // half a million small allocations per chunk from 256 threads were most of the hook's 2.5 ms)
            int32_t bb, be, comp;
            size_t first;
            size_t k0, k1;
        };
        std::vector<size_t> ord;
        std::vector<Chain> chains;
        for (int64_t g = glo; g < ghi; g++) {
            const size_t g0 = (size_t)gstart[(size_t)g], g1 = (size_t)gstart[(size_t)g + 1];
            ord.clear();
            for (size_t x = g0; x < g1; x++) ord.push_back(x);
            std::sort(ord.begin(), ord.end(), [&](size_t x, size_t y) { return la_less(la[x], la[y]); });  // (a, b, comp, abpos, ...)
            chains.clear();
            for (size_t k = 0; k < ord.size(); k++) {
                const dh_la &q = la[ord[k]];
                bool linked = false;
                if (!chains.empty()) {
                    Chain &c = chains.back();
                    const dh_la &p = la[ord[c.k1 - 1]];
                    const int64_t ga = (int64_t)q.abpos - p.aepos, gb = (int64_t)q.bbpos - p.bepos;
                    linked = p.aread == q.aread && (p.flags & DH_FLAG_COMP) == (q.flags & DH_FLAG_COMP) && ga >= -CHAIN_OVERLAP &&
                             gb >= -CHAIN_OVERLAP && ga <= CHAIN_GAP && gb <= CHAIN_GAP && std::llabs(ga - gb) <= CHAIN_INDEL &&
                             q.aepos > p.aepos && q.bepos > p.bepos;
                    if (linked) {
                        c.k1 = k + 1;
                        c.score += (int64_t)(q.aepos - q.abpos) - 2 * (int64_t)q.diffs;
                        c.be = q.bepos;
                    }
                }
                if (!linked)
                    chains.push_back(Chain{(int64_t)(q.aepos - q.abpos) - 2 * (int64_t)q.diffs, q.bbpos, q.bepos,
                                           (int32_t)(q.flags & DH_FLAG_COMP), ord[k], k, k + 1});
            }
            for (size_t x = 0; x < chains.size(); x++) {
                const Chain &p = chains[x];
                bool best = true, drop = false;
                for (size_t y = 0; y < chains.size(); y++) {
                    if (x == y) continue;
                    const Chain &q = chains[y];
                    // ties: the chain whose first LA sorts later (LAsort order) wins
                    if (q.score < p.score || (q.score == p.score && la_less(la[q.first], la[p.first]))) continue;
                    if (q.comp != p.comp) continue;
                    const int32_t lo = std::max(p.bb, q.bb), hi = std::min(p.be, q.be);
                    if (hi - lo > (p.be - p.bb) / 2) {
                        best = false;
                        if (near_ppm > 0 && p.score * 1000000ll < (int64_t)near_ppm * q.score) drop = true;
                    }
                }
                for (size_t m = 0; m < p.k1 - p.k0; m++) {
                    dh_la &l = la[ord[p.k0 + m]];
                    l.flags &= ~(DH_FLAG_START | DH_FLAG_NEXT | DH_FLAG_BEST);
                    l.flags |= (m == 0 ? DH_FLAG_START : DH_FLAG_NEXT) | (best ? DH_FLAG_BEST : 0u) | (drop ? DH_FLAG_DISABLED : 0u);
                }
            }
            // the records of the read in LAsort order: a chain's members are then neighbours (a chain only ever continues the
            // chain before it), START followed by its NEXT records -- how damapper writes them and how every consumer
            // rebuilds the chains (dazzler.d:1728-1758); the trace values stay where they are (toff)
            if (!std::is_sorted(ord.begin(), ord.end())) {
                std::vector<dh_la> tmp(ord.size());
                for (size_t k = 0; k < ord.size(); k++) tmp[k] = la[ord[k]];
                for (size_t k = 0; k < ord.size(); k++) la[g0 + k] = tmp[k];
            }
        }
    });
}

// LAsort order of a B-major (bread, strand, ...) list in O(n): stable counting sort by aread
// keeps (bread, comp) ascending inside every aread; the rare runs with equal (aread, bread, comp)
// are finished with an insertion sort.
static void lasort(dh_la_set *res, int32_t na)
{
    const size_t n = res->la.size();
    LaVec out(n);
    const int64_t chunk = 8192, nchunks = ((int64_t)n + chunk - 1) / chunk;
    if (na <= 4096 && nchunks > 1) {
        // stable counting sort by aread with one histogram per input chunk (threads scatter)
        std::vector<int64_t> hist((size_t)nchunks * ((size_t)na + 1), 0);
        dh_parallel_for(nchunks, 1, [&](int64_t clo, int64_t chi) {
            for (int64_t c = clo; c < chi; c++) {
                int64_t *h = hist.data() + (size_t)c * ((size_t)na + 1);
                const size_t e = std::min(n, (size_t)(c + 1) * (size_t)chunk);
                for (size_t i = (size_t)c * (size_t)chunk; i < e; i++) h[(size_t)res->la[i].aread]++;
            }
        });
        int64_t run = 0;
        for (int32_t a = 0; a <= na; a++)
            for (int64_t c = 0; c < nchunks; c++) {
                int64_t &h = hist[(size_t)c * ((size_t)na + 1) + (size_t)a];
                const int64_t cnt = h;
                h = run;
                run += cnt;
            }
        dh_parallel_for(nchunks, 1, [&](int64_t clo, int64_t chi) {
            for (int64_t c = clo; c < chi; c++) {
                int64_t *h = hist.data() + (size_t)c * ((size_t)na + 1);
                const size_t e = std::min(n, (size_t)(c + 1) * (size_t)chunk);
                for (size_t i = (size_t)c * (size_t)chunk; i < e; i++) {
                    const dh_la &l = res->la[i];
                    out[(size_t)h[(size_t)l.aread]++] = l;
                }
            }
        });
    } else {
        std::vector<int64_t> first((size_t)na + 2, 0);
        for (const dh_la &l : res->la) first[(size_t)l.aread + 1]++;
        for (int32_t a = 0; a <= na; a++) first[(size_t)a + 1] += first[(size_t)a];
        for (const dh_la &l : res->la) out[(size_t)first[(size_t)l.aread]++] = l;
    }
    for (size_t i = 1; i < n; i++) {
        if (!la_less(out[i], out[i - 1])) continue;
        dh_la x = out[i];
        size_t j = i;
        while (j > 0 && la_less(x, out[j - 1])) {
            out[j] = out[j - 1];
            j--;
        }
        out[j] = x;
    }
    // the traces stay where the device compaction put them: every record's toff still points at
    // its (diffs, bbases) pairs, only the records are permuted (saves re-laying out tens of MB)
    res->la.swap(out);
}

// ------------------------------------------------------------------------------------ align


extern "C" int dh_align_db(dh_ctx *ctx, dh_db *A, dh_db *B, const dh_align_opts *opts,
                           int32_t want_best, dh_la_set **out)
{
    return dh_align_db_ex(ctx, A, B, opts, want_best, 1, out);
}

// per-chunk hook: called on a host thread of its own with the records of a finished chunk (B-major,
// whole reads) while the device works on the next chunk; the records may be modified in place
// (records of the chunk, their number, their offset in the result, number of the chunk)
typedef std::function<void(dh_la *, int64_t, int64_t, int64_t)> ChunkHook;
static int align_range(dh_ctx *ctx, dh_db *A, dh_db *B, int32_t first, int32_t count, const dh_align_opts *opts,
                       int32_t want_best, int32_t want_sorted, dh_la_set **out, const ChunkHook *hook = nullptr,
                       dh_la_set **out_tr = nullptr);

// `damapper <ref> <reads>.<block>` (snakemake/Snakefile:1143-1170): the reads [first, first + count)
// of B against all of A; read ids in the records are those of the whole DB, as in a block's .las
extern "C" int dh_align_db_block(dh_ctx *ctx, dh_db *A, dh_db *B, int32_t first, int32_t count,
                                 const dh_align_opts *opts, int32_t want_best, dh_la_set **out)
{
    if (!B || first < 0 || count < 0 || (int64_t)first + count > B->n)
        return fail(DH_EINVAL, "dh_align_db_block: block outside the DB");
    return align_range(ctx, A, B, first, count, opts, want_best, 1, out);
}

// The mapping pass with the alignment filters of `dentist collect` applied on the way
// (damapper per read block, Snakefile:1143-1170, + collectPileUps/filter.d:122-356): all six filters
// decide per read, so the records of a finished chunk of reads are filtered on a host thread while the
// device maps the next chunk.  Same records and flags as dh_align_db_block(want_best = 1) followed by
// dh_collect_filter.  rep_ptr / rep_iv: repeat mask of the contigs for WeaklyAnchored (may be NULL).
extern "C" int dh_map_reads(dh_ctx *ctx, dh_db *contigs, dh_db *reads, int32_t first, int32_t count,
                            const dh_align_opts *opts, const dh_process_opts *popts, const int64_t *rep_ptr,
                            const int32_t *rep_iv, int32_t want_sorted, int64_t *dropped6, dh_la_set **out,
                            dh_pileups **cands)
{
    if (!contigs || !reads || !popts || first < 0 || count < 0 || (int64_t)first + count > reads->n)
        return fail(DH_EINVAL, "dh_map_reads: bad argument");
    if (cands && (want_sorted & 1))
        return fail(DH_EINVAL, "dh_map_reads: candidates index the records in mapping order (want_sorted bit 0 clear)");
    want_sorted &= 1 | 8;  // (bit 0: LAsort order; bit 3: the trace values stay on the device, dh_la_set_trace fetches them on demand)
    if (cands) *cands = nullptr;
    std::mutex mu;
    int64_t dropped[6] = {0, 0, 0, 0, 0, 0};
    int hook_rc = DH_OK;
    std::string hook_msg;  // dh_last_error() is per thread: the hook thread's message travels with its code
    std::vector<dh_pileups *> per_chunk;  // spanning-read candidates of every chunk, LA indices of the result
    struct CandGuard {
        std::vector<dh_pileups *> &v;
        ~CandGuard()
        {
            for (dh_pileups *p : v) dh_pileups_destroy(p);
        }
    } cguard{per_chunk};
    const ChunkHook hook = [&](dh_la *las, int64_t n, int64_t l0, int64_t chunk_no) {
        int64_t d[6] = {0, 0, 0, 0, 0, 0};
        int rc = dh_collect_filter(las, n, contigs->h_off.data(), contigs->n, reads->h_off.data(), reads->n, rep_ptr,
                                   rep_iv, popts, d, nullptr);
        dh_pileups *pc = nullptr;
        if (rc == DH_OK && cands) rc = dh_collect_candidates(las, n, contigs->h_off.data(), contigs->n, popts, &pc);
        if (pc && l0 != 0) dh_pileups_shift(pc, (int32_t)l0);
        std::lock_guard<std::mutex> lk(mu);
        if (rc != DH_OK && hook_rc == DH_OK) {
            hook_rc = rc;
            hook_msg = dh_last_error();
        }
        for (int k = 0; k < 6; k++) dropped[k] += d[k];
        if ((size_t)chunk_no >= per_chunk.size()) per_chunk.resize((size_t)chunk_no + 1, nullptr);
        per_chunk[(size_t)chunk_no] = pc;
    };
    const int rc = align_range(ctx, contigs, reads, first, count, opts, 1, want_sorted, out, &hook);
    if (rc != DH_OK) return rc;
    if (hook_rc != DH_OK) {
        dh_la_set_destroy(*out);
        *out = nullptr;
        return fail(hook_rc, hook_msg.empty() ? "dh_map_reads: a chunk's filters failed" : hook_msg);
    }
    if (cands) {  // chunks hold ascending read ranges: concatenating per gap keeps every gap ordered by read
        if (int rc2 = dh_pileups_concat(per_chunk.data(), (int32_t)per_chunk.size(), cands)) {
            dh_la_set_destroy(*out);
            *out = nullptr;
            return rc2;
        }
    }
    if (dropped6) memcpy(dropped6, dropped, sizeof(dropped));
    return DH_OK;
}

int dh_align_db_ex(dh_ctx *ctx, dh_db *A, dh_db *B, const dh_align_opts *opts, int32_t want_best,
                   int32_t want_sorted, dh_la_set **out)
{
    if (!B) return fail(DH_EINVAL, "dh_align_db: NULL argument");
    return align_range(ctx, A, B, 0, B->n, opts, want_best, want_sorted, out);
}

// derived copies (reverse complement, 2-bit packed forward / reverse) of the reads [r0, r1) of B in
// the context's scratch arena; the returned pointers are shifted so that absolute base offsets of
// the DB index them, exactly like the DB-owned whole copies
struct ChunkCopies {
    const uint8_t *rc = nullptr, *pk = nullptr, *rcpk = nullptr;
    bool has_n = false;
    // the packed words of the chunk themselves (unshifted) -- k_tile turns them into plane words in place
    uint8_t *pk_w0 = nullptr, *rcpk_w0 = nullptr;
    int64_t pk_words = 0;
    bool planes = false;  // the copies are plane-packed already (made so straight from the bytes)
};
// planes: plane-packed copies for k_tile instead of the 2-bit packed ones (a DH-2 mapping that does not keep the packed
// words for the transposed pairs): the conversion passes over both copies -- 8 of the 24 GB a chunk of configs[2] moves
// for its copies -- fall away
static int chunk_copies(dh_ctx *ctx, dh_db *B, int32_t r0, int32_t r1, bool want_packed, bool need_bytes,
                        ChunkCopies *out, bool planes = false)
{
    hipStream_t st = ctx->stream;
    const int64_t o0 = B->h_off[(size_t)r0], o1 = B->h_off[(size_t)r1];
    const int64_t a0 = o0 & ~31ll;  // packed words hold 32 bases: start the chunk on a word boundary
    uint8_t *d_rc, *d_pk, *d_rcpk;
    int32_t *d_flag;
    out->has_n = false;
    auto rc_bytes = [&]() -> int {
        // reverse complement as bytes (only the wave kernels' byte path reads it): every read mirrored
        // inside its own [off, off + len) range
        if (int rc = dh_scratch(ctx, 26, (size_t)(o1 - a0) + 2 * DB_PAD, (void **)&d_rc)) return rc;
        HIPCHK(dhk_memset(st, d_rc, 4, (size_t)(o1 - a0) + 2 * DB_PAD));
        uint8_t *rc_shift = d_rc + DB_PAD - a0;
        dhk_revcomp(st, B->d_bases, rc_shift, B->d_off + r0, r1 - r0, B->max_len);
        HIPCHK(hipGetLastError());
        out->rc = rc_shift;
        return DH_OK;
    };
    if (!want_packed) return rc_bytes();
    // 2-bit packed forward copy and, straight from the forward bytes, the packed reverse complements
    const size_t pbytes = (size_t)((o1 - a0 + 31) / 32) * 8 + 2 * PK_PAD;
    if (int rc = dh_scratch(ctx, 27, pbytes, (void **)&d_pk)) return rc;
    if (int rc = dh_scratch(ctx, 28, pbytes, (void **)&d_rcpk)) return rc;
    if (int rc = dh_scratch(ctx, 3, 4 * sizeof(int32_t), (void **)&d_flag)) return rc;
    HIPCHK(hipMemsetAsync(d_flag + 1, 0, 2 * sizeof(int32_t), st));
    // k_pack2_rc stores the words inside a read whole and ORs into the words reads share: only those (and the padding on
    // both sides) are zeroed -- the memset of the whole buffer was 2 GB per chunk of the mapping
    HIPCHK(hipMemsetAsync(d_rcpk, 0, PK_PAD + 8, st));
    HIPCHK(hipMemsetAsync(d_rcpk + pbytes - PK_PAD - 8, 0, PK_PAD + 8, st));
    if (planes) {
        dhk_pack2_planes(st, B->d_bases + a0, o1 - a0, d_pk + PK_PAD, d_flag + 1);
        // (the reverse-complement planes from the forward planes: the chunk's bytes are read once, not twice --
        // DH_RC_FROM_BYTES=1 keeps the pass over the bytes, tests compare)
        if (getenv("DH_RC_FROM_BYTES"))
            dhk_pack2_rc_planes(st, B->d_bases, B->d_off + r0, r1 - r0, B->max_len, a0, d_rcpk + PK_PAD);
        else
            dhk_planes_rc(st, d_pk + PK_PAD, B->d_off + r0, r1 - r0, B->max_len, a0, d_rcpk + PK_PAD);
    } else {
        dhk_pack2_rc_bounds(st, B->d_off + r0, r1 - r0, a0, d_rcpk + PK_PAD);
        dhk_pack2(st, B->d_bases + a0, o1 - a0, d_pk + PK_PAD, d_flag + 1);
        dhk_pack2_rc(st, B->d_bases, B->d_off + r0, r1 - r0, B->max_len, a0, d_rcpk + PK_PAD);
    }
    out->planes = planes;
    HIPCHK(hipGetLastError());
    int32_t flag = 0;
    HIPCHK(hipMemcpyAsync(&flag, d_flag + 1, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    out->has_n = flag != 0;
    out->pk = d_pk + PK_PAD - (a0 >> 2);
    out->rcpk = d_rcpk + PK_PAD - (a0 >> 2);
    out->pk_words = (o1 - a0 + 31) / 32;
    out->pk_w0 = d_pk + PK_PAD;
    out->rcpk_w0 = d_rcpk + PK_PAD;
    // codes outside 0..3 (here or in A): the wave kernels slide over the byte arrays
    if (out->has_n || need_bytes) return rc_bytes();
    return DH_OK;
}

static int align_range(dh_ctx *ctx, dh_db *A, dh_db *B, int32_t first, int32_t count, const dh_align_opts *opts,
                       int32_t want_best, int32_t want_sorted, dh_la_set **out, const ChunkHook *hook, dh_la_set **out_tr)
{
    auto now_ms = [] {
        return (double)std::chrono::duration_cast<std::chrono::microseconds>(
                   std::chrono::steady_clock::now().time_since_epoch()).count() / 1e3;
    };
    const double wall0 = now_ms() * 1e3;
    double w_a = now_ms(), w_index = 0, w_loop = 0, w_post = 0;
    if (!ctx || !A || !B || !opts || !out) return fail(DH_EINVAL, "dh_align_db: NULL argument");
    if (A->ctx != ctx || B->ctx != ctx) return fail(DH_EINVAL, "dh_align_db: DB of another context");
    const dh_align_opts &o = *opts;
    const int32_t near_ppm = ctx->near_best_ppm >= 0 ? ctx->near_best_ppm : g_near_best_ppm.load();
    if (o.k < 8 || o.k > 28) return fail(DH_EINVAL, "k must be in [8, 28]");
    if (o.algo != 0 && o.algo != 1) return fail(DH_EINVAL, "algo must be 0 (DH-1, wave) or 1 (DH-2, tiled band)");
    const bool tiled = o.algo == 1;
    if (tiled) {
        if (o.width != 64 && o.width != 32) return fail(DH_EINVAL, "algo 1 (DH-2): width is the band, it must be 64 or 32");
        if (o.tspace > dhtile::TS_MAX) return fail(DH_EINVAL, "algo 1 (DH-2): tspace must be <= 128");
    } else if (o.width < 1 || o.width > 62)
        return fail(DH_EINVAL, "width must be in [1, 62]");
    if (o.tspace < 16 || o.tspace > 32767) return fail(DH_EINVAL, "tspace out of range");
    if (o.max_cand < 1 || o.max_cand > 256) return fail(DH_EINVAL, "max_cand must be in [1, 256]");
    if (o.max_la < 1 || o.max_la > 256) return fail(DH_EINVAL, "max_la must be in [1, 256]");
    if (o.pen < 2) return fail(DH_EINVAL, "pen must be >= 2");
    if (o.band_shift < 1 || o.band_shift > 12) return fail(DH_EINVAL, "band_shift out of range");
    if (o.skip_self && A != B) return fail(DH_EINVAL, "skip_self needs A == B");
    if (out_tr && (o.algo != 1 || A == B || hook))
        return fail(DH_EINVAL, "the transposed file is defined for DH-2 (algo 1) mappings of one DB onto another");
    if (o.skip_self < 0 || o.skip_self > 3) return fail(DH_EINVAL, "skip_self must be 0, 1, 2 or 3");
    if (o.skip_self == 3 && (o.algo != 1 || o.strands != 1 || hook || out_tr))
        return fail(DH_EINVAL, "skip_self 3 (a read against itself, datander) is defined for DH-2 (algo 1) on the forward strand (strands 1)");
    if (o.kmer_mod < 1 || o.kmer_mod > 64) return fail(DH_EINVAL, "kmer_mod must be in [1, 64]");
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    dh_align_stats stats = {};
    stats.b_bases = B->h_off[(size_t)first + (size_t)count] - B->h_off[(size_t)first];
    dh_la_set *res = new dh_la_set();
    res->tspace = o.tspace;
    *out = nullptr;
    struct Guard {
        dh_la_set *&r;
        bool ok = false;
        ~Guard()
        {
            if (!ok) delete r;
        }
    } guard{res};
    // the transposed file of a mapping (`damapper -C`): records (read, contig) of the transposed pairs
    dh_la_set *res2 = out_tr ? new dh_la_set() : nullptr;
    struct Guard2 {
        dh_la_set *p;
        bool ok = false;
        ~Guard2()
        {
            if (!ok) delete p;
        }
    } guard2{res2};
    if (res2) res2->tspace = o.tspace;
    if (out_tr) *out_tr = nullptr;
    // hook tasks in flight; joined before the result can move or is handed out (also on error paths)
    struct Tasks {
        hipStream_t cs;
        std::vector<std::thread> v;
        double ms_hooks = 0, ms_copies = 0;  // of the last join: waiting for the hook threads, then for the copy stream
        void join()
        {
            const auto t0 = std::chrono::steady_clock::now();
            for (auto &t : v)
                if (t.joinable()) t.join();
            v.clear();
            const auto t1 = std::chrono::steady_clock::now();
            (void)hipStreamSynchronize(cs);  // copies in flight have landed
            const auto t2 = std::chrono::steady_clock::now();
            ms_hooks = std::chrono::duration<double, std::milli>(t1 - t0).count();
            ms_copies = std::chrono::duration<double, std::milli>(t2 - t1).count();
        }
        ~Tasks() { join(); }
    } tasks{ctx->cstream, {}};
    int64_t nchunk_done = 0;

    HIPCHK(hipEventRecord(ctx->ev[0], st));
    // A sequences start at multiples of 4096 on the virtual axis and sepv is one too, so the
    // position of a hit inside its diagonal band (2^band_shift <= 4096 wide) depends only on the
    // pair (A sequence, B read) -- never on which other sequences share the DB or the launch
    const int32_t sepv = (B->max_len + 64 + 4095) & ~4095;
    // ---- a grouped DB against itself (the pile-up all-vs-all): the seeds come from the per-pile-up k-mer join
    // (dh_join.hip) -- no k-mer directory is built, no line of HBM is looked up at random; bit-identical hits.
    // Plan: slices per group (about JOIN_FILL entries each), part blocks (JP_THREADS chunks of one group each), the
    // rows of the two tables.  DH_NO_JOIN=1 forces the directory path (tests compare the two).
    struct JoinPlan {
        std::vector<int32_t> gfirst, gns, pfirst;
        std::vector<int2> pblk, jblk;
        std::vector<int64_t> psubrow, segrow;
        int64_t npsub = 0, nseg = 0;
    } jp;
    bool use_join = A == B && A->d_group && A->ngroups >= 1 && o.k <= 16 && B->max_len < JOIN_MAX_LEN && first == 0 &&
                    count == B->n && B->n > 0 && !getenv("DH_NO_JOIN");
    if (use_join) {
        const int32_t ng = A->ngroups;
        jp.gfirst.assign((size_t)ng + 1, 0);
        for (int32_t s2 = 0; s2 < A->n && use_join; s2++) {
            if (s2 > 0 && A->h_group[(size_t)s2] < A->h_group[(size_t)s2 - 1]) use_join = false;  // groups must be contiguous
            jp.gfirst[(size_t)A->h_group[(size_t)s2] + 1]++;
        }
        for (int32_t g2 = 0; g2 < ng; g2++) jp.gfirst[(size_t)g2 + 1] += jp.gfirst[(size_t)g2];
        jp.gns.assign((size_t)ng, 1);
        jp.pfirst.assign((size_t)ng + 1, 0);
        jp.segrow.assign((size_t)A->n, 0);
        for (int32_t g2 = 0; g2 < ng && use_join; g2++) {
            const int32_t r0 = jp.gfirst[(size_t)g2], r1 = jp.gfirst[(size_t)g2 + 1];
            if (r1 - r0 > JOIN_MAX_READS) use_join = false;
            int64_t nkm = 0, nch = 0;
            for (int32_t r = r0; r < r1; r++) {
                const int64_t np_ = A->h_off[(size_t)r + 1] - A->h_off[(size_t)r] - o.k + 1;
                if (np_ > 0) {
                    nkm += np_;
                    nch += (np_ + JP_PER - 1) / JP_PER;
                }
            }
            const int64_t ns = std::max<int64_t>(1, (nkm / std::max(1, o.kmer_mod) + JOIN_FILL - 1) / JOIN_FILL);
            if (ns > JOIN_MAX_SLICES) use_join = false;
            jp.gns[(size_t)g2] = (int32_t)ns;
            for (int64_t c0 = 0; c0 < nch; c0 += JP_THREADS) {
                jp.pblk.push_back(int2{g2, (int32_t)c0});
                jp.psubrow.push_back(jp.npsub);
                jp.npsub += ns;
            }
            jp.pfirst[(size_t)g2 + 1] = (int32_t)jp.pblk.size();
            if (nch > 0)
                for (int32_t s2 = 0; s2 < (int32_t)ns; s2++) jp.jblk.push_back(int2{g2, s2});
            for (int32_t r = r0; r < r1; r++) {
                jp.segrow[(size_t)r] = jp.nseg;
                jp.nseg += ns;
            }
        }
        if (jp.pblk.size() > (size_t)INT32_MAX / 2 || jp.jblk.size() > (size_t)INT32_MAX / 2) use_join = false;
    }
    if (int rc = build_index(A, o.k, sepv, o.kmer_mod, use_join)) return rc;
    // B's derived copies (reverse complement, 2-bit packed) live with the DB when the whole DB is
    // one chunk of this call (pile-up and template DBs are re-aligned several times); a block of a
    // larger DB gets them chunk by chunk in the scratch arena, so the resident footprint of a reads
    // DB stays at one byte per base however large it is
    const int64_t nitems_total = 2ll * count;
    // items per launch.  k_tile runs one alignment per lane: a launch needs several alignments per
    // resident lane (262 144 of them) to keep the wavefronts full until the queue drains -- measured on
    // configs[2]: 2^18 items per launch 70 ms of k_tile per step, 2^20 37 ms (the host filters of a chunk
    // still overlap the next chunk's kernels)
    int32_t chunk = o.algo == 1 ? 1 << 20 : 1 << 18;
    if (const char *e = getenv("DH_ALIGN_CHUNK")) chunk = std::max(2, atoi(e)) & ~1;
    // symmetric mode writes records into the slots of other items: everything is one chunk
    if (o.skip_self == 2) {
        if (first != 0 || count != B->n) return fail(DH_EINVAL, "symmetric mode needs the whole DB");
        chunk = (int32_t)std::min<int64_t>(std::max<int64_t>(nitems_total, 2), INT32_MAX - 1);
    }
    // (DH-2 reads B from plane-packed copies made chunk by chunk in the scratch arena)
    const bool db_copies = !tiled && (A == B || (first == 0 && count == B->n && nitems_total <= chunk));
    const bool want_packed = !getenv("DH_WAVE_BYTES");
    // the wave kernel slides over 2-bit packed copies unless a DB holds codes outside 0..3
    if (int rc = dh_ensure_packed(A, false)) return rc;
    if (db_copies) {
        if (int rc = dh_ensure_rc(B)) return rc;
        if (int rc = dh_ensure_packed(B, true)) return rc;
    }
    // up to 30 live diagonals fit a 32-lane half: two alignments per wavefront (k_wave2); its
    // reverse extensions run forward over the reverse complements, so A needs one as well
    const bool dual = tiled || (o.width <= 30 && !getenv("DH_WAVE_SINGLE"));
    if (dual) {
        if (int rc = dh_ensure_rc(A)) return rc;
        if (A->has_n == 0)
            if (int rc = dh_ensure_packed(A, true)) return rc;
    }
    if (tiled && (A->has_n != 0 || !A->d_pk || !A->d_rcpk))
        return fail(DH_EINVAL, "algo 1 (DH-2) needs sequences of a, c, g, t only (2-bit copies), A holds other codes");
    HIPCHK(hipEventRecord(ctx->ev[1], st));
    w_index = now_ms() - w_a;
    w_a = now_ms();

    DhOpts dopt;
    memcpy(&dopt, &o, sizeof(dopt));
    auto index_view = [&]() {
        return IndexView{A->ix.d_fat, A->ix.d_ent, A->ix.d_goff, A->ix.d_page_seq, A->ix.n,
                         A->ix.na,    A->ix.sepv,   A->ix.shift,  A->ix.pbits};
    };
    IndexView iv = index_view();
    const DbView av = A->view(), bv = B->view();

    // capacity planning
    const int64_t maxext =
        std::min<int64_t>(A->max_len, (int64_t)B->max_len + (2ll * B->max_len + o.xdrop) / (o.pen - 1) + 1);
    const int32_t nbmax = (int32_t)(maxext / o.tspace + 3);
    const int32_t trmax = 2 * (2 * nbmax + 2);
    // resident alignment slots: one per wavefront of k_wave (<= 64 VGPRs -> 8 waves/SIMD), one per
    // 32-lane half of k_wave2 (two per wavefront, 6 waves/SIMD)
    // alignments per wavefront of k_wave2: 2 (32 lanes each, width <= 30) or 4 (16 lanes, width <= 14)
    const int32_t per_wave = (o.width <= 14 && !getenv("DH_WAVE_G32")) ? 4 : 2;
    int32_t slots_per_cu = 32;
    // k_wave2: 80 VGPRs -> 6 waves/SIMD = 24 wavefronts per CU (two per wavefront); 96 VGPRs -> 5 waves/SIMD = 20 (four)
    if (o.width <= 30) slots_per_cu = per_wave == 4 ? 20 * 4 : 24 * 2;
    if (const char *e = getenv("DH_WAVE_SLOTS_PER_CU")) slots_per_cu = std::max(4, atoi(e)) & ~3;
    // trace-node pool of one alignment slot.  k_wave2: every lane of the group owns a stretch (a lane
    // crosses each boundary of either grid at most once per diagonal it serves; twice that is the
    // capacity, an overflow is reported); k_wave: one shared pool
    const int32_t poolcap = dual ? (64 / per_wave) * (4 * nbmax + 8) : 96 * nbmax;
    const int32_t nslots = tiled ? 4 : (int32_t)std::min<int64_t>((int64_t)ctx->ncu * slots_per_cu,
                                                      (std::max<int64_t>(nitems_total, 4) + 3) & ~3ll);
    // DH-2: one alignment per lane; wavefronts resident = CUs x waves per CU, no more than the items need
    int32_t tile_waves = 0;
    if (tiled) {
        // symmetric launches spend most of a wavefront's time waiting on the records and scratch of short alignments:
        // all 16 wavefronts the registers allow (configs[2]: pile-up launch -2.5 ms against 12; mapping +1 ms with 16)
        int32_t per_cu = o.skip_self == 2 ? 16 : dhk_tile_waves_per_cu();
        if (const char *e = getenv("DH_TILE_WAVES_PER_CU")) per_cu = std::max(1, atoi(e));
        if (o.skip_self == 2)
            if (const char *e = getenv("DH_TILE_SYM_WAVES_PER_CU")) per_cu = std::max(1, atoi(e));  // development
        // (symmetric mode: the work units are groups of candidates, many per item -- a pile-up read meets every other
        // read of its pile-up -- so the items do not bound the lanes that find work)
        const int64_t lanes_wanted = o.skip_self == 2 ? nitems_total * (int64_t)o.max_cand : nitems_total;
        tile_waves = (int32_t)std::min<int64_t>((int64_t)ctx->ncu * per_cu, (std::max<int64_t>(lanes_wanted, 1) + 63) / 64);
    }
    const int32_t cn = (int32_t)std::min<int64_t>(chunk, std::max<int64_t>(nitems_total, 2));
    DhCand *d_cand;
    int32_t *d_ncand, *d_nhits, *d_status, *d_cdj;
    uint32_t *d_nla, *d_ntr, *d_queue, *d_sums;
    DhNode *d_pool;
    DhLa *d_la, *d_laout;
    uint16_t *d_trslots, *d_trout;
    unsigned long long *d_counters;
#define SCR(id, ptr, count)                                                                      \
    if (int rc_ = dh_scratch(ctx, id, sizeof(*ptr) * std::max<size_t>((size_t)(count), 1), (void **)&ptr)) return rc_;
    SCR(0, d_cand, (size_t)cn * o.max_cand)
    SCR(1, d_ncand, cn)
    SCR(2, d_nhits, cn)
    SCR(3, d_status, 4)
    SCR(4, d_nla, cn + 1)
    SCR(5, d_ntr, cn + 1)
    SCR(6, d_pool, (size_t)nslots * poolcap)
    SCR(7, d_cdj, (size_t)nslots * 8 * nbmax)
    SCR(8, d_queue, 4)
    // (symmetric DH-2 launches keep their records in candidate-indexed slots, sized once the candidates are counted)
    const bool sym_tiled = tiled && o.skip_self == 2;
    // want_sorted & 8 (dh_map_reads): the trace values of every chunk stay on the device in a buffer the result owns
    const bool keep_dev = (want_sorted & 8) && hook && tiled && !out_tr && !sym_tiled;
    if (!sym_tiled) {
        SCR(9, d_la, (size_t)cn * o.max_la)
        SCR(10, d_trslots, (size_t)cn * o.max_la * trmax)
    } else
        d_la = nullptr, d_trslots = nullptr;
    SCR(11, d_counters, 2)
    SCR(12, d_sums, (size_t)cn / 2048 + 4)
    unsigned long long *d_summary;
    SCR(31, d_summary, 4)
    int32_t *d_ovf;
    SCR(29, d_ovf, cn)
    int32_t *d_regs = nullptr;
    if (tiled) SCR(32, d_regs, (size_t)tile_waves * 64 * dhtile::MAXREG * dhtile::REGF)
    dhtile::Cold *d_cold = nullptr;
    if (tiled) SCR(34, d_cold, (size_t)tile_waves * 64)
    DhLa *d_la2 = nullptr, *d_laout2 = nullptr;
    uint16_t *d_trslots2 = nullptr, *d_trout2 = nullptr;
    uint32_t *d_nla2 = nullptr, *d_ntr2 = nullptr;
    uint8_t *d_app = nullptr, *d_arcpp = nullptr;  // plane-packed copies of A (B'' of the transposed pairs)
    if (res2) {
        SCR(35, d_la2, (size_t)cn * o.max_la)
        SCR(36, d_trslots2, (size_t)cn * o.max_la * trmax)
        SCR(37, d_nla2, cn + 1)
        SCR(38, d_ntr2, cn + 1)
        const size_t awords = (size_t)((A->total + 31) / 32), abytes = awords * 8 + 2 * PK_PAD;
        SCR(39, d_app, abytes)
        SCR(40, d_arcpp, abytes)
        HIPCHK(hipMemcpyAsync(d_app, A->d_pk_alloc, abytes, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemcpyAsync(d_arcpp, A->d_rcpk_alloc, abytes, hipMemcpyDeviceToDevice, st));
        dhk_pk2planes(st, d_app + PK_PAD, (int64_t)awords);
        dhk_pk2planes(st, d_arcpp + PK_PAD, (int64_t)awords);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipMemsetAsync(d_status, 0, sizeof(int32_t), st));
    HIPCHK(hipMemsetAsync(d_counters, 0, 2 * sizeof(unsigned long long), st));

    JoinView jv = {};
    int64_t join_hits = 0;
    float ms_join = 0;
    unsigned int jhist[4] = {0, 0, 0, 0};  // reads with more than 2048 / 4096 / 8192 hits, the largest count
    if (use_join) {
        // one upload of the plan tables; device buffers from the scratch arena
        const size_t ng = (size_t)A->ngroups;
        size_t blob_bytes = 0;
        auto place = [&](size_t bytes) {
            const size_t at = blob_bytes;
            blob_bytes += (bytes + 15) & ~(size_t)15;
            return at;
        };
        const size_t o_gfirst = place(sizeof(int32_t) * (ng + 1)), o_gns = place(sizeof(int32_t) * ng),
                     o_pfirst = place(sizeof(int32_t) * (ng + 1)), o_pblk = place(sizeof(int2) * jp.pblk.size()),
                     o_psubrow = place(sizeof(int64_t) * jp.psubrow.size()), o_jblk = place(sizeof(int2) * jp.jblk.size()),
                     o_segrow = place(sizeof(int64_t) * jp.segrow.size());
        std::vector<uint8_t, PinnedAlloc<uint8_t>> blob(blob_bytes);
        memcpy(blob.data() + o_gfirst, jp.gfirst.data(), sizeof(int32_t) * (ng + 1));
        memcpy(blob.data() + o_gns, jp.gns.data(), sizeof(int32_t) * ng);
        memcpy(blob.data() + o_pfirst, jp.pfirst.data(), sizeof(int32_t) * (ng + 1));
        if (!jp.pblk.empty()) memcpy(blob.data() + o_pblk, jp.pblk.data(), sizeof(int2) * jp.pblk.size());
        if (!jp.psubrow.empty()) memcpy(blob.data() + o_psubrow, jp.psubrow.data(), sizeof(int64_t) * jp.psubrow.size());
        if (!jp.jblk.empty()) memcpy(blob.data() + o_jblk, jp.jblk.data(), sizeof(int2) * jp.jblk.size());
        memcpy(blob.data() + o_segrow, jp.segrow.data(), sizeof(int64_t) * jp.segrow.size());
        uint8_t *d_blob;
        uint32_t *d_psub;
        uint64_t *d_entries, *d_segtab, *d_hits;
        unsigned long long *d_cursor;
        SCR(45, d_blob, blob_bytes)
        SCR(46, d_psub, (size_t)jp.npsub)
        SCR(47, d_entries, jp.pblk.size() * (size_t)JP_POS)
        SCR(48, d_segtab, (size_t)jp.nseg)
        SCR(49, d_cursor, 4)  // [0] the hit cursor; [1..2] = four 32-bit counters of k_join_hist
        HIPCHK(hipMemcpyAsync(d_blob, blob.data(), blob_bytes, hipMemcpyHostToDevice, st));
        jv.gfirst = (const int32_t *)(d_blob + o_gfirst);
        jv.gns = (const int32_t *)(d_blob + o_gns);
        jv.pfirst = (const int32_t *)(d_blob + o_pfirst);
        jv.pblk = (const int2 *)(d_blob + o_pblk);
        jv.psubrow = (const int64_t *)(d_blob + o_psubrow);
        jv.jblk = (const int2 *)(d_blob + o_jblk);
        jv.segrow = (const int64_t *)(d_blob + o_segrow);
        jv.psub = d_psub;
        jv.entries = d_entries;
        jv.segtab = d_segtab;
        jv.cursor = d_cursor;
        jv.status = d_status;
        jv.npart = (int32_t)jp.pblk.size();
        jv.njoin = (int32_t)jp.jblk.size();
        // reads of groups without k-mers have no join block: their rows read as "no hits"
        HIPCHK(dhk_memset(st, d_segtab, 0, sizeof(uint64_t) * (size_t)std::max<int64_t>(jp.nseg, 1)));
        HIPCHK(hipEventRecord(ctx->ev[6], st));
        dhk_join_part(st, jv, bv, o.k, o.kmer_mod);
        HIPCHK(hipGetLastError());
        // hit buffer: measured 0.77 hits per base for pile-ups of 60 reads at 13 % error; a rerun sizes it exactly
        int64_t hcap = std::max<int64_t>(1 << 20, (int64_t)(1.25 * (double)A->total));
        if (const char *e = getenv("DH_JOIN_HITCAP")) hcap = std::max<int64_t>(1, atoll(e));  // development / tests
        for (int attempt = 0;; attempt++) {
            SCR(50, d_hits, (size_t)hcap)
            jv.hits = d_hits;
            jv.hits_cap = hcap;
            HIPCHK(hipMemsetAsync(d_cursor, 0, sizeof(unsigned long long), st));
            dhk_join(st, jv, bv, dopt, A->ix.d_goff, sepv);
            HIPCHK(hipGetLastError());
            unsigned long long cur = 0;
            int32_t jstatus = 0;
            HIPCHK(hipMemcpyAsync(&cur, d_cursor, sizeof(cur), hipMemcpyDeviceToHost, st));
            HIPCHK(hipMemcpyAsync(&jstatus, d_status, sizeof(int32_t), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            join_hits = (int64_t)cur;
            if (jstatus & DH_ST_JOIN_OVERFLOW) {  // a slice did not fit its LDS table: directory path for this call
                use_join = false;
                break;
            }
            if (!(jstatus & DH_ST_JOIN_HITCAP)) break;
            if (attempt >= 2) return fail(DH_EOVERFLOW, "k-mer join: hit buffer capacity exceeded twice");
            hcap = (int64_t)cur + 1024;
            HIPCHK(hipMemsetAsync(d_status, 0, sizeof(int32_t), st));
        }
        if (use_join) {
            dhk_join_hist(st, jv, B->d_group, B->n, (unsigned int *)(d_cursor + 1));
            HIPCHK(hipMemcpyAsync(jhist, d_cursor + 1, sizeof(unsigned int) * 4, hipMemcpyDeviceToHost, st));
        }
        HIPCHK(hipEventRecord(ctx->ev[7], st));
        HIPCHK(hipEventSynchronize(ctx->ev[7]));
        HIPCHK(hipEventElapsedTime(&ms_join, ctx->ev[6], ctx->ev[7]));
        if (!use_join) {
            if (getenv("DH_TRACE")) fprintf(stderr, "[join] a slice overflowed its table: falling back to the k-mer directory\n");
            HIPCHK(hipMemsetAsync(d_status, 0, sizeof(int32_t), st));
            if (int rc = build_index(A, o.k, sepv, o.kmer_mod, false)) return rc;
            iv = index_view();
        }
    }

    // expected hits per read (both strands share the LDS buffer): random matches + true seeds (measured
    // 0.075 per sampled k-mer for 15 % error reads at k = 20; reads that need more are redone with their
    // hits in HBM, and a chunk with many of them restarts with the next capacity); pick the LDS hit capacity
    // (ix.n = indexed k-mers; a sampled k-mer of B meets ix.n / (4^k / kmer_mod) of them by chance)
    const double dens = (double)A->ix.n * std::max(1, o.kmer_mod) / std::pow(4.0, o.k) / std::max(1, A->ngroups);
    const double exp_hits = (double)B->max_len / std::max(1, o.kmer_mod) * (2.0 * dens + 0.1);
    int cap = 1024;
    while (cap < 16384 && exp_hits * 1.5 >= cap) cap *= 2;
    if (A == B) {
        // all-vs-all inside pile-ups: a read shares k-mers with every other read of its group; measured
        // ~0.5 hits per base, and the 2048-entry variant (8 blocks per CU) with a few items redone from
        // HBM beats the 8192-entry one by 40 %
        cap = 1024;
        while (cap < 8192 && 0.6 * B->max_len > cap) cap *= 2;
    }
    if (use_join) {
        // the hits are counted already (a whole second pass with the next size cost 9.5 ms at configs[2] when the guess
        // was one size short)
        // (tiers: reads above the first capacity are redone by the 8192-entry variant, reads above that from HBM -- so
        // the first tier is the smallest one that serves at least 70 % of the reads)
        const unsigned int tol = (unsigned int)(0.3 * B->n);
        cap = jhist[0] <= tol ? 2048 : (jhist[1] <= tol ? 4096 : (jhist[2] <= tol ? 8192 : 16384));
    }
    if (const char *e = getenv("DH_SEED_CAP")) cap = atoi(e);  // development: 1024 .. 16384, power of two
    // ---- a mapping pass (A != B, ungrouped): the seeds of a chunk of reads come from the radix-partitioned k-mer join
    // (dh_mjoin.h) -- the reads' k-mers binned by directory slice, every slice joined on chip -- instead of one random
    // directory line per k-mer; bit-identical hits.  Small chunks keep the directory path (the join's fixed costs: 1 024
    // partitions, a page per wavefront); DH_NO_MJOIN=1 forces it, DH_MJOIN_MIN sets the threshold (bases of a chunk).
    bool use_mj = !use_join && A != B && !A->d_group && A->ngroups == 1 && !B->d_group && o.k >= MJ_MINK && o.k <= MJ_MAXK &&
                  o.skip_self == 0 && want_packed && A->ix.n > 0 && A->ix.n < (1ll << 28) && !getenv("DH_NO_MJOIN");
    int64_t mj_min_bases = 64ll << 20;
    if (const char *e = getenv("DH_MJOIN_MIN")) mj_min_bases = atoll(e);
    if (use_mj && !A->ix.d_bitmap) {
        // about 16 buckets per indexed k-mer (7 % of the looked-up k-mers then pass the filter without being in A)
        int32_t nbbits = std::min(MJ_MAXBITS, std::min(2 * o.k, std::max(MJ_PBITS + 5, ceil_log2((uint64_t)A->ix.n) + 4)));
        const size_t words = (size_t)1 << (nbbits - 5);
        HIPCHK(dh_dev_alloc(&A->ix.d_bitmap, sizeof(uint32_t) * words));
        HIPCHK(dhk_memset(st, A->ix.d_bitmap, 0, sizeof(uint32_t) * words));
        dhk_mj_bitmap(st, A->ix.d_ent, A->ix.n, o.k, nbbits, A->ix.d_bitmap);
        HIPCHK(hipGetLastError());
        A->ix.nbbits = nbbits;
    }
    JoinView jv_mj = {};
    bool mj_skip_chunk = false;  // the chunk at hand overflowed a capacity of the join: directory path for it
    uint32_t *d_mjctr_last = nullptr;
    int64_t mj_exp_ent_last = 0, mj_npages_last = 0;

    std::vector<int32_t> h_ncand((size_t)cn), h_nhits((size_t)cn);
    float ms_seed = 0, ms_wave = 0, ms_gather = 0;
    double w_g[6] = {0, 0, 0, 0, 0, 0};  // host wall of the chunk loop's phases (DH_TRACE)

    const int64_t item_first = 2ll * first, item_end = item_first + nitems_total;
    // The device-to-host copy of a chunk's records runs as a copy kernel here; beside it the streaming kernels that make
    // the next chunk's derived copies (pack, reverse complement, planes) ran 2-5 x slower (12 ms between two chunks of
    // configs[2] for 6.5 ms of work).  The copy -- and the hook that waits for it -- of chunk c is therefore issued after
    // chunk c + 1's copies have been made: it overlaps that chunk's seed kernel instead.
    std::function<int()> deferred;
    for (int64_t item0 = item_first; item0 < item_end; item0 += cn) {
        const int32_t ni = (int32_t)std::min<int64_t>(cn, item_end - item0);
        double w_c = now_ms();
        auto lap = [&](int i) {
            const double t = now_ms();
            w_g[i] += t - w_c;
            w_c = t;
        };
        // the partitioned join of this chunk is planned here, ahead of the chunk's derived copies and of the previous chunk's
        // device-to-host copy: its first kernel (k_mj_tile_reads, a binary search per tile) then runs before the copy kernels
        // take the device (beside them it took 4 ms instead of 10 us)
        MjView mv = {};
        bool mj_planned = false;
        {
            const int32_t cr0 = (int32_t)(item0 >> 1), cr1 = (int32_t)((item0 + ni) >> 1);
            const int64_t cb0 = B->h_off[(size_t)cr0], cb1 = B->h_off[(size_t)cr1];
            if (use_mj && !mj_skip_chunk && cb1 - cb0 >= mj_min_bases && cb1 - cb0 < (1ll << 40)) {
                mv.c0 = cb0;
                mv.c1 = cb1;
                mv.r0 = cr0;
                mv.r1 = cr1;
                mv.k = o.k;
                mv.kmer_mod = std::max(1, o.kmer_mod);
                mv.nbbits = A->ix.nbbits;
                // bases per tile: 13/16 of the tile's capacity expected (modimer sampling is a hash), every
                // lane of the block rolls the same number of positions, positions fit MJ_POSBITS
                // (a wavefront stages its eighth of the tile's entries in its own 1 024 slots: 832 expected, 7 sigma of slack)
                int64_t tb = (int64_t)(MJ_CAP / 16 * 13) * mv.kmer_mod;
                if (mv.kmer_mod == 1) tb = MJ_CAP;
                tb = std::min<int64_t>(tb, (1 << MJ_POSBITS) - 64);
                tb = std::max<int64_t>(MJ_THREADS * 8, tb / (MJ_THREADS * 8) * (MJ_THREADS * 8));
                mv.tb = (int32_t)tb;
                const int64_t ntiles = (cb1 - cb0 + tb - 1) / tb;
                mv.ntiles = (int32_t)ntiles;
                mv.ntiles_pad = (int32_t)((ntiles + MJ_BATCH - 1) / MJ_BATCH * MJ_BATCH);
                mv.ngroups = mv.ntiles_pad / MJ_GROUP;
                const int64_t tbg = tb * MJ_GROUP;
                mv.nseg = (int32_t)((B->max_len + tbg - 1) / tbg + 1);
                // hit pool: 20 % of the sampled k-mers hit (measured 6 % at 13 % error and k = 20; raised when a pool ran out: low-error
                // reads) plus the chance matches, a page per wavefront
                // of the probe kernel on top; the same number of hits regrouped by read
                // (dens: chance matches of a sampled k-mer per strand, as for the LDS capacity above)
                const int64_t exp_ent = (cb1 - cb0) / mv.kmer_mod;
                // (a page is left when less than a quarter of it is free: a third more pages than hits)
                // (the pool holds the SURVIVORS of the filter: the hits' k-mers and a few per cent of the others)
                int64_t npages = (int64_t)(1.34 * (ctx->mj_hit_frac + 0.06 + 2.5 * dens) * (double)exp_ent) / MJ_PAGE + (int64_t)ctx->ncu * (MJ_PROBE_THREADS / 64) + 64;
                if (const char *e = getenv("DH_MJOIN_PAGES")) npages = std::max<int64_t>(1, atoll(e));  // development / tests: force the fall-back
                if (ntiles < (1ll << 30) / MJ_P && npages < (1ll << 31) / 2 && mv.nseg <= 512) {
                    mv.npages = (int32_t)npages;
                    mv.rcap = npages * MJ_PAGE;
                    uint32_t *d_mjctr;
                    SCR(64, mv.ent, (size_t)ntiles * MJ_CAP)
                    SCR(65, mv.segoff, (size_t)ntiles * MJ_P)
                    SCR(66, mv.tile_n, (size_t)ntiles)
                    SCR(73, mv.tile_r, (size_t)ntiles)
                    SCR(67, mv.seg, (size_t)MJ_P * mv.ntiles_pad)
                    SCR(68, mv.hseg, (size_t)mv.ngroups * MJ_P)
                    SCR(69, mv.hits, (size_t)npages * MJ_PAGE)
                    SCR(70, mv.rhits, (size_t)mv.rcap)
                    SCR(71, mv.segtab, (size_t)(cr1 - cr0) * mv.nseg)
                    SCR(72, d_mjctr, 16)
                    mv.ctr = d_mjctr;
                    d_mjctr_last = d_mjctr;
                    mj_exp_ent_last = exp_ent;
                    mj_npages_last = npages;
                    mv.bitmap = A->ix.d_bitmap;
                    mv.status = d_status;
                    if (const char *e = getenv("DH_MJ_DBG")) mv.dbg = atoi(e);
                    dhk_mj_tile_reads(st, bv, mv);
                    HIPCHK(hipGetLastError());
                    mj_planned = true;
                }
            }
        }
        ChunkCopies cc;
        uint8_t *d_bpk2 = nullptr, *d_brcpk2 = nullptr;
        if (db_copies) {
            cc.rc = B->d_rc;
            cc.pk = B->d_pk;
            cc.rcpk = B->d_rcpk;
            cc.has_n = B->has_n != 0;
        } else if (int rc = chunk_copies(ctx, B, (int32_t)(item0 >> 1), (int32_t)((item0 + ni) >> 1), want_packed, A->has_n != 0, &cc,
                                         tiled && want_packed && !res2 && !getenv("DH_PLANES_BY_PASS")))
            return rc;
        const bool packed = want_packed && A->has_n == 0 && !cc.has_n && cc.pk && cc.rcpk;
        if (tiled) {
            if (!packed) return fail(DH_EINVAL, "algo 1 (DH-2) needs sequences of a, c, g, t only (2-bit copies), B holds other codes");
            if (res2) {
                // the transposed pairs read this chunk of B as their A'': keep its 2-bit copies
                const size_t pbytes = (size_t)cc.pk_words * 8 + 2 * PK_PAD;
                SCR(41, d_bpk2, pbytes)
                SCR(42, d_brcpk2, pbytes)
                HIPCHK(hipMemcpyAsync(d_bpk2, cc.pk_w0 - PK_PAD, pbytes, hipMemcpyDeviceToDevice, st));
                HIPCHK(hipMemcpyAsync(d_brcpk2, cc.rcpk_w0 - PK_PAD, pbytes, hipMemcpyDeviceToDevice, st));
            }
            if (!cc.planes) {
                dhk_pk2planes(st, cc.pk_w0, cc.pk_words);
                dhk_pk2planes(st, cc.rcpk_w0, cc.pk_words);
            }
            HIPCHK(hipGetLastError());
        }
        if (deferred) {
            HIPCHK(hipEventRecord(ctx->ev[6], st));
            HIPCHK(hipStreamWaitEvent(ctx->cstream, ctx->ev[6], 0));
            const int rc = deferred();
            deferred = nullptr;
            if (rc) return rc;
        }
        // per-chunk arrays are indexed by absolute item inside the kernels: shift the bases
        DhCand *candbase = d_cand - item0 * o.max_cand;
        DhLa *labase = d_la ? d_la - item0 * o.max_la : nullptr;
        uint16_t *trbase = d_trslots ? d_trslots - item0 * (int64_t)o.max_la * trmax : nullptr;
        int32_t *ncandbase = d_ncand - item0, *nhitsbase = d_nhits - item0;
        int32_t *nlabase = (int32_t *)d_nla - item0, *ntrbase = (int32_t *)d_ntr - item0;
        lap(0);
        HIPCHK(hipEventRecord(ctx->ev[2], st));
        HIPCHK(hipMemsetAsync(d_queue, 0, 4 * sizeof(uint32_t), st));
        uint64_t *d_fscr = nullptr;
        if (cap > 4096 && cap <= 8192)
            SCR(30, d_fscr, (size_t)ctx->ncu * DH_SEED_FSCR_BLOCKS_PER_CU * DH_SEED_FSCR_WORDS)
        bool mj_chunk = false;
        if (mj_planned && !cc.has_n) {
            const int32_t cr0 = mv.r0, cr1 = mv.r1;
            HIPCHK(dhk_memset(st, mv.segtab, 0, sizeof(unsigned long long) * (size_t)(cr1 - cr0) * mv.nseg));
            dhk_mj_run(st, bv, iv, dopt, mv, ctx->ncu);
            HIPCHK(hipGetLastError());
            jv_mj = JoinView{};
            jv_mj.segtab = (uint64_t *)mv.segtab;
            jv_mj.hits = mv.rhits;
            jv_mj.status = d_status;
            jv_mj.ns_fixed = mv.nseg;
            jv_mj.read0 = cr0;
            mj_chunk = true;
        }
        const bool jn = use_join || mj_chunk;            // the back end gathers its hits from segments
        const JoinView &jvx = mj_chunk ? jv_mj : jv;
        // (the back end fed from segments exists with 2048, 4096 and 8192 entries of LDS; the 8192-entry one scans in a slab)
        const int tier_max = getenv("DH_SEED_NO16K") ? 8192 : 16384;  // development / tests: without the 16384-entry tier
        // (a mapping chunk through the partitioned join starts with the wavefront-per-read tier: 512 hits, 32 candidate band
        // pairs -- 140 hits per read at 1/8 sampling; a block of 512 threads per read kept 3 reads per CU in flight and spent
        // its time in barriers -- then 2048, 8192, 16384 for what overflows; DH_SEED_NO_WAVE_TIER=1: from 2048 as before)
        // The first tier of a mapping chunk goes by the MEAN hits per read -- 0.075 true seeds per sampled k-mer at 13 % error
        // plus the random matches -- with half as much again of room (`cap` above goes by the longest read of the DB: right for
        // the directory path, whose overflowing reads are staged in HBM, two sizes too large here, where the next tiers take
        // them from a list: the unsampled mapping of configs[2] ran all reads through the 4096-entry variant for 1 100 hits per
        // read).  The wavefront-per-read tier is switched off for the context once a quarter of a chunk's reads overflowed it.
        const double kmers_per_read = (double)(B->h_off[(size_t)((item0 + ni) >> 1)] - B->h_off[(size_t)(item0 >> 1)]) /
                                      std::max(1, ni / 2) / std::max(1, o.kmer_mod);
        const double mean_hits = kmers_per_read * (2.0 * dens + 0.075);
        const bool wave_tier = mj_chunk && ctx->seed_wave_tier && 1.5 * mean_hits <= 512.0 && !getenv("DH_SEED_NO_WAVE_TIER");
        int capj = std::min(std::max(cap, 2048), tier_max);
        if (mj_chunk) {
            capj = 2048;
            while (capj < tier_max && 1.5 * mean_hits > capj) capj *= 2;
            if (wave_tier) capj = 512;
        }
        if (jn && capj > 4096) SCR(30, d_fscr, (size_t)ctx->ncu * DH_SEED_FSCR_BLOCKS_PER_CU * (capj > 8192 ? DH_SEED_FSCR_WORDS16 : DH_SEED_FSCR_WORDS))
        if (jn)
            dhk_seed_join(st, capj, bv, iv, dopt, jvx, (int32_t)item0, ni, candbase, ncandbase, nhitsbase, d_status,
                          d_queue + 1, ctx->ncu, d_fscr, nullptr, 0);
        else
            dhk_seed(st, cap, bv, iv, dopt, (int32_t)item0, ni, candbase, ncandbase, nhitsbase, d_status,
                     d_queue + 1, ctx->ncu, d_fscr);
        HIPCHK(hipGetLastError());
        {
            // items whose hits did not fit the LDS buffer (ncand == -1) are redone with their hits
            // staged in HBM: same kernel code, capacity = the item's own hit count
            int32_t status = 0;
            unsigned long long sm[4] = {0, 0, 0, 0};
            dhk_seed_summary(st, d_ncand, d_nhits, ni, d_summary);
            HIPCHK(hipMemcpyAsync(&status, d_status, sizeof(int32_t), hipMemcpyDeviceToHost, st));
            HIPCHK(hipMemcpyAsync(sm, d_summary, sizeof(sm), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            if (mj_chunk && (status & DH_ST_MJ_POOL) && !(status & DH_ST_MJ_OVERFLOW) && !getenv("DH_MJOIN_PAGES")) {
                // the hit pool ran out (more hits per k-mer than planned: low-error reads, short k-mers): sized by the pages
                // the probe kernel asked for, the chunk runs through the join again -- and the later ones start with that rate
                unsigned long long cnt2[2] = {0, 0};  // hits counted for rhits, survivors that found no page
                HIPCHK(hipMemcpyAsync(cnt2, d_mjctr_last + 10, sizeof(cnt2), hipMemcpyDeviceToHost, st));
                status &= ~DH_ST_MJ_POOL;
                HIPCHK(hipMemcpyAsync(d_status, &status, sizeof(int32_t), hipMemcpyHostToDevice, st));
                HIPCHK(hipStreamSynchronize(st));
                const double have = (double)mj_npages_last * MJ_PAGE / 1.34;
                const double need = 1.2 * std::max(have + (double)cnt2[1], (double)cnt2[0]) / std::max<double>(1.0, (double)mj_exp_ent_last);
                ctx->mj_hit_frac = std::max(ctx->mj_hit_frac * 1.5, need);
                if (getenv("DH_TRACE")) fprintf(stderr, "[mjoin] pool too small (%llu survivors without a page, %llu hits): %.2f per k-mer planned from now on\n", cnt2[1], cnt2[0], ctx->mj_hit_frac);
                if (ctx->mj_hit_frac <= 64.0) {
                    item0 -= cn;
                    continue;
                }
                status |= DH_ST_MJ_OVERFLOW;
            }
            if (mj_chunk && (status & (DH_ST_MJ_OVERFLOW | DH_ST_MJ_POOL))) {
                // a capacity of the partitioned join was exceeded (repeat-rich reads): this chunk again, by the directory
                if (getenv("DH_TRACE")) fprintf(stderr, "[mjoin] a capacity was exceeded: chunk at item %lld redone by the directory path\n", (long long)item0);
                status &= ~(DH_ST_MJ_OVERFLOW | DH_ST_MJ_POOL);
                HIPCHK(hipMemcpyAsync(d_status, &status, sizeof(int32_t), hipMemcpyHostToDevice, st));
                HIPCHK(hipStreamSynchronize(st));
                mj_skip_chunk = true;
                ctx->mj_fallbacks++;
                item0 -= cn;
                continue;
            }
            std::vector<int32_t> big;
            int32_t gcap = 0;
            if (sm[2] > 0) {  // the per-item arrays travel only when some item overflowed its LDS buffer
                HIPCHK(hipMemcpyAsync(h_ncand.data(), d_ncand, sizeof(int32_t) * (size_t)ni, hipMemcpyDeviceToHost, st));
                HIPCHK(hipMemcpyAsync(h_nhits.data(), d_nhits, sizeof(int32_t) * (size_t)ni, hipMemcpyDeviceToHost, st));
                HIPCHK(hipStreamSynchronize(st));
                for (int32_t it = 0; it + 1 < ni; it += 2)  // a read overflows with both of its strands
                    if (h_ncand[(size_t)it] == -1) {
                        big.push_back((int32_t)((item0 + it) >> 1));
                        gcap = std::max(gcap, h_nhits[(size_t)it] + h_nhits[(size_t)it + 1]);
                    }
            }
            if (wave_tier && big.size() * 4 > (size_t)(ni / 2)) {
                ctx->seed_wave_tier = 0;
                if (getenv("DH_TRACE"))
                    fprintf(stderr, "[seeds] %zu of %d reads overflow the wavefront-per-read tier: not used by this context any more\n", big.size(), ni / 2);
            }
            // many items overflow the LDS buffer: the next size is cheaper than HBM staging -- up to 8192; the 16384-entry
            // variant keeps one block per CU resident and pays off only when most items need it
            size_t redo_all = (size_t)ni / 50 + 8;
            if (cap >= 8192) redo_all = (size_t)ni / 4;
            if (const char *e = getenv("DH_SEED_BIG_PCT")) redo_all = (size_t)((double)ni * atof(e) / 200.0);  // development (reads = ni / 2)
            if (getenv("DH_TRACE") && !big.empty())
                fprintf(stderr, "[seeds] cap %d: %zu of %d reads overflow (whole chunk again above %zu)\n", cap, big.size(), ni / 2, redo_all);
            if (!jn && big.size() > redo_all && cap < 16384) {
                cap *= 2;
                item0 -= cn;
                continue;
            }
            if (mj_chunk) ctx->mj_chunks++;
            mj_skip_chunk = false;  // (the next chunk tries the join again)
            if (jn && capj < tier_max && !big.empty()) {
                // further tiers of the join path: the reads above the first capacity that fit the 8192-entry variant, then the
                // 16384-entry one (uncapped pile-ups: ~10 000 hits per read); what is left is staged in HBM
                std::vector<int32_t> huge;
                int32_t gcap2 = 0;
                for (int tier = capj < 2048 ? 2048 : 8192; tier <= tier_max; tier = tier < 8192 ? 8192 : tier * 2) {
                    if (tier <= capj) continue;
                    std::vector<int32_t> mid;
                    huge.clear();
                    gcap2 = 0;
                    for (int32_t r : big) {
                        const size_t it = (size_t)(2 * (int64_t)r - item0);
                        const int32_t nh = h_nhits[it] + h_nhits[it + 1];
                        if (nh <= tier)
                            mid.push_back(r);
                        else {
                            huge.push_back(r);
                            gcap2 = std::max(gcap2, nh);
                        }
                    }
                    if (!mid.empty()) {
                        int32_t *d_mid;
                        uint64_t *d_fscr2;
                        SCR(54, d_mid, mid.size())
                        SCR(30, d_fscr2, (size_t)ctx->ncu * DH_SEED_FSCR_BLOCKS_PER_CU * (tier > 8192 ? DH_SEED_FSCR_WORDS16 : DH_SEED_FSCR_WORDS))
                        HIPCHK(hipMemcpyAsync(d_mid, mid.data(), sizeof(int32_t) * mid.size(), hipMemcpyHostToDevice, st));
                        HIPCHK(hipMemsetAsync(d_queue + 1, 0, sizeof(uint32_t), st));
                        dhk_seed_join(st, tier, bv, iv, dopt, jvx, (int32_t)item0, ni, candbase, ncandbase, nhitsbase, d_status,
                                      d_queue + 1, ctx->ncu, d_fscr2, d_mid, (int32_t)mid.size());
                        HIPCHK(hipGetLastError());
                        HIPCHK(hipStreamSynchronize(st));  // mid goes out of scope
                    }
                    if (getenv("DH_TRACE"))
                        fprintf(stderr, "[seeds] join tiers: %zu reads redone with %d entries, %zu left\n", mid.size(), tier, huge.size());
                    big = huge;
                }
                gcap = gcap2;
            }
            if (!big.empty()) {
                if (gcap > (1 << 22)) return fail(DH_EOVERFLOW, "seed filter: more than 4M k-mer hits for one sequence; lower -t");
                int32_t pow2 = 1;
                while (pow2 < gcap) pow2 <<= 1;  // the bitonic sort pads to a power of two
                // per block: pow2 hits, pow2 64-bit prefix sums, pow2 32-bit band-head positions (k_seed<0>)
                const size_t slab_words = 2 * (size_t)pow2 + ((size_t)pow2 + 1) / 2;
                const size_t per_launch = std::max<size_t>(1, (size_t)(2ull << 30) / (slab_words * 8));
                int32_t *d_list;
                uint64_t *d_gbuf;
                SCR(15, d_list, big.size())
                SCR(16, d_gbuf, std::min(per_launch, big.size()) * slab_words)
                HIPCHK(hipMemcpyAsync(d_list, big.data(), sizeof(int32_t) * big.size(), hipMemcpyHostToDevice, st));
                for (size_t b0 = 0; b0 < big.size(); b0 += per_launch) {
                    const int32_t cnt = (int32_t)std::min(per_launch, big.size() - b0);
                    HIPCHK(hipMemsetAsync(d_queue + 2, 0, sizeof(uint32_t), st));
                    if (jn)
                        dhk_seed_big_join(st, bv, iv, dopt, jvx, d_list + b0, cnt, d_gbuf, pow2, candbase, ncandbase,
                                          nhitsbase, d_status, d_queue + 2, ctx->ncu);
                    else
                        dhk_seed_big(st, bv, iv, dopt, d_list + b0, cnt, d_gbuf, pow2, candbase, ncandbase,
                                     nhitsbase, d_status, d_queue + 2, ctx->ncu);
                    HIPCHK(hipGetLastError());
                }
                HIPCHK(hipMemcpyAsync(&status, d_status, sizeof(int32_t), hipMemcpyDeviceToHost, st));
                HIPCHK(hipStreamSynchronize(st));
                if (status & DH_ST_HIT_OVERFLOW)
                    return fail(DH_EOVERFLOW, "seed filter: capacity exceeded in the HBM-staged pass");
                stats.big_items += 2 * (int64_t)big.size();
            }
        }
        lap(1);
        HIPCHK(hipEventRecord(ctx->ev[3], st));
        HIPCHK(hipMemsetAsync(d_queue, 0, sizeof(uint32_t), st));
        // symmetric mode claims slots with atomics: every counter starts at zero
        HIPCHK(hipMemsetAsync(d_nla, 0, sizeof(uint32_t) * (size_t)(o.skip_self == 2 ? ni + 1 : 0), st));
        HIPCHK(hipMemsetAsync(d_ntr, 0, sizeof(uint32_t) * (size_t)(o.skip_self == 2 ? ni + 1 : 0), st));
        HIPCHK(hipMemsetAsync(d_nla + ni, 0, sizeof(uint32_t), st));
        HIPCHK(hipMemsetAsync(d_ntr + ni, 0, sizeof(uint32_t), st));
        // symmetric all-vs-all: one work unit per (item, A read) group of candidates instead of per
        // item (k_units); d_queue[3] counts them
        void *d_units = nullptr;
        uint32_t *d_candoff = nullptr;
        int32_t *d_reclist = nullptr;
        int64_t nrec_slots = 0;
        if (sym_tiled) {
            // candidate slots: exclusive prefix sums of the items' candidate counts; two record slots per candidate
            SCR(51, d_candoff, (size_t)ni + 1)
            dhk_cand_counts(st, d_ncand, ni, d_candoff);
            dhk_scan(st, d_candoff, (int64_t)ni + 1, d_sums);
            uint32_t ncand_total = 0;
            HIPCHK(hipMemcpyAsync(&ncand_total, d_candoff + ni, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            nrec_slots = 2 * (int64_t)ncand_total;
            if (nrec_slots >= INT32_MAX) return fail(DH_EOVERFLOW, "symmetric alignment: more than 2^30 candidates in one call");
            SCR(9, d_la, (size_t)nrec_slots)
            SCR(10, d_trslots, (size_t)nrec_slots * trmax)
            SCR(52, d_reclist, (size_t)nrec_slots)
            HIPCHK(dhk_memset(st, d_la, 0, sizeof(DhLa) * (size_t)std::max<int64_t>(nrec_slots, 1)));
        }
        if (o.skip_self == 2 && ni > 1) {
            if (tiled) {
                dhtile::Unit *d_u;
                SCR(24, d_u, (size_t)ni * (size_t)o.max_cand)  // at most one unit per candidate
                d_units = d_u;
                dhk_tile_units(st, candbase, ncandbase, (const int32_t *)d_candoff - item0, (int32_t)item0, ni, o.max_cand, A->d_off,
                               B->d_off, d_u, d_queue + 3);
            } else {
                int4 *d_u;
                SCR(24, d_u, (size_t)ni * (size_t)o.max_cand)
                d_units = d_u;
                dhk_units(st, candbase, ncandbase, (int32_t)item0, ni, o.max_cand, d_units, d_queue + 3);
            }
            HIPCHK(hipGetLastError());
        }
        HIPCHK(dhk_memset(st, d_ovf, 0, sizeof(int32_t) * (size_t)ni));
        WaveScratch ws{d_pool, d_cdj, d_queue, (const int4 *)d_units, d_queue + 3, poolcap, nbmax, d_ovf - item0};
        if (tiled) {
            dhtile::Params tp = {};
            tp.aoff = A->d_off;
            tp.boff = B->d_off;
            tp.apk = (const uint32_t *)A->d_pk;
            tp.arcpk = (const uint32_t *)A->d_rcpk;
            tp.bpp = (const dhtile::PlanePair *)cc.pk;
            tp.brcpp = (const dhtile::PlanePair *)cc.rcpk;
            // transposed pairs: a symmetric launch (A == B) reads the same copies in both roles
            tp.apk1 = tp.apk;
            tp.arcpk1 = tp.arcpk;
            tp.bpp1 = tp.bpp;
            tp.brcpp1 = tp.brcpp;
            tp.out_la2 = nullptr;
            tp.out_trace2 = nullptr;
            tp.out_nla2 = tp.out_ntr2 = nullptr;
            if (res2) {
                tp.apk1 = (const uint32_t *)(d_bpk2 + PK_PAD + (cc.pk - cc.pk_w0));
                tp.arcpk1 = (const uint32_t *)(d_brcpk2 + PK_PAD + (cc.rcpk - cc.rcpk_w0));
                tp.bpp1 = (const dhtile::PlanePair *)(d_app + PK_PAD);
                tp.brcpp1 = (const dhtile::PlanePair *)(d_arcpp + PK_PAD);
                tp.out_la2 = d_la2 - item0 * o.max_la;
                tp.out_trace2 = d_trslots2 - item0 * (int64_t)o.max_la * trmax;
                tp.out_nla2 = (int32_t *)d_nla2 - item0;
                tp.out_ntr2 = (int32_t *)d_ntr2 - item0;
                HIPCHK(hipMemsetAsync(d_nla2, 0, sizeof(uint32_t) * (size_t)(ni + 1), st));
                HIPCHK(hipMemsetAsync(d_ntr2, 0, sizeof(uint32_t) * (size_t)(ni + 1), st));
            }
            tp.o = dopt;
            tp.item0 = (int32_t)item0;
            tp.nitems = ni;
            tp.cand = candbase;
            tp.ncand = ncandbase;
            tp.queue = d_queue;
            tp.units = (const dhtile::Unit *)d_units;
            tp.nunits = d_queue + 3;
            tp.candoff = sym_tiled ? (const int32_t *)d_candoff - item0 : nullptr;
            tp.book_min = 1;
            if (const char *e = getenv("DH_TILE_BOOK_MIN")) tp.book_min = std::max(1, std::min(64, atoi(e)));
            tp.qbatch = 64;
            if (const char *e = getenv("DH_TILE_QBATCH")) tp.qbatch = std::max(1, std::min(4096, atoi(e)));  // development
            tp.regs = d_regs;
            tp.cold = d_cold;
            tp.nbmax = nbmax;
            tp.trmax = trmax;
            tp.out_la = sym_tiled ? d_la : labase;
            tp.out_trace = sym_tiled ? d_trslots : trbase;
            tp.out_nla = nlabase;
            tp.out_ntr = ntrbase;
            tp.counters = d_counters;
            tp.status = d_status;
            tp.pflags = o.skip_self == 2 ? B->d_pflags : nullptr;
            tp.tandem = o.skip_self == 3 ? 1 : 0;
            dhk_tile(st, tile_waves, &tp);
        } else if (dual)
            dhk_wave2(st, nslots / per_wave, av, bv, A->d_rc, cc.rc, packed ? A->d_pk : nullptr,
                      packed ? A->d_rcpk : nullptr, packed ? cc.pk : nullptr, packed ? cc.rcpk : nullptr, dopt,
                      (int32_t)item0, ni, candbase, ncandbase, ws, labase, trbase, trmax, nlabase, ntrbase, d_counters,
                      d_status);
        else
        dhk_wave(st, nslots, av, bv, cc.rc, packed ? A->d_pk : nullptr, packed ? cc.pk : nullptr,
                 packed ? cc.rcpk : nullptr, dopt, (int32_t)item0, ni, candbase, ncandbase, ws, labase, trbase,
                 trmax, nlabase, ntrbase, d_counters, d_status);
        HIPCHK(hipGetLastError());
        stats.wave_launches++;
        HIPCHK(hipEventRecord(ctx->ev[4], st));
        // compaction on the device: exclusive scans of the per-item counts, then one copy kernel
        if (sym_tiled) dhk_rec_count(st, d_la, nrec_slots, (int32_t)item0, d_nla, d_ntr);  // records per A-read item
        dhk_scan(st, d_nla, (int64_t)ni + 1, d_sums);
        dhk_scan(st, d_ntr, (int64_t)ni + 1, d_sums);
        uint32_t totals[2] = {0, 0};
        int32_t status = 0;
        HIPCHK(hipMemcpyAsync(&totals[0], d_nla + ni, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(&totals[1], d_ntr + ni, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(&status, d_status, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        unsigned long long sm2[4] = {0, 0, 0, 0};
        dhk_seed_summary(st, d_ncand, d_nhits, ni, d_summary);
        HIPCHK(hipMemcpyAsync(sm2, d_summary, sizeof(sm2), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        lap(2);
        if (status & DH_ST_POOL_OVERFLOW)
            return fail(DH_EOVERFLOW, "wave: trace-tree pool or boundary capacity exceeded");
        stats.hits += (int64_t)sm2[0];
        stats.cands += (int64_t)sm2[1];
        if (sm2[3] > 0) {  // the seed filter gave up on some items (> 256 candidate band pairs): which reads
            HIPCHK(hipMemcpy(h_ncand.data(), d_ncand, sizeof(int32_t) * (size_t)ni, hipMemcpyDeviceToHost));
            for (int32_t it = 0; it < ni; it++)
                if (h_ncand[(size_t)it] == -2) {
                    stats.overflow_items++;
                    res->ovf_reads.push_back((int32_t)((item0 + it) >> 1));
                }
        }
        auto read_ovf = [&]() -> int {  // items whose records did not fit (DH-1: > max_la slots; DH-2: > 512 per read and strand)
            std::vector<int32_t> h_ovf((size_t)ni);
            HIPCHK(hipMemcpy(h_ovf.data(), d_ovf, sizeof(int32_t) * (size_t)ni, hipMemcpyDeviceToHost));
            for (int32_t it = 0; it < ni; it++)
                if (h_ovf[(size_t)it]) {
                    stats.overflow_items++;
                    res->ovf_reads.push_back((int32_t)((item0 + it) >> 1));
                }
            return DH_OK;
        };
        if (o.skip_self == 2 && !sym_tiled)
            if (int rc = read_ovf()) return rc;
        lap(3);
        hipEvent_t copied = nullptr;
        std::function<int()> enqueue_copy;
        bool defer_copy = false;
        if (totals[0] > 0) {
            // the compacted buffers are reused: the previous chunk's copies must have left them
            HIPCHK(hipStreamSynchronize(ctx->cstream));
            SCR(13, d_laout, totals[0])
            const size_t l0 = res->la.size(), t0 = keep_dev ? (size_t)res->d_trace_own_len : res->trace.size();
            if (keep_dev) {
                // the chunk's trace values are compacted straight into the set's own device buffer (grown by copy when
                // the first chunk's yield was a bad guess for the call)
                const int64_t need = (int64_t)t0 + totals[1];
                if (need > res->d_trace_own_cap) {
                    const double f = 1.15 * (double)nitems_total / std::max<double>(1.0, (double)(item0 - item_first + ni));
                    const int64_t cap = std::max<int64_t>(need + 65536, (int64_t)(f * (double)need) + 65536);
                    uint16_t *nb = nullptr;
                    HIPCHK(dh_dev_alloc((void **)&nb, sizeof(uint16_t) * (size_t)cap));
                    if (res->d_trace_own) {
                        HIPCHK(hipMemcpyAsync(nb, res->d_trace_own, sizeof(uint16_t) * t0, hipMemcpyDeviceToDevice, st));
                        HIPCHK(hipStreamSynchronize(st));
                        dh_dev_free(res->d_trace_own);
                    }
                    res->d_trace_own = nb;
                    res->d_trace_own_cap = cap;
                    res->device = ctx->device;
                }
                d_trout = res->d_trace_own + t0;
            } else
                SCR(14, d_trout, totals[1])
            if (sym_tiled) {
                uint32_t *d_cur;
                SCR(53, d_cur, (size_t)ni)
                HIPCHK(dhk_memset(st, d_cur, 0, sizeof(uint32_t) * (size_t)ni));
                dhk_rec_scatter(st, d_la, nrec_slots, (int32_t)item0, d_nla, d_cur, d_reclist);
                dhk_compact_sym(st, d_la, d_trslots, trmax, d_reclist, ni, d_nla, d_ntr, (int64_t)t0, d_laout, d_trout, d_ovf);
            } else
                dhk_compact(st, d_la, d_trslots, trmax, o.max_la, o.skip_self == 2 ? 1 : 0, ni, d_nla, d_ntr, (int64_t)t0,
                            d_laout, d_trout);
            HIPCHK(hipGetLastError());
            if (l0 == 0 && ni < nitems_total) {
                // first of several chunks: reserve for the whole call (this chunk's yield + 15 %) so that
                // the result never moves while it grows
                const double f = 1.15 * (double)nitems_total / ni;
                res->la.reserve((size_t)(f * totals[0]) + 1024);
                if (!keep_dev) res->trace.reserve((size_t)(f * totals[1]) + 65536);
            }
            if (l0 + totals[0] > res->la.capacity() || (!keep_dev && t0 + totals[1] > res->trace.capacity()))
                tasks.join();  // the records are about to move: copies and hooks in flight finish first
            const bool dev_only = (want_sorted & 2) && t0 == 0 && item0 == item_first && ni == nitems_total;
            const bool rec_dev = dev_only && (want_sorted & 4) && sym_tiled && l0 == 0 && !hook;
            if (!rec_dev) res->la.resize(l0 + totals[0]);
            if (keep_dev)
                res->d_trace_own_len = (int64_t)t0 + totals[1];
            else if (!dev_only)
                res->trace.resize(t0 + totals[1]);
            lap(4);
            // device-to-host on the copy stream: it overlaps the next chunk's kernels
            hipEvent_t compacted = ctx->cev[nchunk_done & 1];
            copied = ctx->cev[2 + (nchunk_done & 1)];
            HIPCHK(hipEventRecord(compacted, st));
            res->d_trace = (t0 == 0 && item0 == item_first && ni == nitems_total) ? d_trout : nullptr;
            if (totals[1] > 0 && dev_only) res->d_trace_len = (int64_t)totals[1];
            const hipEvent_t copied_ev = copied;
            const uint32_t nla_c = totals[0], ntr_c = totals[1];
            hipStream_t cst = ctx->cstream;
            enqueue_copy = [res, l0, t0, nla_c, ntr_c, dev_only, keep_dev, compacted, copied_ev, cst, d_laout, d_trout]() -> int {
                HIPCHK(hipStreamWaitEvent(cst, compacted, 0));
                HIPCHK(hipMemcpyAsync(res->la.data() + l0, d_laout, sizeof(dh_la) * (size_t)nla_c, hipMemcpyDeviceToHost, cst));
                // the chunk's hook (chain flags, filters, candidates) reads the records only: it starts when they have
                // arrived, while the trace values -- ten times the bytes -- are still on their way (Tasks::join waits
                // for the stream before anybody sees the result)
                HIPCHK(hipEventRecord(copied_ev, cst));
                // (want_sorted & 2: the caller reads the trace from the device copy -- the pile-up all-vs-all, whose host
                // side needs 1 / n of the values: the overlaps of the reference reads -- so the 2 x 160 MB of configs[2] stay)
                if (ntr_c > 0 && !dev_only && !keep_dev)
                    HIPCHK(hipMemcpyAsync(res->trace.data() + t0, d_trout, sizeof(uint16_t) * (size_t)ntr_c, hipMemcpyDeviceToHost, cst));
                return DH_OK;
            };
            defer_copy = hook && tiled && !db_copies && !res2 && !sym_tiled && item0 + cn < item_end && !getenv("DH_NO_DEFER_COPY");
            if (rec_dev) {  // the caller works on the device copy of the records (dh_process_cropped's funnel)
                res->d_la = d_laout;
                res->d_la_n = (int64_t)totals[0];
                res->d_item_off = d_nla;
                HIPCHK(hipEventRecord(copied_ev, cst));
            } else if (!defer_copy)
                if (int rc = enqueue_copy()) return rc;
        }
        if (res2) {
            // the transposed records of the chunk: same compaction, copied on this stream (not the benched path)
            dhk_scan(st, d_nla2, (int64_t)ni + 1, d_sums);
            dhk_scan(st, d_ntr2, (int64_t)ni + 1, d_sums);
            uint32_t tot2[2] = {0, 0};
            HIPCHK(hipMemcpyAsync(&tot2[0], d_nla2 + ni, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            HIPCHK(hipMemcpyAsync(&tot2[1], d_ntr2 + ni, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            if (tot2[0] > 0) {
                SCR(43, d_laout2, tot2[0])
                SCR(44, d_trout2, tot2[1])
                const size_t l2 = res2->la.size(), t2 = res2->trace.size();
                dhk_compact(st, d_la2, d_trslots2, trmax, o.max_la, 0, ni, d_nla2, d_ntr2, (int64_t)t2, d_laout2, d_trout2);
                HIPCHK(hipGetLastError());
                res2->la.resize(l2 + tot2[0]);
                res2->trace.resize(t2 + tot2[1]);
                HIPCHK(hipMemcpyAsync(res2->la.data() + l2, d_laout2, sizeof(dh_la) * (size_t)tot2[0], hipMemcpyDeviceToHost, st));
                if (tot2[1] > 0)
                    HIPCHK(hipMemcpyAsync(res2->trace.data() + t2, d_trout2, sizeof(uint16_t) * (size_t)tot2[1],
                                          hipMemcpyDeviceToHost, st));
            }
        }
        HIPCHK(hipEventRecord(ctx->ev[5], st));
        HIPCHK(hipStreamSynchronize(st));
        if (sym_tiled && totals[0] > 0)  // (set by the compaction)
            if (int rc = read_ovf()) return rc;
        lap(5);
        nchunk_done++;
        if (hook && totals[0] > 0) {
            dh_la *p = res->la.data() + (res->la.size() - totals[0]);
            const int64_t cnt = (int64_t)totals[0];
            const ChunkHook h = *hook;
            const int dev = ctx->device;
            const int64_t l0h = (int64_t)(res->la.size() - totals[0]), chunk_no = nchunk_done - 1;
            const bool best = want_best != 0;
            const hipEvent_t copied_h = copied;
            auto make_hook = [&tasks, h, p, cnt, copied_h, dev, l0h, chunk_no, best, near_ppm]() {
                tasks.v.emplace_back([h, p, cnt, copied_h, dev, l0h, chunk_no, best, near_ppm] {
                    (void)hipSetDevice(dev);
                    const auto t0 = std::chrono::steady_clock::now();
                    (void)hipEventSynchronize(copied_h);  // the records of this chunk have arrived
                    const auto t1 = std::chrono::steady_clock::now();
                    if (best) select_best_range(p, (size_t)cnt, near_ppm);  // chain flags: a per-read decision too
                    const auto t2 = std::chrono::steady_clock::now();
                    h(p, cnt, l0h, chunk_no);
                    if (getenv("DH_TRACE"))
                        fprintf(stderr, "[chunk hook %lld] %lld records: wait %.2f chains %.2f filters + candidates %.2f ms\n",
                                (long long)chunk_no, (long long)cnt, std::chrono::duration<double, std::milli>(t1 - t0).count(),
                                std::chrono::duration<double, std::milli>(t2 - t1).count(),
                                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t2).count());
                });
            };
            if (defer_copy)  // (the event the hook waits for is recorded when the copy is issued: both wait for the next chunk)
                deferred = [enqueue_copy, make_hook]() -> int {
                    if (int rc = enqueue_copy()) return rc;
                    make_hook();
                    return DH_OK;
                };
            else
                make_hook();
        } else if (defer_copy)
            deferred = enqueue_copy;
        float t;
        HIPCHK(hipEventElapsedTime(&t, ctx->ev[2], ctx->ev[3]));
        ms_seed += t;
        HIPCHK(hipEventElapsedTime(&t, ctx->ev[3], ctx->ev[4]));
        ms_wave += t;
        HIPCHK(hipEventElapsedTime(&t, ctx->ev[4], ctx->ev[5]));
        ms_gather += t;
    }
#undef SCR
    w_loop = now_ms() - w_a;
    w_a = now_ms();
    unsigned long long counters[2] = {0, 0};
    HIPCHK(hipMemcpy(counters, d_counters, sizeof(counters), hipMemcpyDeviceToHost));
    stats.wave_cells = (int64_t)counters[0];
    stats.alignments = (int64_t)counters[1];

    tasks.join();
    const double tail_hooks = tasks.ms_hooks, tail_copies = tasks.ms_copies;
    if (want_best && !hook) select_best_range(res->la.data(), res->la.size(), near_ppm);
    if (want_sorted & 1) lasort(res, A->n);
    if (res2) {
        // chain flags of the transposed set: the same rule with the roles of the sequences exchanged (chains of a read
        // on one contig, ordered along the read); then LAsort order
        auto swap_roles = [&]() {
            for (dh_la &l : res2->la) {
                std::swap(l.aread, l.bread);
                std::swap(l.abpos, l.bbpos);
                std::swap(l.aepos, l.bepos);
            }
        };
        if (want_best) {
            swap_roles();  // grouped by read already (items are (read, strand) in order)
            select_best_range(res2->la.data(), res2->la.size(), near_ppm);
            swap_roles();
        }
        std::sort(res2->la.begin(), res2->la.end(), la_less);
    }
    w_post = now_ms() - w_a;
    stats.las = res->d_la_n > 0 ? res->d_la_n : (int64_t)res->la.size();
    float t;
    HIPCHK(hipEventElapsedTime(&t, ctx->ev[0], ctx->ev[1]));
    stats.ms_index = t;
    stats.ms_seed = ms_seed + ms_join;  // the k-mer join is seeding work
    stats.ms_wave = ms_wave;
    stats.ms_gather = ms_gather;
    stats.ms_total = stats.ms_index + ms_seed + ms_wave + ms_gather;
    ctx->stats = stats;
    {
        dh_cum_stats &c = ctx->cum;
        c.ms_index += stats.ms_index;
        c.ms_seed += stats.ms_seed;
        c.ms_wave += stats.ms_wave;
        c.ms_gather += stats.ms_gather;
        c.wave_launches += stats.wave_launches;
        c.wave_cells += stats.wave_cells;
        c.alignments += stats.alignments;
        c.las += stats.las;
        c.hits += stats.hits;
        c.b_bases += stats.b_bases;
        c.trace_values += res->d_trace_len > 0 ? res->d_trace_len : (res->d_trace_own_len > 0 ? res->d_trace_own_len : (int64_t)res->trace.size());
        std::atomic<int64_t> abp{0};
        const dh_la *lp = res->la.data();
        dh_parallel_for((int64_t)res->la.size(), 1 << 16, [&](int64_t lo, int64_t hi) {
            int64_t sum = 0;
            for (int64_t i = lo; i < hi; i++) sum += lp[i].aepos - lp[i].abpos;
            abp += sum;
        });
        c.aligned_bp += abp.load();
    }
#ifdef DH_SEED_PROF
    if (getenv("DH_TRACE")) {
        dhk_seed_prof_dump();
        dhk_join_prof_dump();
        dhk_tile_prof_dump();
    }
#endif
    if (getenv("DH_TRACE"))
        fprintf(stderr,
                "[dh_align_db] A=%d seqs/%lld bp B=%d seqs/%lld bp hits=%lld cands=%lld aln=%lld las=%lld cells=%lld | "
                "index %.2f seed %.2f (join %.2f: %lld hits) wave %.2f gather %.2f ms, wall %.2f ms (host: index %.2f loop %.2f post %.2f; "
                "loop: copies %.2f seed %.2f wave %.2f stats %.2f resize %.2f d2h %.2f; tail: hooks %.2f copies %.2f)\n",
                A->n, (long long)A->total, B->n, (long long)B->total, (long long)stats.hits, (long long)stats.cands,
                (long long)stats.alignments, (long long)stats.las, (long long)stats.wave_cells, stats.ms_index,
                stats.ms_seed, ms_join, (long long)join_hits, stats.ms_wave, stats.ms_gather,
                ((double)std::chrono::duration_cast<std::chrono::microseconds>(
                     std::chrono::steady_clock::now().time_since_epoch()).count() - wall0) / 1e3,
                w_index, w_loop, w_post, w_g[0], w_g[1], w_g[2], w_g[3], w_g[4], w_g[5], tail_hooks, tail_copies);
    guard.ok = true;
    *out = res;
    if (out_tr) {
        guard2.ok = true;
        *out_tr = res2;
    }
    return DH_OK;
}

// `damapper -C <ref> <reads>`: the mapping and, as a second set, the records of the transposed pairs (read, contig) --
// for every accepted local alignment the tiled alignment (DH-2) of A'' = the read on its forward strand against B'' = the
// contig (complemented for a reverse-strand mapping) through the same seed, accepted on its own; one pass over the reads
// (the reference's tools write <reads>.<ref>.las from the same alignments: source/dentist/dazzler.d:6158-6170,
// getLasFile :4339-4354).  opts->algo must be 1.
extern "C" int dh_align_db_transposed(dh_ctx *ctx, dh_db *A, dh_db *B, const dh_align_opts *opts, int32_t want_best,
                                      dh_la_set **out, dh_la_set **out_transposed)
{
    if (!B || !out_transposed) return fail(DH_EINVAL, "dh_align_db_transposed: NULL argument");
    return align_range(ctx, A, B, 0, B->n, opts, want_best, 1, out, nullptr, out_transposed);
}

// ------------------------------------------------------------------------------------ .las

// header int64 novl + int32 tspace; record = 40 bytes (9 x int32 + pad); trace values are u8 when
// tspace <= 125 (TRACE_XOVR) else u16 -- dazzler.d:1665-1834, 1988-2032, 2130-2170.
extern "C" int dh_las_write(const char *path, const dh_la *las, int64_t n, const uint16_t *trace,
                            int32_t tspace)
{
    if (!path || (n > 0 && (!las || !trace))) return fail(DH_EINVAL, "dh_las_write: NULL argument");
    FILE *f = fopen(path, "wb");
    if (!f) return fail(DH_EIO, std::string("cannot open ") + path);
    bool ok = fwrite(&n, 8, 1, f) == 1 && fwrite(&tspace, 4, 1, f) == 1;
    const bool small = tspace <= 125;
    std::vector<uint8_t> tmp;
    for (int64_t i = 0; ok && i < n; i++) {
        const dh_la &l = las[i];
        const int32_t rec[10] = {l.tlen, l.diffs, l.abpos, l.bbpos, l.aepos,
                                 l.bepos, (int32_t)l.flags, l.aread, l.bread, 0};
        ok = fwrite(rec, 4, 10, f) == 10;
        const uint16_t *t = trace + l.toff;
        if (small) {
            tmp.resize((size_t)l.tlen);
            for (int32_t j = 0; j < l.tlen; j++) {
                if (t[j] > 255) {
                    fclose(f);
                    return fail(DH_EINVAL, "dh_las_write: trace value exceeds 8 bits at tspace <= 125");
                }
                tmp[(size_t)j] = (uint8_t)t[j];
            }
            ok = ok && (l.tlen == 0 || fwrite(tmp.data(), 1, (size_t)l.tlen, f) == (size_t)l.tlen);
        } else
            ok = ok && (l.tlen == 0 || fwrite(t, 2, (size_t)l.tlen, f) == (size_t)l.tlen);
    }
    if (fclose(f) != 0) ok = false;
    return ok ? DH_OK : fail(DH_EIO, std::string("short write to ") + path);
}

extern "C" int dh_las_read(const char *path, dh_la_set **out)
{
    if (!path || !out) return fail(DH_EINVAL, "dh_las_read: NULL argument");
    FILE *f = fopen(path, "rb");
    if (!f) return fail(DH_EIO, std::string("cannot open ") + path);
    int64_t novl = 0;
    int32_t ts = 0;
    if (fread(&novl, 8, 1, f) != 1 || fread(&ts, 4, 1, f) != 1) {
        fclose(f);
        return fail(DH_EIO, std::string("error reading LAS file `") + path + "`: unexpected end of file");
    }
    dh_la_set *s = new dh_la_set();
    s->tspace = ts;
    const bool small = ts <= 125;
    std::vector<uint8_t> tmp;
    for (int64_t i = 0; i < novl; i++) {
        int32_t rec[10];
        if (fread(rec, 4, 10, f) != 10) {
            fclose(f);
            delete s;
            return fail(DH_EIO, std::string("error reading LAS file `") + path +
                                    "`: unexpected end of file; expected overlapHead");
        }
        dh_la l = {};
        l.tlen = rec[0];
        l.diffs = rec[1];
        l.abpos = rec[2];
        l.bbpos = rec[3];
        l.aepos = rec[4];
        l.bepos = rec[5];
        l.flags = (uint32_t)rec[6];
        l.aread = rec[7];
        l.bread = rec[8];
        l.toff = (int64_t)s->trace.size();
        if (l.tlen < 0 || l.tlen % 2) {
            fclose(f);
            delete s;
            return fail(DH_EIO, "illegal value for tlen: must be multiple of 2");
        }
        s->trace.resize(s->trace.size() + (size_t)l.tlen);
        uint16_t *t = s->trace.data() + l.toff;
        bool ok;
        if (small) {
            tmp.resize((size_t)l.tlen);
            ok = l.tlen == 0 || fread(tmp.data(), 1, (size_t)l.tlen, f) == (size_t)l.tlen;
            for (int32_t j = 0; ok && j < l.tlen; j++) t[j] = tmp[(size_t)j];
        } else
            ok = l.tlen == 0 || fread(t, 2, (size_t)l.tlen, f) == (size_t)l.tlen;
        if (!ok) {
            fclose(f);
            delete s;
            return fail(DH_EIO, std::string("error reading LAS file `") + path +
                                    "`: unexpected end of file; expected tracePoints");
        }
        s->la.push_back(l);
    }
    fclose(f);
    *out = s;
    return DH_OK;
}

// LAmerge in memory: the result sets of the read blocks (dh_align_db_block) merged into one set in
// LAsort order; traces are concatenated in set order and every record's toff is rebased.
extern "C" int dh_la_set_merge(const dh_la_set *const *sets, int32_t nsets, dh_la_set **out)
{
    if (!sets || nsets < 1 || !out) return fail(DH_EINVAL, "dh_la_set_merge: bad argument");
    int32_t tspace = -1;
    size_t nla = 0, ntr = 0;
    for (int32_t i = 0; i < nsets; i++) {
        if (!sets[i]) return fail(DH_EINVAL, "dh_la_set_merge: NULL set");
        if (int rc = dh_la_set_ensure_host_trace(const_cast<dh_la_set *>(sets[i]))) return rc;
        if (!sets[i]->la.empty()) {
            if (tspace >= 0 && sets[i]->tspace != tspace)
                return fail(DH_EINVAL, "dh_la_set_merge: sets with different trace spacing");
            tspace = sets[i]->tspace;
        }
        nla += sets[i]->la.size();
        ntr += sets[i]->trace.size();
    }
    dh_la_set *res = new dh_la_set();
    res->tspace = tspace >= 0 ? tspace : sets[0]->tspace;
    res->la.resize(nla);
    res->trace.resize(ntr);
    size_t l0 = 0, t0 = 0;
    int32_t na = 0;
    for (int32_t i = 0; i < nsets; i++) {
        const dh_la_set *x = sets[i];
        if (!x->trace.empty()) memcpy(res->trace.data() + t0, x->trace.data(), sizeof(uint16_t) * x->trace.size());
        for (size_t j = 0; j < x->la.size(); j++) {
            dh_la l = x->la[j];
            l.toff += (int64_t)t0;
            na = std::max(na, l.aread + 1);
            res->la[l0 + j] = l;
        }
        l0 += x->la.size();
        t0 += x->trace.size();
    }
    // every input is in LAsort order: a stable sort of the concatenation is the merge
    std::stable_sort(res->la.begin(), res->la.end(), la_less);
    *out = res;
    return DH_OK;
}

// LAmerge (workflow rule snakemake/Snakefile:1173-1185): the alignment files of the read blocks
// (one per GPU / per block) merged into one file in LAsort order.  Host only.
extern "C" int dh_las_merge(const char *const *paths, int32_t npaths, const char *out_path)
{
    if (!paths || npaths < 1 || !out_path) return fail(DH_EINVAL, "dh_las_merge: bad argument");
    std::vector<dh_la_set *> sets((size_t)npaths, nullptr);
    struct Guard {
        std::vector<dh_la_set *> &s;
        ~Guard()
        {
            for (dh_la_set *x : s) dh_la_set_destroy(x);
        }
    } guard{sets};
    int32_t tspace = -1;
    size_t total = 0;
    for (int32_t i = 0; i < npaths; i++) {
        if (int rc = dh_las_read(paths[i], &sets[(size_t)i])) return rc;
        if (tspace >= 0 && sets[(size_t)i]->tspace != tspace && !sets[(size_t)i]->la.empty())
            return fail(DH_EINVAL, "dh_las_merge: files with different trace spacing");
        if (!sets[(size_t)i]->la.empty() || tspace < 0) tspace = sets[(size_t)i]->tspace;
        total += sets[(size_t)i]->la.size();
    }
    // records of all files with the index of their file; traces stay in their sets
    std::vector<std::pair<dh_la, int32_t>> all;
    all.reserve(total);
    for (int32_t i = 0; i < npaths; i++)
        for (const dh_la &l : sets[(size_t)i]->la) all.emplace_back(l, i);
    std::stable_sort(all.begin(), all.end(),
                     [](const std::pair<dh_la, int32_t> &x, const std::pair<dh_la, int32_t> &y) { return la_less(x.first, y.first); });
    std::vector<dh_la> las(all.size());
    std::vector<uint16_t> trace;
    for (size_t i = 0; i < all.size(); i++) {
        dh_la l = all[i].first;
        const uint16_t *t = sets[(size_t)all[i].second]->trace.data() + l.toff;
        l.toff = (int64_t)trace.size();
        trace.insert(trace.end(), t, t + l.tlen);
        las[i] = l;
    }
    static const uint16_t none = 0;
    return dh_las_write(out_path, las.data(), (int64_t)las.size(), trace.empty() ? &none : trace.data(), tspace);
}
