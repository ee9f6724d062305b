// dh_comm.cpp -- the multi-GPU entry of the C ABI: a communicator over RCCL (one process per GPU, xGMI) and
// dh_shard_run, the whole `collect` + `process` of one rank's share of the reads with its three exchanges.
//
// What it replaces in the reference: the file system between the jobs of the workflow -- `LAmerge` of the per-block
// mappings (snakemake/Snakefile:1173-1185), `dentist process --batch` jobs (:1315-1358) and `dentist merge-insertions`
// (commands/mergeInsertions.d:60-164).  A D host binds these entry points like every other one: rank 0
// calls dh_comm_unique_id, hands the 128 bytes to the other processes by whatever means it has (a file, MPI, a socket),
// every process calls dh_comm_create and then dh_shard_run per batch.
//
// Exchanges (payloads of configs[2] at N = 8: 2.5 MB of joins, 21 MB of cropped reads, 0.3 MB of closed gaps per rank):
//   all-gather(v)   int64 sizes (ncclAllGather), then the padded blobs (ncclAllGather) -- joins / candidates, closed gaps
//   all-to-all(v)   int64 sizes (ncclAllGather of the send-size rows), then grouped ncclSend / ncclRecv -- cropped reads
// Host blobs are staged through page-locked memory into device buffers of the context's arena; the collectives run on
// the context's stream.  RCCL is loaded with dlopen on first use: the library itself has no link-time dependency on it
// (the .las / DB tools and the CPU-side tests load libdentist_hip.so on machines without RCCL).
//
// A second back end, the LOCAL hub, serves the same calls between threads of one process (tests and the N-rank
// emulation on one GPU: every "rank" is a host thread with its own context): dh_shard_run is the same code over both.
#include <dlfcn.h>

#include <condition_variable>
#include <cstring>
#include <mutex>
#include <numeric>

#include "dh_internal.h"

#include <rccl/rccl.h>  // types and prototypes only; the symbols are resolved with dlsym

namespace {

struct Rccl {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err;
    bool load()
    {
        if (lib) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) {
            err = std::string("cannot load librccl: ") + (dlerror() ? dlerror() : "?");
            return false;
        }
#define SYM(f)                                                                   \
    f = (decltype(f))dlsym(lib, "nccl" #f);                                      \
    if (!f) {                                                                    \
        err = "librccl lacks nccl" #f;                                           \
        return false;                                                            \
    }
        SYM(GetUniqueId) SYM(CommInitRank) SYM(CommDestroy) SYM(AllGather) SYM(GroupStart) SYM(GroupEnd) SYM(Send) SYM(Recv)
        SYM(GetErrorString) SYM(CommAbort)
#undef SYM
        return true;
    }
};
Rccl &rccl()
{
    static Rccl r;
    return r;
}
std::mutex g_rccl_mu;

// threads of one process standing in for ranks: a barrier and one slot per rank
struct LocalHub {
    int32_t world;
    std::mutex mu;
    std::condition_variable cv;
    int32_t arrived = 0;
    int64_t gen = 0;
    std::vector<const uint8_t *> ptr;                     // all-gather: payload of rank r
    std::vector<int64_t> size;
    std::vector<const uint8_t *const *> dest_ptr;         // all-to-all: per-destination payloads of rank r
    std::vector<const int64_t *> dest_size;
    int32_t refs = 0;
    explicit LocalHub(int32_t w) : world(w), ptr((size_t)w), size((size_t)w), dest_ptr((size_t)w), dest_size((size_t)w) {}
    void barrier()
    {
        std::unique_lock<std::mutex> lk(mu);
        const int64_t g = gen;
        if (++arrived == world) {
            arrived = 0;
            gen++;
            cv.notify_all();
        } else
            cv.wait(lk, [&] { return gen != g; });
    }
};

}  // namespace

struct dh_comm {
    int32_t rank = 0, world = 1;
    dh_ctx *ctx = nullptr;
    ncclComm_t nccl = nullptr;
    LocalHub *hub = nullptr;
    // RCCL back end: the buffers of the size / status exchanges, made with the communicator -- nothing is allocated
    // between deciding to enter an exchange and the collective itself
    int64_t *d_sz = nullptr;   // device: world * (world + 1) + 2 * world words
    int64_t *h_sz = nullptr;   // page-locked: the same
    bool aborted = false;      // a collective failed and the communicator was aborted: every later call fails at once
};

#define NCCLCHK(expr)                                                                                          \
    do {                                                                                                       \
        ncclResult_t r_ = (expr);                                                                              \
        if (r_ != ncclSuccess) return dh_fail(DH_EHIP, std::string(#expr) + ": " + rccl().GetErrorString(r_)); \
    } while (0)

extern "C" int dh_comm_unique_id(uint8_t *id128)
{
    if (!id128) return dh_fail(DH_EINVAL, "dh_comm_unique_id: NULL argument");
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (!rccl().load()) return dh_fail(DH_ENODEV, rccl().err);
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    NCCLCHK(rccl().GetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return DH_OK;
}

extern "C" int dh_comm_create(const uint8_t *id128, int32_t rank, int32_t world, dh_ctx *ctx, dh_comm **out)
{
    if (!id128 || !ctx || !out || world < 1 || rank < 0 || rank >= world) return dh_fail(DH_EINVAL, "dh_comm_create: bad argument");
    {
        std::lock_guard<std::mutex> lk(g_rccl_mu);
        if (!rccl().load()) return dh_fail(DH_ENODEV, rccl().err);
    }
    HIPCHK(hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    dh_comm *c = new dh_comm();
    c->rank = rank;
    c->world = world;
    c->ctx = ctx;
    const size_t nsz = sizeof(int64_t) * ((size_t)world * (size_t)(world + 1) + 2 * (size_t)world);
    if (hipMalloc((void **)&c->d_sz, nsz) != hipSuccess || hipHostMalloc((void **)&c->h_sz, nsz, hipHostMallocDefault) != hipSuccess) {
        if (c->d_sz) (void)hipFree(c->d_sz);
        delete c;
        return dh_fail(DH_ENOMEM, "dh_comm_create: no memory for the size exchanges");
    }
    const ncclResult_t r = rccl().CommInitRank(&c->nccl, world, id, rank);
    if (r != ncclSuccess) {
        (void)hipFree(c->d_sz);
        (void)hipHostFree(c->h_sz);
        delete c;
        return dh_fail(DH_EHIP, std::string("ncclCommInitRank: ") + rccl().GetErrorString(r));
    }
    *out = c;
    return DH_OK;
}

// `world` communicators of one process that talk through memory (one per host thread / context)
extern "C" int dh_comm_create_local(int32_t world, dh_ctx *const *ctxs, dh_comm **out)
{
    if (world < 1 || !out) return dh_fail(DH_EINVAL, "dh_comm_create_local: bad argument");  // (ctxs may be NULL: collectives only)
    LocalHub *hub = new LocalHub(world);
    hub->refs = world;
    for (int32_t r = 0; r < world; r++) {
        dh_comm *c = new dh_comm();
        c->rank = r;
        c->world = world;
        c->ctx = ctxs ? ctxs[r] : nullptr;
        c->hub = hub;
        out[r] = c;
    }
    return DH_OK;
}

extern "C" void dh_comm_destroy(dh_comm *c)
{
    if (!c) return;
    if (c->nccl) (void)rccl().CommDestroy(c->nccl);
    if (c->d_sz) (void)hipFree(c->d_sz);
    if (c->h_sz) (void)hipHostFree(c->h_sz);
    if (c->hub) {
        bool last;
        {
            std::lock_guard<std::mutex> lk(c->hub->mu);
            last = --c->hub->refs == 0;
        }
        if (last) delete c->hub;
    }
    delete c;
}
extern "C" int32_t dh_comm_rank(const dh_comm *c) { return c ? c->rank : -1; }
extern "C" int32_t dh_comm_world(const dh_comm *c) { return c ? c->world : 0; }

// all-gather(v) of one blob per rank.  *out: one malloc'd block holding the blobs in rank order (dh_shard_free),
// sizes[world] their lengths.
// A rank that failed locally before an exchange must not leave its peers waiting in it: every exchange starts with the
// all-gather of the sizes, and a NEGATIVE size says "this rank failed with status -size".  Every rank then returns an
// error from the same exchange (its own status, or DH_EPEER-like DH_EINVAL naming the first failed rank) and none
// enters the payload collective.  local_rc = the status of what the rank did since the last exchange.
static int peers_failed(const int64_t *sizes, int32_t W, int32_t rank, int local_rc, const char *who)
{
    for (int32_t r = 0; r < W; r++)
        if (sizes[r] < 0) {
            if (local_rc) return local_rc;  // (the message of the local failure is already set)
            return dh_fail(DH_EINVAL, std::string(who) + ": rank " + std::to_string(r) + " failed before the exchange (status " +
                                          std::to_string(-sizes[r]) + "); rank " + std::to_string(rank) + " gives up with it");
        }
    return DH_OK;
}


// ---- RCCL back end: how a rank may fail without stranding its peers.
//   * BETWEEN two exchanges (anything the rank does on its own): the status travels with the next exchange's sizes
//     (negative size), every rank returns from that exchange (peers_failed above).
//   * between the SIZE exchange and the PAYLOAD collective (the buffers are sized by what arrived: scratch, page-locked
//     staging, the result block): everything is allocated first, then ONE more word per rank is exchanged -- 0 or the
//     status -- and only if every rank holds its buffers does anybody enter the payload collective (status_exchange).
//   * INSIDE a collective (an RCCL or HIP call fails: a link or the device is gone): the rank aborts the communicator
//     (ncclCommAbort) so that the peers' pending collectives end with an error instead of waiting forever; the
//     communicator is dead from then on (every later call fails at once, on every rank that learns of it the same way).
static int comm_abort(dh_comm *c, int rc)
{
    if (c->nccl) {
        (void)rccl().CommAbort(c->nccl);
        c->nccl = nullptr;
    }
    c->aborted = true;
    return rc;
}
#define COLL_HIP(expr)                                                                                                   \
    do {                                                                                                                 \
        hipError_t e_ = (expr);                                                                                          \
        if (e_ != hipSuccess)                                                                                            \
            return comm_abort(c, dh_fail(DH_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_) + " (communicator aborted)")); \
    } while (0)
#define COLL_NCCL(expr)                                                                                                  \
    do {                                                                                                                 \
        ncclResult_t r_ = (expr);                                                                                        \
        if (r_ != ncclSuccess)                                                                                           \
            return comm_abort(c, dh_fail(DH_EHIP, std::string(#expr) + ": " + rccl().GetErrorString(r_) + " (communicator aborted)")); \
    } while (0)
static int comm_alive(dh_comm *c, const char *who)
{
    if (c->aborted || !c->nccl) return dh_fail(DH_EHIP, std::string(who) + ": the communicator was aborted after a failed collective");
    if (!c->ctx) return dh_fail(DH_EINVAL, std::string(who) + ": the communicator has no context");
    return DH_OK;
}
// all-gather of `n` words per rank through the communicator's own buffers: out[r * n + i] = word i of rank r
static int words_exchange(dh_comm *c, const int64_t *mine, size_t n, int64_t *out)
{
    const size_t W = (size_t)c->world;
    hipStream_t st = c->ctx->stream;
    int64_t *d_all = c->d_sz, *d_mine = c->d_sz + W * W, *h_all = c->h_sz, *h_mine = c->h_sz + W * W;
    memcpy(h_mine, mine, sizeof(int64_t) * n);
    COLL_HIP(hipMemcpyAsync(d_mine, h_mine, sizeof(int64_t) * n, hipMemcpyHostToDevice, st));
    COLL_NCCL(rccl().AllGather(d_mine, d_all, n, ncclInt64, c->nccl, st));
    COLL_HIP(hipMemcpyAsync(h_all, d_all, sizeof(int64_t) * n * W, hipMemcpyDeviceToHost, st));
    COLL_HIP(hipStreamSynchronize(st));
    memcpy(out, h_all, sizeof(int64_t) * n * W);
    return DH_OK;
}
// the second status word: every rank says whether it holds the buffers of the payload collective
static int status_exchange(dh_comm *c, int local_rc, const char *who)
{
    const int32_t W = c->world;
    const int64_t mine = local_rc ? -(int64_t)std::max(1, local_rc > 0 ? local_rc : -local_rc) : 0;
    std::vector<int64_t> all((size_t)W);
    if (int rc = words_exchange(c, &mine, 1, all.data())) return rc;
    for (int32_t r = 0; r < W; r++)
        if (all[(size_t)r] < 0) {
            if (local_rc) return local_rc;
            return dh_fail(DH_EINVAL, std::string(who) + ": rank " + std::to_string(r) + " could not stage its payload (status " +
                                          std::to_string(-all[(size_t)r]) + "); rank " + std::to_string(c->rank) + " gives up with it");
        }
    return DH_OK;
}

static int all_gather_st(dh_comm *c, const uint8_t *payload, int64_t nbytes, int local_rc, uint8_t **out, int64_t *sizes)
{
    if (!c || !out || !sizes || nbytes < 0 || (nbytes > 0 && !payload)) {
        if (!c || !sizes || !out) return dh_fail(DH_EINVAL, "dh_comm_all_gather: bad argument");
        if (!local_rc) local_rc = dh_fail(DH_EINVAL, "dh_comm_all_gather: bad argument");
    }
    if (local_rc) nbytes = -(int64_t)(local_rc > 0 ? local_rc : -local_rc);
    if (local_rc && nbytes == 0) nbytes = -1;
    const int32_t W = c->world;
    *out = nullptr;
    if (c->hub) {
        LocalHub &h = *c->hub;
        h.ptr[(size_t)c->rank] = payload;
        h.size[(size_t)c->rank] = nbytes;
        h.barrier();
        int64_t total = 0;
        bool failed = false;
        for (int32_t r = 0; r < W; r++) {
            sizes[r] = h.size[(size_t)r];
            failed = failed || sizes[r] < 0;
            total += std::max<int64_t>(sizes[r], 0);
        }
        uint8_t *buf = failed ? nullptr : (uint8_t *)malloc((size_t)std::max<int64_t>(total, 1));
        if (buf) {
            int64_t at = 0;
            for (int32_t r = 0; r < W; r++) {
                if (sizes[r]) memcpy(buf + at, h.ptr[(size_t)r], (size_t)sizes[r]);
                at += sizes[r];
            }
        }
        h.barrier();  // every rank has copied (or given up): the payloads may go -- reached on every path
        if (failed) return peers_failed(sizes, W, c->rank, local_rc, "dh_comm_all_gather");
        if (!buf) return dh_fail(DH_ENOMEM, "dh_comm_all_gather: out of memory");
        *out = buf;
        return DH_OK;
    }
    if (int rc = comm_alive(c, "dh_comm_all_gather")) return rc;
    dh_ctx *ctx = c->ctx;
    COLL_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    // (1) sizes, a negative one = "this rank failed with status -size"
    if (int rc = words_exchange(c, &nbytes, 1, sizes)) return rc;
    if (int rc = peers_failed(sizes, W, c->rank, local_rc, "dh_comm_all_gather")) return rc;
    int64_t cap = 1, total = 0;
    for (int32_t r = 0; r < W; r++) {
        cap = std::max(cap, sizes[r]);
        total += sizes[r];
    }
    cap = (cap + 15) & ~15ll;
    // (2) every buffer the payload collective needs, then the second status word: nobody enters the payload collective
    // unless every rank holds them.  Page-locked staging both ways (the blobs the glue hands over are plain malloc'd memory)
    uint8_t *d_send = nullptr, *d_recv = nullptr;
    const size_t nstage = (size_t)std::max<int64_t>(std::max(nbytes, total), 1);
    uint8_t *stage = nullptr, *buf = nullptr;
    int rc2 = dh_scratch(ctx, 56, (size_t)cap, (void **)&d_send);
    if (!rc2) rc2 = dh_scratch(ctx, 57, (size_t)cap * (size_t)W, (void **)&d_recv);
    if (!rc2 && !(stage = (uint8_t *)dh_pinned_alloc(nstage))) rc2 = dh_fail(DH_ENOMEM, "dh_comm_all_gather: out of page-locked memory");
    if (!rc2 && !(buf = (uint8_t *)malloc((size_t)std::max<int64_t>(total, 1)))) rc2 = dh_fail(DH_ENOMEM, "dh_comm_all_gather: out of memory");
    struct Staged {
        uint8_t *p, *b;
        size_t n;
        ~Staged()
        {
            if (p) dh_pinned_free(p, n);
            free(b);
        }
    } staged{stage, buf, nstage};
    if (int rc = status_exchange(c, rc2, "dh_comm_all_gather")) return rc;
    // (3) the payload; a failure in here aborts the communicator
    if (nbytes) {
        memcpy(stage, payload, (size_t)nbytes);
        COLL_HIP(hipMemcpyAsync(d_send, stage, (size_t)nbytes, hipMemcpyHostToDevice, st));
    }
    COLL_NCCL(rccl().AllGather(d_send, d_recv, (size_t)cap, ncclUint8, c->nccl, st));
    COLL_HIP(hipStreamSynchronize(st));  // (the staging buffer is reused for the way back)
    int64_t at = 0;
    for (int32_t r = 0; r < W; r++) {
        if (sizes[r]) COLL_HIP(hipMemcpyAsync(stage + at, d_recv + (size_t)r * (size_t)cap, (size_t)sizes[r], hipMemcpyDeviceToHost, st));
        at += sizes[r];
    }
    COLL_HIP(hipStreamSynchronize(st));
    if (total) memcpy(buf, stage, (size_t)total);
    *out = buf;
    staged.b = nullptr;
    return DH_OK;
}
extern "C" int dh_comm_all_gather(dh_comm *c, const uint8_t *payload, int64_t nbytes, uint8_t **out, int64_t *sizes)
{
    if (!c || !out || !sizes || nbytes < 0 || (nbytes > 0 && !payload)) return dh_fail(DH_EINVAL, "dh_comm_all_gather: bad argument");
    return all_gather_st(c, payload, nbytes, DH_OK, out, sizes);
}

// all-to-all(v): per_dest[r] / send_sizes[r] = the blob for rank r.  *out: one malloc'd block with the blobs received in
// source-rank order (dh_shard_free), recv_sizes[world] their lengths.
static int all_to_all_st(dh_comm *c, const uint8_t *const *per_dest, const int64_t *send_sizes_in, int local_rc, uint8_t **out,
                         int64_t *recv_sizes)
{
    if (!c || !out || !recv_sizes) return dh_fail(DH_EINVAL, "dh_comm_all_to_all: NULL argument");
    const int32_t W = c->world;
    if (!local_rc && (!per_dest || !send_sizes_in)) local_rc = dh_fail(DH_EINVAL, "dh_comm_all_to_all: NULL argument");
    if (!local_rc)
        for (int32_t r = 0; r < W; r++)
            if (send_sizes_in[r] < 0) local_rc = dh_fail(DH_EINVAL, "dh_comm_all_to_all: negative size");
    // a failed rank sends its status as a negative size to everybody
    std::vector<int64_t> fail_sizes;
    const int64_t *send_sizes = send_sizes_in;
    if (local_rc) {
        fail_sizes.assign((size_t)W, -(int64_t)std::max(1, local_rc > 0 ? local_rc : -local_rc));
        send_sizes = fail_sizes.data();
    }
    *out = nullptr;
    if (c->hub) {
        LocalHub &h = *c->hub;
        h.dest_ptr[(size_t)c->rank] = per_dest;
        h.dest_size[(size_t)c->rank] = send_sizes;
        h.barrier();
        int64_t total = 0;
        bool failed = false;
        for (int32_t r = 0; r < W; r++) {
            recv_sizes[r] = h.dest_size[(size_t)r][c->rank];
            failed = failed || recv_sizes[r] < 0;
            total += std::max<int64_t>(recv_sizes[r], 0);
        }
        uint8_t *buf = failed ? nullptr : (uint8_t *)malloc((size_t)std::max<int64_t>(total, 1));
        if (buf) {
            int64_t at = 0;
            for (int32_t r = 0; r < W; r++) {
                if (recv_sizes[r]) memcpy(buf + at, h.dest_ptr[(size_t)r][c->rank], (size_t)recv_sizes[r]);
                at += recv_sizes[r];
            }
        }
        h.barrier();  // reached on every path
        if (failed) return peers_failed(recv_sizes, W, c->rank, local_rc, "dh_comm_all_to_all");
        if (!buf) return dh_fail(DH_ENOMEM, "dh_comm_all_to_all: out of memory");
        *out = buf;
        return DH_OK;
    }
    if (int rc = comm_alive(c, "dh_comm_all_to_all")) return rc;
    dh_ctx *ctx = c->ctx;
    COLL_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    // (1) every rank's row of send sizes: recv_sizes[r] = row r, column rank (a negative row = that rank's status)
    std::vector<int64_t> rows((size_t)W * (size_t)W);
    if (int rc = words_exchange(c, send_sizes, (size_t)W, rows.data())) return rc;
    int64_t stot = 0, rtot = 0;
    for (int32_t r = 0; r < W; r++) recv_sizes[r] = rows[(size_t)r * W + c->rank];
    if (int rc = peers_failed(recv_sizes, W, c->rank, local_rc, "dh_comm_all_to_all")) return rc;
    for (int32_t r = 0; r < W; r++) {
        stot += send_sizes[r];
        rtot += recv_sizes[r];
    }
    // (2) the buffers, then the second status word
    uint8_t *d_send = nullptr, *d_recv = nullptr, *stage = nullptr, *buf = nullptr;
    const size_t nstage = (size_t)std::max<int64_t>(std::max(stot, rtot), 1);
    int rc2 = dh_scratch(ctx, 56, (size_t)std::max<int64_t>(stot, 16), (void **)&d_send);
    if (!rc2) rc2 = dh_scratch(ctx, 57, (size_t)std::max<int64_t>(rtot, 16), (void **)&d_recv);
    if (!rc2 && !(stage = (uint8_t *)dh_pinned_alloc(nstage))) rc2 = dh_fail(DH_ENOMEM, "dh_comm_all_to_all: out of page-locked memory");
    if (!rc2 && !(buf = (uint8_t *)malloc((size_t)std::max<int64_t>(rtot, 1)))) rc2 = dh_fail(DH_ENOMEM, "dh_comm_all_to_all: out of memory");
    struct Staged {
        uint8_t *p, *b;
        size_t n;
        ~Staged()
        {
            if (p) dh_pinned_free(p, n);
            free(b);
        }
    } staged{stage, buf, nstage};
    if (int rc = status_exchange(c, rc2, "dh_comm_all_to_all")) return rc;
    // (3) the payload; a failure in here aborts the communicator
    int64_t at = 0;
    for (int32_t r = 0; r < W; r++) {
        if (send_sizes[r]) memcpy(stage + at, per_dest[r], (size_t)send_sizes[r]);
        at += send_sizes[r];
    }
    if (stot) COLL_HIP(hipMemcpyAsync(d_send, stage, (size_t)stot, hipMemcpyHostToDevice, st));
    COLL_NCCL(rccl().GroupStart());
    int64_t so = 0, ro = 0;
    ncclResult_t grc = ncclSuccess;  // an error inside the group still closes it
    for (int32_t r = 0; r < W && grc == ncclSuccess; r++) {
        if (send_sizes[r]) grc = rccl().Send(d_send + so, (size_t)send_sizes[r], ncclUint8, r, c->nccl, st);
        if (grc == ncclSuccess && recv_sizes[r]) grc = rccl().Recv(d_recv + ro, (size_t)recv_sizes[r], ncclUint8, r, c->nccl, st);
        so += send_sizes[r];
        ro += recv_sizes[r];
    }
    {
        const ncclResult_t erc = rccl().GroupEnd();
        COLL_NCCL(grc);
        COLL_NCCL(erc);
    }
    COLL_HIP(hipStreamSynchronize(st));
    if (rtot) COLL_HIP(hipMemcpyAsync(stage, d_recv, (size_t)rtot, hipMemcpyDeviceToHost, st));
    COLL_HIP(hipStreamSynchronize(st));
    if (rtot) memcpy(buf, stage, (size_t)rtot);
    *out = buf;
    staged.b = nullptr;
    return DH_OK;
}
extern "C" int dh_comm_all_to_all(dh_comm *c, const uint8_t *const *per_dest, const int64_t *send_sizes, uint8_t **out,
                                  int64_t *recv_sizes)
{
    if (!c || !per_dest || !send_sizes || !out || !recv_sizes) return dh_fail(DH_EINVAL, "dh_comm_all_to_all: NULL argument");
    return all_to_all_st(c, per_dest, send_sizes, DH_OK, out, recv_sizes);
}

// ------------------------------------------------------------------------------------ the sharded collect + process

namespace {
struct FreeGuard {
    void *p = nullptr;
    ~FreeGuard() { free(p); }
};
// blob pointers of a gathered block
std::vector<const uint8_t *> blob_ptrs(const uint8_t *block, const int64_t *sizes, int32_t W)
{
    std::vector<const uint8_t *> p((size_t)W);
    int64_t at = 0;
    for (int32_t r = 0; r < W; r++) {
        p[(size_t)r] = block + at;
        at += sizes[r];
    }
    return p;
}

// the closed gaps of one rank as a blob, and the merge of all ranks' blobs in PILE-UP order: a rank's records are its
// owned pile-ups in ascending pile-up index (dh_shard_unpack_cropped), owners are assigned by cost (not contiguously), so
// the k-th record of rank r is the k-th pile-up with owner == r.  The result is then what dh_process_pileups returns on
// one GPU -- the same order whatever the world size, also when several pile-ups share contig_left (extension joins,
// contig-skipping or anti-parallel joins) -- with the flank overlaps and read ids re-based as dh_process_pileups does
// when it appends its parts.
struct ClosedHead {
    int64_t nrec, nbases, nflank, ntr, nids, has_ids;
};
void pack_closed(const dh_insertions &L, std::vector<uint8_t> &blob)
{
    ClosedHead h;
    h.nrec = (int64_t)L.rec.size();
    h.nbases = (int64_t)L.bases.size();
    h.nflank = (int64_t)L.flank.size();
    h.ntr = (int64_t)L.flank_tr.size();
    h.has_ids = L.ids_off.size() == L.rec.size() + 1 ? 1 : 0;
    h.nids = h.has_ids ? (int64_t)L.ids.size() : 0;
    const size_t bytes = sizeof(h) + sizeof(dh_insertion) * L.rec.size() + 4 * L.rec.size() + (h.has_ids ? 4 * (L.rec.size() + 1) : 0) +
                         sizeof(dh_la) * L.flank.size() + 2 * L.flank_tr.size() + 4 * (size_t)h.nids + L.bases.size();
    blob.resize(bytes);
    uint8_t *p = blob.data();
    auto put = [&](const void *src, size_t nb) {
        if (nb) memcpy(p, src, nb);
        p += nb;
    };
    put(&h, sizeof(h));
    put(L.rec.data(), sizeof(dh_insertion) * L.rec.size());
    std::vector<int32_t> fo(L.rec.size(), -1);
    for (size_t i = 0; i < L.rec.size() && i < L.flank_of.size(); i++) fo[i] = L.flank_of[i];
    put(fo.data(), 4 * fo.size());
    if (h.has_ids) put(L.ids_off.data(), 4 * L.ids_off.size());
    put(L.flank.data(), sizeof(dh_la) * L.flank.size());
    put(L.flank_tr.data(), 2 * L.flank_tr.size());
    if (h.nids) put(L.ids.data(), 4 * (size_t)h.nids);
    put(L.bases.data(), L.bases.size());
}
int merge_closed(const std::vector<const uint8_t *> &bp, const std::vector<int64_t> &sizes, int32_t W, const int32_t *owner, int32_t npiles,
                 dh_insertions &res)
{
    struct Part {
        ClosedHead h;
        const dh_insertion *rec;
        const int32_t *flank_of, *ids_off, *ids;
        const dh_la *flank;
        const uint16_t *tr;
        const uint8_t *bases;
        int64_t next = 0;
    };
    std::vector<Part> part((size_t)W);
    bool all_ids = true;
    for (int32_t r = 0; r < W; r++) {
        Part &q = part[(size_t)r];
        if (sizes[(size_t)r] < (int64_t)sizeof(ClosedHead)) return dh_fail(DH_EINVAL, "dh_shard_run: short closed-gap blob");
        memcpy(&q.h, bp[(size_t)r], sizeof(ClosedHead));
        const ClosedHead &h = q.h;
        const int64_t lim = sizes[(size_t)r];
        if (h.nrec < 0 || h.nbases < 0 || h.nflank < 0 || h.ntr < 0 || h.nids < 0 || (h.has_ids != 0 && h.has_ids != 1) || h.nrec > lim ||
            h.nbases > lim || h.nflank > lim || h.ntr > lim || h.nids > lim ||
            (int64_t)sizeof(ClosedHead) + (int64_t)sizeof(dh_insertion) * h.nrec + 4 * h.nrec + (h.has_ids ? 4 * (h.nrec + 1) : 0) +
                    (int64_t)sizeof(dh_la) * h.nflank + 2 * h.ntr + 4 * h.nids + h.nbases != lim)
            return dh_fail(DH_EINVAL, "dh_shard_run: malformed closed-gap blob");
        const uint8_t *p = bp[(size_t)r] + sizeof(ClosedHead);
        q.rec = (const dh_insertion *)p;
        p += sizeof(dh_insertion) * (size_t)h.nrec;
        q.flank_of = (const int32_t *)p;
        p += 4 * (size_t)h.nrec;
        q.ids_off = h.has_ids ? (const int32_t *)p : nullptr;
        p += h.has_ids ? 4 * (size_t)(h.nrec + 1) : 0;
        q.flank = (const dh_la *)p;
        p += sizeof(dh_la) * (size_t)h.nflank;
        q.tr = (const uint16_t *)p;
        p += 2 * (size_t)h.ntr;
        q.ids = (const int32_t *)p;
        p += 4 * (size_t)h.nids;
        q.bases = p;
        all_ids = all_ids && h.has_ids;
    }
    std::vector<int64_t> owned((size_t)W, 0);
    for (int32_t g = 0; g < npiles; g++) {
        if (owner[g] < 0 || owner[g] >= W) return dh_fail(DH_EINVAL, "dh_shard_run: owner out of range");
        owned[(size_t)owner[g]]++;
    }
    for (int32_t r = 0; r < W; r++)
        if (part[(size_t)r].h.nrec != owned[(size_t)r])
            return dh_fail(DH_EINVAL, "dh_shard_run: rank " + std::to_string(r) + " returned " + std::to_string(part[(size_t)r].h.nrec) +
                                          " closed-gap records for " + std::to_string(owned[(size_t)r]) + " owned pile-ups");
    res.rec.reserve((size_t)npiles);
    res.flank_of.reserve((size_t)npiles);
    if (all_ids) res.ids_off.assign(1, 0);
    for (int32_t g = 0; g < npiles; g++) {
        Part &q = part[(size_t)owner[g]];
        const int64_t i = q.next++;
        dh_insertion x;
        memcpy(&x, q.rec + i, sizeof(x));
        if (x.cons_off < 0 || x.cons_len < 0 || x.cons_off + x.cons_len > q.h.nbases) return dh_fail(DH_EINVAL, "dh_shard_run: malformed closed-gap record");
        const int64_t b0 = (int64_t)res.bases.size();
        res.bases.insert(res.bases.end(), q.bases + x.cons_off, q.bases + x.cons_off + x.cons_len);
        x.cons_off = b0;
        res.rec.push_back(x);
        int32_t fo;
        memcpy(&fo, q.flank_of + i, 4);
        if (fo >= 0) {
            const int32_t nf = (x.join & DH_JOIN_EXTENSION) ? 1 : 2;
            if ((int64_t)fo + nf > q.h.nflank) return dh_fail(DH_EINVAL, "dh_shard_run: malformed closed-gap record");
            res.flank_of.push_back((int32_t)res.flank.size());
            for (int32_t f = 0; f < nf; f++) {
                dh_la l;
                memcpy(&l, q.flank + fo + f, sizeof(l));
                if (l.toff < 0 || l.tlen < 0 || l.toff + l.tlen > q.h.ntr) return dh_fail(DH_EINVAL, "dh_shard_run: malformed closed-gap record");
                const int64_t t0 = (int64_t)res.flank_tr.size();
                res.flank_tr.insert(res.flank_tr.end(), q.tr + l.toff, q.tr + l.toff + l.tlen);
                l.toff = t0;
                res.flank.push_back(l);
            }
        } else
            res.flank_of.push_back(-1);
        if (all_ids) {
            int32_t i0, i1;
            memcpy(&i0, q.ids_off + i, 4);
            memcpy(&i1, q.ids_off + i + 1, 4);
            if (i0 < 0 || i1 < i0 || i1 > q.h.nids) return dh_fail(DH_EINVAL, "dh_shard_run: malformed closed-gap record");
            res.ids.insert(res.ids.end(), q.ids + i0, q.ids + i1);
            res.ids_off.push_back((int32_t)res.ids.size());
        }
    }
    return DH_OK;
}
}  // namespace

// `collect` + `process` for this rank's share of the reads: reads_db holds the reads [read_first,
// read_first + n) of the whole reads DB, las / trace its mapping with bread already ids of the whole DB.
//   cands != NULL           spanning-read collector: the candidates dh_map_reads listed (read ids local: shifted here)
//   cands == NULL           scaffold-graph collector (`dentist collect`): read_off = offsets of this rank's reads,
//                           input_gaps / ngaps, sopts as for dh_scaffold_pileups
// Exchanges: all-gather of the joins / candidates -> the same plan on every rank; every rank crops ITS reads of every
// pile-up; all-to-all of the cropped reads to the pile-ups' owners; dh_process_cropped; all-gather of the closed gaps.
// *out: the records of ALL ranks ordered by gap with their consensus bases (identical on every rank); info4 (optional):
// pile-ups, pile-ups owned by this rank, entries, bytes of cropped reads sent.
extern "C" int dh_shard_run(dh_comm *c, dh_db *contigs, dh_db *reads, int32_t read_first, const int64_t *contig_off,
                            int32_t ncontigs, const dh_la *las, int64_t n, const uint16_t *trace, const dh_process_opts *opts,
                            const dh_pileups *cands, const int64_t *read_off, const int32_t *input_gaps, int32_t ngaps,
                            const dh_scaffold_opts *sopts, dh_insertions **out, int64_t *info4)
{
    if (!c || !contigs || !reads || !contig_off || !opts || !out || (n > 0 && (!las || !trace)) || (!cands && !read_off))
        return dh_fail(DH_EINVAL, "dh_shard_run: NULL argument");
    dh_ctx *ctx = c->ctx;
    if (!ctx) return dh_fail(DH_EINVAL, "dh_shard_run: the communicator has no context");
    const int32_t W = c->world;
    *out = nullptr;
    std::vector<int64_t> sizes((size_t)W);
    // Every local failure between two exchanges is carried INTO the next exchange (all_gather_st / all_to_all_st): the
    // peers learn of it from the size all-gather and every rank returns from the same place; nobody is left waiting.
    // ---- 1. this rank's joins / candidates, gathered; the plan
    FreeGuard mine, gathered;
    int64_t nmine = 0;
    int lrc = cands ? dh_shard_pack_candidates(cands, las, n, read_first, (uint8_t **)&mine.p, &nmine)
                    : dh_shard_read_joins(las, n, contig_off, ncontigs, read_off, read_first, reads->n, (uint8_t **)&mine.p, &nmine);
    if (int rc = all_gather_st(c, (const uint8_t *)mine.p, lrc ? 0 : nmine, lrc, (uint8_t **)&gathered.p, sizes.data())) return rc;
    dh_shard_plan *plan = nullptr;
    {
        const std::vector<const uint8_t *> bp = blob_ptrs((const uint8_t *)gathered.p, sizes.data(), W);
        dh_scaffold_opts so;
        if (!cands) {
            if (sopts)
                so = *sopts;
            else {
                dh_default_scaffold_opts(&so);
                so.min_spanning_reads = opts->min_reads;
            }
        }
        lrc = cands ? dh_shard_plan_create(bp.data(), sizes.data(), W, opts, &plan)
                    : dh_shard_graph_plan_create(bp.data(), sizes.data(), W, ncontigs, input_gaps, ngaps, &so, opts, &plan);
    }
    struct PlanGuard {
        dh_shard_plan *p;
        ~PlanGuard() { dh_shard_plan_destroy(p); }
    } pg{plan};
    const dh_pileups *piles = lrc ? nullptr : dh_shard_plan_pileups(plan);
    const int32_t *owner = lrc ? nullptr : dh_shard_plan_owner(plan);
    const int32_t npiles = lrc ? 0 : dh_pileups_count(piles);
    // ---- 2. crop this rank's reads of every pile-up; to the owners
    struct CropGuard {
        dh_cropped *p;
        ~CropGuard() { dh_cropped_destroy(p); }
    };
    std::vector<dh_insertion> rec((size_t)npiles);
    std::vector<uint8_t *> dest((size_t)W, nullptr);
    std::vector<int64_t> ssz((size_t)W, 0), rsz((size_t)W, 0);
    FreeGuard sent, recvd;
    if (!lrc) {
        dh_cropped *crop = nullptr;
        lrc = dh_crop_pileups(ctx, contigs, reads, read_first, dh_shard_plan_las(plan), dh_shard_plan_nlas(plan), trace, piles, opts, &crop);
        CropGuard cg{crop};
        if (!lrc) {
            if (npiles) memcpy(rec.data(), dh_cropped_records(crop), sizeof(dh_insertion) * (size_t)npiles);
            lrc = dh_shard_pack_cropped(crop, owner, W, dest.data(), ssz.data());
            if (!lrc) sent.p = dest[0];
        }
    }
    if (int rc = all_to_all_st(c, (const uint8_t *const *)dest.data(), ssz.data(), lrc, (uint8_t **)&recvd.p, rsz.data())) return rc;
    // ---- 3. the pile-ups this rank owns
    dh_cropped *own = nullptr;
    {
        const std::vector<const uint8_t *> bp = blob_ptrs((const uint8_t *)recvd.p, rsz.data(), W);
        lrc = dh_shard_unpack_cropped(bp.data(), rsz.data(), W, rec.data(), npiles, owner, c->rank, &own);
    }
    dh_insertions *local = nullptr;
    {
        CropGuard og{own};
        if (!lrc) lrc = dh_process_cropped(ctx, contigs, own, opts, &local);
    }
    struct InsGuard {
        dh_insertions *p;
        ~InsGuard() { dh_insertions_destroy(p); }
    } ig{local};
    // ---- 4. closed gaps of all ranks (the role of merge-insertions, mergeInsertions.d:60-164): everything insertions.db
    // keeps of a pile-up travels -- record, consensus bases, the flank overlaps with their trace points, the read ids
    std::vector<uint8_t> blob;
    if (!lrc) pack_closed(*local, blob);
    FreeGuard closed;
    if (int rc = all_gather_st(c, blob.data(), (int64_t)blob.size(), lrc, (uint8_t **)&closed.p, sizes.data())) return rc;
    dh_insertions *res = new dh_insertions();
    {
        const std::vector<const uint8_t *> bp = blob_ptrs((const uint8_t *)closed.p, sizes.data(), W);
        if (int rc = merge_closed(bp, sizes, W, owner, npiles, *res)) {
            delete res;
            return rc;
        }
    }
    if (info4) {
        int32_t owned = 0;
        for (int32_t p = 0; p < npiles; p++) owned += owner[p] == c->rank ? 1 : 0;
        int64_t entries = 0;
        for (int32_t p = 0; p < npiles; p++) entries += dh_pileups_get(piles, p, nullptr, nullptr);
        info4[0] = npiles;
        info4[1] = owned;
        info4[2] = entries;
        info4[3] = std::accumulate(ssz.begin(), ssz.end(), (int64_t)0);
    }
    *out = res;
    return DH_OK;
}
