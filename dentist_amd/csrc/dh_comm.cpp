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
        SYM(GetErrorString)
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
    const ncclResult_t r = rccl().CommInitRank(&c->nccl, world, id, rank);
    if (r != ncclSuccess) {
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
extern "C" int dh_comm_all_gather(dh_comm *c, const uint8_t *payload, int64_t nbytes, uint8_t **out, int64_t *sizes)
{
    if (!c || !out || !sizes || nbytes < 0 || (nbytes > 0 && !payload)) return dh_fail(DH_EINVAL, "dh_comm_all_gather: bad argument");
    const int32_t W = c->world;
    *out = nullptr;
    if (c->hub) {
        LocalHub &h = *c->hub;
        h.ptr[(size_t)c->rank] = payload;
        h.size[(size_t)c->rank] = nbytes;
        h.barrier();
        int64_t total = 0;
        for (int32_t r = 0; r < W; r++) total += (sizes[r] = h.size[(size_t)r]);
        uint8_t *buf = (uint8_t *)malloc((size_t)std::max<int64_t>(total, 1));
        if (!buf) return dh_fail(DH_ENOMEM, "dh_comm_all_gather: out of memory");
        int64_t at = 0;
        for (int32_t r = 0; r < W; r++) {
            if (sizes[r]) memcpy(buf + at, h.ptr[(size_t)r], (size_t)sizes[r]);
            at += sizes[r];
        }
        h.barrier();  // every rank has copied: the payloads may go
        *out = buf;
        return DH_OK;
    }
    dh_ctx *ctx = c->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    // sizes first
    int64_t *d_sz;
    if (int rc = dh_scratch(ctx, 55, sizeof(int64_t) * (size_t)(W + 1), (void **)&d_sz)) return rc;
    HIPCHK(hipMemcpyAsync(d_sz + W, &nbytes, sizeof(int64_t), hipMemcpyHostToDevice, st));
    NCCLCHK(rccl().AllGather(d_sz + W, d_sz, 1, ncclInt64, c->nccl, st));
    HIPCHK(hipMemcpyAsync(sizes, d_sz, sizeof(int64_t) * (size_t)W, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    int64_t cap = 1, total = 0;
    for (int32_t r = 0; r < W; r++) {
        cap = std::max(cap, sizes[r]);
        total += sizes[r];
    }
    cap = (cap + 15) & ~15ll;
    uint8_t *d_send, *d_recv;
    if (int rc = dh_scratch(ctx, 56, (size_t)cap, (void **)&d_send)) return rc;
    if (int rc = dh_scratch(ctx, 57, (size_t)cap * (size_t)W, (void **)&d_recv)) return rc;
    // page-locked staging both ways (the blobs the glue hands over are plain malloc'd memory)
    uint8_t *stage = (uint8_t *)dh_pinned_alloc((size_t)std::max<int64_t>(std::max(nbytes, total), 1));
    if (!stage) return dh_fail(DH_ENOMEM, "dh_comm_all_gather: out of page-locked memory");
    struct Unpin {
        uint8_t *p;
        size_t n;
        ~Unpin() { dh_pinned_free(p, n); }
    } unpin{stage, (size_t)std::max<int64_t>(std::max(nbytes, total), 1)};
    if (nbytes) {
        memcpy(stage, payload, (size_t)nbytes);
        HIPCHK(hipMemcpyAsync(d_send, stage, (size_t)nbytes, hipMemcpyHostToDevice, st));
    }
    NCCLCHK(rccl().AllGather(d_send, d_recv, (size_t)cap, ncclUint8, c->nccl, st));
    HIPCHK(hipStreamSynchronize(st));  // (the staging buffer is reused for the way back)
    int64_t at = 0;
    for (int32_t r = 0; r < W; r++) {
        if (sizes[r]) HIPCHK(hipMemcpyAsync(stage + at, d_recv + (size_t)r * (size_t)cap, (size_t)sizes[r], hipMemcpyDeviceToHost, st));
        at += sizes[r];
    }
    HIPCHK(hipStreamSynchronize(st));
    uint8_t *buf = (uint8_t *)malloc((size_t)std::max<int64_t>(total, 1));
    if (!buf) return dh_fail(DH_ENOMEM, "dh_comm_all_gather: out of memory");
    if (total) memcpy(buf, stage, (size_t)total);
    *out = buf;
    return DH_OK;
}

// all-to-all(v): per_dest[r] / send_sizes[r] = the blob for rank r.  *out: one malloc'd block with the blobs received in
// source-rank order (dh_shard_free), recv_sizes[world] their lengths.
extern "C" int dh_comm_all_to_all(dh_comm *c, const uint8_t *const *per_dest, const int64_t *send_sizes, uint8_t **out,
                                  int64_t *recv_sizes)
{
    if (!c || !per_dest || !send_sizes || !out || !recv_sizes) return dh_fail(DH_EINVAL, "dh_comm_all_to_all: NULL argument");
    const int32_t W = c->world;
    *out = nullptr;
    if (c->hub) {
        LocalHub &h = *c->hub;
        h.dest_ptr[(size_t)c->rank] = per_dest;
        h.dest_size[(size_t)c->rank] = send_sizes;
        h.barrier();
        int64_t total = 0;
        for (int32_t r = 0; r < W; r++) total += (recv_sizes[r] = h.dest_size[(size_t)r][c->rank]);
        uint8_t *buf = (uint8_t *)malloc((size_t)std::max<int64_t>(total, 1));
        if (!buf) return dh_fail(DH_ENOMEM, "dh_comm_all_to_all: out of memory");
        int64_t at = 0;
        for (int32_t r = 0; r < W; r++) {
            if (recv_sizes[r]) memcpy(buf + at, h.dest_ptr[(size_t)r][c->rank], (size_t)recv_sizes[r]);
            at += recv_sizes[r];
        }
        h.barrier();
        *out = buf;
        return DH_OK;
    }
    dh_ctx *ctx = c->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    // every rank's row of send sizes: recv_sizes[r] = row r, column rank
    int64_t *d_sz;
    if (int rc = dh_scratch(ctx, 55, sizeof(int64_t) * (size_t)W * (size_t)(W + 1), (void **)&d_sz)) return rc;
    std::vector<int64_t> rows((size_t)W * (size_t)W);
    HIPCHK(hipMemcpyAsync(d_sz + (size_t)W * W, send_sizes, sizeof(int64_t) * (size_t)W, hipMemcpyHostToDevice, st));
    NCCLCHK(rccl().AllGather(d_sz + (size_t)W * W, d_sz, (size_t)W, ncclInt64, c->nccl, st));
    HIPCHK(hipMemcpyAsync(rows.data(), d_sz, sizeof(int64_t) * rows.size(), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    int64_t stot = 0, rtot = 0;
    for (int32_t r = 0; r < W; r++) {
        if (send_sizes[r] < 0) return dh_fail(DH_EINVAL, "dh_comm_all_to_all: negative size");
        stot += send_sizes[r];
        rtot += (recv_sizes[r] = rows[(size_t)r * W + c->rank]);
    }
    uint8_t *d_send, *d_recv;
    if (int rc = dh_scratch(ctx, 56, (size_t)std::max<int64_t>(stot, 16), (void **)&d_send)) return rc;
    if (int rc = dh_scratch(ctx, 57, (size_t)std::max<int64_t>(rtot, 16), (void **)&d_recv)) return rc;
    const size_t nstage = (size_t)std::max<int64_t>(std::max(stot, rtot), 1);
    uint8_t *stage = (uint8_t *)dh_pinned_alloc(nstage);
    if (!stage) return dh_fail(DH_ENOMEM, "dh_comm_all_to_all: out of page-locked memory");
    struct Unpin {
        uint8_t *p;
        size_t n;
        ~Unpin() { dh_pinned_free(p, n); }
    } unpin{stage, nstage};
    int64_t at = 0;
    for (int32_t r = 0; r < W; r++) {
        if (send_sizes[r]) memcpy(stage + at, per_dest[r], (size_t)send_sizes[r]);
        at += send_sizes[r];
    }
    if (stot) HIPCHK(hipMemcpyAsync(d_send, stage, (size_t)stot, hipMemcpyHostToDevice, st));
    NCCLCHK(rccl().GroupStart());
    int64_t so = 0, ro = 0;
    for (int32_t r = 0; r < W; r++) {
        if (send_sizes[r]) NCCLCHK(rccl().Send(d_send + so, (size_t)send_sizes[r], ncclUint8, r, c->nccl, st));
        if (recv_sizes[r]) NCCLCHK(rccl().Recv(d_recv + ro, (size_t)recv_sizes[r], ncclUint8, r, c->nccl, st));
        so += send_sizes[r];
        ro += recv_sizes[r];
    }
    NCCLCHK(rccl().GroupEnd());
    HIPCHK(hipStreamSynchronize(st));
    if (rtot) HIPCHK(hipMemcpyAsync(stage, d_recv, (size_t)rtot, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    uint8_t *buf = (uint8_t *)malloc((size_t)std::max<int64_t>(rtot, 1));
    if (!buf) return dh_fail(DH_ENOMEM, "dh_comm_all_to_all: out of memory");
    if (rtot) memcpy(buf, stage, (size_t)rtot);
    *out = buf;
    return DH_OK;
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
    // ---- 1. this rank's joins / candidates, gathered; the plan
    FreeGuard mine, gathered;
    int64_t nmine = 0;
    if (cands) {
        if (int rc = dh_shard_pack_candidates(cands, las, n, read_first, (uint8_t **)&mine.p, &nmine)) return rc;
    } else if (int rc = dh_shard_read_joins(las, n, contig_off, ncontigs, read_off, read_first, reads->n, (uint8_t **)&mine.p, &nmine))
        return rc;
    if (int rc = dh_comm_all_gather(c, (const uint8_t *)mine.p, nmine, (uint8_t **)&gathered.p, sizes.data())) return rc;
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
        if (int rc = cands ? dh_shard_plan_create(bp.data(), sizes.data(), W, opts, &plan)
                           : dh_shard_graph_plan_create(bp.data(), sizes.data(), W, ncontigs, input_gaps, ngaps, &so, opts, &plan))
            return rc;
    }
    struct PlanGuard {
        dh_shard_plan *p;
        ~PlanGuard() { dh_shard_plan_destroy(p); }
    } pg{plan};
    const dh_pileups *piles = dh_shard_plan_pileups(plan);
    const int32_t *owner = dh_shard_plan_owner(plan);
    const int32_t npiles = dh_pileups_count(piles);
    // ---- 2. crop this rank's reads of every pile-up; to the owners
    dh_cropped *crop = nullptr;
    if (int rc = dh_crop_pileups(ctx, contigs, reads, read_first, dh_shard_plan_las(plan), dh_shard_plan_nlas(plan), trace, piles,
                                 opts, &crop))
        return rc;
    struct CropGuard {
        dh_cropped *p;
        ~CropGuard() { dh_cropped_destroy(p); }
    };
    std::vector<dh_insertion> rec((size_t)npiles);
    if (npiles) memcpy(rec.data(), dh_cropped_records(crop), sizeof(dh_insertion) * (size_t)npiles);
    std::vector<uint8_t *> dest((size_t)W, nullptr);
    std::vector<int64_t> ssz((size_t)W, 0), rsz((size_t)W, 0);
    FreeGuard sent, recvd;
    {
        CropGuard cg{crop};
        if (int rc = dh_shard_pack_cropped(crop, owner, W, dest.data(), ssz.data())) return rc;
        sent.p = dest[0];
    }
    if (int rc = dh_comm_all_to_all(c, (const uint8_t *const *)dest.data(), ssz.data(), (uint8_t **)&recvd.p, rsz.data())) return rc;
    // ---- 3. the pile-ups this rank owns
    dh_cropped *own = nullptr;
    {
        const std::vector<const uint8_t *> bp = blob_ptrs((const uint8_t *)recvd.p, rsz.data(), W);
        if (int rc = dh_shard_unpack_cropped(bp.data(), rsz.data(), W, rec.data(), npiles, owner, c->rank, &own)) return rc;
    }
    dh_insertions *local = nullptr;
    {
        CropGuard og{own};
        if (int rc = dh_process_cropped(ctx, contigs, own, opts, &local)) return rc;
    }
    struct InsGuard {
        dh_insertions *p;
        ~InsGuard() { dh_insertions_destroy(p); }
    } ig{local};
    // ---- 4. closed gaps of all ranks (the role of merge-insertions): int64 record bytes, records, bases
    std::vector<uint8_t> blob(8 + sizeof(dh_insertion) * local->rec.size() + local->bases.size());
    {
        const int64_t rb = (int64_t)(sizeof(dh_insertion) * local->rec.size());
        memcpy(blob.data(), &rb, 8);
        if (rb) memcpy(blob.data() + 8, local->rec.data(), (size_t)rb);
        if (!local->bases.empty()) memcpy(blob.data() + 8 + rb, local->bases.data(), local->bases.size());
    }
    FreeGuard closed;
    if (int rc = dh_comm_all_gather(c, blob.data(), (int64_t)blob.size(), (uint8_t **)&closed.p, sizes.data())) return rc;
    dh_insertions *res = new dh_insertions();
    {
        const std::vector<const uint8_t *> bp = blob_ptrs((const uint8_t *)closed.p, sizes.data(), W);
        std::vector<dh_insertion> all;
        for (int32_t r = 0; r < W; r++) {
            if (sizes[r] < 8) {
                delete res;
                return dh_fail(DH_EINVAL, "dh_shard_run: short closed-gap blob");
            }
            int64_t rb;
            memcpy(&rb, bp[(size_t)r], 8);
            if (rb < 0 || rb % (int64_t)sizeof(dh_insertion) || 8 + rb > sizes[r]) {
                delete res;
                return dh_fail(DH_EINVAL, "dh_shard_run: malformed closed-gap blob");
            }
            const size_t nr = (size_t)rb / sizeof(dh_insertion), b0 = res->bases.size();
            const size_t a0 = all.size();
            all.resize(a0 + nr);
            if (nr) memcpy(all.data() + a0, bp[(size_t)r] + 8, (size_t)rb);
            for (size_t x = a0; x < all.size(); x++) all[x].cons_off += (int64_t)b0;
            res->bases.insert(res->bases.end(), bp[(size_t)r] + 8 + rb, bp[(size_t)r] + sizes[r]);
        }
        // insertions.sort() by start node (processPileUps/package.d:156): stable by gap, ranks in order
        std::vector<size_t> order(all.size());
        std::iota(order.begin(), order.end(), (size_t)0);
        std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return all[x].contig_left < all[y].contig_left; });
        res->rec.reserve(all.size());
        for (size_t x : order) res->rec.push_back(all[x]);
        res->flank_of.assign(res->rec.size(), -1);
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
