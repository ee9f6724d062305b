"""Multi-GPU layer: one process per GPU, the hot path sharded the way the reference's workflow
shards it, with the filesystem replaced by two small exchanges over RCCL/xGMI (gloo in CPU tests).

  mapping   read blocks over ranks, contigs + k-mer index replicated -- one `damapper` job per read
            block (snakemake/Snakefile:1143-1170).  No collective.
  collect   every rank finds the spanning reads among ITS reads (per-read decision); the candidate
            entries (gap, read id, the two anchoring LA records; ~100 B each, a few MB in total) are
            all-gathered in rank order = read-id order, which is what `LAmerge` + `dentist collect`
            see (Snakefile:1173-1185).  Every rank then applies the same min/max-reads cut and gets
            identical pile-ups.
  process   pile-ups are bin-packed over ranks by cost n^2 * L (greedy, largest first) -- the role of
            `process --batch` (Snakefile:1315-1334).  Each rank crops ITS reads of every pile-up on
            its GPU (dh_crop_pileups) and sends the cropped reads (gap + anchors, not whole reads) to
            the pile-up's owner: one all-to-all(v).  Owners run dh_process_cropped.
  gather    closed-gap records of all ranks, ordered by gap = `dentist merge-insertions`
            (commands/mergeInsertions.d:60-164, insertions.sort() processPileUps/package.d:156).

The result is bit-identical to the single-GPU run on the same inputs (tests/test_parallel_gloo.py
for the host logic, tests/test_parity_shard_gpu.py for two shards run on one GPU).
"""
import ctypes

import numpy as np

from ._lib import INSERTION_DTYPE, LA_DTYPE, lib

CAND_DTYPE = np.dtype([("gap", "<i4"), ("read", "<i4"), ("L", LA_DTYPE), ("R", LA_DTYPE)])
CROP_DTYPE = np.dtype([("pile", "<i4"), ("entry", "<i4"), ("read", "<i4"), ("len", "<i4")])


def shard_range(n, rank, world):
    """Contiguous block partition of n units (reads) over the ranks: (first, end)."""
    return n * rank // world, n * (rank + 1) // world


def assign_owners(costs, world):
    """Greedy bin-packing (largest cost first, ties by index; least-loaded rank, ties by rank):
    owner[i] for every unit.  Deterministic, so every rank computes the same assignment."""
    costs = np.asarray(costs, dtype=np.int64)
    order = sorted(range(len(costs)), key=lambda i: (-int(costs[i]), i))
    load = [0] * world
    owner = np.zeros(len(costs), dtype=np.int32)
    for i in order:
        r = min(range(world), key=lambda x: (load[x], x))
        owner[i] = r
        load[r] += int(costs[i])
    return owner


# ------------------------------------------------------------------ collectives on byte payloads

def _device(dist):
    import torch
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def all_gather_bytes(payload, world):
    """all-gather(v) of one uint8 array per rank: sizes first, then the padded bytes."""
    import torch
    import torch.distributed as dist
    payload = np.ascontiguousarray(payload, dtype=np.uint8)
    if world == 1 or not dist.is_initialized():
        return [payload]
    dev = _device(dist)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([len(payload)], dtype=torch.int64, device=dev))
    sizes = [int(s.item()) for s in sizes]
    cap = max(1, max(sizes))
    mine = torch.zeros(cap, dtype=torch.uint8, device=dev)
    if len(payload):
        mine[:len(payload)] = torch.from_numpy(payload.copy()).to(dev)
    out = [torch.zeros(cap, dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(out, mine)
    return [out[r][:sizes[r]].cpu().numpy() for r in range(world)]


def all_to_all_bytes(per_dest, world):
    """all-to-all(v): per_dest[r] = uint8 array for rank r; returns the arrays received, by source.
    RCCL: one all_to_all_single with split sizes; gloo has no all-to-all, the CPU tests fall back to
    an all-gather of everything and pick their own parts."""
    import torch
    import torch.distributed as dist
    per_dest = [np.ascontiguousarray(x, dtype=np.uint8) for x in per_dest]
    if world == 1 or not dist.is_initialized():
        return [per_dest[0]]
    rank = dist.get_rank()
    dev = _device(dist)
    if dist.get_backend() != "nccl":
        head = np.asarray([len(x) for x in per_dest], dtype=np.int64)
        blobs = all_gather_bytes(np.concatenate([head.view(np.uint8)] + per_dest), world)
        out = []
        for src in range(world):
            h = blobs[src][:8 * world].view(np.int64)
            start = 8 * world + int(h[:rank].sum())
            out.append(blobs[src][start:start + int(h[rank])].copy())
        return out
    send_sizes = torch.tensor([len(x) for x in per_dest], dtype=torch.int64, device=dev)
    recv_sizes = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_to_all_single(recv_sizes, send_sizes)
    rs = [int(x) for x in recv_sizes.tolist()]
    ss = [len(x) for x in per_dest]
    send = torch.from_numpy(np.concatenate(per_dest) if sum(ss) else np.zeros(0, np.uint8)).to(dev)
    recv = torch.zeros(sum(rs), dtype=torch.uint8, device=dev)
    dist.all_to_all_single(recv, send, output_split_sizes=rs, input_split_sizes=ss)
    buf = recv.cpu().numpy()
    out, at = [], 0
    for n in rs:
        out.append(buf[at:at + n].copy())
        at += n
    return out


# ------------------------------------------------------------------ the sharded `collect` + `process`

def pack_candidates(cands, las, read_shift=0):
    """Candidate entries of this rank (dentist_amd.Pileups(..., candidates=True) on its LAs, whose
    bread are ids of the whole reads DB) as CAND_DTYPE records in (gap, read) order.  read_shift: added
    to the read ids of candidates that were collected before the LAs got their whole-DB ids."""
    cl, cnt, tri = cands.flat()
    rec = np.zeros(len(tri), dtype=CAND_DTYPE)
    rec["gap"] = np.repeat(cl, cnt)
    rec["read"] = tri[:, 0] + read_shift
    rec["L"] = las[tri[:, 1]]
    rec["R"] = las[tri[:, 2]]
    return rec


def merge_candidates(per_rank):
    """All ranks' candidates -> (LA array, contig_left, count, triples).  Ranks hold ascending read
    ranges and list their candidates by read, so a stable sort by gap of the concatenation in rank
    order keeps every gap's entries ordered by read id -- the order `dentist collect` sees after
    LAmerge."""
    allc = np.concatenate(per_rank) if len(per_rank) else np.zeros(0, dtype=CAND_DTYPE)
    # the two LA records of an entry are adjacent in the packed record: one strided copy gives L0 R0 L1 R1 ...
    raw = np.ascontiguousarray(allc).view(np.uint8).reshape(len(allc), CAND_DTYPE.itemsize)
    las = np.ascontiguousarray(raw[:, 8:8 + 2 * LA_DTYPE.itemsize]).reshape(-1).view(LA_DTYPE)
    order = np.argsort(allc["gap"], kind="stable")
    gaps, counts = np.unique(allc["gap"][order], return_counts=True)
    triples = np.stack([allc["read"][order], 2 * order, 2 * order + 1], axis=1).astype(np.int32)
    return las, gaps.astype(np.int32), counts.astype(np.int32), triples


def pile_costs(piles, las):
    """n^2 * L per pile-up: n reads, L = mean read span between the anchors + 1 kb."""
    _, cnt, tri = piles.flat()
    if len(cnt) == 0:
        return np.zeros(0, dtype=np.int64)
    span = np.maximum(las["bbpos"][tri[:, 2]].astype(np.int64) - las["bepos"][tri[:, 1]], 0)
    starts = np.concatenate([[0], np.cumsum(cnt)[:-1]])
    mean = np.add.reduceat(span, starts) / np.maximum(cnt, 1) + 1000.0
    return (cnt.astype(np.int64) ** 2 * mean).astype(np.int64)


def _runs(bases, off, idx):
    """Slices of `bases` covering the sequences idx[0], idx[1], ... (off = their offsets), with runs
    of consecutive indices merged into one slice."""
    idx = np.asarray(idx, dtype=np.int64)
    if len(idx) == 0:
        return []
    brk = np.nonzero(np.diff(idx) != 1)[0] + 1
    starts = np.concatenate([[0], brk])
    ends = np.concatenate([brk, [len(idx)]])
    return [bases[off[idx[a]]:off[idx[b - 1] + 1]] for a, b in zip(starts, ends)]


_COMM = {}


def rccl_comm(ctx, rank, world):
    """The communicator of the C ABI (dh_comm_create) for this process, created once: rank 0 draws the RCCL unique id
    (dh_comm_unique_id), torch.distributed only carries those 128 bytes to the other processes -- what a D host would do
    with a file or MPI."""
    import torch.distributed as dist
    from ._lib import Comm
    key = (id(ctx), rank, world)
    if key not in _COMM:
        box = [None]
        if rank == 0:
            try:
                box = [Comm.unique_id()]
            except Exception as e:   # noqa: BLE001 -- the other ranks wait in the broadcast: they must get an answer
                box = [e]
        dist.broadcast_object_list(box, src=0)
        if not isinstance(box[0], (bytes, bytearray)):
            raise RuntimeError("dh_comm_unique_id failed on rank 0: %r" % (box[0],))
        _COMM[key] = Comm.create(ctx, rank, world, box[0])
    return _COMM[key]


# error codes of the C ABI that say "the communicator / RCCL / the device path did not work" (include/dentist_hip.h:
# DH_ENODEV -2, DH_EHIP -3) -- the only failures of a first dh_shard_run the ranks answer by switching to
# torch.distributed's collectives; a data error (DH_EINVAL, DH_EOVERFLOW, DH_EIO, DH_ENOMEM) would fail there as well
_COMM_ERROR_CODES = (-2, -3)


def _agree_min(value, dist):
    """MIN over the ranks of one small integer (every rank must call it at the same point)."""
    import torch
    t = torch.tensor([int(value)], dtype=torch.int32, device=_device(dist))
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item())


def sharded_process(ctx, contigs_db, reads_db, read_first, contig_off, las, trace, popts, rank, world, cands=None, graph=None):
    """`collect` + `process` for one rank's share of the reads.  las/trace: this rank's mapping
    result with bread ALREADY shifted to ids of the whole reads DB; reads_db holds the reads
    [read_first, read_first + n).  Returns (records, bases, info): the closed-gap records of ALL
    ranks ordered by gap (identical on every rank) with ref_read_id as whole-DB ids."""
    import torch.distributed as dist
    if world > 1 and dist.is_initialized() and dist.get_backend() == "nccl" and _c_abi_collectives(ctx, rank, world):
        # one process per GPU over RCCL: the whole sequence behind the C ABI (dh_shard_run); this module is a thin caller
        import sys
        from ._lib import DhError, shard_run_prepare
        key = (id(ctx), rank, world)
        # PRE-FLIGHT, agreed on by all ranks: whatever can fail before the first exchange inside dh_shard_run (array
        # conversion, option names, NULL handles) fails HERE, and a rank that failed tells the others before anybody
        # enters a collective of the C ABI -- nobody is left waiting in an RCCL call for a rank that never arrives
        call, perr = None, None
        try:
            call = shard_run_prepare(rccl_comm(ctx, rank, world), contigs_db, reads_db, read_first, contig_off, las, trace, popts,
                                     cands=cands, graph=graph)
        except Exception as e:   # noqa: BLE001
            perr = e
        if _agree_min(0 if perr is not None else 1, dist) == 0:
            if perr is not None:
                raise perr
            raise RuntimeError("sharded_process: another rank failed before dh_shard_run (its own error names the cause)")
        res, err = None, None
        try:
            res = call()
        except Exception as e:   # noqa: BLE001
            err = e
        if not _C_ABI_RAN.get(key):
            # the FIRST run of this process: the ranks agree on how it went (a rank that fails inside dh_shard_run still takes
            # part in the next size exchange with its status, so every rank comes back from the same exchange, dh_comm.cpp).
            # 2 = fine, 1 = a communicator / RCCL / HIP failure, 0 = a data error.  Only a communicator failure sends all
            # ranks to the torch.distributed collectives (for good); a data error is raised on every rank as it is -- it
            # would fail the same way there, and computing it twice hides where it came from
            mine = 2 if err is None else (1 if isinstance(err, DhError) and err.code in _COMM_ERROR_CODES else 0)
            # (a rank that only learned of a peer's failure reports 2 here: DH_EINVAL "rank r failed before the exchange")
            if err is not None and isinstance(err, DhError) and "failed before the exchange" in str(err):
                mine = 2
            worst = _agree_min(mine, dist)
            if worst == 2 and err is None:
                _C_ABI_RAN[key] = True
            elif worst == 1:
                print("[dentist_amd] rank %d: dh_shard_run failed in the communicator on some rank (%s); the exchanges go through "
                      "torch.distributed from here on" % (rank, repr(err) if err is not None else "not this one"),
                      file=sys.stderr, flush=True)
                _C_ABI_OK[key] = False
                res = None
            else:
                if err is not None:
                    raise err
                raise RuntimeError("sharded_process: dh_shard_run failed on another rank with a data error")
        elif err is not None:
            raise err
        if res is not None:
            rec, bases, info = res
            info["collectives"] = "dh_comm (RCCL behind the C ABI)"
            return rec, bases, info
    gen = sharded_process_steps(ctx, contigs_db, reads_db, read_first, contig_off, las, trace, popts, rank, world, cands, graph)
    try:
        req = next(gen)
        while True:
            kind, payload = req
            req = gen.send(all_gather_bytes(payload, world) if kind == "all_gather" else all_to_all_bytes(payload, world))
    except StopIteration as done:
        rec, bases, info = done.value
        info["collectives"] = ("torch.distributed (%s), host steps in dh_shard_*" % dist.get_backend()
                               if world > 1 and dist.is_initialized() else "none (one rank)")
        return rec, bases, info


_C_ABI_OK = {}
_C_ABI_RAN = {}


def _c_abi_collectives(ctx, rank, world):
    """Whether the ranks run the exchanges behind the C ABI (dh_comm over RCCL).  Decided ONCE per process, by all ranks
    together: rank 0's DH_SHARD_COLLECTIVES setting is broadcast first (the variable may differ between the ranks'
    environments; every rank must take the same sequence of collectives from here on), then every rank tries to create its
    communicator and the outcomes are min-reduced over torch.distributed, so that either all ranks take dh_shard_run or all
    take the torch.distributed collectives (RCCL as well; the host steps between them are the same dh_shard_* functions).  A
    rank that cannot create the communicator says so on stderr -- the result records which path ran
    (info["collectives"]).  DH_SHARD_COLLECTIVES=torch (on rank 0) forces the second path."""
    import os
    import sys
    import torch.distributed as dist
    key = (id(ctx), rank, world)
    if key not in _C_ABI_OK:
        box = [os.environ.get("DH_SHARD_COLLECTIVES", "") if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        ok, why = 1, ""
        if box[0] == "torch":
            ok, why = 0, "DH_SHARD_COLLECTIVES=torch on rank 0"
        else:
            try:
                rccl_comm(ctx, rank, world)
            except Exception as e:   # noqa: BLE001 -- whatever went wrong, the other ranks must learn of it
                ok, why = 0, repr(e)
        agreed = _agree_min(ok, dist)
        if not ok:
            print("[dentist_amd] rank %d: no dh_comm communicator (%s); the exchanges go through torch.distributed" % (rank, why),
                  file=sys.stderr, flush=True)
        _C_ABI_OK[key] = bool(agreed)
    return _C_ABI_OK[key]


def emulate_ranks(gens):
    """Drive one sharded_process_steps generator per emulated rank in lockstep inside ONE process
    (tests: two shards on one GPU); collectives are served from memory.  Returns their results."""
    world = len(gens)
    reqs = [next(g) for g in gens]
    results = [None] * world
    while any(r is not None for r in reqs):
        kinds = {r[0] for r in reqs if r is not None}
        assert len(kinds) == 1 and all(r is not None for r in reqs), "ranks diverged"
        if kinds == {"all_gather"}:
            answers = [[np.asarray(reqs[src][1], dtype=np.uint8).copy() for src in range(world)] for _ in range(world)]
        else:
            answers = [[np.asarray(reqs[src][1][dst], dtype=np.uint8).copy() for src in range(world)] for dst in range(world)]
        nxt = []
        for r, g in enumerate(gens):
            try:
                nxt.append(g.send(answers[r]))
            except StopIteration as done:
                results[r] = done.value
                nxt.append(None)
        reqs = nxt
    return results


def sharded_process_steps(ctx, contigs_db, reads_db, read_first, contig_off, las, trace, popts, rank, world, cands=None,
                          graph=None):
    """Generator form of sharded_process: yields ("all_gather", bytes) / ("all_to_all", [bytes per
    destination]) and expects the list of arrays received (by source rank) to be sent back.

    graph = None: the spanning-read collector (candidates).  graph = dict(read_off=offsets of this rank's reads,
    input_gaps=..., plus scaffold options): the scaffold-graph collector of `dentist collect` -- the raw joins of a
    rank's reads are a per-read computation (collectReadAlignments), they are all-gathered instead of the
    candidates, and every rank builds the same scaffold and the same gap pile-ups (extension entries included)."""
    import os
    import time
    from . import Cropped, Pileups
    _t = [time.perf_counter()]
    _laps = []

    def lap(what):
        if os.environ.get("DH_TRACE"):
            t = time.perf_counter()
            _laps.append("%s %.1f" % (what, (t - _t[0]) * 1e3))
            _t[0] = t
    # cands: candidates dh_map_reads collected on the way (read ids still local to this rank's reads DB).
    # The host work between the collectives is C++ behind dh_shard_* (the numpy restatement of it -- pack_candidates,
    # merge_candidates, assign_owners, pile_costs above -- is what tests/test_parallel_gloo.py checks it against)
    from ._lib import ShardPlan, shard_pack_candidates, shard_pack_cropped, shard_read_joins, shard_unpack_cropped
    if graph is not None:
        g = dict(graph)
        read_off, input_gaps = g.pop("read_off"), g.pop("input_gaps", None)
        g.setdefault("min_spanning_reads", popts.min_reads)
        mine = shard_read_joins(las, contig_off, read_off, read_first)
        lap("read joins")
        blobs = yield ("all_gather", mine)
        _t[0] = time.perf_counter()
        ncand = sum(int(np.frombuffer(memoryview(b)[:8], dtype=np.int64)[0]) for b in blobs if len(b) >= 16)  # JoinHead.njoins
        plan = ShardPlan(blobs, popts, graph=(len(contig_off) - 1, input_gaps, g))
    else:
        shift = 0 if cands is None else read_first
        if cands is None:
            cands = Pileups(las, contig_off, popts, candidates=True)
        mine = shard_pack_candidates(cands, las, shift)
        lap("candidates")
        blobs = yield ("all_gather", mine)
        _t[0] = time.perf_counter()
        ncand = sum(len(b) for b in blobs) // CAND_DTYPE.itemsize
        plan = ShardPlan(blobs, popts)
    glas, piles, owner = plan.las, plan.piles, plan.owner
    lap("plan (merge, cut, owners)")
    # the entries of this rank inside glas are copies of its own records: their toff still points
    # into its own trace array, which is all dh_crop_pileups needs (other ranks' traces stay there)
    crop = Cropped.crop(ctx, contigs_db, reads_db, read_first, glas, trace, piles, popts)
    lap("crop")
    L = lib()
    npl = L.dh_cropped_npiles(crop._h)
    rec = (np.frombuffer(ctypes.string_at(L.dh_cropped_records(crop._h), npl * INSERTION_DTYPE.itemsize),
                         dtype=INSERTION_DTYPE).copy() if npl else np.zeros(0, dtype=INSERTION_DTYPE))
    per_dest = shard_pack_cropped(crop, owner, world)
    crop.close()
    lap("pack per dest")
    got = yield ("all_to_all", per_dest)
    _t[0] = time.perf_counter()
    mine_piles = np.nonzero(owner == rank)[0]
    own = shard_unpack_cropped(got, rec, owner, rank)
    lap("unpack + create")
    lrec, lbases = own.process(ctx, contigs_db, popts)
    lap("process")
    own.close()
    blobs = yield ("all_gather", _pack_closed(lrec, lbases))
    _t[0] = time.perf_counter()
    grec, gbases, origin = _unpack_closed(blobs)
    lap("unpack closed gaps")
    # pile-up order (= the single-GPU order, whatever the world size): rank r's records are its owned pile-ups in
    # ascending pile-up index; several pile-ups may share contig_left, so the start node alone does not order them
    pile_of = np.concatenate([np.nonzero(owner == r)[0] for r in range(world)]) if len(grec) else np.zeros(0, dtype=np.int64)
    if len(pile_of) != len(grec):   # (as merge_closed of dh_comm.cpp: an order that depends on the world size must not go unnoticed)
        raise RuntimeError("sharded_process: the ranks returned %d closed-gap records for %d owned pile-ups" % (len(grec), len(pile_of)))
    order = np.argsort(pile_of, kind="stable")
    nentries = int(piles.flat()[1].sum())
    lap("order")
    plan.close()
    lap("plan released")
    if _laps and rank == int(os.environ.get("DH_TRACE_RANK", min(1, world - 1))):   # (rank 0 pays the first-use allocations of a process)
        import sys
        print("[sharded rank %d] " % rank + ", ".join(_laps), file=sys.stderr)
    info = {"piles": len(rec), "owned": len(mine_piles), "candidates": int(ncand), "entries": nentries,
            "cropped_bytes_sent": int(sum(len(x) for x in per_dest)), "owner": owner}
    return grec[order], gbases, info


def all_gather_closed_gaps(rec, bases, rank, world, device=None):
    """Every rank receives all ranks' insertion records and consensus bases.

    Returns (records, bases, origin): records concatenated in rank order with ``cons_off``
    rebased into the concatenated ``bases``; ``origin[i]`` = rank that produced record i.
    """
    import torch.distributed as dist

    if world == 1 or not dist.is_initialized():
        return rec.copy(), bases.copy(), np.zeros(len(rec), dtype=np.int32)
    return _unpack_closed(all_gather_bytes(_pack_closed(rec, bases), world))


def _pack_closed(rec, bases):
    rb = np.frombuffer(np.ascontiguousarray(rec, dtype=INSERTION_DTYPE).tobytes(), dtype=np.uint8)
    bb = np.ascontiguousarray(bases, dtype=np.uint8)
    return np.concatenate([np.asarray([len(rb)], dtype=np.int64).view(np.uint8), rb, bb])


def _unpack_closed(blobs):
    recs, seqs, origin, base_off = [], [], [], 0
    for r, buf in enumerate(blobs):
        nr = int(buf[:8].view(np.int64)[0])
        rr = np.frombuffer(buf[8:8 + nr].tobytes(), dtype=INSERTION_DTYPE).copy()
        rr["cons_off"] += base_off
        recs.append(rr)
        seqs.append(buf[8 + nr:].copy())
        origin.append(np.full(len(rr), r, dtype=np.int32))
        base_off += len(buf) - 8 - nr
    return np.concatenate(recs), np.concatenate(seqs), np.concatenate(origin)
