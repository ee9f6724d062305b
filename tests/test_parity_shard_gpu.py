"""The sharded (multi-GPU) hot path against the single-GPU path -- bit exact.

`world` shards are emulated inside one process on one GPU (dentist_amd.parallel.emulate_ranks serves
the collectives from memory): every emulated rank maps its own block of reads, the candidate
entries are all-gathered, pile-ups are bin-packed over the ranks, cropped reads go to the owners
(all-to-all) and the closed gaps are gathered -- the result must equal the unsharded run."""
import numpy as np
import pytest

import dentist_amd
from dentist_amd import parallel, sim

pytestmark = pytest.mark.gpu


def single(ctx, w, mo, po):
    A, B = ctx.db(w.contigs), ctx.db(w.reads)
    las, trace = ctx.align_db(A, B, mo, select_best=True)
    piles = dentist_amd.Pileups(las, w.contigs.off, po)
    rec, bases = dentist_amd.process_pileups(ctx, A, B, las, trace, piles, po)
    return rec, bases


def sharded(ctx, w, mo, po, world):
    A = ctx.db(w.contigs)
    gens, keep = [], []
    for rank in range(world):
        lo, hi = parallel.shard_range(w.reads.n, rank, world)
        share = sim.SeqDb(w.reads.bases[w.reads.off[lo]:w.reads.off[hi]], w.reads.off[lo:hi + 1] - w.reads.off[lo])
        B = ctx.db(share)
        las, trace = ctx.align_db(A, B, mo, select_best=True)
        las = las.copy()
        las["bread"] += lo   # ids of the whole reads DB, as in the .las of a block
        keep.append((B, las, trace))
        gens.append(parallel.sharded_process_steps(ctx, A, B, lo, w.contigs.off, las, trace, po, rank, world))
    return parallel.emulate_ranks(gens)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_run_equals_the_single_gpu_run(gpu_ctx, world):
    w = sim.Workload(800_000, 8, 4000, 6000, seed=61, spacing=20000, gap_max=1500)
    mo = dentist_amd.default_align_opts(kmer_mod=2)
    po = dentist_amd.default_process_opts(max_reads=12)   # the quality cut is exercised: ~30 spanning reads per gap
    rec, bases = single(gpu_ctx, w, mo, po)
    assert (rec["status"] == 0).sum() >= 6
    results = sharded(gpu_ctx, w, mo, po, world)
    owners = results[0][2]["owner"]
    assert len(set(owners.tolist())) == world, "every emulated rank must own pile-ups"
    for grec, gbases, info in results:
        assert len(grec) == len(rec)
        assert np.array_equal(grec["contig_left"], rec["contig_left"])
        for f in rec.dtype.names:
            if f not in ("cons_off", "pad"):
                assert np.array_equal(grec[f], rec[f]), f
        for a, b in zip(grec, rec):
            assert np.array_equal(gbases[a["cons_off"]:a["cons_off"] + a["cons_len"]],
                                  bases[b["cons_off"]:b["cons_off"] + b["cons_len"]])
        assert info["cropped_bytes_sent"] > 0


def test_crop_then_process_equals_the_fused_call(gpu_ctx):
    """dh_crop_pileups + dh_process_cropped (also through the host: dh_cropped_create) == dh_process_pileups."""
    w = sim.Workload(500_000, 5, 2500, 6000, seed=67, spacing=20000, gap_max=1200)
    mo = dentist_amd.default_align_opts()
    po = dentist_amd.default_process_opts()
    A, B = gpu_ctx.db(w.contigs), gpu_ctx.db(w.reads)
    las, trace = gpu_ctx.align_db(A, B, mo)
    piles = dentist_amd.Pileups(las, w.contigs.off, po)
    rec, bases = dentist_amd.process_pileups(gpu_ctx, A, B, las, trace, piles, po)
    crop = dentist_amd.Cropped.crop(gpu_ctx, A, B, 0, las, trace, piles, po)
    parts = crop.arrays()
    rec2, bases2 = crop.process(gpu_ctx, A, po)
    assert np.array_equal(rec2, rec) and np.array_equal(bases2, bases)
    rec3, bases3 = dentist_amd.Cropped.create(*parts).process(gpu_ctx, A, po)
    assert np.array_equal(rec3, rec) and np.array_equal(bases3, bases)
    # cropped reads are [patch] + slice + [patch] of the reads in (pile, entry) order
    prec, cpile, centry, cread, coff, cbases = parts
    assert np.all(np.diff(cpile.astype(np.int64) * 1000 + centry) > 0)
    assert len(cpile) == sum(len(piles.get(i)[1]) for i in range(len(piles)) if prec[i]["status"] == 0)


def test_sharded_run_with_candidates_collected_while_mapping(gpu_ctx, monkeypatch):
    """bench.py's flow: dh_map_reads (filters and spanning-read candidates per chunk, records in mapping
    order) on every rank, candidates handed to the sharded collect + process; equal to the single-GPU
    flow mapping -> dh_collect_filter -> collect -> process, whatever the chunking."""
    w = sim.Workload(800_000, 8, 4000, 6000, seed=71, spacing=20000, gap_max=1500)
    mo = dentist_amd.default_align_opts(kmer_mod=2)
    po = dentist_amd.default_process_opts(max_reads=12)
    A, B = gpu_ctx.db(w.contigs), gpu_ctx.db(w.reads)
    las, trace = gpu_ctx.align_db(A, B, mo, select_best=True)
    las, _, _ = dentist_amd.collect_filter(las, w.contigs.off, w.reads.off, po)
    rec, bases = dentist_amd.process_pileups(gpu_ctx, A, B, las, trace, dentist_amd.Pileups(las, w.contigs.off, po), po)
    assert (rec["status"] == 0).sum() >= 6
    monkeypatch.setenv("DH_ALIGN_CHUNK", "700")
    # one rank: candidates of dh_map_reads + the min / max reads cut == collect on the filtered records
    l1, t1, _, c1 = gpu_ctx.map_reads(A, B, mo, po, sorted=False, candidates=True)
    rec1, bases1 = dentist_amd.process_pileups(gpu_ctx, A, B, l1, t1, c1.select(l1, po), po)
    for f in rec.dtype.names:
        if f not in ("cons_off", "pad"):
            assert np.array_equal(rec1[f], rec[f]), f
    assert np.array_equal(bases1, bases)
    world, gens, keep = 3, [], []
    for rank in range(world):
        lo, hi = parallel.shard_range(w.reads.n, rank, world)
        share = sim.SeqDb(w.reads.bases[w.reads.off[lo]:w.reads.off[hi]], w.reads.off[lo:hi + 1] - w.reads.off[lo])
        Br = gpu_ctx.db(share)
        lr, tr, _, cr = gpu_ctx.map_reads(A, Br, mo, po, sorted=False, candidates=True)
        lr = lr.copy()
        lr["bread"] += lo
        keep.append((Br, lr, tr, cr))
        gens.append(parallel.sharded_process_steps(gpu_ctx, A, Br, lo, w.contigs.off, lr, tr, po, rank, world, cr))
    for grec, gbases, info in parallel.emulate_ranks(gens):
        for f in rec.dtype.names:
            if f not in ("cons_off", "pad"):
                assert np.array_equal(grec[f], rec[f]), f
        for a, b in zip(grec, rec):
            assert np.array_equal(gbases[a["cons_off"]:a["cons_off"] + a["cons_len"]],
                                  bases[b["cons_off"]:b["cons_off"] + b["cons_len"]])


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_scaffold_graph_collector_equals_the_single_gpu_run(gpu_ctx, world):
    """bench.py's default flow for every N: mapping + filters per rank (dh_map_reads), the raw scaffold joins of a
    rank's reads all-gathered (dh_shard_read_joins), the same scaffold / gap pile-ups with extension entries on every
    rank (dh_shard_graph_plan_create), crop, all-to-all, process -- against the N = 1 flow (dh_scaffold_pileups +
    dh_scaffold_gap_pileups + select + dh_process_pileups).  DH-2 in every stage."""
    w = sim.Workload(800_000, 8, 4000, 6000, seed=73, spacing=20000, gap_max=1500)
    mo = dentist_amd.default_align_opts(kmer_mod=2, k=20, algo=1, width=64)
    po = dentist_amd.default_process_opts(max_reads=30, algo=1)   # (the cap takes spanning reads first: 30 keeps extension entries in)
    gaps = np.stack([np.arange(w.contigs.n - 1), np.arange(1, w.contigs.n)], axis=1).astype(np.int32)
    A, B = gpu_ctx.db(w.contigs), gpu_ctx.db(w.reads)
    las, trace, _, _ = gpu_ctx.map_reads(A, B, mo, po, sorted=False, candidates=True)
    gp, _ = dentist_amd.scaffold_spanning_pileups(las, w.contigs.off, w.reads.off, gaps, with_extensions=True,
                                                  min_spanning_reads=po.min_reads)
    piles = gp.select(las, po)
    tri = piles.flat()[2]
    assert (tri[:, 1] < 0).any() and (tri[:, 2] < 0).any()
    rec, bases = dentist_amd.process_pileups(gpu_ctx, A, B, las, trace, piles, po)
    assert (rec["status"] == 0).sum() >= 6
    gens, keep = [], []
    for rank in range(world):
        lo, hi = parallel.shard_range(w.reads.n, rank, world)
        share = sim.SeqDb(w.reads.bases[w.reads.off[lo]:w.reads.off[hi]], w.reads.off[lo:hi + 1] - w.reads.off[lo])
        Br = gpu_ctx.db(share)
        lr, tr, _, _ = gpu_ctx.map_reads(A, Br, mo, po, sorted=False, candidates=True)
        lr = lr.copy()
        lr["bread"] += lo
        keep.append((Br, lr, tr))
        gens.append(parallel.sharded_process_steps(gpu_ctx, A, Br, lo, w.contigs.off, lr, tr, po, rank, world,
                                                   graph=dict(read_off=share.off, input_gaps=gaps)))
    results = parallel.emulate_ranks(gens)
    assert len(set(results[0][2]["owner"].tolist())) == world
    for grec, gbases, info in results:
        assert info["entries"] == len(tri)
        for f in rec.dtype.names:
            if f not in ("cons_off", "pad"):
                assert np.array_equal(grec[f], rec[f]), f
        for a, b in zip(grec, rec):
            assert np.array_equal(gbases[a["cons_off"]:a["cons_off"] + a["cons_len"]],
                                  bases[b["cons_off"]:b["cons_off"] + b["cons_len"]])


def _same(results, rec, bases):
    for grec, gbases, info in results:
        assert len(grec) == len(rec) and np.array_equal(grec["contig_left"], rec["contig_left"])
        for f in rec.dtype.names:
            if f not in ("cons_off", "pad"):
                assert np.array_equal(grec[f], rec[f]), f
        for a, b in zip(grec, rec):
            assert np.array_equal(gbases[a["cons_off"]:a["cons_off"] + a["cons_len"]],
                                  bases[b["cons_off"]:b["cons_off"] + b["cons_len"]])


def _graph_single(ctx, w, mo, po, gaps_in):
    A, B = ctx.db(w.contigs), ctx.db(w.reads)
    las, trace, _ = ctx.map_reads(A, B, mo, po, sorted=False, candidates=False)[:3]
    gp, _ = dentist_amd.scaffold_spanning_pileups(las, w.contigs.off, w.reads.off, gaps_in, with_extensions=True,
                                                  min_spanning_reads=po.min_reads)
    return dentist_amd.process_pileups(ctx, A, B, las, trace, gp.select(las, po), po)


@pytest.mark.parametrize("world,collector", [(2, "graph"), (3, "graph"), (2, "spanning")])
def test_dh_shard_run_between_host_threads_equals_the_single_gpu_run(world, collector):
    """The C-ABI entry of the multi-GPU path (dh_comm_create_local + dh_shard_run): every rank is a host thread with its
    own context on the one GPU, the exchanges go through the in-process hub -- the same dh_shard_run code that runs over
    RCCL between processes.  Scaffold-graph collector (the benched one) and spanning-read collector."""
    import threading
    w = sim.Workload(800_000, 8, 4000, 6000, seed=61, spacing=20000, gap_max=1500)
    mo = dentist_amd.default_align_opts(kmer_mod=4, k=20, width=64, xdrop=60, algo=1)
    po = dentist_amd.default_process_opts(max_reads=12, algo=1)
    gaps_in = np.stack([np.arange(w.contigs.n - 1), np.arange(1, w.contigs.n)], axis=1).astype(np.int32)
    ctx0 = dentist_amd.Context(0)
    if collector == "graph":
        rec, bases = _graph_single(ctx0, w, mo, po, gaps_in)
    else:
        A, B = ctx0.db(w.contigs), ctx0.db(w.reads)
        las, trace, _, cands = ctx0.map_reads(A, B, mo, po, sorted=False, candidates=True)
        rec, bases = dentist_amd.process_pileups(ctx0, A, B, las, trace, cands.select(las, po), po)
    assert (rec["status"] == 0).sum() >= 6
    ctxs = [dentist_amd.Context(0) for _ in range(world)]
    comms = dentist_amd.Comm.local(world, ctxs)
    results, errors = [None] * world, []

    def run(rank):
        try:
            ctx = ctxs[rank]
            lo, hi = parallel.shard_range(w.reads.n, rank, world)
            share = sim.SeqDb(w.reads.bases[w.reads.off[lo]:w.reads.off[hi]], w.reads.off[lo:hi + 1] - w.reads.off[lo])
            A, B = ctx.db(w.contigs), ctx.db(share)
            m = ctx.map_reads(A, B, mo, po, sorted=False, candidates=collector != "graph")
            las, trace = m[0].copy(), m[1]
            las["bread"] += lo
            if collector == "graph":
                results[rank] = dentist_amd.shard_run(comms[rank], A, B, lo, w.contigs.off, las, trace, po,
                                                      graph=dict(read_off=share.off, input_gaps=gaps_in))
            else:
                results[rank] = dentist_amd.shard_run(comms[rank], A, B, lo, w.contigs.off, las, trace, po, cands=m[3])
        except Exception as e:  # noqa: BLE001
            errors.append((rank, repr(e)))
            raise
    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not errors, errors
    _same(results, rec, bases)
    assert sum(r[2]["owned"] for r in results) == results[0][2]["piles"] and all(r[2]["owned"] > 0 for r in results)
    for c in comms:
        c.close()


def test_dh_shard_run_with_alignment_chains_equals_the_single_gpu_run():
    """Reads with a 2-5 kb indel next to a gap map as CHAINS (START + NEXT records, dazzler.d:1728-1758); the sharded
    collector has to treat them as the single-GPU one does: joins collected on one unit per chain, every member of the
    chain in the join blob (dh_shard_read_joins), the members laid out behind the first record on the receiving side
    (dh_shard_graph_plan_create) so that the cropper translates the crop point through the member that covers it.  With
    single records these reads would be dropped or cropped through the wrong member (round-4 advisor finding)."""
    import threading
    from helpers import plant_long_indels
    w = sim.Workload(600_000, 6, 1500, 12_000, seed=20260930, spacing=60000, gap_max=1500)
    reads, planted = plant_long_indels(w, np.random.default_rng(5))
    assert len(planted) >= 12
    mo = dentist_amd.default_align_opts(kmer_mod=4, k=20, width=64, xdrop=60, algo=1)
    po = dentist_amd.default_process_opts(algo=1, rounds=2, max_reads=0)
    gaps_in = np.stack([np.arange(w.contigs.n - 1), np.arange(1, w.contigs.n)], axis=1).astype(np.int32)
    ctx0 = dentist_amd.Context(0)
    A, B = ctx0.db(w.contigs), ctx0.db(reads)
    las, trace, _ = ctx0.map_reads(A, B, mo, po, sorted=False, candidates=False)[:3]
    gp, _ = dentist_amd.scaffold_spanning_pileups(las, w.contigs.off, reads.off, gaps_in, with_extensions=True,
                                                  min_spanning_reads=po.min_reads)
    piles = gp.select(las, po)
    tri = piles.flat()[2]
    nxt = lambda i: (i >= 0) & (i + 1 < len(las)) & ((las["flags"][np.minimum(i + 1, len(las) - 1)] & 0xC) == 0x8)  # noqa: E731
    assert int(nxt(tri[:, 1]).sum() + nxt(tri[:, 2]).sum()) >= 10, "the planted reads must enter the pile-ups as chains"
    rec, bases = dentist_amd.process_pileups(ctx0, A, B, las, trace, piles, po)
    assert (rec["status"] == 0).sum() >= 5
    for world in (2, 3):
        ctxs = [dentist_amd.Context(0) for _ in range(world)]
        comms = dentist_amd.Comm.local(world, ctxs)
        results, errors = [None] * world, []

        def run(rank):
            try:
                ctx = ctxs[rank]
                lo, hi = parallel.shard_range(reads.n, rank, world)
                share = sim.SeqDb(reads.bases[reads.off[lo]:reads.off[hi]], reads.off[lo:hi + 1] - reads.off[lo])
                Ar, Br = ctx.db(w.contigs), ctx.db(share)
                m = ctx.map_reads(Ar, Br, mo, po, sorted=False, candidates=False)
                lr, tr = m[0].copy(), m[1]
                lr["bread"] += lo
                results[rank] = dentist_amd.shard_run(comms[rank], Ar, Br, lo, w.contigs.off, lr, tr, po,
                                                      graph=dict(read_off=share.off, input_gaps=gaps_in))
            except Exception as e:  # noqa: BLE001
                errors.append((rank, repr(e)))
                raise
        threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=600)
        assert not errors, errors
        _same(results, rec, bases)
        for c in comms:
            c.close()


def _scrambled_single(ctx, w, contigs, mo, po):
    A, B = ctx.db(contigs), ctx.db(w.reads)
    las, trace, _ = ctx.map_reads(A, B, mo, po, sorted=False, candidates=False)[:3]
    piles, _ = dentist_amd.scaffold_all_pileups(las, contigs.off, w.reads.off, None, only="both", min_spanning_reads=po.min_reads)
    return dentist_amd.process_pileups(ctx, A, B, las, trace, piles.select(las, po), po)


@pytest.mark.parametrize("world,how", [(2, "steps"), (3, "steps"), (2, "dh_shard_run"), (3, "dh_shard_run")])
def test_sharded_general_joins_equal_the_single_gpu_run(gpu_ctx, world, how):
    """`process --batch` hands ANY pile-up to any job (snakemake/Snakefile:1315-1334, processPileUps/package.d:146-159): the
    sharded path with dh_scaffold_opts.only_joins = 3 plans every pile-up of the scaffold graph -- here the scrambled
    assembly of tests/test_parity_joins_gpu.py: an anti-parallel join (a reverse-complemented contig), joins that skip
    contig ids (two contigs in swapped order) and the two extension pile-ups at the trimmed ends -- , every rank crops its
    reads of them, owners process them; the result equals the single-GPU run (dh_scaffold_all_pileups + select +
    dh_process_pileups) record for record, join for join, base for base.  Through parallel.sharded_process_steps (the
    exchanges served from memory) and through dh_shard_run between host threads (the C ABI's own sequence)."""
    import threading
    from test_parity_joins_gpu import scrambled_assembly
    w = sim.Workload(500_000, 4, 1400, 9000, seed=20260931, spacing=70000, gap_max=1200)
    contigs, _ = scrambled_assembly(w)
    mo = dentist_amd.default_align_opts(kmer_mod=4, k=20, width=64, xdrop=60, algo=1)
    po = dentist_amd.default_process_opts(algo=1, rounds=2, max_reads=0)
    rec, bases = _scrambled_single(gpu_ctx, w, contigs, mo, po)
    assert (rec["status"] == 0).sum() == 6
    assert (rec["join"] != 0).any() and (rec["contig_right"] < 0).sum() == 2 and (rec["contig_right"] > rec["contig_left"] + 1).any()
    graph = lambda off: dict(read_off=off, input_gaps=None, only_joins=3, min_spanning_reads=po.min_reads)  # noqa: E731
    if how == "steps":
        A = gpu_ctx.db(contigs)
        gens, keep = [], []
        for rank in range(world):
            lo, hi = parallel.shard_range(w.reads.n, rank, world)
            share = sim.SeqDb(w.reads.bases[w.reads.off[lo]:w.reads.off[hi]], w.reads.off[lo:hi + 1] - w.reads.off[lo])
            Br = gpu_ctx.db(share)
            lr, tr, _ = gpu_ctx.map_reads(A, Br, mo, po, sorted=False, candidates=False)[:3]
            lr = lr.copy()
            lr["bread"] += lo
            keep.append((Br, lr, tr))
            gens.append(parallel.sharded_process_steps(gpu_ctx, A, Br, lo, contigs.off, lr, tr, po, rank, world, graph=graph(share.off)))
        results = parallel.emulate_ranks(gens)
    else:
        ctxs = [dentist_amd.Context(0) for _ in range(world)]
        comms = dentist_amd.Comm.local(world, ctxs)
        results, errors = [None] * world, []

        def run(rank):
            try:
                ctx = ctxs[rank]
                lo, hi = parallel.shard_range(w.reads.n, rank, world)
                share = sim.SeqDb(w.reads.bases[w.reads.off[lo]:w.reads.off[hi]], w.reads.off[lo:hi + 1] - w.reads.off[lo])
                A, B = ctx.db(contigs), ctx.db(share)
                m = ctx.map_reads(A, B, mo, po, sorted=False, candidates=False)
                las, trace = m[0].copy(), m[1]
                las["bread"] += lo
                results[rank] = dentist_amd.shard_run(comms[rank], A, B, lo, contigs.off, las, trace, po, graph=graph(share.off))
            except Exception as e:  # noqa: BLE001
                errors.append((rank, repr(e)))
                raise
        threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=600)
        assert not errors, errors
        for c in comms:
            c.close()
    for grec, gbases, info in results:
        assert len(grec) == len(rec)
        for f in rec.dtype.names:
            if f not in ("cons_off", "pad"):
                assert np.array_equal(grec[f], rec[f]), f
        for a, b in zip(grec, rec):
            assert np.array_equal(gbases[a["cons_off"]:a["cons_off"] + a["cons_len"]], bases[b["cons_off"]:b["cons_off"] + b["cons_len"]])


def test_dh_shard_run_over_rccl_at_world_one(gpu_ctx):
    """The RCCL back end of the same entry, executable on a one-GPU box: a communicator of one rank (ncclCommInitRank
    with world 1) -- ncclAllGather of the sizes and of the padded blobs, grouped ncclSend / ncclRecv to itself -- must
    give the single-GPU result.  (World sizes above one run this code only on a multi-GPU node: the driver's SCALE
    run; the exchange logic above one rank is what the host-thread test covers.)"""
    w = sim.Workload(600_000, 6, 3000, 6000, seed=67, spacing=20000, gap_max=1500)
    mo = dentist_amd.default_align_opts(kmer_mod=4, k=20, width=64, xdrop=60, algo=1)
    po = dentist_amd.default_process_opts(algo=1)
    gaps_in = np.stack([np.arange(w.contigs.n - 1), np.arange(1, w.contigs.n)], axis=1).astype(np.int32)
    rec, bases = _graph_single(gpu_ctx, w, mo, po, gaps_in)
    comm = dentist_amd.Comm.create(gpu_ctx, 0, 1, dentist_amd.Comm.unique_id())
    # the collectives on ragged payloads first
    got = comm.all_gather(np.arange(1000, dtype=np.uint8))
    assert len(got) == 1 and np.array_equal(got[0], np.arange(1000, dtype=np.uint8))
    got = comm.all_to_all([np.arange(77, dtype=np.uint8)])
    assert len(got) == 1 and np.array_equal(got[0], np.arange(77, dtype=np.uint8))
    assert len(comm.all_gather(np.zeros(0, np.uint8))[0]) == 0
    A, B = gpu_ctx.db(w.contigs), gpu_ctx.db(w.reads)
    las, trace, _ = gpu_ctx.map_reads(A, B, mo, po, sorted=False, candidates=False)[:3]
    res = dentist_amd.shard_run(comm, A, B, 0, w.contigs.off, las, trace, po, graph=dict(read_off=w.reads.off, input_gaps=gaps_in))
    _same([res], rec, bases)
    assert res[2]["owned"] == res[2]["piles"] == len(rec)
    comm.close()


def test_collectives_with_empty_ragged_and_large_payloads_hub_and_rccl(gpu_ctx):
    """Readiness of the exchange code for the first real multi-GPU run: the two ragged collectives of the C ABI with empty
    payloads, ranks that send or receive nothing at all, ragged sizes and a payload beyond 2 GB (64-bit sizes and
    offsets) -- through the in-process hub at world 8 (and 2, for the large one) and, blob by blob, through the RCCL back
    end at world 1 (ncclAllGather of the sizes and of the padded blobs, grouped ncclSend / ncclRecv): identical bytes.
    A size-0 peer or a 2 GB offset must not be what the first SCALE run dies of."""
    import threading
    world = 8
    rng = np.random.default_rng(11)
    sizes = [0, 1, 4097, 0, 123457, 64, 1 << 20, 3]
    payload = [rng.integers(0, 256, n, dtype=np.uint8) for n in sizes]
    per_dest = [[rng.integers(0, 256, 0 if (src == 3 or dst == 5) else int(rng.integers(0, 3)) * int(rng.integers(1, 70000)),
                              dtype=np.uint8) for dst in range(world)] for src in range(world)]
    comms = dentist_amd.Comm.local(world)
    got_g, got_a, errors = [None] * world, [None] * world, []

    def run(r):
        try:
            for _ in range(2):   # the hub is reusable
                got_g[r] = comms[r].all_gather(payload[r])
                got_a[r] = comms[r].all_to_all(per_dest[r])
        except Exception as e:  # noqa: BLE001
            errors.append((r, repr(e)))
            raise
    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not errors, errors
    for r in range(world):
        assert [len(x) for x in got_g[r]] == sizes
        for s in range(world):
            assert np.array_equal(got_g[r][s], payload[s]) and np.array_equal(got_a[r][s], per_dest[s][r])
    assert all(len(x) == 0 for x in got_a[5]) and all(len(got_a[r][3]) == 0 for r in range(world))
    for c in comms:
        c.close()
    # the same blobs through RCCL at world 1
    comm = dentist_amd.Comm.create(gpu_ctx, 0, 1, dentist_amd.Comm.unique_id())
    for r in range(world):
        g = comm.all_gather(payload[r])
        assert len(g) == 1 and np.array_equal(g[0], got_g[0][r])
        a = comm.all_to_all([per_dest[r][r]])
        assert len(a) == 1 and np.array_equal(a[0], got_a[r][r])
    # beyond 2 GB: sizes and offsets are 64 bits wide on both back ends
    big = np.arange((1 << 31) + 200_000_003, dtype=np.uint32).view(np.uint8)[:(1 << 31) + 200_000_003].copy()
    g = comm.all_gather(big)
    assert len(g) == 1 and len(g[0]) == len(big) and np.array_equal(g[0][-1000:], big[-1000:]) and np.array_equal(g[0][::4099], big[::4099])
    a = comm.all_to_all([big])
    assert len(a[0]) == len(big) and np.array_equal(a[0][::4099], big[::4099])
    del g, a
    comm.close()
    two = dentist_amd.Comm.local(2)
    out = [None, None]

    def run2(r):
        out[r] = two[r].all_gather(big if r == 0 else payload[2])
    ts = [threading.Thread(target=run2, args=(r,)) for r in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=600)
    for r in range(2):
        assert len(out[r][0]) == len(big) and np.array_equal(out[r][0][::4099], big[::4099]) and np.array_equal(out[r][1], payload[2])
    for c in two:
        c.close()
