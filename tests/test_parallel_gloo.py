"""The N > 1 path on CPU: world_size 2, gloo backend (the same code runs over RCCL on GPUs)."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from dentist_amd import INSERTION_DTYPE
    from dentist_amd.parallel import all_gather_closed_gaps, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # rank r "closed" r + 2 gaps with consensus lengths 10 * (i + 1) + r
    n = rank + 2
    rec = np.zeros(n, dtype=INSERTION_DTYPE)
    bases, off = [], 0
    for i in range(n):
        ln = 10 * (i + 1) + rank
        rec[i]["contig_left"] = 100 * rank + i
        rec[i]["cons_len"] = ln
        rec[i]["cons_off"] = off
        bases.append(np.full(ln, (rank + i) % 4, dtype=np.uint8))
        off += ln
    allrec, allbases, origin = all_gather_closed_gaps(rec, np.concatenate(bases), rank, world)
    lo, hi = shard_range(10, rank, world)
    q.put((rank, allrec.tobytes(), allbases.tobytes(), origin.tolist(), (lo, hi)))
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_closed_gaps_world2():
    from dentist_amd import INSERTION_DTYPE
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    out.sort()
    assert out[0][1] == out[1][1] and out[0][2] == out[1][2], "ranks disagree on the gathered result"
    rec = np.frombuffer(out[0][1], dtype=INSERTION_DTYPE)
    bases = np.frombuffer(out[0][2], dtype=np.uint8)
    assert out[0][3] == [0, 0, 1, 1, 1]
    assert rec["contig_left"].tolist() == [0, 1, 100, 101, 102]
    for r, org in zip(rec, out[0][3]):
        i = int(r["contig_left"]) % 100
        seq = bases[r["cons_off"]:r["cons_off"] + r["cons_len"]]
        assert len(seq) == 10 * (i + 1) + org and np.all(seq == (org + i) % 4)
    assert out[0][4] == (0, 5) and out[1][4] == (5, 10)


def _worker_bytes(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from dentist_amd.parallel import all_gather_bytes, all_to_all_bytes
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = np.arange(7 * rank + 3, dtype=np.uint8) + rank          # ragged; rank 0 sends 3, rank 1 sends 10 bytes
    gathered = all_gather_bytes(mine, world)
    per_dest = [np.full(5 * rank + dst, 10 * rank + dst, dtype=np.uint8) for dst in range(world)]  # (0,0) is empty
    got = all_to_all_bytes(per_dest, world)
    q.put((rank, [g.tolist() for g in gathered], [g.tolist() for g in got]))
    dist.barrier()
    dist.destroy_process_group()


def test_ragged_all_gather_and_all_to_all_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bytes, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, gathered, got in out:
        assert gathered == [list(range(3)), [x + 1 for x in range(10)]]
        assert got == [[10 * src + rank] * (5 * src + rank) for src in range(2)]


def test_owner_assignment_and_candidate_merge_are_deterministic():
    from dentist_amd import LA_DTYPE
    from dentist_amd.parallel import CAND_DTYPE, assign_owners, emulate_ranks, merge_candidates
    owner = assign_owners([5, 9, 9, 1, 4, 4], 3)
    assert owner.tolist() == [2, 0, 1, 1, 2, 0]     # largest first: 9->r0, 9->r1, 5->r2, 4->r2, 4->r0, 1->r1
    loads = [sum(c for c, o in zip([5, 9, 9, 1, 4, 4], owner) if o == r) for r in range(3)]
    assert max(loads) - min(loads) <= 4
    # candidates of two ranks (rank 1 holds the higher read ids); gap 7 appears on both
    def cand(gap, read, ab):
        c = np.zeros(1, dtype=CAND_DTYPE)
        c["gap"], c["read"] = gap, read
        c["L"]["abpos"], c["R"]["abpos"] = ab, ab + 1
        return c
    r0 = np.concatenate([cand(3, 1, 10), cand(7, 0, 20), cand(7, 2, 30)])
    r1 = np.concatenate([cand(7, 5, 40), cand(9, 6, 50)])
    las, gaps, counts, triples = merge_candidates([r0, r1])
    assert gaps.tolist() == [3, 7, 9] and counts.tolist() == [1, 3, 1] and las.dtype == LA_DTYPE
    assert triples[1:4, 0].tolist() == [0, 2, 5]     # read-id order across the ranks
    for read, il, ir in triples.tolist():
        assert ir == il + 1 and las[ir]["abpos"] == las[il]["abpos"] + 1
    # the lockstep driver: two generators exchanging through "collectives"
    def gen(rank):
        got = yield ("all_gather", np.asarray([rank + 1], dtype=np.uint8))
        s = sum(int(g[0]) for g in got)
        got = yield ("all_to_all", [np.asarray([10 * rank + d], dtype=np.uint8) for d in range(2)])
        return s, [int(g[0]) for g in got]
    assert emulate_ranks([gen(0), gen(1)]) == [(3, [0, 10]), (3, [1, 11])]


def test_cxx_shard_glue_equals_the_numpy_restatement():
    """dh_shard_pack_candidates / dh_shard_plan_create / dh_shard_pack_cropped / dh_shard_unpack_cropped (the host
    work of a rank between the collectives, C++) against pack_candidates / merge_candidates / select / pile_costs /
    assign_owners and the Python packing (the restatement in dentist_amd/parallel.py) on random mappings."""
    import dentist_amd
    from dentist_amd import INSERTION_DTYPE, LA_DTYPE, Cropped, Pileups
    from dentist_amd._lib import ShardPlan, shard_pack_candidates, shard_pack_cropped, shard_unpack_cropped
    from dentist_amd.parallel import (CAND_DTYPE, CROP_DTYPE, assign_owners, merge_candidates, pack_candidates, pile_costs)
    rng = np.random.default_rng(7)
    ncontigs, world = 9, 3
    clen = rng.integers(20000, 40000, ncontigs)
    coff = np.concatenate([[0], np.cumsum(clen)]).astype(np.int64)
    po = dentist_amd.default_process_opts(max_reads=5)
    per_rank, blobs = [], []
    for rank in range(world):
        las = []
        for rd in range(40):
            g = int(rng.integers(0, ncontigs - 1))
            comp = int(rng.integers(0, 2))
            anchor = int(rng.integers(600, 5000))
            L = np.zeros(1, dtype=LA_DTYPE)
            R = np.zeros(1, dtype=LA_DTYPE)
            L["aread"], L["bread"], L["flags"] = g, rd, comp
            L["abpos"], L["aepos"] = clen[g] - anchor, clen[g] - int(rng.integers(0, 50))
            L["bbpos"], L["bepos"], L["diffs"] = 10, 10 + anchor, int(rng.integers(0, anchor // 5))
            R["aread"], R["bread"], R["flags"] = g + 1, rd, comp
            R["abpos"], R["aepos"] = int(rng.integers(0, 50)), anchor
            R["bbpos"], R["bepos"], R["diffs"] = 10 + anchor + int(rng.integers(0, 3000)), 30000, int(rng.integers(0, anchor // 5))
            las += [L, R]
        las = np.concatenate(las)
        c = Pileups(las, coff, po, candidates=True)
        shift = 40 * rank
        exp = pack_candidates(c, las, shift)
        got = shard_pack_candidates(c, las, shift)
        assert got.tobytes() == exp.view(np.uint8).tobytes() and len(exp) > 20
        per_rank.append(exp)
        blobs.append(got)
    glas, gaps, counts, triples = merge_candidates(per_rank)
    piles = Pileups.from_flat(gaps, counts, triples).select(glas, po)
    owner = assign_owners(pile_costs(piles, glas), world)
    plan = ShardPlan(blobs, po)
    assert np.array_equal(plan.las, glas) and np.array_equal(plan.owner, owner) and len(set(owner.tolist())) == world
    a, b = plan.piles.flat(), piles.flat()
    assert all(np.array_equal(x, y) for x, y in zip(a, b)) and a[1].max() == 5
    # cropped reads to their owners and back: every owner ends up with its pile-ups' reads in (pile, entry) order
    npl = len(piles)
    rec = np.zeros(npl, dtype=INSERTION_DTYPE)
    rec["contig_left"] = a[0]
    crops = []
    for rank in range(world):   # rank r "cropped" the entries whose read id falls into its range
        pile, entry, read, seqs = [], [], [], []
        at = 0
        for p_, n in enumerate(a[1]):
            for e in range(n):
                rd = int(a[2][at + e][0])
                if rd // 40 == rank:
                    pile.append(p_)
                    entry.append(e)
                    read.append(rd)
                    seqs.append(rng.integers(0, 4, int(rng.integers(20, 60))).astype(np.uint8))
            at += n
        off = np.concatenate([[0], np.cumsum([len(x) for x in seqs])]).astype(np.int64)
        kind = [(p_ + e + r_) % 3 for p_, e, r_ in zip(pile, entry, read)]      # 0 spanning, 1 / 2 extension entries
        crops.append((Cropped.create(rec, pile, entry, read, off, np.concatenate(seqs), kind=kind), pile, entry, read, seqs, kind))
    sent = [shard_pack_cropped(c[0], owner, world) for c in crops]
    for src, (c, pile, entry, read, seqs, kind) in enumerate(crops):   # blob format = the Python packing
        for dst in range(world):
            sel = [i for i in range(len(pile)) if owner[pile[i]] == dst]
            head = np.zeros(len(sel), dtype=CROP_DTYPE)
            head["pile"], head["read"] = [pile[i] for i in sel], [read[i] for i in sel]
            head["entry"] = [entry[i] | (kind[i] << 28) for i in sel]        # the entry's kind rides in the top bits
            head["len"] = [len(seqs[i]) for i in sel]
            exp = np.concatenate([np.asarray([len(sel)], dtype=np.int64).view(np.uint8), head.view(np.uint8)] + [seqs[i] for i in sel])
            assert sent[src][dst].tobytes() == exp.tobytes()
    for dst in range(world):
        own = shard_unpack_cropped([sent[src][dst] for src in range(world)], rec, owner, dst)
        orec, opile, oentry, oread, ooff, obases = own.arrays()
        mine = np.nonzero(owner == dst)[0]
        assert np.array_equal(orec["contig_left"], rec["contig_left"][mine])
        assert np.all(np.diff(opile.astype(np.int64) * 1000 + oentry) > 0)
        want = sum(int(a[1][p_]) for p_ in mine)
        assert len(opile) == want and ooff[-1] == len(obases)
        assert np.array_equal(own.kind(), (mine[opile] + oentry + oread) % 3)
    plan.close()


def test_sharded_scaffold_collector_equals_the_single_rank_builder():
    """dh_shard_read_joins per rank + dh_shard_graph_plan_create on the gathered joins against dh_scaffold_pileups +
    dh_scaffold_gap_pileups + dh_pileups_select on the merged alignments (the N = 1 path of bench.py): the same pile-ups,
    entry by entry (read id, kind of entry, the LA records behind it), and kinds survive the cropped-read blobs."""
    import dentist_amd
    from dentist_amd import sim
    from dentist_amd._lib import ShardPlan, shard_read_joins
    from oracle import pyoracle as oz
    w = sim.Workload(160_000, 6, 420, 5000, seed=77, spacing=20000, gap_min=50, gap_max=1500)
    o = oz.default_opts(k=14, kmer_mod=1)
    las, _, _ = oz.align_db(w.contigs, w.reads, o, nthreads=4, sort=False, select_best=True)
    las = np.ascontiguousarray(las[np.argsort(las["bread"], kind="stable")])
    coff, roff = w.contigs.off, w.reads.off
    gaps = np.stack([np.arange(w.contigs.n - 1), np.arange(1, w.contigs.n)], axis=1).astype(np.int32)
    # the same mapping with ALIGNMENT CHAINS (damapper's START / NEXT records, dazzler.d:1728-1758): every third long
    # alignment is cut into two chained records around a 40-base hole, as a read with a long indel would be reported
    parts = []
    nchained = 0
    for i, l in enumerate(las):
        if i % 3 == 0 and l["aepos"] - l["abpos"] > 2500 and not (l["flags"] & 0x20):
            a, b = np.array([l, l])
            ma, mb = (l["abpos"] + l["aepos"]) // 2, (l["bbpos"] + l["bepos"]) // 2
            a["aepos"], a["bepos"], a["diffs"] = ma - 20, mb - 20, l["diffs"] // 2
            b["abpos"], b["bbpos"], b["diffs"] = ma + 20, mb + 20, l["diffs"] - l["diffs"] // 2
            a["flags"] = (int(l["flags"]) & 0xFFFFFFF7) | 0x4
            b["flags"] = (int(l["flags"]) & 0xFFFFFFFB) | 0x8
            parts += [a, b]
            nchained += 1
        else:
            parts.append(l)
    las_chained = np.ascontiguousarray(np.array(parts, dtype=las.dtype))
    assert nchained > 50
    nchain_seen = 0
    for max_reads, las in ((0, las), (6, las), (0, las_chained), (6, las_chained)):
        chained = las is las_chained
        po = dentist_amd.default_process_opts(max_reads=max_reads)
        gp, _ = dentist_amd.scaffold_spanning_pileups(las, coff, roff, gaps, with_extensions=True, min_spanning_reads=po.min_reads)
        ecl, ecnt, etri = gp.select(las, po).flat()
        assert len(ecl) >= 4 and (etri[:, 1] < 0).any() and (etri[:, 2] < 0).any()   # extension entries of both kinds
        for world in (1, 2, 3):
            blobs = []
            for r in range(world):
                lo, hi = w.reads.n * r // world, w.reads.n * (r + 1) // world
                mine = las[(las["bread"] >= lo) & (las["bread"] < hi)]
                blobs.append(shard_read_joins(mine, coff, roff[lo:hi + 1] - roff[lo], lo))
            plan = ShardPlan(blobs, po, graph=(w.contigs.n, gaps, {"min_spanning_reads": po.min_reads}))
            cl, cnt, tri = plan.piles.flat()
            assert np.array_equal(cl, ecl) and np.array_equal(cnt, ecnt) and np.array_equal(tri[:, 0], etri[:, 0])
            for col in (1, 2):
                assert np.array_equal(tri[:, col] < 0, etri[:, col] < 0)
                m = tri[:, col] >= 0
                assert np.array_equal(plan.las[tri[m, col]], las[etri[m, col]])
                if chained:   # the members of an entry's chain follow its first record on both sides, and nothing else does
                    nxt_e = np.minimum(etri[m, col] + 1, len(las) - 1)
                    nxt_p = np.minimum(tri[m, col] + 1, len(plan.las) - 1)
                    cont = lambda L, i, j: ((L["flags"][j] & 0xC) == 0x8) & (L["aread"][i] == L["aread"][j]) & \
                        (L["bread"][i] == L["bread"][j]) & ((L["flags"][i] & 1) == (L["flags"][j] & 1)) & (i != j)  # noqa: E731
                    ce, cp = cont(las, etri[m, col], nxt_e), cont(plan.las, tri[m, col], nxt_p)
                    assert np.array_equal(ce, cp)
                    nchain_seen += int(ce.sum())
                    assert np.array_equal(plan.las[nxt_p[cp]], las[nxt_e[ce]])
            assert len(plan.owner) == len(cl) and set(plan.owner.tolist()) <= set(range(world))
            plan.close()
    assert nchain_seen > 20   # entries whose chain has a second member, on both sides


def test_shard_graph_glue_rejects_malformed_input():
    import pytest
    """Error behaviour of the sharded collector's entry points: bad read ranges, truncated or corrupt join blobs."""
    import dentist_amd
    from dentist_amd import LA_DTYPE
    from dentist_amd._lib import ShardPlan, shard_read_joins
    coff = np.asarray([0, 5000, 9000], dtype=np.int64)
    roff = np.asarray([0, 3000, 6000], dtype=np.int64)
    la = np.zeros(2, dtype=LA_DTYPE)
    la["aread"], la["bread"] = [0, 1], [7, 7]           # read 7 of a rank that holds reads [5, 7)
    la["abpos"], la["aepos"], la["bbpos"], la["bepos"] = [4000, 0], [5000, 1200], [0, 1500], [1000, 2700]
    with pytest.raises(dentist_amd.DhError):
        shard_read_joins(la, coff, roff, 5)
    la["bread"] = 6
    blob = shard_read_joins(la, coff, roff, 5)
    assert (len(blob) - 16) % 128 == 0 and len(blob) > 16   # JoinHead + JoinRec records, no chain members here
    po = dentist_amd.default_process_opts()
    with pytest.raises(dentist_amd.DhError):
        ShardPlan([blob[:-8]], po, graph=(2, None, {}))             # not a whole number of records
    bad = blob.copy()
    bad[16:20] = np.asarray([99], dtype=np.int32).view(np.uint8)    # edge names a contig outside the assembly
    with pytest.raises(dentist_amd.DhError):
        ShardPlan([bad], po, graph=(2, None, {}))
    plan = ShardPlan([blob], po, graph=(2, None, {"min_spanning_reads": 1}))
    assert len(plan.piles) <= 1
    plan.close()


def test_in_process_hub_collectives_of_the_c_abi():
    """dh_comm_create_local: the communicator of the C ABI between host threads (no GPU, no RCCL): ragged all-gather(v)
    and all-to-all(v) of byte blobs, empty payloads included -- the exchange code dh_shard_run runs on."""
    import threading
    import dentist_amd
    world = 3
    comms = dentist_amd.Comm.local(world)
    rng = np.random.default_rng(5)
    payload = [rng.integers(0, 256, n, dtype=np.uint8) for n in (1000, 0, 37)]
    per_dest = [[rng.integers(0, 256, (src * 7 + dst * 13) % 50 * (src != dst), dtype=np.uint8) for dst in range(world)]
                for src in range(world)]
    got_g, got_a = [None] * world, [None] * world

    def run(r):
        for _ in range(3):   # the hub is reusable
            got_g[r] = comms[r].all_gather(payload[r])
            got_a[r] = comms[r].all_to_all(per_dest[r])
    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=60)
    for r in range(world):
        assert all(np.array_equal(got_g[r][s], payload[s]) for s in range(world))
        assert all(np.array_equal(got_a[r][s], per_dest[s][r]) for s in range(world))
    for c in comms:
        c.close()


def test_sharded_plan_keeps_copies_of_one_record_apart():
    """A read that runs through a short contig contributes that contig's alignment to TWO joins (into it and out of
    it); the gathered records hold a copy per join, one behind the other.  When that record is flagged NEXT without START
    (the rest of a chain whose head a filter removed), the second copy would read as the continuation of the first:
    dh_shard_graph_plan_create puts a separator record between them.  Fabricated alignments (no sequence is read):
    the plan of 1-3 ranks against the single-rank builder, record by record, and nothing follows an entry's record that
    does not follow it in the input."""
    import dentist_amd
    from dentist_amd._lib import LA_DTYPE, ShardPlan, shard_read_joins
    lens = np.array([30000, 30000, 2000, 30000, 30000, 1800, 30000], dtype=np.int64)   # contigs 2 and 5 are short
    gap, LR = 1000, 15000
    cstart = np.concatenate([[0], np.cumsum(lens + gap)[:-1]])
    rng = np.random.default_rng(11)
    starts = np.sort(rng.integers(0, int(cstart[-1] + lens[-1]) - LR, 900))
    recs = []
    for r, p in enumerate(starts):
        for c in range(len(lens)):
            lo, hi = max(p, cstart[c]), min(p + LR, cstart[c] + lens[c])
            if hi - lo < 500:
                continue
            la = np.zeros(1, dtype=LA_DTYPE)
            la["aread"], la["bread"] = c, r
            la["abpos"], la["aepos"] = lo - cstart[c], hi - cstart[c]
            la["bbpos"], la["bepos"] = lo - p, hi - p
            la["diffs"] = (hi - lo) // 8
            # the alignment of a whole short contig inside a read: flagged as the rest of a chain
            if lens[c] < 5000 and lo == cstart[c] and hi == cstart[c] + lens[c]:
                la["flags"] = 0x8
            recs.append(la)
    las = np.ascontiguousarray(np.concatenate(recs))
    assert int((las["flags"] == 0x8).sum()) > 20
    coff = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    roff = (np.arange(len(starts) + 1, dtype=np.int64) * LR)
    gaps = np.stack([np.arange(len(lens) - 1), np.arange(1, len(lens))], axis=1).astype(np.int32)
    po = dentist_amd.default_process_opts(max_reads=0)
    gp, _ = dentist_amd.scaffold_spanning_pileups(las, coff, roff, gaps, with_extensions=True, min_spanning_reads=po.min_reads)
    ecl, ecnt, etri = gp.select(las, po).flat()
    assert set(ecl.tolist()) >= {1, 2, 4, 5}   # the joins into and out of both short contigs
    cont = lambda L, i, j: ((L["flags"][j] & 0xC) == 0x8) & (L["aread"][i] == L["aread"][j]) & \
        (L["bread"][i] == L["bread"][j]) & ((L["flags"][i] & 1) == (L["flags"][j] & 1)) & (i != j)  # noqa: E731
    for world in (1, 2, 3):
        blobs = []
        for r in range(world):
            lo, hi = len(starts) * r // world, len(starts) * (r + 1) // world
            mine = np.ascontiguousarray(las[(las["bread"] >= lo) & (las["bread"] < hi)])
            blobs.append(shard_read_joins(mine, coff, roff[lo:hi + 1] - roff[lo], lo))
        plan = ShardPlan(blobs, po, graph=(len(lens), gaps, {"min_spanning_reads": po.min_reads}))
        cl, cnt, tri = plan.piles.flat()
        assert np.array_equal(cl, ecl) and np.array_equal(cnt, ecnt) and np.array_equal(tri[:, 0], etri[:, 0])
        nsep = int(((plan.las["aread"] == -1) & (plan.las["flags"] == 0x20)).sum())
        assert nsep > 20   # one per read that runs through a short contig
        for col in (1, 2):
            m = tri[:, col] >= 0
            assert np.array_equal(tri[:, col] < 0, etri[:, col] < 0)
            assert np.array_equal(plan.las[tri[m, col]], las[etri[m, col]])
            nxt_e = np.minimum(etri[m, col] + 1, len(las) - 1)
            nxt_p = np.minimum(tri[m, col] + 1, len(plan.las) - 1)
            assert np.array_equal(cont(las, etri[m, col], nxt_e), cont(plan.las, tri[m, col], nxt_p))
        plan.close()


def _worker_agree(rank, world, port, q, forced):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import dentist_amd.parallel as par
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.pop("DH_SHARD_COLLECTIVES", None)
    if forced is True or forced == "rank%d" % rank:   # ("rank0" / "rank1": the variable differs between the ranks' environments)
        os.environ["DH_SHARD_COLLECTIVES"] = "torch"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    par._device = lambda d: torch.device("cpu")   # (the agreement runs over whatever backend carries torch.distributed)

    class NoCtx:   # no device context here: dh_comm_create must refuse it, on both ranks, and both must learn of it
        _h = None
    first = par._c_abi_collectives(NoCtx(), rank, world)
    again = par._c_abi_collectives(NoCtx(), rank, world)   # decided once per process: no second round of collectives
    q.put((rank, first, again))
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_agree_on_the_collectives_when_the_communicator_cannot_be_made():
    """parallel._c_abi_collectives: every rank tries to create its dh_comm (rank 0 draws the id and broadcasts it, or its
    failure), the outcomes are min-reduced, and all ranks take the same path.  On a machine without a GPU no rank can
    create one: both say False, by agreement and by the DH_SHARD_COLLECTIVES=torch switch -- also when the switch is set in
    one rank's environment only (rank 0's setting is broadcast and decides: no rank skips a collective the other enters)."""
    for forced in (False, True, "rank0", "rank1"):
        port = _free_port()
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_worker_agree, args=(r, 2, port, q, forced)) for r in range(2)]
        for p in procs:
            p.start()
        got = sorted(q.get(timeout=180) for _ in range(2))
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        assert got == [(0, False, False), (1, False, False)]
