"""General joins through `process`: pile-ups of ANY edge of the scaffold graph -- gaps between contig
ends in the same orientation, anti-parallel joins ((c, end) -> (d, end), (c, begin) -> (d, begin)), joins that skip
contig ids, extension pile-ups at scaffold ends -- cropped, aligned, polished and anchored on their flanks
(processPileUps/cropper.d:113-175, package.d:283-374, 631-805; common/insertions.d:110-284).  Product (C ABI:
dh_scaffold_all_pileups -> dh_process_pileups) against the oracle (oracle/scaffold.py:build -> oracle/process.py with
the join), bit for bit, and against the truth the contigs were cut from."""
import os

import numpy as np
import pytest

import dentist_amd
from dentist_amd import sim
from helpers import assert_same_las
from oracle import collect_filters as cf
from oracle import process as pr
from oracle import pyoracle as oz
from oracle import scaffold as sc

pytestmark = pytest.mark.gpu


def scrambled_assembly(w, trim=3000):
    """The contigs of a workload as an assembler might hand them over: contig 1 reverse-complemented, contigs 2 and 3
    in swapped order, the outer ends of the first and last contig trimmed (reads hang over them).  Returns (SeqDb,
    placement) with placement[new id] = (old id, reversed, offset of the kept part in the old contig)."""
    old = [w.contigs.seq(i) for i in range(w.contigs.n)]
    assert len(old) == 5
    new = [old[0][trim:], sim.revcomp(old[1]), old[3], old[2], old[4][:-trim]]
    placement = [(0, False, trim), (1, True, 0), (3, False, 0), (2, False, 0), (4, False, 0)]
    return sim.SeqDb.from_list(new), placement


def expected_entries(contigs, reads, flas, min_reads):
    """{join: [(read, LA flank 0, LA flank 1), ...]} from the oracle's scaffold builder (0-based contigs, seeds 0 / 1)."""
    units, _, ranges = cf.chain_units(flas)
    chains = [sc.chain(ranges[c][0], int(l["aread"]) + 1, contigs.length(int(l["aread"])), int(l["bread"]) + 1,
                       reads.length(int(l["bread"])), bool(l["flags"] & 1), int(l["abpos"]), int(l["aepos"]),
                       int(l["bbpos"]), int(l["bepos"]), disabled=bool(l["flags"] & 0x20)) for c, l in enumerate(units)]
    out = {}
    for e, ras in sc.build(contigs.n, chains, [], min_spanning_reads=min_reads):
        (c0, p0), (c1, p1) = e["start"], e["end"]
        if c0 == c1:
            if (p0, p1) == (sc.PRE, sc.BEGIN):
                join = (c0 - 1, 0, -1, 0)
            elif (p0, p1) == (sc.END, sc.POST):
                join = (c0 - 1, 1, -1, 0)
            else:
                continue
        else:
            join = (c0 - 1, 0 if p0 == sc.BEGIN else 1, c1 - 1, 0 if p1 == sc.BEGIN else 1)
        ent = []
        for ra in ras:
            fl = []
            for chain, seed in ra:
                f = 0 if (chain["a_id"] - 1, seed) == (join[0], join[1]) else (1 if (chain["a_id"] - 1, seed) == (join[2], join[3]) else -1)
                fl.append((f, chain))
            if any(f < 0 for f, _ in fl) or len({f for f, _ in fl}) != len(fl):
                continue
            if len(fl) == 2:
                same = fl[0][1]["complement"] == fl[1][1]["complement"]
                if same != (join[1] != join[3]):
                    continue
            t = [ra[0][0]["b_id"] - 1, -1, -1]
            for f, chain in fl:
                t[1 + f] = chain["id"]
            ent.append(tuple(t))
        ent.sort(key=lambda t: t[0])   # stable: by read, then the builder's order
        if ent:
            out[join] = ent
    return out


def test_general_joins_against_the_oracle_and_the_truth(gpu_ctx):
    w = sim.Workload(500_000, 4, 1400, 9000, seed=20260931, spacing=70000, gap_max=1200)
    contigs, placement = scrambled_assembly(w)
    mo = dentist_amd.default_align_opts(kmer_mod=4, k=20, width=64, xdrop=60, algo=1)
    po = dentist_amd.default_process_opts(algo=1, rounds=2, max_reads=0)
    A, B = gpu_ctx.db(contigs), gpu_ctx.db(w.reads)
    las, trace, dropped = gpu_ctx.map_reads(A, B, mo, po, sorted=False, candidates=False)[:3]
    oo = oz.default_opts(kmer_mod=4, k=20, width=64, xdrop=60, algo=1)
    olas, otrace, _ = oz.align_db(contigs, w.reads, oo, nthreads=os.cpu_count() or 1, sort=False, select_best=True)
    flas, odropped, _ = cf.collect_filter(olas, contigs.off, w.reads.off)
    assert [int(x) for x in dropped] == [int(x) for x in odropped]
    assert_same_las((las, trace), (flas, otrace))
    piles, _ = dentist_amd.scaffold_all_pileups(las, contigs.off, w.reads.off, None, only="both", min_spanning_reads=po.min_reads)
    exp = expected_entries(contigs, w.reads, flas, po.min_reads)
    got = {}
    for i in range(len(piles)):
        _, tri = piles.get(i)
        got[piles.get_join(i)] = [tuple(int(x) for x in t) for t in tri.tolist()]
    assert got == exp
    # the joins the scrambling planted: anti-parallel end-end and begin-begin, a reversed pair of ids, a skipping join,
    # and the two trimmed ends as extension pile-ups
    planted = [(0, 1, 1, 1), (1, 0, 3, 0), (2, 0, 3, 1), (2, 1, 4, 0), (0, 0, -1, 0), (4, 1, -1, 0)]
    for j in planted:
        assert j in got and len(got[j]) >= 3, j
    rec, bases = dentist_amd.process_pileups(gpu_ctx, A, B, las, trace, piles, po)
    closed = 0
    for i in range(len(piles)):
        join = piles.get_join(i)
        r = rec[i]
        ex = pr.process_pile(got[join], flas, otrace, contigs, w.reads, join, rounds=po.rounds, nthreads=os.cpu_count() or 1, algo=1)
        assert (r["contig_left"], r["contig_right"]) == (join[0], join[2])
        assert r["join"] == (1 if join[1] == 0 else 0) | (2 if join[2] >= 0 and join[3] == 1 else 0) | (4 if join[2] < 0 else 0)
        assert (r["status"] == 0) == (ex["status"] == "ok"), (join, int(r["status"]), ex["status"])
        if r["status"] != 0:
            continue
        assert (r["crop_left"], r["crop_right"], r["nreads"]) == (ex["cropL"], ex["cropR"], ex["pile"].n)
        assert r["ref_read"] == ex["ref_idx"]
        cons = bases[r["cons_off"]:r["cons_off"] + r["cons_len"]]
        assert np.array_equal(cons, ex["consensus"]), f"join {join}: consensus differs"
        assert (r["left_aepos"], r["right_abpos"], r["ins_begin"], r["ins_end"], r["comp"]) == \
               (ex["left_aepos"], ex["right_abpos"], ex["ins_begin"], ex["ins_end"], ex["comp"])
        # ---- against the truth: the insertion (in the direction of contig 0) is the genome between the two splice sites
        cseq = sim.revcomp(cons) if r["comp"] else cons
        ins = cseq[r["ins_begin"]:r["ins_end"]]

        def genome_pos(c, x):
            old, rev, off = placement[c]
            return int(w.contig_start[old]) + (w.contigs.length(old) - x if rev else off + x)
        g0 = genome_pos(join[0], int(r["left_aepos"]))
        rev0 = placement[join[0]][1]
        if join[2] >= 0:
            g1 = genome_pos(join[2], int(r["right_abpos"]))
        else:   # an extension leaves the contig on its seeded side
            outward_up = (join[1] == 1) != rev0
            g1 = g0 + len(ins) if outward_up else g0 - len(ins)
        seg = w.truth[min(g0, g1):max(g0, g1)]
        ed, _ = oz.nw(sim.revcomp(seg) if rev0 else seg, ins)
        # (the far end of an extension is covered by ever fewer reads, at last by the reference read alone)
        assert ed <= max(6, (0.03 if join[2] >= 0 else 0.08) * len(seg)), (join, ed, len(seg), len(ins))
        closed += 1
    assert closed >= 5, closed
    for j in planted:
        i = [k for k in range(len(piles)) if piles.get_join(k) == j][0]
        assert rec[i]["status"] == 0, (j, int(rec[i]["status"]))


def test_scrambled_assembly_comes_out_as_the_genome(gpu_ctx, tmp_path):
    """End to end: mapping -> every pile-up of the scaffold graph -> process -> `dentist output` with --join-policy contigs
    and --only both.  The five scrambled contigs (one reverse-complemented, two in swapped order, two trimmed ends) come
    out as ONE scaffold that reads like the genome: every contig piece is found in the truth verbatim, in order, on one
    strand, and every insertion matches the truth between its neighbours."""
    import re
    w = sim.Workload(500_000, 4, 1400, 9000, seed=20260931, spacing=70000, gap_max=1200)
    contigs, placement = scrambled_assembly(w)
    mo = dentist_amd.default_align_opts(kmer_mod=4, k=20, width=64, xdrop=60, algo=1)
    po = dentist_amd.default_process_opts(algo=1, rounds=3, max_reads=0)
    A, B = gpu_ctx.db(contigs), gpu_ctx.db(w.reads)
    las, trace, _ = gpu_ctx.map_reads(A, B, mo, po, sorted=False, candidates=False)[:3]
    piles, _ = dentist_amd.scaffold_all_pileups(las, contigs.off, w.reads.off, None, only="both", min_spanning_reads=po.min_reads)
    rec, bases, ids = dentist_amd.process_pileups(gpu_ctx, A, B, las, trace, piles, po, read_ids=True)
    assert (rec["status"] == 0).sum() == 6
    fa = str(tmp_path / "out.fasta")
    dropped = dentist_amd.output_assembly(fa, contigs, list(range(contigs.n)), ["ctg%d" % i for i in range(contigs.n)], [0] * contigs.n,
                                          rec, bases, read_ids=ids, join_policy="contigs", only="both", line_width=0)
    assert dropped == 0
    text = open(fa).read().strip().split("\n")
    assert len(text) == 2 and text[0] == ">ctg0\tscaffold-1"
    truth = sim.decode(w.truth)
    pieces = re.findall(r"[acgt]+|[ACGT]+", text[1])
    assert len(pieces) == 11 and [p[0].isupper() for p in pieces] == [True, False] * 5 + [True]
    # the walk starts at the front extension of contig 0, i.e. runs along the genome's forward strand
    at = 0
    prev_end = None
    edits = ins_bases = 0
    for k, p in enumerate(pieces):
        if p[0].islower():
            pos = truth.find(p, at)
            assert pos >= 0, "contig piece %d is not in the truth downstream of the previous one" % k
            if prev_end is not None:
                ins = pieces[k - 1].lower()
                ed, _ = oz.nw(sim.encode(truth[prev_end:pos]), sim.encode(ins))
                edits += ed
                ins_bases += pos - prev_end
                assert ed <= max(6, 0.03 * (pos - prev_end)), (k, ed, pos - prev_end)
            prev_end = pos + len(p)
            at = prev_end
    assert ins_bases > 2000 and edits <= 0.01 * ins_bases, (edits, ins_bases)
    # the two extensions: 3 kb were trimmed from either end, the consensus wins some of them back
    first_pos = truth.find(pieces[1])
    head, tail = pieces[0].lower(), pieces[-1].lower()
    assert len(head) >= 100 and len(tail) >= 100
    assert oz.nw(sim.encode(truth[first_pos - len(head):first_pos]), sim.encode(head))[0] <= 0.08 * len(head)
    assert oz.nw(sim.encode(truth[prev_end:prev_end + len(tail)]), sim.encode(tail))[0] <= 0.08 * len(tail)


def test_containers_of_general_joins_and_the_repeat_mask_in_the_cropper(gpu_ctx, tmp_path):
    """(1) pile-ups.db / insertions.db of general joins: every SeededAlignment carries the seed of its flank, every
    insertion the nodes of its join (makeJoin, base.d:2680-2722: begin = 1, end = 2; a front extension is (pre = 0) ->
    begin, a back extension end -> (post = 3)) and one overlap per flank.  (2) `process --mask`: a repeat-mask interval
    over the common trace point of a flank moves the crop to the next trace point outside it (getCommonTracePoint,
    cropper.d:446-500) -- product (dh_process_pileups_masked) == oracle, crop points, consensus and splice coordinates."""
    w = sim.Workload(500_000, 4, 1400, 9000, seed=20260931, spacing=70000, gap_max=1200)
    contigs, placement = scrambled_assembly(w)
    mo = dentist_amd.default_align_opts(kmer_mod=4, k=20, width=64, xdrop=60, algo=1)
    po = dentist_amd.default_process_opts(algo=1, rounds=2, max_reads=0)
    A, B = gpu_ctx.db(contigs), gpu_ctx.db(w.reads)
    las, trace, _ = gpu_ctx.map_reads(A, B, mo, po, sorted=False, candidates=False)[:3]
    piles, _ = dentist_amd.scaffold_all_pileups(las, contigs.off, w.reads.off, None, only="both", min_spanning_reads=po.min_reads)
    joins = [piles.get_join(i) for i in range(len(piles))]
    # ---- (1) containers
    pdb, idb = str(tmp_path / "pile-ups.db"), str(tmp_path / "insertions.db")
    piles.write_db(pdb, las, trace, contigs.off, w.reads.off, tspace=100)
    got = dentist_amd.pileupdb_read(pdb)
    assert got["pile_counts"].tolist() == [len(piles.get(i)[1]) for i in range(len(piles))]
    at = ra = 0
    for i, j in enumerate(joins):
        _, tri = piles.get(i)
        for t in tri:
            n = got["ra_counts"][ra]
            assert n == int(t[1] >= 0) + int(t[2] >= 0)
            for f in (0, 1):
                if t[1 + f] >= 0:
                    sa = got["seeded"][at]
                    assert (sa["contig_a_id"] - 1, sa["seed"]) == (j[2 * f], j[2 * f + 1]) and sa["contig_b_id"] - 1 == t[0]
                    at += 1
            ra += 1
    rec, bases = dentist_amd.process_pileups(gpu_ctx, A, B, las, trace, piles, po, insertions_db=(idb, contigs.off, po.tspace_pile))
    ins = dentist_amd.insertiondb_read(idb)
    ok = [i for i in range(len(piles)) if rec[i]["status"] == 0]
    assert len(ins["insertions"]) == len(ok) == 6
    at = 0
    for x, i in zip(ins["insertions"], ok):
        c0, s0, c1, s1 = joins[i]
        if c1 < 0:
            want = (c0 + 1, 0 if s0 == 0 else 2, c0 + 1, 1 if s0 == 0 else 3, 1)
        else:
            want = (c0 + 1, 1 if s0 == 0 else 2, c1 + 1, 1 if s1 == 0 else 2, 2)
        assert (x["start_contig"], x["start_part"], x["end_contig"], x["end_part"], x["noverlaps"]) == want
        assert x["seq_len"] == rec[i]["cons_len"]
        for f in range(x["noverlaps"]):
            sa = ins["seeded"][at]
            assert (sa["contig_a_id"] - 1, sa["seed"]) == ((c0, s0) if f == 0 else (c1, s1))
            at += 1
    # ---- (2) the mask: over the crop point of flank 0 of the skipping join and of flank 1 of the end-end join
    mask = {}
    for i, j in enumerate(joins):
        if j == (2, 1, 4, 0):
            mask[2] = [(int(rec[i]["crop_left"]) - 150, int(rec[i]["crop_left"]) + 60)]
        if j == (0, 1, 1, 1):
            mask[1] = [(int(rec[i]["crop_right"]) - 120, int(rec[i]["crop_right"]) + 250)]
    assert len(mask) == 2
    ptr = np.zeros(contigs.n + 1, dtype=np.int64)
    iv = []
    for c in range(contigs.n):
        iv += [x for b_e in mask.get(c, []) for x in b_e]
        ptr[c + 1] = len(iv) // 2
    rec2, bases2 = dentist_amd.process_pileups(gpu_ctx, A, B, las, trace, piles, po, repeat_mask=(ptr, np.asarray(iv, dtype=np.int32)))
    olas, otrace = las, trace
    moved = 0
    for i, j in enumerate(joins):
        _, tri = piles.get(i)
        ex = pr.process_pile([tuple(int(x) for x in t) for t in tri.tolist()], olas, otrace, contigs, w.reads, j, rounds=po.rounds,
                             nthreads=os.cpu_count() or 1, algo=1, mask=mask)
        r = rec2[i]
        assert (r["status"] == 0) == (ex["status"] == "ok"), (j, int(r["status"]), ex["status"])
        assert (r["crop_left"], r["crop_right"]) == (ex["cropL"], ex["cropR"])
        if j in ((2, 1, 4, 0), (0, 1, 1, 1)):
            f = 0 if j[0] == 2 else 1
            b, e = mask[j[2 * f]][0]
            new = int(r["crop_left"] if f == 0 else r["crop_right"])
            old = int(rec[i]["crop_left"] if f == 0 else rec[i]["crop_right"])
            assert b <= old < e and not (b <= new < e)
            moved += 1
        else:
            assert (r["crop_left"], r["crop_right"]) == (rec[i]["crop_left"], rec[i]["crop_right"])
        if r["status"] == 0:
            assert np.array_equal(bases2[r["cons_off"]:r["cons_off"] + r["cons_len"]], ex["consensus"])
            assert (r["left_aepos"], r["right_abpos"], r["ins_begin"], r["ins_end"]) == (ex["left_aepos"], ex["right_abpos"], ex["ins_begin"], ex["ins_end"])
    assert moved == 2
