"""Parity of the pile-up consensus path (dh_collect_spanning + dh_process_pileups through the C
ABI) with the oracle's restatement of `dentist process` -- bit exact: pile-up membership, crop
points, reference read, every consensus base and the insertion coordinates."""
import os

import numpy as np
import pytest

import dentist_amd
from dentist_amd import sim
from helpers import plant_long_indels, plant_gap_insertions, assert_same_las
from oracle import process as pr
from oracle import pyoracle as oz

pytestmark = pytest.mark.gpu

STATUS = {0: "ok", 1: "no common trace point", 2: "pile too small",
          3: "empty pileup alignment after filtering"}


def run_case(ctx, w, rounds, algo=0, truth_slack=1.0, band=64, **po_kw):
    """algo: 0 = DH-1 (waves) everywhere, 1 = DH-2 (tiled band) for the mapping and every process stage (band rows: 64 or 32)."""
    g = dentist_amd.default_align_opts(**(dict(algo=1, width=band) if algo else {}))
    A, B = ctx.db(w.contigs), ctx.db(w.reads)
    las, trace = ctx.align_db(A, B, g)
    # the mapping LAs themselves are covered by test_parity_map_gpu; the oracle re-derives them
    olas, otrace, _ = oz.align_db(w.contigs, w.reads, oz.default_opts(width=g.width, algo=algo), nthreads=os.cpu_count() or 1)
    assert_same_las((las, trace), (olas, otrace))
    po = dentist_amd.default_process_opts(rounds=rounds, algo=algo, **(dict(width=32) if band == 32 else {}), **po_kw)
    piles = dentist_amd.Pileups(las, w.contigs.off, po)
    exp_piles = pr.collect_spanning(olas, otrace, w.contigs, w.reads)
    assert len(piles) == len(exp_piles) and len(piles) > 0
    rec, bases = dentist_amd.process_pileups(ctx, A, B, las, trace, piles, po)
    assert len(rec) == len(piles)
    closed = 0
    for i in range(len(piles)):
        gap, tri = piles.get(i)
        assert [tuple(t) for t in tri.tolist()] == [tuple(int(x) for x in e) for e in exp_piles[gap]]
        exp = pr.process_pile(exp_piles[gap], olas, otrace, w.contigs, w.reads, gap, rounds=rounds,
                              nthreads=os.cpu_count() or 1, algo=algo, band=band)
        r = rec[i]
        assert r["contig_left"] == gap
        if exp["status"] != "ok":
            assert r["status"] != 0, (gap, exp["status"])
            if r["status"] in STATUS:
                assert STATUS[int(r["status"])] == exp["status"]
            continue
        assert r["status"] == 0, (gap, int(r["status"]))
        assert (r["crop_left"], r["crop_right"]) == (exp["cropL"], exp["cropR"])
        assert r["nreads"] == exp["pile"].n and r["ref_read"] == exp["ref_idx"]
        assert r["ref_read_id"] == exp["read_ids"][exp["ref_idx"]]
        cons = bases[r["cons_off"]:r["cons_off"] + r["cons_len"]]
        assert np.array_equal(cons, exp["consensus"]), f"gap {gap}: consensus differs"
        assert (r["left_aepos"], r["right_abpos"], r["ins_begin"], r["ins_end"]) == \
               (exp["left_aepos"], exp["right_abpos"], exp["ins_begin"], exp["ins_end"])
        cseq = sim.revcomp(cons) if r["comp"] else cons
        ins = cseq[r["ins_begin"]:r["ins_end"]]
        assert np.array_equal(ins, exp["insertion"])
        # against the truth: the spliced region must be close to what was cut out
        truth = w.truth[w.contig_start[gap] + r["left_aepos"]: w.gap_end[gap] + r["right_abpos"]]
        ed, _ = oz.nw(truth, ins)
        assert ed <= truth_slack * max(3, (0.05 if rounds == 1 else 0.02) * len(truth)), (gap, ed, len(truth))
        closed += 1
    assert closed >= 1
    return rec


@pytest.mark.parametrize("rounds", [1, 2, 3])
def test_process_small_gaps(gpu_ctx, rounds):
    w = sim.Workload(300_000, 3, 1200, 6000, seed=17, spacing=20000, gap_max=800)
    rec = run_case(gpu_ctx, w, rounds)
    assert (rec["status"] == 0).sum() >= 2


def test_process_longer_gaps(gpu_ctx):
    w = sim.Workload(600_000, 4, 2400, 8000, seed=7, spacing=20000, gap_max=2000)
    rec = run_case(gpu_ctx, w, 2)
    assert (rec["status"] == 0).sum() >= 3


@pytest.mark.parametrize("rounds,seed", [(1, 17), (3, 17), (2, 7)])
def test_process_with_the_tiled_band_in_every_stage(gpu_ctx, rounds, seed):
    """dh_process_opts.algo = 1: pile-up all-vs-all (symmetric: the second record of a pair is the tiled
    alignment of the transposed pair), re-alignment to the template and flank alignment all run on k_tile;
    crop points, reference read, every consensus base and the splice coordinates equal the oracle's."""
    w = sim.Workload(400_000, 4, 1600, 6000, seed=seed, spacing=20000, gap_max=1200)
    rec = run_case(gpu_ctx, w, rounds, algo=1)
    assert (rec["status"] == 0).sum() >= 3


def test_process_with_the_band_of_32_rows(gpu_ctx):
    """dh_process_opts.width = 32 with algo 1: pile-up all-vs-all, re-alignment to the template and flank alignment on the
    narrow band (k_tile<., 32>); everything equals the oracle's restatement at W = 32."""
    w = sim.Workload(400_000, 4, 1600, 6000, seed=17, spacing=20000, gap_max=1200)
    rec = run_case(gpu_ctx, w, 3, algo=1, band=32)
    assert (rec["status"] == 0).sum() >= 3


def test_process_reference_fixture_gap_is_closed_perfectly(gpu_ctx, tmp_path):
    """tests/test-commands.sh:17-44, 62-65: the 97 bp gap at [2000, 2097) of the 4 097 bp contig is
    reconstructed exactly from 20x / 13 %-error reads (md5 of gap-closed.fasta)."""
    import hashlib
    raw = open(os.path.join(os.path.dirname(__file__), "golden", "test_commands_assembly_reference.fasta")).read()
    header, seq = raw.split("\n")[0], "".join(raw.split("\n")[1:])
    codes = sim.encode(seq)
    contigs = sim.SeqDb.from_list([codes[:2000], codes[2097:]])
    # simulator -m25000 -s12500 -e.13 -c20: reads are longer than the contig, so every read spans
    reads, _ = sim.reads(1724161952, codes, 40, 3000, 0, min_len=500)
    g = dentist_amd.default_align_opts()
    A, B = gpu_ctx.db(contigs), gpu_ctx.db(reads)
    las, trace = gpu_ctx.align_db(A, B, g)
    po = dentist_amd.default_process_opts(rounds=2, flank_window=20000)
    piles = dentist_amd.Pileups(las, contigs.off, po)
    assert len(piles) == 1
    rec, bases = dentist_amd.process_pileups(gpu_ctx, A, B, las, trace, piles, po)
    r = rec[0]
    assert r["status"] == 0
    cons = bases[r["cons_off"]:r["cons_off"] + r["cons_len"]]
    cseq = sim.revcomp(cons) if r["comp"] else cons
    ins = sim.decode(cseq[r["ins_begin"]:r["ins_end"]])
    # output.d:782-925: lower-case contig slices, upper-case insertion, 50 columns, header rule :743-759
    out = seq[:r["left_aepos"]] + ins.upper() + seq[2097 + r["right_abpos"]:]
    assert out.lower() == seq, "gap not reconstructed exactly"
    # the library's `dentist output` writer produces the reference's gap-closed.fasta byte for byte
    path = str(tmp_path / "gap-closed.fasta")
    bed = str(tmp_path / "closed-gaps.bed")
    dentist_amd.output_fasta(path, contigs, [0, 0], [header[1:]], [97], rec, bases, bed_path=bed)
    data = open(path, "rb").read()
    assert data.decode() == f"{header}\tscaffold-1\n" + "\n".join(out[i:i + 50] for i in range(0, len(out), 50)) + "\n"
    # the flank overlaps reach the contig ends, so exactly the gap [2000, 2097) is inserted (upper case)
    assert (r["left_aepos"], r["right_abpos"]) == (2000, 0)
    assert hashlib.md5(data).hexdigest() == "c3836dc00a3f5e1e2aa8f2a802da4d67"   # tests/test-commands.sh:62-65
    b = open(bed).read().split("\t")
    # output.d:879-891: start = currentScaffoldCoord - 1, end = nextScaffoldCoord (1-based, one past the 0-based end)
    assert b[0] == header[1:] and int(b[1]) == r["left_aepos"] and int(b[2]) == r["left_aepos"] + len(ins) + 1
    # the full writer: same FASTA bytes, the AGP of the reference's layout (output.d:454-573) and a BED line that
    # lists every read of the pile-up (`%(%d-%)`, output.d:879-891)
    rec2, bases2, ids = dentist_amd.process_pileups(gpu_ctx, A, B, las, trace, piles, po, read_ids=True)
    assert np.array_equal(rec2, rec) and len(ids[0]) == r["nreads"]
    path2, bed2, agp = (str(tmp_path / n) for n in ("gap-closed2.fasta", "closed-gaps2.bed", "gap-closed.agp"))
    assert dentist_amd.output_assembly(path2, contigs, [0, 0], [header[1:]], [97], rec2, bases2, read_ids=ids, bed_path=bed2,
                                       agp_path=agp, agp_dazzler=True, input_assembly="reference.dam") == 0
    assert open(path2, "rb").read() == data
    idlist = "-".join(str(int(x) + 1) for x in sorted(ids[0].tolist()))
    assert open(bed2).read() == f"chr3R\t2000\t2098\tcontigs-1-2|reads-{idlist}\n"
    sb, se = (r["cons_len"] - r["ins_end"], r["cons_len"] - r["ins_begin"]) if r["comp"] else (r["ins_begin"], r["ins_end"])
    assert [ln for ln in open(agp).read().split("\n") if ln and not ln.startswith("#")] == [
        "chr3R\t1\t2000\t1\tW\t1\t0\t2000\t-\tna",
        f"chr3R\t2001\t2097\t2\tO\treads-{idlist}\t{sb}\t{se}\t{'+' if r['comp'] else '-'}\tclone_contig",
        "chr3R\t2098\t4097\t3\tW\t2\t2097\t4097\t-\tna",
    ]


def test_config1_full_size_properties(gpu_ctx):
    """BASELINE config[1] (10 Mb assembly, 100 gaps, 100 k x 10 kb reads at 13 %): size-independent
    properties of the whole hot path -- every mapped read lies where the simulator put it, trace
    invariants, all gaps closed with a consensus within 0.5 % of the truth, idempotence."""
    w = sim.Workload(10_000_000, 100, 100_000, 10_000, seed=20260929)
    A, B = gpu_ctx.db(w.contigs), gpu_ctx.db(w.reads)
    mo = dentist_amd.default_align_opts(kmer_mod=4)
    po = dentist_amd.default_process_opts()
    las, trace = gpu_ctx.align_db(A, B, mo, select_best=True)
    from helpers import check_trace_invariants
    check_trace_invariants(las[:: max(1, len(las) // 3000)], trace, 100)
    assert len(set(las["bread"].tolist())) >= 0.995 * w.reads.n
    s, e = w.read_truth[las["bread"], 0], w.read_truth[las["bread"], 1]
    cs = w.contig_start[las["aread"]]
    ok = ((las["flags"] & 1) == w.read_truth[las["bread"], 2]) & (cs + las["abpos"] >= s - 80) & (cs + las["aepos"] <= e + 80)
    assert ok.mean() > 0.999
    piles = dentist_amd.Pileups(las, w.contigs.off, po)
    assert len(piles) == 100
    rec, bases = dentist_amd.process_pileups(gpu_ctx, A, B, las, trace, piles, po)
    closed = rec[rec["status"] == 0]
    assert len(closed) >= 99
    edits = total = 0
    for r in closed:
        g = int(r["contig_left"])
        cons = bases[r["cons_off"]:r["cons_off"] + r["cons_len"]]
        cseq = sim.revcomp(cons) if r["comp"] else cons
        ins = cseq[r["ins_begin"]:r["ins_end"]]
        truth = w.truth[w.contig_start[g] + r["left_aepos"]: w.gap_end[g] + r["right_abpos"]]
        ed, _ = oz.nw(truth, ins)
        edits += ed
        total += len(truth)
    assert edits <= 0.005 * total, (edits, total)
    # a second pass with every cache dropped gives the same bits
    A.drop_cache()
    B.drop_cache()
    las2, trace2 = gpu_ctx.align_db(A, B, mo, select_best=True)
    assert_same_las((las2, trace2), (las, trace))
    rec2, bases2 = dentist_amd.process_pileups(gpu_ctx, A, B, las2, trace2, dentist_amd.Pileups(las2, w.contigs.off, po), po)
    assert np.array_equal(rec2, rec) and np.array_equal(bases2, bases)
    # a batch of 64 or more pile-ups is processed as two concurrent halves (two contexts, two host threads): one after
    # the other (DH_PROCESS_SERIAL) gives the same records, consensus sequences, read ids and containers
    ids = dentist_amd.process_pileups(gpu_ctx, A, B, las, trace, piles, po, read_ids=True)
    os.environ["DH_PROCESS_SERIAL"] = "1"
    try:
        ids1 = dentist_amd.process_pileups(gpu_ctx, A, B, las, trace, piles, po, read_ids=True)
    finally:
        del os.environ["DH_PROCESS_SERIAL"]
    for got, exp in ((ids[0], ids1[0]), (ids[1], ids1[1]), (ids[2][0], ids1[2][0]), (ids[2][1], ids1[2][1])):
        assert np.array_equal(got, exp)
    assert np.array_equal(ids1[0], rec) and np.array_equal(ids1[1], bases) and len(ids[2][0]) >= 3 * 99
    # four concurrent parts (DH_PROCESS_PARTS, at least 16 pile-ups each): the same bits again
    os.environ["DH_PROCESS_PARTS"] = "4"
    try:
        ids3 = dentist_amd.process_pileups(gpu_ctx, A, B, las, trace, piles, po, read_ids=True)
    finally:
        del os.environ["DH_PROCESS_PARTS"]
    for got, exp in ((ids3[0], ids1[0]), (ids3[1], ids1[1]), (ids3[2][0], ids1[2][0]), (ids3[2][1], ids1[2][1])):
        assert np.array_equal(got, exp)


def test_config2_full_size_properties(gpu_ctx, cfg2_workload):
    """BASELINE configs[2], the headline workload (100 Mb assembly, 1 000 gaps, 1 M x 15 kb reads at
    13 %, 15.7 Gbp): the mapping runs as a loop over read blocks against the persistent contig index
    (snakemake/Snakefile:1143-1170) merged in memory (LAmerge, :1173-1185) and must equal the single
    call bit for bit; then the size-independent properties of the whole hot path: placement of every
    mapped read, trace invariants, >= 99 % of the gaps closed, consensus <= 0.1 % from the truth,
    idempotence of the process stage."""
    from helpers import check_trace_invariants
    w = cfg2_workload
    A, B = gpu_ctx.db(w.contigs), gpu_ctx.db(w.reads)
    mo = dentist_amd.default_align_opts(kmer_mod=4, k=20)
    po = dentist_amd.default_process_opts()
    nblocks = 8
    bounds = [w.reads.n * b // nblocks for b in range(nblocks + 1)]
    handles = [gpu_ctx.align_db_block(A, B, bounds[b], bounds[b + 1] - bounds[b], mo, select_best=True, raw=True)
               for b in range(nblocks)]
    las, trace = dentist_amd.merge_las(handles)
    check_trace_invariants(las[:: max(1, len(las) // 3000)], trace, 100)
    assert len(set(las["bread"].tolist())) >= 0.995 * w.reads.n
    key = las["aread"].astype(np.int64) << 32 | las["bread"]
    assert np.all(np.diff(key) >= 0), "merged blocks are not in LAsort order"
    s, e = w.read_truth[las["bread"], 0], w.read_truth[las["bread"], 1]
    cs = w.contig_start[las["aread"]]
    ok = ((las["flags"] & 1) == w.read_truth[las["bread"], 2]) & (cs + las["abpos"] >= s - 80) & (cs + las["aepos"] <= e + 80)
    assert ok.mean() > 0.999
    piles = dentist_amd.Pileups(las, w.contigs.off, po)
    assert len(piles) == 1000
    rec, bases = dentist_amd.process_pileups(gpu_ctx, A, B, las, trace, piles, po)
    closed = rec[rec["status"] == 0]
    assert len(closed) >= 990
    edits = total = 0
    for r in closed:
        g = int(r["contig_left"])
        cons = bases[r["cons_off"]:r["cons_off"] + r["cons_len"]]
        cseq = sim.revcomp(cons) if r["comp"] else cons
        ins = cseq[r["ins_begin"]:r["ins_end"]]
        truth = w.truth[w.contig_start[g] + r["left_aepos"]: w.gap_end[g] + r["right_abpos"]]
        ed, _ = oz.nw(truth, ins)
        edits += ed
        total += len(truth)
    assert edits <= 0.001 * total, (edits, total)
    # the whole DB in one call (internal chunk loop, every cache dropped) gives the same records
    A.drop_cache()
    B.drop_cache()
    las2, trace2 = gpu_ctx.align_db(A, B, mo, select_best=True)
    assert len(las2) == len(las)
    for f in ("tlen", "diffs", "abpos", "bbpos", "aepos", "bepos", "flags", "aread", "bread"):
        assert np.array_equal(las2[f], las[f]), f
    idx = np.arange(0, len(las), max(1, len(las) // 20000))
    for i in idx:
        a, b = las[i], las2[i]
        assert np.array_equal(trace[a["toff"]:a["toff"] + a["tlen"]], trace2[b["toff"]:b["toff"] + b["tlen"]])
    rec2, bases2 = dentist_amd.process_pileups(gpu_ctx, A, B, las2, trace2, dentist_amd.Pileups(las2, w.contigs.off, po), po)
    assert np.array_equal(rec2, rec) and np.array_equal(bases2, bases)


def test_containers_of_collect_and_process(gpu_ctx, tmp_path):
    """pile-ups.db of the collect result and insertions.db of the process result (binio/pileupdb.d,
    insertiondb.d): what an unmodified `dentist process` / `dentist output` would read back."""
    w = sim.Workload(400_000, 4, 1600, 6000, seed=83, spacing=20000, gap_max=900)
    g = dentist_amd.default_align_opts()
    A, B = gpu_ctx.db(w.contigs), gpu_ctx.db(w.reads)
    las, trace = gpu_ctx.align_db(A, B, g, select_best=True)
    po = dentist_amd.default_process_opts(rounds=2)
    piles = dentist_amd.Pileups(las, w.contigs.off, po)
    pdb = str(tmp_path / "pile-ups.db")
    piles.write_db(pdb, las, trace, w.contigs.off, w.reads.off)
    back = dentist_amd.pileupdb_read(pdb)
    assert len(back["pile_counts"]) == len(piles) and np.all(back["ra_counts"] == 2)
    at = 0
    for i in range(len(piles)):
        gap, tri = piles.get(i)
        assert back["pile_counts"][i] == len(tri)
        for read, il, ir in tri.tolist():
            left, right = back["seeded"][at], back["seeded"][at + 1]
            at += 2
            assert (left["contig_a_id"], right["contig_a_id"]) == (gap + 1, gap + 2)
            assert left["contig_b_id"] == right["contig_b_id"] == read + 1 and (left["seed"], right["seed"]) == (1, 0)
            assert left["contig_b_len"] == w.reads.length(read) and left["flags"] == (int(las[il]["flags"]) & 1)
    assert int(back["las"]["ntp"].sum()) * 2 == len(back["trace"])
    idb = str(tmp_path / "insertions.db")
    rec, bases = dentist_amd.process_pileups(gpu_ctx, A, B, las, trace, piles, po, insertions_db=(idb, w.contigs.off, 126))
    ins = dentist_amd.insertiondb_read(idb)
    closed = rec[rec["status"] == 0]
    assert len(ins["insertions"]) == len(closed) >= 3
    boff = ioff = 0
    for k, r in enumerate(closed):
        q = ins["insertions"][k]
        gap = int(r["contig_left"])
        assert (q["start_contig"], q["start_part"], q["end_contig"], q["end_part"]) == (gap + 1, 2, gap + 2, 1)
        assert q["seq_len"] == r["cons_len"] and q["noverlaps"] == 2
        assert np.array_equal(ins["bases"][boff:boff + q["seq_len"]], bases[r["cons_off"]:r["cons_off"] + r["cons_len"]])
        boff += int(q["seq_len"])
        ids = ins["read_ids"][ioff:ioff + q["nread_ids"]]
        ioff += int(q["nread_ids"])
        assert np.all(np.diff(ids.astype(np.int64)) > 0) and int(r["ref_read_id"]) + 1 in ids.tolist() and len(ids) == r["nreads"]
        left, right = ins["seeded"][2 * k], ins["seeded"][2 * k + 1]
        ll, rl = ins["las"][2 * k], ins["las"][2 * k + 1]
        # the cropping positions `dentist output` derives from the overlaps (insertions.d:230-284):
        # contig A: end of the left overlap / begin of the right one; contig B likewise
        assert (ll["a_end"], rl["a_begin"], ll["b_end"], rl["b_begin"]) == \
               (r["left_aepos"], r["right_abpos"], r["ins_begin"], r["ins_end"])
        assert left["contig_a_len"] == w.contigs.length(gap) and left["contig_b_len"] == r["cons_len"]
        assert (left["seed"], right["seed"], left["tspace"]) == (1, 0, 126) and left["flags"] == r["comp"]


@pytest.mark.gpu
def test_scaffold_graph_builder_on_a_mapping(gpu_ctx):
    """The scaffold-graph builder (pileups.d:173-208) on real mapping output: one gap join per gap of
    the linear assembly, no forks; its spanning read alignments are a subset of what the spanning
    collector pairs (the builder follows pileups.d:870 literally: a read whose first alignment starts
    after read position 0 opens with an extension, so its two alignments become two extension entries
    of the gap pile-up instead of one spanning entry), and every LA it uses is enabled."""
    w = sim.Workload(2_000_000, 20, 20_000, 10_000, seed=77)
    A, B = gpu_ctx.db(w.contigs), gpu_ctx.db(w.reads)
    mo = dentist_amd.default_align_opts(kmer_mod=4, k=20)
    po = dentist_amd.default_process_opts()
    las, trace = gpu_ctx.align_db(A, B, mo, select_best=True)
    las, dropped, _ = dentist_amd.collect_filter(las, w.contigs.off, w.reads.off, po, inplace=True)
    cand = dentist_amd.Pileups(las, w.contigs.off, po, candidates=True)
    ig = np.stack([np.arange(w.contigs.n - 1), np.arange(1, w.contigs.n)], axis=1)
    joins, ent = dentist_amd.scaffold_pileups(las, w.contigs.off, w.reads.off, ig, min_spanning_reads=po.min_reads)
    gaps = joins[joins["type"] == 1]
    assert len(gaps) == 20 and np.array_equal(gaps["contig1"], gaps["contig0"] + 1)
    assert np.all(gaps["part0"] == 2) and np.all(gaps["part1"] == 1)
    gp, skipped = dentist_amd.scaffold_spanning_pileups(las, w.contigs.off, w.reads.off, ig, min_spanning_reads=po.min_reads)
    assert skipped == len(joins) - len(gaps)
    (cl_a, cnt_a, tri_a), (cl_b, cnt_b, tri_b) = cand.flat(), gp.flat()
    assert np.array_equal(cl_a, cl_b)
    sa = set(map(tuple, np.asarray(tri_a).reshape(-1, 3).tolist()))
    sb = set(map(tuple, np.asarray(tri_b).reshape(-1, 3).tolist()))
    assert sb and sb <= sa
    for e in ent:
        assert not las[e["la0"]]["flags"] & 0x20 and (e["n"] == 1 or not las[e["la1"]]["flags"] & 0x20)
    # the spanning reads the builder keeps start exactly at read position 0 on the read's strand
    for rd, il, ir in sb:
        first = las[il] if not las[il]["flags"] & 1 else las[ir]
        blen = int(w.reads.off[rd + 1] - w.reads.off[rd])
        assert (first["bbpos"] == 0) if not first["flags"] & 1 else (first["bepos"] == blen)


@pytest.mark.gpu
@pytest.mark.parametrize("algo,max_partners", [(0, 0), (1, 0), (1, 12), (1, 5)])
def test_pile_ups_of_the_graph_builder_with_extension_entries(gpu_ctx, algo, max_partners):
    """(max_partners > 0: dh_process_opts.max_partners -- a read of a pile-up is aligned with the first max_partners reads
    only, in the order allowed reference reads, then the others; oracle/process.py applies the same rule.)
    Read-id parity on the path bench.py runs: mapping -> six collect filters -> scaffold-graph builder
    (pileups.d:173-208) -> gap pile-ups WITH the extension-type read alignments mergeExtensionsWithGaps put
    into them (scaffold.d:789-816; pileups.d:870 makes a read whose first alignment starts after read
    position 0 open with an extension) -> crop -> process.  Membership equals oracle/scaffold.py:build() on
    the same alignments; crop points, pile-up reads, reference read (a spanning read: package.d:461-472),
    consensus and splice coordinates equal oracle/process.py with the same entries."""
    from oracle import scaffold as sc
    w = sim.Workload(1_000_000, 10, 8000, 8000, seed=43)
    mo = dentist_amd.default_align_opts(kmer_mod=4, k=20, **(dict(algo=1, width=64) if algo else {}))
    po = dentist_amd.default_process_opts(rounds=2, algo=algo, max_reads=0, max_partners=max_partners)
    A, B = gpu_ctx.db(w.contigs), gpu_ctx.db(w.reads)
    las, trace = gpu_ctx.align_db(A, B, mo, select_best=True)
    las, dropped, _ = dentist_amd.collect_filter(las, w.contigs.off, w.reads.off, po, inplace=True)
    gaps_in = np.stack([np.arange(w.contigs.n - 1), np.arange(1, w.contigs.n)], axis=1)
    piles, skipped = dentist_amd.scaffold_spanning_pileups(las, w.contigs.off, w.reads.off, gaps_in, with_extensions=True,
                                                           min_spanning_reads=po.min_reads)
    # ---- membership against the oracle's builder
    chains = [sc.chain(i, int(l["aread"]) + 1, w.contigs.length(int(l["aread"])), int(l["bread"]) + 1,
                       w.reads.length(int(l["bread"])), bool(l["flags"] & 1), int(l["abpos"]), int(l["aepos"]),
                       int(l["bbpos"]), int(l["bepos"]), disabled=bool(l["flags"] & 0x20)) for i, l in enumerate(las)]
    exp = {}
    for e, ras in sc.build(w.contigs.n, chains, [(int(a) + 1, int(b) + 1) for a, b in gaps_in], min_spanning_reads=po.min_reads):
        (c0, p0), (c1, p1) = e["start"], e["end"]
        if not (p0 == sc.END and p1 == sc.BEGIN and c1 == c0 + 1):
            continue
        ent = []
        for ra in ras:
            if len(ra) == 2:
                a, b = sorted(ra, key=lambda s: s[0]["a_id"])
                ent.append((a[0]["b_id"] - 1, a[0]["id"], b[0]["id"]))
            elif ra[0][0]["a_id"] == c0:
                ent.append((ra[0][0]["b_id"] - 1, ra[0][0]["id"], -1))
            else:
                ent.append((ra[0][0]["b_id"] - 1, -1, ra[0][0]["id"]))
        exp[c0 - 1] = sorted(ent, key=lambda t: t[0])   # stable: by read, then the builder's order
    got = {}
    for i in range(len(piles)):
        g, tri = piles.get(i)
        got[int(g)] = [tuple(int(x) for x in t) for t in tri.tolist()]
    assert set(got) == set(exp) and len(got) == 10
    for g in got:
        assert sorted(got[g]) == sorted(exp[g]), g
    next_ = sum(1 for v in got.values() for t in v if t[1] < 0 or t[2] < 0)
    assert next_ > 0, "the case must contain extension entries"
    # ---- process with these pile-ups against the oracle's driver
    rec, bases = dentist_amd.process_pileups(gpu_ctx, A, B, las, trace, piles, po)
    closed = 0
    for i in range(len(piles)):
        g, tri = piles.get(i)
        ex = pr.process_pile([tuple(int(x) for x in t) for t in tri.tolist()], las, trace, w.contigs, w.reads, int(g),
                             rounds=po.rounds, nthreads=os.cpu_count() or 1, algo=algo, max_partners=max_partners)
        r = rec[i]
        assert (r["status"] == 0) == (ex["status"] == "ok"), (g, int(r["status"]), ex["status"])
        if r["status"] != 0:
            continue
        assert max_partners == 0 or ex["pile"].n > max_partners   # (the option bites in every pile-up of this case)
        assert (r["crop_left"], r["crop_right"], r["nreads"]) == (ex["cropL"], ex["cropR"], ex["pile"].n)
        assert r["ref_read"] == ex["ref_idx"] and (ex["kinds"][ex["ref_idx"]] & 3) == 0
        cons = bases[r["cons_off"]:r["cons_off"] + r["cons_len"]]
        assert np.array_equal(cons, ex["consensus"]), f"gap {g}: consensus differs"
        assert (r["left_aepos"], r["right_abpos"], r["ins_begin"], r["ins_end"]) == \
               (ex["left_aepos"], ex["right_abpos"], ex["ins_begin"], ex["ins_end"])
        closed += 1
    assert closed >= 9


def test_the_benched_chain_against_the_oracle(gpu_ctx):
    """bench.py's exact call chain and options on a workload the oracle finishes in seconds: k = 20, modimer sampling
    1 / 8, band 64, x-drop 60, DH-2 through dh_map_reads (chain flags + the six collect filters on the way) ->
    dh_scaffold_gap_pileups (graph builder with extension entries) -> select with the default read cap (60: 100x
    coverage puts more entries than that into every gap) -> dh_process_pileups with its defaults (3 rounds, dust, two
    concurrent halves need >= 64 pile-ups, so this runs them in one piece).  Oracle side: oz.align_db with the same
    options and chain flags, oracle/collect_filters.py, oracle/scaffold.py:build, the cap restated below,
    oracle/process.py.  Bit-exact: records and traces of the mapping, filter counts, pile-up entries, crop points,
    reference read, consensus bases, splice coordinates."""
    from oracle import collect_filters as cf
    from oracle import scaffold as sc
    w = sim.Workload(2_000_000, 20, 20_000, 10_000, seed=20260929)
    mo = dentist_amd.default_align_opts(kmer_mod=8, k=20, width=64, xdrop=60, algo=1)
    po = dentist_amd.default_process_opts(algo=1)
    assert po.max_reads == 60 and po.rounds == 3
    A, B = gpu_ctx.db(w.contigs), gpu_ctx.db(w.reads)
    las, trace, dropped = gpu_ctx.map_reads(A, B, mo, po, sorted=False, candidates=False)[:3]
    # ---- mapping + filters
    oo = oz.default_opts(kmer_mod=8, k=20, width=64, xdrop=60, algo=1)
    olas, otrace, _ = oz.align_db(w.contigs, w.reads, oo, nthreads=os.cpu_count() or 1, sort=False, select_best=True)
    flas, odropped, _ = cf.collect_filter(olas, w.contigs.off, w.reads.off)
    assert [int(x) for x in dropped] == [int(x) for x in odropped]
    assert_same_las((las, trace), (flas, otrace))
    # ---- pile-ups of the graph builder, then the cap
    gaps_in = np.stack([np.arange(w.contigs.n - 1), np.arange(1, w.contigs.n)], axis=1).astype(np.int32)
    gp, _ = dentist_amd.scaffold_spanning_pileups(las, w.contigs.off, w.reads.off, gaps_in, with_extensions=True,
                                                  min_spanning_reads=po.min_reads)
    piles = gp.select(las, po)
    chains = [sc.chain(i, int(l["aread"]) + 1, w.contigs.length(int(l["aread"])), int(l["bread"]) + 1,
                       w.reads.length(int(l["bread"])), bool(l["flags"] & 1), int(l["abpos"]), int(l["aepos"]),
                       int(l["bbpos"]), int(l["bepos"]), disabled=bool(l["flags"] & 0x20)) for i, l in enumerate(flas)]
    exp = {}
    for e, ras in sc.build(w.contigs.n, chains, [(int(a) + 1, int(b) + 1) for a, b in gaps_in], min_spanning_reads=po.min_reads):
        (c0, p0), (c1, p1) = e["start"], e["end"]
        if not (p0 == sc.END and p1 == sc.BEGIN and c1 == c0 + 1):
            continue
        ent = []
        for ra in ras:
            if len(ra) == 2:
                a, b = sorted(ra, key=lambda s_: s_[0]["a_id"])
                ent.append((a[0]["b_id"] - 1, a[0]["id"], b[0]["id"]))
            elif ra[0][0]["a_id"] == c0:
                ent.append((ra[0][0]["b_id"] - 1, ra[0][0]["id"], -1))
            else:
                ent.append((ra[0][0]["b_id"] - 1, -1, ra[0][0]["id"]))
        ent.sort(key=lambda t: (t[0], t[1] < 0))   # read order; the halves of a spanning read that opens with an extension: left one first
        if len(ent) > po.max_reads:   # the cap: distinct reads first, then the lowest error rate of the anchoring alignments
            def err(t):
                ln = sum(int(flas[i]["aepos"] - flas[i]["abpos"]) for i in t[1:] if i >= 0)
                df = sum(int(flas[i]["diffs"]) for i in t[1:] if i >= 0)
                return df * 1000000 // max(ln, 1)
            second = [False] * len(ent)   # every entry of a read but its best one ranks behind all first entries
            x0 = 0
            while x0 < len(ent):
                x1 = x0
                while x1 < len(ent) and ent[x1][0] == ent[x0][0]:
                    x1 += 1
                best = min(range(x0, x1), key=lambda x: (err(ent[x]), x))
                for x in range(x0, x1):
                    second[x] = x != best
                x0 = x1
            ext = [t[1] < 0 or t[2] < 0 for t in ent]   # ... and extension entries behind the reads that span the gap
            order = sorted(range(len(ent)), key=lambda x: (2 * ext[x] + second[x], err(ent[x]), x))[:po.max_reads]
            ent = [ent[x] for x in sorted(order)]
        exp[c0 - 1] = ent
    got = {}
    for i in range(len(piles)):
        g, tri = piles.get(i)
        got[int(g)] = [tuple(int(x) for x in t) for t in tri.tolist()]
    assert set(got) == set(exp) and len(got) == 20
    capped = 0
    for g in got:
        assert got[g] == exp[g], g
        capped += len(got[g]) == po.max_reads
    assert capped >= 10, "the cap must bite in this case"
    # ---- process
    rec, bases = dentist_amd.process_pileups(gpu_ctx, A, B, las, trace, piles, po)
    closed = edits = tb = 0
    for i in range(len(piles)):
        g, tri = piles.get(i)
        ex = pr.process_pile(got[int(g)], flas, otrace, w.contigs, w.reads, int(g), rounds=po.rounds,
                             nthreads=os.cpu_count() or 1, algo=1)
        r = rec[i]
        assert (r["status"] == 0) == (ex["status"] == "ok"), (g, int(r["status"]), ex["status"])
        if r["status"] != 0:
            continue
        assert (r["crop_left"], r["crop_right"], r["nreads"]) == (ex["cropL"], ex["cropR"], ex["pile"].n)
        assert r["ref_read"] == ex["ref_idx"]
        cons = bases[r["cons_off"]:r["cons_off"] + r["cons_len"]]
        assert np.array_equal(cons, ex["consensus"]), f"gap {g}: consensus differs"
        assert (r["left_aepos"], r["right_abpos"], r["ins_begin"], r["ins_end"]) == \
               (ex["left_aepos"], ex["right_abpos"], ex["ins_begin"], ex["ins_end"])
        cseq = sim.revcomp(cons) if r["comp"] else cons
        truth = w.truth[w.contig_start[int(g)] + r["left_aepos"]: w.gap_end[int(g)] + r["right_abpos"]]
        ed, _ = oz.nw(truth, cseq[r["ins_begin"]:r["ins_end"]])
        edits += ed
        tb += len(truth)
        closed += 1
    assert closed == 20 and edits <= 0.002 * tb, (closed, edits, tb)


def test_chains_with_a_long_indel_next_to_a_gap(gpu_ctx):
    """Alignment chains as the unit from the mapping to the cropper (dazzler.d:1728-1758 builds them from START / NEXT;
    base.d:306-421; SeededAlignment.from(chain) base.d:1964-2050; getCommonTracePoint over ReferenceRegions
    cropper.d:446-500; AlignmentChain.translateTracePoint base.d:866-880): reads that span a gap get a 2-5 kb insertion of
    foreign bases, or lose 2-5 kb of contig bases, 1.5-3.5 kb away from the gap -- each maps as two collinear records on
    that flank, ONE chain.  As single records the part next to the gap would be improper (it begins in the middle of
    read and contig) and the read lost; as a chain the read enters the gap's pile-up, its A region is the union of the
    members' intervals, and the crop position is translated through the member that covers it (regions with holes
    and the repeat mask: tests/test_collect_filters.py, CPU).  Product (dh_map_reads -> dh_scaffold_gap_pileups -> dh_process_pileups) == oracle (collect_filters on
    chains, scaffold.build on one chain() per unit, process.crop_pile on regions), bit for bit."""
    from oracle import collect_filters as cf
    from oracle import scaffold as sc
    w = sim.Workload(600_000, 6, 1500, 12_000, seed=20260930, spacing=60000, gap_max=1500)
    reads, planted = plant_long_indels(w, np.random.default_rng(5))
    assert len(planted) >= 12
    mo = dentist_amd.default_align_opts(kmer_mod=4, k=20, width=64, xdrop=60, algo=1)
    po = dentist_amd.default_process_opts(algo=1, rounds=2, max_reads=0)
    A, B = gpu_ctx.db(w.contigs), gpu_ctx.db(reads)
    las, trace, dropped = gpu_ctx.map_reads(A, B, mo, po, sorted=False, candidates=False)[:3]
    oo = oz.default_opts(kmer_mod=4, k=20, width=64, xdrop=60, algo=1)
    olas, otrace, _ = oz.align_db(w.contigs, reads, oo, nthreads=os.cpu_count() or 1, sort=False, select_best=True)
    flas, odropped, _ = cf.collect_filter(olas, w.contigs.off, reads.off)
    assert [int(x) for x in dropped] == [int(x) for x in odropped]
    assert_same_las((las, trace), (flas, otrace))
    # ---- pile-ups: one chain() per unit for the oracle's builder, ids = the first record of the chain
    gaps_in = np.stack([np.arange(w.contigs.n - 1), np.arange(1, w.contigs.n)], axis=1).astype(np.int32)
    piles, _ = dentist_amd.scaffold_spanning_pileups(las, w.contigs.off, reads.off, gaps_in, with_extensions=True,
                                                     min_spanning_reads=po.min_reads)
    units, _, ranges = cf.chain_units(flas)
    multi = {i for i, j in ranges if j - i > 1}
    assert len(multi) >= 10
    chains = [sc.chain(ranges[c][0], int(l["aread"]) + 1, w.contigs.length(int(l["aread"])), int(l["bread"]) + 1,
                       reads.length(int(l["bread"])), bool(l["flags"] & 1), int(l["abpos"]), int(l["aepos"]),
                       int(l["bbpos"]), int(l["bepos"]), disabled=bool(l["flags"] & 0x20)) for c, l in enumerate(units)]
    exp = {}
    for e, ras in sc.build(w.contigs.n, chains, [(int(a) + 1, int(b) + 1) for a, b in gaps_in], min_spanning_reads=po.min_reads):
        (c0, p0), (c1, p1) = e["start"], e["end"]
        if not (p0 == sc.END and p1 == sc.BEGIN and c1 == c0 + 1):
            continue
        ent = []
        for ra in ras:
            if len(ra) == 2:
                a, b = sorted(ra, key=lambda s_: s_[0]["a_id"])
                ent.append((a[0]["b_id"] - 1, a[0]["id"], b[0]["id"]))
            elif ra[0][0]["a_id"] == c0:
                ent.append((ra[0][0]["b_id"] - 1, ra[0][0]["id"], -1))
            else:
                ent.append((ra[0][0]["b_id"] - 1, -1, ra[0][0]["id"]))
        ent.sort(key=lambda t: (t[0], t[1] < 0))
        exp[c0 - 1] = ent
    got = {}
    for i in range(len(piles)):
        g, tri = piles.get(i)
        got[int(g)] = [tuple(int(x) for x in t) for t in tri.tolist()]
    assert set(got) == set(exp) and len(got) == 6
    chained_entries = 0
    for g in got:
        assert got[g] == exp[g], g
        chained_entries += sum(1 for t in got[g] if t[1] in multi or t[2] in multi)
    assert chained_entries >= 10, "the planted reads must enter the pile-ups as chains"
    in_piles = {t[0] for v in got.values() for t in v}
    assert len(in_piles & set(planted)) >= 10
    # ---- crop + process
    rec, bases = dentist_amd.process_pileups(gpu_ctx, A, B, las, trace, piles, po)
    closed = through_later_member = 0
    for i in range(len(piles)):
        g, tri = piles.get(i)
        ex = pr.process_pile(got[int(g)], flas, otrace, w.contigs, reads, int(g), rounds=po.rounds,
                             nthreads=os.cpu_count() or 1, algo=1)
        r = rec[i]
        assert (r["status"] == 0) == (ex["status"] == "ok"), (g, int(r["status"]), ex["status"])
        if r["status"] != 0:
            continue
        assert (r["crop_left"], r["crop_right"], r["nreads"]) == (ex["cropL"], ex["cropR"], ex["pile"].n)
        assert r["ref_read"] == ex["ref_idx"]
        cons = bases[r["cons_off"]:r["cons_off"] + r["cons_len"]]
        assert np.array_equal(cons, ex["consensus"]), f"gap {g}: consensus differs"
        assert (r["left_aepos"], r["right_abpos"], r["ins_begin"], r["ins_end"]) == \
               (ex["left_aepos"], ex["right_abpos"], ex["ins_begin"], ex["ins_end"])
        # chained entries whose crop position lies on a later member of the chain (the part next to the gap)
        for t in got[int(g)]:
            for i0, pos in ((t[1], ex["cropL"]), (t[2], ex["cropR"])):
                if i0 in multi and not (flas[i0]["abpos"] <= pos <= flas[i0]["aepos"]):
                    through_later_member += 1
        closed += 1
    assert closed >= 5 and through_later_member >= 5, (closed, through_later_member)


def test_consensus_band_classes_and_scalar_fill_agree(gpu_ctx, monkeypatch):
    """The three fills of the per-tile Needleman-Wunsch (bit-parallel with one / two 64-cell words per matrix row, scalar
    for bands above 63) against the oracle's full-matrix NW on noisy reads (20 % error: read-read tiles reach 60+
    differences, so all three classes occur), and the scalar fill forced for everything (DH_CONS_SCALAR) gives the
    same bits."""
    w = sim.Workload(150_000, 2, 700, 4000, seed=47, err=0.20, spacing=15000, gap_max=700)
    rec = run_case(gpu_ctx, w, 2, algo=1, truth_slack=4.0)     # 20 % reads: parity is the point, not the polish
    monkeypatch.setenv("DH_CONS_SCALAR", "1")
    rec2 = run_case(gpu_ctx, w, 2, algo=1, truth_slack=4.0)
    for f in rec.dtype.names:
        if f != "pad":
            assert np.array_equal(rec[f], rec2[f]), f


def test_canonical_indel_placement_on_column_sets_equals_the_bytewise_passes(gpu_ctx, monkeypatch):
    """k_seg_vote2 places indels at the start of template homopolymer runs: since round 6 on sets of columns in registers
    (one count-leading-zeros per walk; votes cast from the sets), before byte by byte over LDS.  ONT-like reads (indels
    biased into homopolymer runs: most indels move) against the oracle, and DH_VOTE_BYTEWISE=1 -- the former passes, still
    the path of tiles above 126 columns -- gives the same bits."""
    w = sim.Workload(150_000, 2, 600, 4000, seed=53, err=0.12, p_ins=0.30, p_del=0.40, hp_bias=0.5, spacing=15000, gap_max=700)
    rec = run_case(gpu_ctx, w, 3, algo=1, truth_slack=2.0)
    monkeypatch.setenv("DH_VOTE_BYTEWISE", "1")
    rec2 = run_case(gpu_ctx, w, 3, algo=1, truth_slack=2.0)
    assert int((rec["status"] == 0).sum()) >= 1
    for f in rec.dtype.names:
        if f != "pad":
            assert np.array_equal(rec[f], rec2[f]), f


@pytest.mark.parametrize("ts,algo", [(100, 1), (128, 1), (200, 0)])
def test_other_trace_spacings_of_the_pile_up_alignments(gpu_ctx, monkeypatch, ts, algo):
    """dh_process_opts.tspace_pile other than the default 126: tiles of up to 100 columns take k_seg_vote2<13> (column sets of
    104 bits); 128 columns -- the most DH-2 takes -- and 200 (DH-1) no longer fit the 128-bit sets: the byte-wise kernel
    k_seg_vote2<0> (and the two-word / scalar fills of the per-tile Needleman-Wunsch more often).  The oracle's process
    sequence with the same spacing gives the same bits."""
    monkeypatch.setattr(pr, "TS_PILE", ts)
    w = sim.Workload(200_000, 2, 800, 5000, seed=61, spacing=20000, gap_max=900)
    rec = run_case(gpu_ctx, w, 2, algo=algo, truth_slack=2.0, tspace_pile=ts)   # (parity is the point: the polish is tuned at 126)
    assert int((rec["status"] == 0).sum()) >= 1


def test_bubble_resolver_on_a_mapping_with_a_masked_contig(gpu_ctx):
    """resolveBubbles end to end (pileups.d:1100-1590): a 3 kb contig is covered by the repeat mask, so the mapping seeds
    nothing on it and the reads that cross it SKIP it -- their join (left contig end -> right contig begin) closes a cycle
    with the input-gap joins around the masked contig.  dh_scaffold_pileups_resolved maps the skipping reads onto that
    contig again without the mask on the device (dh_remap_skipping_reads), collects their read alignments from old + new
    alignments and replaces the skipping pile-up by the two gap pile-ups.  Membership equals oracle/scaffold.py's
    BubbleResolver driven by the oracle's own re-mapping (oz.align_db on the same subsets); the process stage then closes
    both gaps from the augmented alignments."""
    from oracle import collect_filters as cf
    from oracle import scaffold as sc
    rng = np.random.default_rng(99)
    g = sim.genome(4242, 60000)
    cuts = [(0, 16000), (16300, 19300), (19600, 38000), (38400, 60000)]
    contigs = sim.SeqDb.from_list([g[a:b].copy() for a, b in cuts])
    reads, _ = sim.reads(777, g, 240, 11000)
    # the repeat mask: all of contig 1
    contigs.mask = (np.asarray([0, 0, 1, 1, 1], dtype=np.int64), np.asarray([0, 3000, 0, 0], dtype=np.int32))
    mo = dentist_amd.default_align_opts(k=20, kmer_mod=2, width=64, xdrop=60, algo=1)
    po = dentist_amd.default_process_opts(algo=1, rounds=2)
    A, B = gpu_ctx.db(contigs), gpu_ctx.db(reads)
    las, trace = gpu_ctx.align_db(A, B, mo, select_best=True)
    las, dropped, _ = dentist_amd.collect_filter(las, contigs.off, reads.off, po, inplace=True)
    assert not np.any(las["aread"] == 1), "the masked contig must not be seeded"
    gaps_in = np.asarray([[0, 1], [1, 2], [2, 3]], dtype=np.int32)
    plain, _ = dentist_amd.scaffold_spanning_pileups(las, contigs.off, reads.off, gaps_in, with_extensions=True,
                                                     min_spanning_reads=po.min_reads)
    assert sorted(int(plain.get(i)[0]) for i in range(len(plain))) == [2], "without the resolver only the last gap has a pile-up"
    piles, skipped, las_all, trace_all, resolved = dentist_amd.scaffold_spanning_pileups(
        las, contigs.off, reads.off, gaps_in, with_extensions=True, min_spanning_reads=po.min_reads,
        resolve=dict(ctx=gpu_ctx, contigs=A, reads=B, map_opts=mo, trace=trace, allowance=100))
    assert resolved == 1 and len(las_all) > len(las)
    # ---- oracle: same mapping (checked bit-exact elsewhere), its own re-mapping, its own resolver
    oo = oz.default_opts(k=20, kmer_mod=2, width=64, xdrop=60, algo=1)
    n = len(las)

    def chains_of(arr, first_id):
        return [sc.chain(first_id + i, int(l["aread"]) + 1, contigs.length(int(l["aread"])), int(l["bread"]) + 1,
                         reads.length(int(l["bread"])), bool(l["flags"] & 1), int(l["abpos"]), int(l["aepos"]),
                         int(l["bbpos"]), int(l["bepos"]), disabled=bool(l["flags"] & 0x20)) for i, l in enumerate(arr)]
    added = []

    def remap(pile, inter):
        rids = sorted({sa[0]["b_id"] - 1 for ra in pile for sa in ra})
        cids = [c - 1 for c in inter]
        sa_ = sim.SeqDb.from_list([contigs.seq(c) for c in cids])
        sb_ = sim.SeqDb.from_list([reads.seq(r) for r in rids])
        ol, ot, _ = oz.align_db(sa_, sb_, oo, nthreads=os.cpu_count() or 1, sort=True, select_best=True)
        ol = ol.copy()
        i = 0
        while i < len(ol):   # chains: START then its NEXT records; enabled iff the chain covers its contig (allowance 100)
            j = i + 1
            while j < len(ol) and (ol[j]["flags"] & 0x8) and not (ol[j]["flags"] & 0x4):
                j += 1
            alen = sa_.length(int(ol[i]["aread"]))
            if not (ol[i]["abpos"] <= 100 and ol[j - 1]["aepos"] >= alen - 100):
                ol["flags"][i:j] |= 0x20
            i = j
        ol["aread"] = np.asarray(cids, dtype=np.int32)[ol["aread"]]
        ol["bread"] = np.asarray(rids, dtype=np.int32)[ol["bread"]]
        new = chains_of(ol, n + len(added))
        added.extend(new)
        return new
    exp = {}
    for e, ras in sc.build(contigs.n, chains_of(las, 0), [(int(a) + 1, int(b) + 1) for a, b in gaps_in],
                           min_spanning_reads=po.min_reads, remap=remap):
        (c0, p0), (c1, p1) = e["start"], e["end"]
        if not (p0 == sc.END and p1 == sc.BEGIN and c1 == c0 + 1):
            continue
        ent = []
        for ra in ras:
            if len(ra) == 2:
                a, b = sorted(ra, key=lambda s_: s_[0]["a_id"])
                ent.append((a[0]["b_id"] - 1, a[0]["id"], b[0]["id"]))
            elif ra[0][0]["a_id"] == c0:
                ent.append((ra[0][0]["b_id"] - 1, ra[0][0]["id"], -1))
            else:
                ent.append((ra[0][0]["b_id"] - 1, -1, ra[0][0]["id"]))
        exp[c0 - 1] = sorted(ent, key=lambda t: (t[0], t[1] < 0))
    got = {}
    for i in range(len(piles)):
        gg, tri = piles.get(i)
        got[int(gg)] = [tuple(int(x) for x in t) for t in tri.tolist()]
    assert set(got) == set(exp) == {0, 1, 2}
    assert len(las_all) == n + len(added)
    for gg in got:
        assert got[gg] == exp[gg], gg
    assert any(t[2] >= n for t in got[0]) and any(t[1] >= n for t in got[1]), "the new pile-ups use the re-mapped alignments"
    # ---- the process stage on the augmented alignments closes the gaps around the masked contig
    # (with the mask on the contigs DB the flank re-alignment -- daligner -mrep -- cannot seed on the masked contig and the
    # two gaps end as DH_PILE_FLANKS_NOT_UNIQUE; `process` without that mask closes them)
    rec, bases = dentist_amd.process_pileups(gpu_ctx, A, B, las_all, trace_all, piles.select(las_all, po), po)
    assert [int(r["status"]) for r in rec] == [4, 4, 0]
    contigs.mask = None
    A2 = gpu_ctx.db(contigs)
    rec, bases = dentist_amd.process_pileups(gpu_ctx, A2, B, las_all, trace_all, piles.select(las_all, po), po)
    assert [int(r["contig_left"]) for r in rec] == [0, 1, 2] and int((rec["status"] == 0).sum()) == 3
    for r in rec:
        gq = int(r["contig_left"])
        cons = bases[r["cons_off"]:r["cons_off"] + r["cons_len"]]
        ins = (sim.revcomp(cons) if r["comp"] else cons)[r["ins_begin"]:r["ins_end"]]
        truth = g[cuts[gq][0] + r["left_aepos"]: cuts[gq + 1][0] + r["right_abpos"]]
        ed, _ = oz.nw(truth, ins)
        assert ed <= max(3, 0.02 * len(truth)), (gq, ed, len(truth))


def test_flank_window_equals_whole_contig_flanks_with_a_repeat_outside_the_window(gpu_ctx):
    """`dentist process` aligns the consensus against the WHOLE flanking contigs (commandline.d:2918-2935); this build hands
    the aligner the last / first flank_window (20 kb) bases by default.  Contigs of 100 kb+, and 2 kb next to a gap planted
    a second time 57 kb further inside the contig (outside the window, inside the whole contig): the copy yields a second
    overlap of consensus and contig, which is not a flank overlap (it does not reach the contig's end) -- crop points,
    reference read, consensus and splice coordinates are identical for flank_window = 20 000 and 0 (whole contigs), and
    the whole-contig run equals the oracle's."""
    g = sim.genome(31337, 330_000).copy()
    g[40_000:42_000] = g[97_500:99_500]
    cuts = [(0, 100_000), (100_400, 215_000), (215_300, 330_000)]
    contigs = sim.SeqDb.from_list([g[a:b].copy() for a, b in cuts])
    reads, _ = sim.reads(4711, g, 1300, 10_000)
    mo = dentist_amd.default_align_opts(k=20, kmer_mod=4, width=64, xdrop=60, algo=1)
    A, B = gpu_ctx.db(contigs), gpu_ctx.db(reads)
    las, trace = gpu_ctx.align_db(A, B, mo, select_best=True)
    res = {}
    for fw in (20_000, 0):
        po = dentist_amd.default_process_opts(algo=1, rounds=2, flank_window=fw)
        piles = dentist_amd.Pileups(las, contigs.off, po)
        assert len(piles) == 2
        res[fw] = dentist_amd.process_pileups(gpu_ctx, A, B, las, trace, piles, po)
    (r1, b1), (r0, b0) = res[20_000], res[0]
    assert np.all(r1["status"] == 0) and np.all(r0["status"] == 0)
    for f in r1.dtype.names:
        if f != "pad":
            assert np.array_equal(r1[f], r0[f]), f
    assert np.array_equal(b1, b0)
    # the oracle with whole contigs
    olas, otrace, _ = oz.align_db(contigs, reads, oz.default_opts(k=20, kmer_mod=4, width=64, xdrop=60, algo=1),
                                  nthreads=os.cpu_count() or 1, select_best=True)
    assert_same_las((las, trace), (olas, otrace))
    exp_piles = pr.collect_spanning(olas, otrace, contigs, reads)
    for i, r in enumerate(r0):
        gq = int(r["contig_left"])
        ex = pr.process_pile(exp_piles[gq], olas, otrace, contigs, reads, gq, rounds=2, nthreads=os.cpu_count() or 1, algo=1,
                             flank_window=0)
        assert ex["status"] == "ok"
        assert np.array_equal(b0[r["cons_off"]:r["cons_off"] + r["cons_len"]], ex["consensus"])
        assert (r["left_aepos"], r["right_abpos"], r["ins_begin"], r["ins_end"]) == \
               (ex["left_aepos"], ex["right_abpos"], ex["ins_begin"], ex["ins_end"])


def test_device_funnel_host_funnel_and_the_fall_back_agree(gpu_ctx, monkeypatch):
    """The alignment funnel of computeQVs (error filter, chains per pair, proper-overlap flags) runs on the device for
    DH-2 (k_pile_funnel); DH_HOST_FUNNEL=1 keeps the records' round trip and the host code, DH_FUNNEL_FALLBACK=1 lets
    the kernel run and then takes the fall-back the kernel's capacities would trigger.  Noisy reads (20 % errors: pairs
    with several local alignments, i.e. real chaining work): every field of every record and every consensus base equal."""
    w = sim.Workload(200_000, 3, 900, 5000, seed=53, err=0.20, spacing=20000, gap_max=900)
    mo = dentist_amd.default_align_opts(algo=1, width=64)
    po = dentist_amd.default_process_opts(rounds=2, algo=1, max_reads=0)
    A, B = gpu_ctx.db(w.contigs), gpu_ctx.db(w.reads)
    las, trace = gpu_ctx.align_db(A, B, mo)
    piles = dentist_amd.Pileups(las, w.contigs.off, po)
    assert len(piles) >= 2
    out = []
    for env in (None, "DH_HOST_FUNNEL", "DH_FUNNEL_FALLBACK"):
        if env:
            monkeypatch.setenv(env, "1")
        out.append(dentist_amd.process_pileups(gpu_ctx, A, B, las, trace, piles, po))
        if env:
            monkeypatch.delenv(env)
    assert (out[0][0]["status"] == 0).sum() >= 2
    for rec, bases in out[1:]:
        assert rec.tobytes() == out[0][0].tobytes() and np.array_equal(bases, out[0][1])


def test_trace_values_left_on_the_device(gpu_ctx, monkeypatch):
    """dh_map_reads with want_sorted & 8: the trace values of every chunk stay in HBM in a buffer the result set owns, and
    dh_process_pileups_set brings over what the cropper reads (cropper.d:446-550: the records of the pile-up reads) and
    nothing else.  Several chunks (DH_ALIGN_CHUNK): identical records; the trace downloaded on demand equals the one the
    plain call copies chunk by chunk; `process` through the set equals `process` on the host arrays, bit for bit."""
    monkeypatch.setenv("DH_ALIGN_CHUNK", "700")
    w = sim.Workload(500_000, 5, 1500, 7000, seed=77, spacing=20000, gap_max=1500)
    mo = dentist_amd.default_align_opts(kmer_mod=4, k=20, width=64, xdrop=60, algo=1)
    po = dentist_amd.default_process_opts(algo=1, rounds=2)
    A, B = gpu_ctx.db(w.contigs), gpu_ctx.db(w.reads)
    las0, trace0, dropped0 = gpu_ctx.map_reads(A, B, mo, po, sorted=False)[:3]
    las1, dtrace, dropped1 = gpu_ctx.map_reads(A, B, mo, po, sorted=False, trace_on_device=True)[:3]
    assert isinstance(dtrace, dentist_amd.DeviceTrace) and dtrace.on_device() and len(dtrace) == len(trace0)
    assert las1.tobytes() == las0.tobytes() and list(dropped0) == list(dropped1)
    gaps = np.stack([np.arange(w.contigs.n - 1), np.arange(1, w.contigs.n)], axis=1).astype(np.int32)
    gp, _ = dentist_amd.scaffold_spanning_pileups(las0, w.contigs.off, w.reads.off, gaps, with_extensions=True,
                                                  min_spanning_reads=po.min_reads)
    piles = gp.select(las0, po)
    assert len(piles) >= 3
    r0, b0 = dentist_amd.process_pileups(gpu_ctx, A, B, las0, trace0, piles, po)
    r1, b1 = dentist_amd.process_pileups(gpu_ctx, A, B, las1, dtrace, piles, po)
    assert dtrace.on_device()                      # (process took what it needed, the rest is still in HBM only)
    assert (r0["status"] == 0).sum() >= 3 and r1.tobytes() == r0.tobytes() and np.array_equal(b0, b1)
    assert np.array_equal(dtrace.numpy(), trace0) and not dtrace.on_device()
    # sorted order with the trace on the device: the records are permuted, their offsets still name the same values
    las2, dtrace2, _ = gpu_ctx.map_reads(A, B, mo, po, sorted=True, trace_on_device=True)[:3]
    las3, trace3, _ = gpu_ctx.map_reads(A, B, mo, po, sorted=True)[:3]
    assert_same_las((las2, dtrace2.numpy()), (las3, trace3))


def test_chains_below_the_default_min_relative_score(gpu_ctx):
    """dh_process_opts.min_relative_score_ppm (--min-relative-score of `dentist process`, commandline.d:2141-2153;
    buildAlignmentChains chaining.d:151-312: components of the chainability relation, the chains within that fraction of
    the pair's best chain, alternate chains with the records they share written once per chain).  Half of the reads of
    every pile-up carry 1.5 kb of foreign bases inside the gap: with a plain read they align as two records that no chain
    joins -- at the default the tile QVs (computeQVs, package.d:486-505) see the better one only, at 0.3 both, which changes
    the ranking of the reference reads.  Product == oracle at 0.3 (crop points, reference read, every consensus base, the
    splice), and the product's own result differs from its default's in at least one pile-up."""
    w = sim.Workload(300_000, 2, 1500, 7000, seed=53, spacing=20000, gap_min=1500, gap_max=2500)
    reads, planted = plant_gap_insertions(w, np.random.default_rng(3))
    assert len(planted) >= 6
    g = dentist_amd.default_align_opts(algo=1, width=64)
    A, B = gpu_ctx.db(w.contigs), gpu_ctx.db(reads)
    las, trace = gpu_ctx.align_db(A, B, g)
    olas, otrace, _ = oz.align_db(w.contigs, reads, oz.default_opts(width=64, algo=1), nthreads=os.cpu_count() or 1)
    assert_same_las((las, trace), (olas, otrace))
    po = dentist_amd.default_process_opts(rounds=2, algo=1, max_reads=30, min_relative_score_ppm=300000)
    po1 = dentist_amd.default_process_opts(rounds=2, algo=1, max_reads=30)
    piles = dentist_amd.Pileups(las, w.contigs.off, po)
    exp_piles = pr.collect_spanning(olas, otrace, w.contigs, reads, max_reads=30)
    assert len(piles) == len(exp_piles) == 2
    rec, bases = dentist_amd.process_pileups(gpu_ctx, A, B, las, trace, piles, po)
    rec1, bases1 = dentist_amd.process_pileups(gpu_ctx, A, B, las, trace, piles, po1)
    differs = 0
    for i in range(len(piles)):
        gap, _ = piles.get(i)
        exp = pr.process_pile(exp_piles[gap], olas, otrace, w.contigs, reads, gap, rounds=2, nthreads=os.cpu_count() or 1, algo=1,
                              min_rel_score=0.3)
        exp1 = pr.process_pile(exp_piles[gap], olas, otrace, w.contigs, reads, gap, rounds=2, nthreads=os.cpu_count() or 1, algo=1)
        for r, b, e in ((rec[i], bases, exp), (rec1[i], bases1, exp1)):
            assert e["status"] == "ok" and r["status"] == 0
            assert (r["crop_left"], r["crop_right"]) == (e["cropL"], e["cropR"])
            assert r["nreads"] == e["pile"].n and r["ref_read"] == e["ref_idx"]
            assert np.array_equal(b[r["cons_off"]:r["cons_off"] + r["cons_len"]], e["consensus"]), f"gap {gap}: consensus differs"
            assert (r["left_aepos"], r["right_abpos"], r["ins_begin"], r["ins_end"]) == \
                   (e["left_aepos"], e["right_abpos"], e["ins_begin"], e["ins_end"])
        assert ((exp["chained_las"]["flags"] & 0x20) == 0).sum() > ((exp1["chained_las"]["flags"] & 0x20) == 0).sum()
        differs += int(rec[i]["ref_read"] != rec1[i]["ref_read"])
    assert differs >= 1
    with pytest.raises(dentist_amd.DhError):
        dentist_amd.process_pileups(gpu_ctx, A, B, las, trace, piles, dentist_amd.default_process_opts(min_relative_score_ppm=1000001))


def test_pairs_without_a_reference_read_candidate_are_not_aligned_and_nothing_changes(gpu_ctx, monkeypatch):
    """The pile-up all-vs-all (`daligner pile.db pile.db`, processPileUps/package.d:474-485) feeds the error filter, the
    chaining, the tile QVs and the first consensus round -- all of which read the overlaps of the ALLOWED reference reads
    only (the reads that span the gap, :461-472; ranking :518-568).  The product therefore does not align two reads of
    which neither may serve as reference read, and makes only the allowed read's record of a mixed pair; with
    DH_PILE_ALL_PAIRS=1 it aligns every pair, as daligner would.  Graph pile-ups with extension entries, no read cap:
    every record field and every consensus base must be the same either way, with visibly less alignment work."""
    w = sim.Workload(700_000, 7, 4000, 9000, seed=97, spacing=30000, gap_max=1200)
    mo = dentist_amd.default_align_opts(kmer_mod=4, k=20, width=64, xdrop=60, algo=1)
    po = dentist_amd.default_process_opts(algo=1, max_reads=0)
    gaps = np.stack([np.arange(w.contigs.n - 1), np.arange(1, w.contigs.n)], axis=1).astype(np.int32)
    A, B = gpu_ctx.db(w.contigs), gpu_ctx.db(w.reads)
    las, trace, _ = gpu_ctx.map_reads(A, B, mo, po, sorted=False, candidates=False)[:3]
    gp, _ = dentist_amd.scaffold_spanning_pileups(las, w.contigs.off, w.reads.off, gaps, with_extensions=True,
                                                  min_spanning_reads=po.min_reads)
    piles = gp.select(las, po)
    tri = piles.flat()[2]
    assert ((tri[:, 1] < 0) | (tri[:, 2] < 0)).sum() > 0.2 * len(tri)     # extension entries: reads that cannot be the reference
    gpu_ctx.cum_stats(reset=True)
    rec, bases = dentist_amd.process_pileups(gpu_ctx, A, B, las, trace, piles, po)
    cut = gpu_ctx.cum_stats().as_dict()
    monkeypatch.setenv("DH_PILE_ALL_PAIRS", "1")
    gpu_ctx.cum_stats(reset=True)
    rec2, bases2 = dentist_amd.process_pileups(gpu_ctx, A, B, las, trace, piles, po)
    full = gpu_ctx.cum_stats().as_dict()
    assert (rec["status"] == 0).sum() >= 5
    for f in rec.dtype.names:
        if f not in ("cons_off", "pad"):
            assert np.array_equal(rec[f], rec2[f]), f
    for a, b in zip(rec, rec2):
        assert np.array_equal(bases[a["cons_off"]:a["cons_off"] + a["cons_len"]], bases2[b["cons_off"]:b["cons_off"] + b["cons_len"]])
    assert cut["alignments"] < full["alignments"], (cut["alignments"], full["alignments"])
