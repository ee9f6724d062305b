"""`dentist output` (dh_output_assembly, host only): the three writers and the join policy on a hand-made case.
Reference: source/dentist/commands/output.d:305-348 (graph + enforceJoinPolicy, common/scaffold.d:642-715),
:454-573 (AGP), :743-759 (header), :782-925 (FASTA / BED), common/insertions.d:110-284 (splice arithmetic)."""
import numpy as np

import dentist_amd
from dentist_amd import sim


def case():
    rng = np.random.default_rng(3)
    c = [rng.integers(0, 4, n).astype(np.uint8) for n in (120, 80, 60, 90)]
    contigs = sim.SeqDb.from_list(c)
    scaffold_of = [0, 0, 0, 1]            # scaffold A = contigs 1-3, scaffold B = contig 4
    headers = ["scafA\textra", "scafB"]
    gap_len = [30, 20, 0, 0]
    cons = [rng.integers(0, 4, 50).astype(np.uint8), rng.integers(0, 4, 40).astype(np.uint8)]
    rec = np.zeros(3, dtype=dentist_amd.INSERTION_DTYPE)
    # gap 1|2 closed by a forward consensus, gap 3|4 (between the scaffolds) by a complemented one; gap 2|3 failed
    rec[0]["contig_left"], rec[0]["left_aepos"], rec[0]["right_abpos"] = 0, 110, 5
    rec[0]["ins_begin"], rec[0]["ins_end"], rec[0]["cons_off"], rec[0]["cons_len"], rec[0]["ref_read_id"] = 10, 42, 0, 50, 7
    rec[1]["contig_left"], rec[1]["status"] = 1, 4
    rec[2]["contig_left"], rec[2]["left_aepos"], rec[2]["right_abpos"], rec[2]["comp"] = 2, 55, 3, 1
    rec[2]["ins_begin"], rec[2]["ins_end"], rec[2]["cons_off"], rec[2]["cons_len"], rec[2]["ref_read_id"] = 4, 30, 50, 40, 2
    ids = (np.asarray([7, 3, 11, 2, 9], dtype=np.int32), np.asarray([0, 3, 3, 5], dtype=np.int64))
    return contigs, scaffold_of, headers, gap_len, rec, np.concatenate(cons), ids, c, cons


def text(contigs, lo, hi):
    return sim.decode(contigs[lo:hi])


def test_scaffold_gaps_policy_fasta_agp_bed(tmp_path):
    contigs, sof, hd, gl, rec, bases, ids, c, cons = case()
    fa, bed, agp = (str(tmp_path / n) for n in ("out.fasta", "out.bed", "out.agp"))
    dropped = dentist_amd.output_assembly(fa, contigs, sof, hd, gl, rec, bases, read_ids=ids, bed_path=bed, agp_path=agp,
                                          agp_dazzler=True, line_width=0, tool="dentist-hip test", input_assembly="ref.dam")
    assert dropped == 1   # the insertion between scaffold A and B: joinPolicy scaffoldGaps
    ins = sim.decode(cons[0][10:42]).upper()
    a = text(c[0], 0, 110) + ins + text(c[1], 5, 80) + "n" * 20 + text(c[2], 0, 60)
    assert open(fa).read() == f">scafA\tscaffold-1\n{a}\n>scafB\tscaffold-4\n{text(c[3], 0, 90)}\n"
    assert open(bed).read() == "scafA\t110\t143\tcontigs-1-2|reads-4-8-12\n"
    lines = open(agp).read().split("\n")
    assert lines[0] == "##agp-version\t2.1" and lines[1] == "# TOOL: dentist-hip test" and lines[2] == "# INPUT_ASSEMBLY: ref.dam"
    assert lines[3].startswith("# object\tobject_beg\tobject_end\tpart_number\tcomponent_type")
    assert lines[4:] == [
        "scafA\t1\t110\t1\tW\t1\t0\t110\t-\tna",
        "scafA\t111\t142\t2\tO\treads-4-8-12\t10\t42\t-\tclone_contig",
        "scafA\t143\t217\t3\tW\t2\t155\t230\t-\tna",            # contig 2 begins at 120 + 30 in its scaffold
        "scafA\t218\t237\t4\tN\t20\tscaffold\tyes\tna\tunspecified",
        "scafA\t238\t297\t5\tW\t3\t250\t310\t-\tna",
        "scafB\t1\t90\t1\tW\t4\t0\t90\t-\tna",
        "",
    ]


def test_scaffolds_policy_joins_two_scaffolds(tmp_path):
    contigs, sof, hd, gl, rec, bases, ids, c, cons = case()
    fa, bed, agp = (str(tmp_path / n) for n in ("out.fasta", "out.bed", "out.agp"))
    dropped = dentist_amd.output_assembly(fa, contigs, sof, hd, gl, rec, bases, read_ids=ids, bed_path=bed, agp_path=agp,
                                          join_policy="scaffolds", agp_skip_read_ids=True, line_width=0)
    assert dropped == 0
    ins0 = sim.decode(cons[0][10:42]).upper()
    ins2 = sim.decode(sim.revcomp(cons[1])[4:30]).upper()
    a = text(c[0], 0, 110) + ins0 + text(c[1], 5, 80) + "n" * 20 + text(c[2], 0, 55) + ins2 + text(c[3], 3, 90)
    assert open(fa).read() == f">scafA\tscaffold-1\n{a}\n"
    assert open(bed).read().split("\n")[1] == f"scafA\t{110 + 32 + 75 + 20 + 55}\t{110 + 32 + 75 + 20 + 55 + 26 + 1}\tcontigs-3-4|reads-3-10"
    last = [ln for ln in open(agp).read().split("\n") if ln][-2:]
    # the slice of a complemented insertion is given on the stored consensus (insertions.d:230-284)
    assert last[0] == "scafA\t293\t318\t6\tO\t2 reads\t10\t36\t+\tclone_contig"
    assert last[1] == "scafA\t319\t405\t7\tW\tscafB\t3\t90\t-\tna"


def test_reference_read_alone_without_ids_and_wrapping(tmp_path):
    contigs, sof, hd, gl, rec, bases, ids, c, cons = case()
    fa, bed = str(tmp_path / "o.fasta"), str(tmp_path / "o.bed")
    dentist_amd.output_assembly(fa, contigs, sof, hd, gl, rec, bases, bed_path=bed, line_width=50)
    assert open(bed).read() == "scafA\t110\t143\tcontigs-1-2|reads-8\n"
    body = open(fa).read().split("\n")
    assert all(len(ln) <= 50 for ln in body if not ln.startswith(">"))
