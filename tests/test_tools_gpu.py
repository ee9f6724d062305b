"""The drop-in executables (tools/daligner, tools/damapper): argv + file contract of
source/dentist/dazzler.d:6121-6170 and getLasFile :4339-4354, checked against the direct C-ABI call."""
import os
import subprocess

import numpy as np
import pytest

import dentist_amd
from dentist_amd import sim
from helpers import assert_same_las, tandem_reads

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def fasta_dam(db, name="scaf"):
    out = []
    for i in range(db.n):
        s = sim.decode(db.seq(i))
        out.append(f">{name}{i}\n" + "\n".join(s[k:k + 80] for k in range(0, len(s), 80)))
    return "\n".join(out) + "\n"


def fasta_db(db):
    out = []
    for i in range(db.n):
        s = sim.decode(db.seq(i))
        out.append(f">sim/{i + 1}/0_{len(s)} RQ=0.850\n" + "\n".join(s[k:k + 100] for k in range(0, len(s), 100)))
    return "\n".join(out) + "\n"


def test_damapper_executable_matches_the_library(gpu_ctx, tmp_path):
    w = sim.Workload(200_000, 2, 300, 4000, seed=31, spacing=15000)
    ref, rds = str(tmp_path / "ref.dam"), str(tmp_path / "reads.db")
    dentist_amd.dazz_create_dam(ref, fasta_dam(w.contigs))
    dentist_amd.dazz_split(ref, cutoff=20)
    dentist_amd.dazz_create_db(rds, fasta_db(w.reads))
    dentist_amd.dazz_split(rds, cutoff=20)
    # literal instance of the reference's call: tests/test-commands.sh:197
    r = subprocess.run([os.path.join(ROOT, "tools", "damapper"), "-C", "-T1", "-e0.7", "-mdust", "ref", "reads.1"],
                       cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    las, trace, ts = dentist_amd.las_read(str(tmp_path / "ref.reads.1.las"))
    assert ts == 100
    A, B = gpu_ctx.db(w.contigs), gpu_ctx.db(w.reads)

    def library(Adb):
        """What the executable runs: DH-2 (tiled band), chains with damapper's default -n.85, discarded chains not written."""
        dentist_amd.lib().dh_set_near_best(850000)
        try:
            el, et = gpu_ctx.align_db(Adb, B, dentist_amd.default_align_opts(algo=1, width=64), select_best=True)
        finally:
            dentist_amd.lib().dh_set_near_best(0)
        return el[(el["flags"] & 0x20) == 0], et
    exp = library(A)
    assert_same_las((las, trace), exp)
    assert np.all((las["flags"] & (0x4 | 0x8)) != 0)   # every record is part of a chain
    # -C: the transposed file of the same pass (dh_align_db_transposed), chains with -n.85 applied to it as well
    back, btrace, _ = dentist_amd.las_read(str(tmp_path / "reads.1.ref.las"))
    dentist_amd.lib().dh_set_near_best(850000)
    try:
        (_, _), (bl, bt) = gpu_ctx.align_db_transposed(A, B, dentist_amd.default_align_opts(algo=1, width=64), select_best=True)
    finally:
        dentist_amd.lib().dh_set_near_best(0)
    assert_same_las((back, btrace), (bl[(bl["flags"] & 0x20) == 0], bt))
    assert len(back) > 0.9 * len(las) and set(back["aread"].tolist()) <= set(range(w.reads.n))
    # -m<track>: an existing track is applied (fewer seeds, same reads mapped), a missing one is reported
    db = dentist_amd.DazzDb(ref)
    ptr = np.zeros(db.n + 1, dtype=np.int64)
    iv = []
    for i in range(db.n):
        iv += [1000, 1400, 5000, 5600]
        ptr[i + 1] = len(iv) // 2
    dentist_amd.dazz_write_mask(ref, "dentist-self", ptr, np.asarray(iv, dtype=np.int32))
    r = subprocess.run([os.path.join(ROOT, "tools", "damapper"), "-T1", "-e0.7", "-mdentist-self", "-mtan", "ref", "reads.1"],
                       cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "`tan` not found for ref" in r.stderr and "`dentist-self` not found for ref" not in r.stderr, r.stderr
    w.contigs.mask = (ptr, np.asarray(iv, dtype=np.int32))
    Am = gpu_ctx.db(w.contigs)
    expm = library(Am)
    lasm, tracem, _ = dentist_amd.las_read(str(tmp_path / "ref.reads.1.las"))
    assert_same_las((lasm, tracem), expm)
    # a missing DB is an error with a non-zero exit status (DazzlerCommandException, dazzler.d:6586-6591)
    r = subprocess.run([os.path.join(ROOT, "tools", "damapper"), "-C", "nope", "reads.1"], cwd=tmp_path,
                       capture_output=True, text=True)
    assert r.returncode != 0 and "not found" in r.stderr


def test_daligner_pile_up_call(gpu_ctx, tmp_path):
    g = sim.genome(21, 20000)
    pile, _ = sim.reads(22, g, 25, 5000)
    p = str(tmp_path / "pileup-1b-2f.db")
    dentist_amd.dazz_create_db(p, fasta_db(pile))
    dentist_amd.dazz_split(p, cutoff=0)
    # commandline.d:2886-2902: daligner -T<a> -B -s126 -l500 -e0.7 -mdust db db
    r = subprocess.run([os.path.join(ROOT, "tools", "daligner"), "-T1", "-B", "-s126", "-l500", "-e0.7", "-mdust", p, p],
                       cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    las, trace, ts = dentist_amd.las_read(str(tmp_path / "pileup-1b-2f.pileup-1b-2f.las"))
    assert ts == 126
    raw = open(tmp_path / "pileup-1b-2f.pileup-1b-2f.las", "rb").read()
    assert len(raw) == 12 + 40 * len(las) + 2 * len(trace)      # 16-bit traces above tspace 125
    d = gpu_ctx.db(pile)
    # one DB against itself = symmetric mode: each pair aligned once, both records written
    exp = gpu_ctx.align_db(d, d, dentist_amd.default_align_opts(tspace=126, skip_self=2, max_la=64, max_cand=128))
    assert len(las) % 2 == 0 and len(las) > 0
    assert_same_las((las, trace), exp)


def test_daligner_on_two_dbs_writes_both_files_from_one_pass(gpu_ctx, tmp_path):
    """`daligner <flags> A B` without -A writes A.B.las and B.A.las (dazzler.d:6121-6140; the workflow's block pairs of the
    self alignment, Snakefile:998-1022).  Both come out of one pass: the second file holds the transposed pair of every
    alignment (dh_align_db_transposed, DH-2).  Files == library call; every pair of the first file has its transposed record."""
    g = sim.genome(41, 30000)
    ra, _ = sim.reads(42, g, 20, 5000)
    rb, _ = sim.reads(43, g, 24, 5000)
    pa, pb = str(tmp_path / "a.db"), str(tmp_path / "b.db")
    for p, r in ((pa, ra), (pb, rb)):
        dentist_amd.dazz_create_db(p, fasta_db(r))
        dentist_amd.dazz_split(p, cutoff=0)
    r = subprocess.run([os.path.join(ROOT, "tools", "daligner"), "-T1", "-s126", "-l500", "-e0.7", "a", "b"], cwd=tmp_path,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    ab, abt, _ = dentist_amd.las_read(str(tmp_path / "a.b.las"))
    ba, bat, _ = dentist_amd.las_read(str(tmp_path / "b.a.las"))
    A, B = gpu_ctx.db(ra), gpu_ctx.db(rb)
    (el, et), (tl, tt) = gpu_ctx.align_db_transposed(A, B, dentist_amd.default_align_opts(tspace=126, min_len=500, algo=1, width=64))
    assert_same_las((ab, abt), (el, et))
    assert_same_las((ba, bat), (tl, tt))
    assert len(ab) > 50 and len(ba) > 0.9 * len(ab)
    pairs = set(zip(ba["bread"].tolist(), ba["aread"].tolist(), (ba["flags"] & 1).tolist()))
    have = sum((a, b, c) in pairs for a, b, c in zip(ab["aread"].tolist(), ab["bread"].tolist(), (ab["flags"] & 1).tolist()))
    assert have > 0.95 * len(ab)
    # -A: the first file only
    os.remove(tmp_path / "b.a.las")
    r = subprocess.run([os.path.join(ROOT, "tools", "daligner"), "-A", "-T1", "-s126", "-l500", "-e0.7", "a", "b"], cwd=tmp_path,
                       capture_output=True, text=True)
    assert r.returncode == 0 and not os.path.exists(tmp_path / "b.a.las")


def test_datander_tanmask_sequence(gpu_ctx, tmp_path):
    """The tandem-mask sequence of the workflow (snakemake/Snakefile:1056-1123): `datander -T<n> -s126 -l500 -e0.7 <dam>.<block>`
    (commandline.d:2866-2876) writes TAN.<dam>.<block>.las, `TANmask` turns it into the block's `tan` track.  The file equals
    the library call (dh_align_opts.skip_self = 3 with datander's own -k12 -w4), the mask covers every planted array and
    nothing in the reads without one."""
    db, truth = tandem_reads()
    ref = str(tmp_path / "asm.dam")
    dentist_amd.dazz_create_dam(ref, fasta_dam(db))
    dentist_amd.dazz_split(ref, cutoff=20)
    r = subprocess.run([os.path.join(ROOT, "tools", "datander"), "-T1", "-s126", "-l500", "-e0.7", "asm.1"],
                       cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    las, trace, ts = dentist_amd.las_read(str(tmp_path / "TAN.asm.1.las"))
    assert ts == 126 and len(las) > 0
    d = gpu_ctx.db(db)
    exp = gpu_ctx.align_db(d, d, dentist_amd.default_align_opts(skip_self=3, strands=1, k=12, band_shift=4, min_len=500,
                                                                tspace=126, algo=1, width=64))
    assert_same_las((las, trace), exp)
    r = subprocess.run([os.path.join(ROOT, "tools", "TANmask"), "-v", "-ntan", ref, "TAN.asm.1.las"], cwd=tmp_path,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert os.path.exists(tmp_path / ".asm.1.tan.anno")       # the block's track; Catrack makes the DB's (Snakefile:1111-1123)
    r = subprocess.run([os.path.join(ROOT, "tools", "Catrack"), "-v", ref, "tan"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    ptr, iv = dentist_amd.DazzDb(ref).read_mask("tan")
    planted = {t[0]: t for t in truth}
    for i in range(db.n):
        ivs = [tuple(iv[2 * x:2 * x + 2]) for x in range(ptr[i], ptr[i + 1])]
        if i not in planted:
            assert ivs == []
            continue
        _, b, e, _ = planted[i]
        covered = sum(min(y, e) - max(x, b) for x, y in ivs if min(y, e) > max(x, b))
        assert covered >= 0.9 * (e - b) and all(b - 80 <= x and y <= e + 80 for x, y in ivs)


def tool(name, *args, cwd=None, stdin=None):
    r = subprocess.run([os.path.join(ROOT, "tools", name), *args], cwd=cwd, input=stdin, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, (name, args, r.stderr)
    return r.stdout


def test_process_tool_sequence_of_one_pile_up(gpu_ctx, tmp_path):
    """The literal per-pile-up command lines of `dentist process` (SURVEY Appendix A;
    processPileUps/package.d:474-516, 600-619): fasta2DB -i | DBsplit | DBdust | daligner -mdust |
    DAScover | DASqv -c | DBdump -r -i | computeintrinsicqv | daccord --eprofonly |
    daccord -f -I<i>,<i> | fasta2DAM -i | DBsplit -a -- against the stage-level library calls."""
    g = sim.genome(41, 12000)
    g[3000:3060] = np.tile([0, 1], 30)          # a microsatellite for DBdust to find
    pile, _ = sim.reads(42, g[1000:7000], 18, 6000)
    p = str(tmp_path / "pileup-3b-4f.db")
    tool("fasta2DB", "-i", p, stdin=fasta_db(pile))
    tool("DBsplit", "-x0", "-a", p)
    tool("DBdust", p)
    dz = dentist_amd.DazzDb(p)
    ptr, iv = dz.read_mask("dust")
    d = gpu_ctx.db(pile)
    d.dust()
    eptr, eiv = d.get_mask()
    assert np.array_equal(ptr, eptr) and np.array_equal(iv, eiv) and len(iv) > 0
    tool("daligner", "-T1", "-B", "-s126", "-l500", "-e0.7", "-mdust", p, p, cwd=tmp_path)
    las_path = str(tmp_path / "pileup-3b-4f.pileup-3b-4f.las")
    las, trace, ts = dentist_amd.las_read(las_path)
    exp = gpu_ctx.align_db(d, d, dentist_amd.default_align_opts(tspace=126, skip_self=2, max_la=64, max_cand=128))
    assert_same_las((las, trace), exp)
    # DAScover / DASqv -c<cov>, then the QVs as DENTIST reads them: DBdump -r -i (dazzler.d:2877-2897)
    tool("DAScover", "-v", p, las_path)
    tool("DASqv", "-v", f"-c{pile.n}", p, las_path)
    maxtiles = max((pile.length(i) + 125) // 126 for i in range(pile.n))
    qv = dentist_amd.tile_qv(gpu_ctx, d, las, trace, 126, pile.n, maxtiles)
    lines = [ln.split() for ln in tool("DBdump", "-r", "-i", p).splitlines() if ln and ln[0] in "RI"]
    dec = lambda c: ord(c) - ord("a") if c.islower() else 26 + ord(c) - ord("A")   # noqa: E731
    for i in range(pile.n):
        assert lines[2 * i] == ["R", str(i + 1)]
        nt = (pile.length(i) + 125) // 126
        assert int(lines[2 * i + 1][1]) == nt
        assert [dec(c) for c in lines[2 * i + 1][2]] == [min(int(x), 50) for x in qv[i, :nt]]
    # consensus of read 2: computeintrinsicqv, error profile pass, then FASTA on stdout into fasta2DAM
    tool("computeintrinsicqv", f"-d{pile.n}", p, las_path)
    tool("daccord", "-t1", "-I2,2", "--eprofonly", las_path, p)
    assert os.path.exists(las_path + ".eprof")
    fa = tool("daccord", "-f", "-t1", "-I2,2", las_path, p)
    out_dam = str(tmp_path / "pileup-3b-4f-daccord-I2-2.dam")
    tool("fasta2DAM", "-i", out_dam, stdin=fa)
    tool("DBsplit", "-a", out_dam)
    cons = dentist_amd.DazzDb(out_dam)
    exp_cons = dentist_amd.consensus(gpu_ctx, d, las, trace, 126, 2, rounds=3)
    assert cons.n == 1 and np.array_equal(cons.seq(0), exp_cons)
    # flank re-alignment call: daligner -A ... contigs consensus writes only contigs.consensus.las
    contigs = sim.SeqDb.from_list([g[:2500], g[5500:]])
    cdam = str(tmp_path / "contigs-3b-4f.dam")
    tool("fasta2DAM", "-i", cdam, stdin=fasta_dam(contigs))
    tool("DBsplit", "-a", cdam)
    tool("DBdust", cdam)
    tool("daligner", "-A", "-B", "-s126", "-T1", "-mdust", "-mrep", "-l126", "-e0.7", cdam, out_dam, cwd=tmp_path)
    fl, _, _ = dentist_amd.las_read(str(tmp_path / "contigs-3b-4f.pileup-3b-4f-daccord-I2-2.las"))
    assert len(fl) >= 1 and not os.path.exists(tmp_path / "pileup-3b-4f-daccord-I2-2.contigs-3b-4f.las")
