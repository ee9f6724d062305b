"""The drop-in executables (tools/daligner, tools/damapper): argv + file contract of
source/dentist/dazzler.d:6121-6170 and getLasFile :4339-4354, checked against the direct C-ABI call."""
import os
import subprocess

import numpy as np
import pytest

import dentist_amd
from dentist_amd import sim
from helpers import assert_same_las

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
    exp = gpu_ctx.align_db(A, B, dentist_amd.default_align_opts(), select_best=True)
    assert_same_las((las, trace), exp)
    assert os.path.exists(tmp_path / "reads.1.ref.las")
    back, _, _ = dentist_amd.las_read(str(tmp_path / "reads.1.ref.las"))
    assert len(back) > 0 and set(back["aread"].tolist()) <= set(range(w.reads.n))
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
    expm = gpu_ctx.align_db(Am, B, dentist_amd.default_align_opts(), select_best=True)
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
