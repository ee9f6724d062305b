"""DAZZ_DB on-disk format: pinned by the reference's own md5s (tests/test-commands.sh:54-61)."""
import hashlib
import os

import numpy as np

import dentist_amd
from dentist_amd import sim

GOLD = os.path.join(os.path.dirname(__file__), "golden")
MD5 = {".assembly-reference.bps": "345a2b75e35a895ccada7d6a53b61b3f",
       ".assembly-reference.hdr": "07f7012c84f6fd336e21379c354a5ea9",
       ".assembly-reference.idx": "0066d6d4ac213b558dd10f48e2a0b7b1"}


def _md5(p):
    return hashlib.md5(open(p, "rb").read()).hexdigest()


def test_fasta2dam_image_matches_reference_checksums(tmp_path):
    """`fasta2DAM -i data/assembly-reference.dam` of the embedded assembly: .idx, .bps and .hdr are
    byte-identical to what DAZZ_DB writes (the reference pins their md5)."""
    fasta = open(os.path.join(GOLD, "test_commands_assembly_reference.fasta")).read()
    path = str(tmp_path / "assembly-reference.dam")
    dentist_amd.dazz_create_dam(path, fasta)
    for name, want in MD5.items():
        assert _md5(str(tmp_path / name)) == want, name
    stub = open(path).read().split("\n")
    assert stub[0] == "files =         1" and stub[2].startswith("blocks =")   # dazzler.d:4383-4479 grammar
    db = dentist_amd.DazzDb(path)
    seq = "".join(fasta.split("\n")[1:])
    assert db.n == 1 and sim.decode(db.seq(0)) == seq and db.headers == [">chr3R"]


def test_dam_splits_scaffolds_at_n_runs_and_dbsplit_trims(tmp_path):
    fasta = open(os.path.join(GOLD, "test_commands_assembly_reference.fasta")).read()
    gap = open(os.path.join(GOLD, "test_commands_gap_seq.txt")).read().strip()
    seq = "".join(fasta.split("\n")[1:])
    test_seq = seq.replace(gap, "n" * len(gap))        # _test_assembly, tests/test-commands.sh:92-99
    short = "acgtacgtac"                                # a 10 bp contig that -x20 must hide
    text = ">chr3R\n" + test_seq + "\n>tiny\n" + short + "nn" + seq[:50] + "\n"
    path = str(tmp_path / "assembly-test.dam")
    dentist_amd.dazz_create_dam(path, text)
    db = dentist_amd.DazzDb(path)
    assert db.n == 4
    assert [int(x) for x in db.fpulse] == [0, 2097, 0, 12] and [int(x) for x in db.origin] == [0, 1, 0, 1]
    assert sim.decode(db.seq(0)) == seq[:2000] and sim.decode(db.seq(1)) == seq[2097:]
    assert db.headers[:3] == [">chr3R", ">chr3R", ">tiny"]
    dentist_amd.dazz_split(path, cutoff=20, all_reads=True, size_mb=200)   # DBsplit -x20
    db = dentist_amd.DazzDb(path)
    assert db.n == 3 and sim.decode(db.seq(2)) == seq[:50]                 # trimmed ids skip the 10 bp contig
    stub = open(path).read()
    assert "blocks =         1" in stub and "cutoff =        20" in stub
    blk = dentist_amd.DazzDb(str(tmp_path / "assembly-test.1"))
    assert blk.n == 3 and blk.first_id == 0


def test_db_with_pacbio_headers_and_blocks(tmp_path):
    g = sim.genome(3, 5000)
    rd, _ = sim.reads(4, g, 12, 900)
    recs = []
    for i in range(rd.n):
        s = sim.decode(rd.seq(i))
        recs.append(f">sim/{i + 1}/0_{len(s)} RQ=0.850\n" + "\n".join(s[k:k + 100] for k in range(0, len(s), 100)))
    path = str(tmp_path / "reads.db")
    dentist_amd.dazz_create_db(path, "\n".join(recs) + "\n")
    dentist_amd.dazz_split(path, cutoff=20, all_reads=True, size_mb=200)
    db = dentist_amd.DazzDb(path)
    assert db.n == rd.n and [int(x) for x in db.origin] == list(range(1, rd.n + 1))
    for i in range(rd.n):
        assert np.array_equal(db.seq(i), rd.seq(i))
    with open(path) as f:
        assert f.readline().startswith("files =")


def test_mask_track_files_follow_the_reference_layout(tmp_path):
    """writeMask / readMask, source/dentist/dazzler.d:4943-5170: .anno = int32 nreads, int32 0,
    int64 byte offsets[nreads + 1]; .data = int32 (begin, end) pairs; files `.<db>.<mask>.anno/.data`."""
    import struct
    g = sim.genome(9, 3000)
    text = ">a\n" + sim.decode(g[:1000]) + "\n>b\n" + sim.decode(g[1000:1010]) + "\n>c\n" + sim.decode(g[1010:]) + "\n"
    path = str(tmp_path / "ref.dam")
    dentist_amd.dazz_create_dam(path, text)
    dentist_amd.dazz_split(path, cutoff=20)            # hides the 10 bp contig `b`
    ptr = np.asarray([0, 2, 2, 3], dtype=np.int64)     # written for the untrimmed DB (3 contigs)
    iv = np.asarray([10, 50, 700, 800, 5, 25], dtype=np.int32)
    dentist_amd.dazz_write_mask(path, "dentist-self", ptr, iv)
    anno = open(tmp_path / ".ref.dentist-self.anno", "rb").read()
    data = open(tmp_path / ".ref.dentist-self.data", "rb").read()
    assert struct.unpack("<ii", anno[:8]) == (3, 0)
    assert struct.unpack("<4q", anno[8:]) == (0, 16, 16, 24)
    assert struct.unpack("<6i", data) == (10, 50, 700, 800, 5, 25)
    db = dentist_amd.DazzDb(path)                        # trimmed view: contigs a and c
    p2, iv2 = db.read_mask("dentist-self")
    assert p2.tolist() == [0, 2, 3] and iv2.tolist() == [10, 50, 700, 800, 5, 25]
    import pytest
    with pytest.raises(dentist_amd.DhError):
        db.read_mask("missing")
