"""DH-2 without a GPU: the lane state machine of dentist_amd/csrc/dh_tile.h (the code k_tile runs --
tile set-up, bit-vector column step, scan, trace pairs, candidate loop, records) compiled for the CPU
(tests/native/tile_host.cpp) against the oracle's plain-DP restatement (oracle/align.c: extend_tiled).
Bit-exact: every record field and every trace value."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from dentist_amd import sim
from helpers import assert_same_las, check_trace_invariants
from oracle import pyoracle as oz

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DHCAND = np.dtype([("score", "<i4"), ("aseq", "<i4"), ("apos", "<i4"), ("bpos", "<i4")])


@pytest.fixture(scope="module")
def host_lib():
    path = os.path.join(ROOT, "tests", "native", "libdh_tile_host.so")
    if not os.path.exists(path):
        subprocess.run(["make", "-C", ROOT, "tests/native/libdh_tile_host.so"], check=True)
    L = ctypes.CDLL(path)
    L.dh_tile_host_align.restype = ctypes.c_long
    L.dh_tile_host_align.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_int32, ctypes.POINTER(oz.Opts), ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p]
    return L


def run_host(L, A, B, o):
    cand, ncand = oz.seed_candidates_all(A, B, o)
    dc = np.zeros(cand.shape, dtype=DHCAND)
    for f in ("score", "aseq", "apos", "bpos"):
        dc[f] = cand[f]
    dc = np.ascontiguousarray(dc)
    maxlen = int(max((A.off[1:] - A.off[:-1]).max(), (B.off[1:] - B.off[:-1]).max()))
    nbmax = maxlen // o.tspace + 3
    las = np.zeros(2 * B.n * o.max_la, dtype=oz.LA_DTYPE)
    cap = int(2 * B.n * o.max_la * 2 * (2 * nbmax + 2))
    trace = np.zeros(cap, dtype=np.uint16)
    counters = np.zeros(2, dtype=np.uint64)
    n = L.dh_tile_host_align(A.bases.ctypes.data, A.off.ctypes.data, A.n, B.bases.ctypes.data, B.off.ctypes.data, B.n,
                             ctypes.byref(o), dc.ctypes.data, ncand.ctypes.data, nbmax, las.ctypes.data,
                             trace.ctypes.data, cap, counters.ctypes.data)
    assert n >= 0, n
    return las[:n], trace, counters


@pytest.mark.parametrize("kw", [
    dict(seed=3, tspace=100, err=0.13),
    dict(seed=4, tspace=126, err=0.13),
    dict(seed=5, tspace=100, err=0.20, xdrop=60),
    dict(seed=6, tspace=64, err=0.05),
    dict(seed=7, tspace=128, err=0.13, min_len=100),
    dict(seed=3, tspace=100, err=0.13, width=32),      # the band of 32 rows (32-bit vectors)
    dict(seed=5, tspace=126, err=0.20, xdrop=60, width=32),
    dict(seed=8, tspace=128, err=0.13, min_len=100, width=32),
])
def test_lane_state_machine_equals_oracle(host_lib, kw):
    w = sim.Workload(300_000, 4, 500, 5000, seed=kw["seed"], err=kw["err"], spacing=20000, gap_max=800)
    o = oz.default_opts(algo=1, width=kw.get("width", 64), k=16, kmer_mod=2, tspace=kw["tspace"], xdrop=kw.get("xdrop", 120),
                        min_len=kw.get("min_len", 500))
    exp_las, exp_trace, stats = oz.align_db(w.contigs, w.reads, o, nthreads=4, sort=False)
    las, trace, counters = run_host(host_lib, w.contigs, w.reads, o)
    assert len(exp_las) >= 450
    assert_same_las((las, trace), (exp_las, exp_trace))
    check_trace_invariants(las, trace, o.tspace)
    assert int(counters[0]) == stats[3] and int(counters[1]) == stats[2]


def test_short_and_ragged_inputs(host_lib):
    """Reads shorter than a tile / than the band, reads hanging over both contig ends, a read equal to
    its contig, tiny contigs: the ends of A' and B' inside the first tile on either side of the seed."""
    rng = np.random.default_rng(11)
    g = rng.integers(0, 4, 6000).astype(np.uint8)
    contigs = sim.SeqDb.from_list([g[:3000], g[3100:3160], g[3200:6000], g[100:140]])
    reads = [g[2900:3000], g[2950:3160], g[0:3000], g[3150:3300], g[10:70], sim.revcomp(g[3300:5900]), g[3100:3160],
             g[2990:3110], g[20:52]]
    B = sim.SeqDb.from_list(reads)
    for width in (64, 32):
        o = oz.default_opts(algo=1, width=width, k=12, hmin=20, min_len=20, tspace=100)
        exp_las, exp_trace, _ = oz.align_db(contigs, B, o, nthreads=1, sort=False)
        las, trace, _ = run_host(host_lib, contigs, B, o)
        assert len(exp_las) >= 8
        assert_same_las((las, trace), (exp_las, exp_trace))
        check_trace_invariants(las, trace, 100)


@pytest.mark.parametrize("seed,grouped,width", [(21, False, 64), (57, True, 64), (21, False, 32)])
def test_symmetric_all_vs_all(host_lib, seed, grouped, width):
    """skip_self = 2 (the pile-up stage, daligner -s126 pile x pile): every unordered pair is seeded once;
    DH-2 then aligns the pair and, for the second record, the transposed pair through the same seed (trace
    on the other read's grid, both axes mirrored for complemented overlaps).  Records land in slots claimed
    at acceptance, so only their set is compared."""
    from helpers import la_rows
    g = sim.genome(seed, 20000)
    reads, _ = sim.reads(seed + 1, g, 30, 6000)
    if grouped:
        reads = sim.SeqDb(reads.bases, reads.off, group=np.arange(reads.n) % 2)
    o = oz.default_opts(algo=1, width=width, tspace=126, skip_self=2, min_len=500, max_la=64, max_cand=128)
    exp_las, exp_trace, stats = oz.align_db(reads, reads, o, nthreads=4, sort=False)
    las, trace, counters = run_host(host_lib, reads, reads, o)
    assert len(exp_las) > reads.n
    assert sorted(la_rows(las, trace)) == sorted(la_rows(exp_las, exp_trace))
    check_trace_invariants(las, trace, 126)
    assert int(counters[0]) == stats[3] and int(counters[1]) == stats[2]
    comp = las[(las["flags"] & 1) != 0]
    # (the record of the transposed pair is accepted on its own length and error: a pair can lose one of its two)
    assert len(comp) > 0 and (len(las) % 2 == 0 or width == 32)


def test_transposed_records_of_a_mapping(host_lib):
    """`damapper -C`: next to every record (contig, read) the record (read, contig) of the transposed pair -- A'' = the
    read on its forward strand, B'' = the contig (complemented for reverse-strand mappings), through the same seed,
    accepted on its own; trace on the read's grid.  Lane code (mode 1 of a plain launch) against oz_align_db2."""
    w = sim.Workload(300_000, 4, 400, 5000, seed=9, err=0.13, spacing=20000, gap_max=800)
    o = oz.default_opts(algo=1, width=64, k=16, kmer_mod=2, tspace=100)
    (exp, exp_t), (exp2, exp2_t) = oz.align_db_transposed(w.contigs, w.reads, o, nthreads=4)
    L = host_lib
    L.dh_tile_host_align2.restype = ctypes.c_long
    L.dh_tile_host_align2.argtypes = L.dh_tile_host_align.argtypes + [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_long)]
    A, B = w.contigs, w.reads
    cand, ncand = oz.seed_candidates_all(A, B, o)
    dc = np.zeros(cand.shape, dtype=DHCAND)
    for f in ("score", "aseq", "apos", "bpos"):
        dc[f] = cand[f]
    dc = np.ascontiguousarray(dc)
    maxlen = int(max((A.off[1:] - A.off[:-1]).max(), (B.off[1:] - B.off[:-1]).max()))
    nbmax = maxlen // o.tspace + 3
    cap = int(2 * B.n * o.max_la * 2 * (2 * nbmax + 2))
    las, las2 = (np.zeros(2 * B.n * o.max_la, dtype=oz.LA_DTYPE) for _ in range(2))
    trace, trace2 = (np.zeros(cap, dtype=np.uint16) for _ in range(2))
    counters = np.zeros(2, dtype=np.uint64)
    n2 = ctypes.c_long(0)
    n = L.dh_tile_host_align2(A.bases.ctypes.data, A.off.ctypes.data, A.n, B.bases.ctypes.data, B.off.ctypes.data, B.n,
                              ctypes.byref(o), dc.ctypes.data, ncand.ctypes.data, nbmax, las.ctypes.data, trace.ctypes.data,
                              cap, counters.ctypes.data, las2.ctypes.data, trace2.ctypes.data, ctypes.byref(n2))
    assert n >= 0 and len(exp) >= 350 and len(exp2) >= 0.95 * len(exp)
    from helpers import la_rows
    assert sorted(la_rows(las[:n], trace)) == sorted(la_rows(exp, exp_t))
    assert sorted(la_rows(las2[:n2.value], trace2)) == sorted(la_rows(exp2, exp2_t))
    check_trace_invariants(las2[:n2.value], trace2, o.tspace)
    t2 = las2[:n2.value]
    assert np.all(t2["aread"] < w.reads.n) and np.all(t2["bread"] < w.contigs.n) and ((t2["flags"] & 1) != 0).any()
    # the two records of a pair describe the same overlap: equal spans up to the slack of independent band paths
    key = {(int(r["bread"]), int(r["aread"]), int(r["flags"]) & 1): r for r in las[:n]}
    close = 0
    for r in t2:
        f = key.get((int(r["aread"]), int(r["bread"]), int(r["flags"]) & 1))
        if f is not None and abs((r["aepos"] - r["abpos"]) - (f["bepos"] - f["bbpos"])) <= 64:
            close += 1
    assert close >= 0.9 * len(t2)
