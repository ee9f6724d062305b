"""The radix-partitioned k-mer join that seeds a mapping pass (csrc/dh_mjoin.h; the damapper role,
dazzler.d:6158-6170) against the CPU oracle and against the directory lookups it replaces -- bit exact: the same
hits per read, hence the same candidates, alignments and trace values.

The join takes chunks of at least 64 Mbp by default; DH_MJOIN_MIN=0 sends the small inputs of these tests through it
and dh_get_mjoin_counts tells that it ran."""
import os

import numpy as np
import pytest

import dentist_amd
from dentist_amd import sim
from helpers import assert_same_las, check_trace_invariants
from oracle import pyoracle as oz
from test_parity_map_gpu import both_opts

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _join_for_small_inputs(monkeypatch):
    monkeypatch.setenv("DH_MJOIN_MIN", "0")


def run_both(ctx, A, B, expect_join=True, **kw):
    g, o = both_opts(**kw)
    exp = oz.align_db(A, B, o, nthreads=os.cpu_count() or 1)
    dA, dB = ctx.db(A), ctx.db(B)
    ctx.mjoin_counts(reset=True)
    got = ctx.align_db(dA, dB, g)
    st = ctx.align_stats()
    chunks, fallbacks = ctx.mjoin_counts()
    assert (chunks > 0) == expect_join, (chunks, fallbacks)
    assert (st.hits, st.cands, st.alignments, st.wave_cells) == tuple(int(x) for x in exp[2])
    assert_same_las(got, exp[:2])
    check_trace_invariants(got[0], got[1], g.tspace)
    return got, (chunks, fallbacks)


@pytest.mark.parametrize("k,mod,algo", [(20, 8, 1), (20, 1, 1), (20, 4, 0), (14, 1, 1), (14, 2, 0), (17, 3, 1), (22, 4, 1), (12, 1, 1)])
def test_mapping_through_the_partitioned_join(gpu_ctx, k, mod, algo):
    w = sim.Workload(400_000, 4, 700, 6000, seed=11 + k, spacing=20000)
    (las, _), (chunks, fallbacks) = run_both(gpu_ctx, w.contigs, w.reads, k=k, kmer_mod=mod, algo=algo,
                                           width=64 if algo == 1 else 30)
    assert fallbacks == 0 and len(set(las["bread"].tolist())) >= 0.98 * w.reads.n


def test_the_tiers_of_the_seed_back_end(monkeypatch):
    """A mapping chunk's first tier goes by the mean hits per read: here the wavefront-per-read variant (k_seed<512, JOIN, 64
    threads, 32 candidate band pairs>; 300 hits per read on average), and the reads it cannot hold -- read lengths are
    log-normal, the long ones bring 600-1 000 hits -- go through the 2 048-entry tier from a list.  Bit-exact against the
    oracle, and identical to the same call without the tier (DH_SEED_NO_WAVE_TIER=1) and by the directory (DH_NO_MJOIN=1).
    A context of its own: the tier is switched off per context when a quarter of a chunk overflows it."""
    g = sim.genome(5, 600_000)
    contigs = sim.SeqDb.from_list([g[:290_000], g[300_000:]])
    rd, _ = sim.reads(6, g, 900, 4000, 2500, min_len=600)
    lens = rd.off[1:] - rd.off[:-1]
    assert (lens > 9000).sum() >= 20 and np.median(lens) < 4500
    ctx = dentist_amd.Context(0)
    try:
        kw = dict(k=20, kmer_mod=1, algo=1, width=64, xdrop=60)
        (las, trace), (chunks, fallbacks) = run_both(ctx, contigs, rd, **kw)
        assert chunks > 0 and fallbacks == 0
        st = ctx.align_stats()
        assert st.hits / rd.n > 150 and st.hits / rd.n < 340          # (the mean that selects the tier)
        go = dentist_amd.default_align_opts(**kw)
        A, B = ctx.db(contigs), ctx.db(rd)
        for env in ("DH_SEED_NO_WAVE_TIER", "DH_NO_MJOIN"):
            monkeypatch.setenv(env, "1")
            other = ctx.align_db(A, B, go)
            monkeypatch.delenv(env)
            assert_same_las((las, trace), other)
    finally:
        ctx.close()


def test_the_round_6_paths_against_their_switches(monkeypatch):
    """Round 6: the filter of the join maps the entries of 64 tiles flat onto the lanes (k_mj_filter2; DH_MJ_DBG=2 = the filter
    of round 5, 16 lanes per segment), the seed sort gives buckets above 256 hits a second counting pass over their own
    diagonal range (DH_SEED_NO_REFINE=1 = the bitonic network as before) and a k_tile wavefront takes 64 work units per queue
    atomic (DH_TILE_QBATCH=1 = one atomic per pass).  Unsampled reads of 12 kb on a 2 Mb assembly: ~700 true hits per read on
    a few hundred neighbouring diagonals next to chance hits anywhere -- one slice of the read's diagonal range holds them
    all, the case the second pass is for; reads spanning two contigs bring two such clusters.  Bit-exact against the oracle,
    identical under every switch."""
    w = sim.Workload(2_000_000, 6, 500, 12000, seed=77, spacing=300_000)
    ctx = dentist_amd.Context(0)
    try:
        kw = dict(k=20, kmer_mod=1, algo=1, width=64, xdrop=60)
        (las, trace), (chunks, fallbacks) = run_both(ctx, w.contigs, w.reads, **kw)
        assert chunks > 0 and fallbacks == 0
        assert ctx.align_stats().hits / w.reads.n > 400
        go = dentist_amd.default_align_opts(**kw)
        A, B = ctx.db(w.contigs), ctx.db(w.reads)
        for env, val in (("DH_MJ_DBG", "2"), ("DH_SEED_NO_REFINE", "1"), ("DH_TILE_QBATCH", "1"), ("DH_NO_MJOIN", "1")):
            monkeypatch.setenv(env, val)
            other = ctx.align_db(A, B, go)
            monkeypatch.delenv(env)
            assert_same_las((las, trace), other)
        # the presence bitmap of the index, made per partition in LDS (k_mj_bitmap_part) or by scattered global atomics as in
        # round 5: it is made with the index of a DB, so a fresh DB object
        monkeypatch.setenv("DH_MJ_BITMAP_ATOMICS", "1")
        other = ctx.align_db(ctx.db(w.contigs), B, go)
        monkeypatch.delenv("DH_MJ_BITMAP_ATOMICS")
        assert_same_las((las, trace), other)
    finally:
        ctx.close()


def test_release_scratch_between_calls():
    """dh_ctx_release_scratch hands the context's grow-only device scratch back (a long-lived host between workloads; the
    partitioned join of an unsampled mapping of configs[2] keeps 100 GB): the next call allocates again and gives the same bits."""
    w = sim.Workload(300_000, 3, 600, 6000, seed=91, spacing=20000)
    ctx = dentist_amd.Context(0)
    try:
        g = dentist_amd.default_align_opts(k=20, kmer_mod=4, algo=1, width=64)
        A, B = ctx.db(w.contigs), ctx.db(w.reads)
        a = ctx.align_db(A, B, g)
        ctx.release_scratch()
        ctx.release_scratch()          # (idempotent)
        b = ctx.align_db(A, B, g)
        assert_same_las(a, b)
        po = dentist_amd.default_process_opts(algo=1, rounds=1)
        piles = dentist_amd.Pileups(a[0], w.contigs.off, po)
        r0, b0 = dentist_amd.process_pileups(ctx, A, B, a[0], a[1], piles, po)
        ctx.release_scratch()
        r1, b1 = dentist_amd.process_pileups(ctx, A, B, a[0], a[1], piles, po)
        assert r0.tobytes() == r1.tobytes() and np.array_equal(b0, b1) and (r0["status"] == 0).sum() >= 1
    finally:
        ctx.close()


def test_equal_to_the_directory_lookups_on_every_field(gpu_ctx, monkeypatch):
    """The same call with DH_NO_MJOIN=1 (one random directory line per k-mer): identical records, trace and counters."""
    w = sim.Workload(600_000, 5, 1500, 9000, seed=71, spacing=20000)
    g = dentist_amd.default_align_opts(k=20, kmer_mod=8, algo=1, width=64, xdrop=60)
    A, B = gpu_ctx.db(w.contigs), gpu_ctx.db(w.reads)
    gpu_ctx.mjoin_counts(reset=True)
    a = gpu_ctx.align_db(A, B, g)
    sa = gpu_ctx.align_stats()
    assert gpu_ctx.mjoin_counts() == (1, 0)
    monkeypatch.setenv("DH_NO_MJOIN", "1")
    B.drop_cache()
    A.drop_cache()
    b = gpu_ctx.align_db(A, B, g)
    sb = gpu_ctx.align_stats()
    assert gpu_ctx.mjoin_counts() == (1, 0)
    assert_same_las(a, b)
    assert (sa.hits, sa.cands, sa.alignments, sa.wave_cells) == (sb.hits, sb.cands, sb.alignments, sb.wave_cells)


def test_ragged_reads_short_empty_and_tile_straddling(gpu_ctx, monkeypatch):
    """Reads shorter than k, empty reads, reads of a few bases between long ones, a long read over many tiles, read
    starts at every offset relative to the 8 192-base tiles (kmer_mod 1): no k-mer may cross a read boundary."""
    rng = np.random.default_rng(3)
    w = sim.Workload(300_000, 3, 300, 7000, seed=91, spacing=15000)
    seqs = []
    for i in range(w.reads.n):
        seqs.append(w.reads.seq(i))
        if i % 7 == 0:
            seqs.append(np.zeros(0, dtype=np.uint8))
        if i % 5 == 0:
            seqs.append(rng.integers(0, 4, int(rng.integers(1, 40))).astype(np.uint8))
        if i % 11 == 0:
            seqs.append(w.reads.seq(i)[:int(rng.integers(19, 23))])     # k - 1 .. k + 2 bases of a real read
    truth = w.truth
    seqs.append(truth[1000:61000].copy())                               # 60 kb: several tiles and tile groups
    reads = sim.SeqDb.from_list(seqs)
    (las, _), (chunks, fallbacks) = run_both(gpu_ctx, w.contigs, reads, k=20, kmer_mod=1, algo=1, width=64)
    assert fallbacks == 0
    monkeypatch.setenv("DH_ALIGN_CHUNK", "200")      # several chunks: chunk ends in the middle of the DB
    run_both(gpu_ctx, w.contigs, reads, k=20, kmer_mod=2, algo=1, width=64)


def test_soft_masked_reads_and_repeats_with_the_t_cap(gpu_ctx):
    """-m tracks on the reads (k-mers touching a masked base are not looked up) and a contig set with a 5-copy repeat:
    multi-entry buckets, the -t cap per orientation class, several hits per looked-up k-mer."""
    rng = np.random.default_rng(17)
    unit = rng.integers(0, 4, 3000).astype(np.uint8)
    parts = []
    for c in range(6):
        parts.append(rng.integers(0, 4, 40000).astype(np.uint8))
        if c < 5:
            parts.append(unit if c % 2 == 0 else sim.revcomp(unit))
    genome = np.concatenate(parts)
    contigs = sim.SeqDb.from_list([genome[:110000], genome[112000:]])
    reads, _ = sim.reads(99, genome, 400, 6000, 0, min_len=6000)
    for tcap in (3, 8):
        run_both(gpu_ctx, contigs, reads, k=14, kmer_mod=1, algo=1, width=64, tcap=tcap)
    ptr, iv = [0], []
    for i in range(reads.n):
        n, pos = reads.length(i), 0
        while True:
            pos += int(rng.integers(300, 2500))
            ln = int(rng.integers(10, 400))
            if pos + ln >= n:
                break
            iv += [pos, pos + ln]
            pos += ln
        ptr.append(len(iv) // 2)
    reads.mask = (np.asarray(ptr, dtype=np.int64), np.asarray(iv + [0, 0], dtype=np.int32))
    run_both(gpu_ctx, contigs, reads, k=20, kmer_mod=2, algo=1, width=64)
    masked_hits = gpu_ctx.align_stats().hits
    reads.mask = None
    run_both(gpu_ctx, contigs, reads, k=20, kmer_mod=2, algo=1, width=64)
    assert masked_hits < 0.95 * gpu_ctx.align_stats().hits


def test_reads_with_low_complexity_tails_fill_one_partition(gpu_ctx, monkeypatch):
    """Every read carries a 110-base poly-A tail: ~95 copies of one k-mer per read, all in ONE partition of the join -- the 64
    tiles a wavefront of k_mj_filter2 takes hold several thousand entries of that partition instead of ~512, so the flat
    lane mapping walks several windows of its segment marks (MJ_F2_MARKS = 1 024 flat indices each) and the segments are
    dozens of entries long.  The assembly holds poly-A stretches too (hits, the -t cap).  Bit-exact against the oracle, the
    chunk stays with the join (no fall-back), identical to the round-5 filter and to the directory lookups."""
    rng = np.random.default_rng(41)
    g = rng.integers(0, 4, 400_000).astype(np.uint8)
    for at in (50_000, 180_000, 310_000):
        g[at:at + 60] = 0
    contigs = sim.SeqDb.from_list([g[:195_000], g[200_000:]])
    rd, _ = sim.reads(42, g, 700, 4000, 0, min_len=4000)
    tail = np.zeros(110, dtype=np.uint8)
    reads = sim.SeqDb.from_list([np.concatenate([rd.seq(i), tail]) for i in range(rd.n)])
    kw = dict(k=20, kmer_mod=1, algo=1, width=64)
    (las, trace), (chunks, fallbacks) = run_both(gpu_ctx, contigs, reads, **kw)
    assert chunks > 0 and fallbacks == 0
    go = dentist_amd.default_align_opts(**kw)
    A, B = gpu_ctx.db(contigs), gpu_ctx.db(reads)
    for env, val in (("DH_MJ_DBG", "2"), ("DH_NO_MJOIN", "1")):
        monkeypatch.setenv(env, val)
        other = gpu_ctx.align_db(A, B, go)
        monkeypatch.delenv(env)
        assert_same_las((las, trace), other)


def test_capacity_overflow_falls_back_to_the_directory(gpu_ctx, monkeypatch):
    """A hit pool of one page cannot hold the hits of the chunk: the chunk is redone by the directory lookups, the result
    is the same and the fall-back is counted."""
    monkeypatch.setenv("DH_MJOIN_PAGES", "1")
    w = sim.Workload(300_000, 3, 600, 6000, seed=23, spacing=15000)
    _, (chunks, fallbacks) = run_both(gpu_ctx, w.contigs, w.reads, expect_join=False, k=20, kmer_mod=1, algo=1, width=64)
    assert fallbacks >= 1


def test_map_reads_with_filters_through_the_join(gpu_ctx, monkeypatch):
    """dh_map_reads (mapping + chain flags + the six collect filters per chunk) with the join underneath == the same call on
    the directory path."""
    w = sim.Workload(500_000, 5, 1500, 8000, seed=83, spacing=20000)
    mo = dentist_amd.default_align_opts(kmer_mod=8, k=20, width=64, xdrop=60, algo=1)
    po = dentist_amd.default_process_opts(algo=1)
    A, B = gpu_ctx.db(w.contigs), gpu_ctx.db(w.reads)
    gpu_ctx.mjoin_counts(reset=True)
    a = gpu_ctx.map_reads(A, B, mo, po, sorted=False, candidates=False)
    assert gpu_ctx.mjoin_counts()[0] >= 1
    monkeypatch.setenv("DH_NO_MJOIN", "1")
    A.drop_cache()
    B.drop_cache()
    b = gpu_ctx.map_reads(A, B, mo, po, sorted=False, candidates=False)
    assert_same_las(a[:2], b[:2])
    assert [int(x) for x in a[2]] == [int(x) for x in b[2]]
