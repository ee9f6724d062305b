"""DH-2 (dh_align_opts.algo = 1: tile-by-tile banded bit-parallel extension, one alignment per lane,
k_tile) against the oracle's plain-DP restatement (oracle/align.c: extend_tiled).  Bit-exact: every
record field, every trace value, the hit / candidate / alignment / cell counters.  The reference's
call sites for this arithmetic: damapper (dazzler.d:6158-6170, Snakefile:1143-1170), daligner -A
(processPileUps/package.d:655-667)."""
import os

import numpy as np
import pytest

import dentist_amd
from dentist_amd import sim
from helpers import assert_same_las, check_trace_invariants, tandem_reads
from oracle import pyoracle as oz
from test_parity_map_gpu import run_both

pytestmark = pytest.mark.gpu

T = dict(algo=1, width=64)


def test_mapping_reads_to_contigs(gpu_ctx):
    w = sim.Workload(400_000, 4, 800, 5000, seed=7, spacing=20000)
    las, _ = run_both(gpu_ctx, w.contigs, w.reads, **T)
    assert len(set(las["bread"].tolist())) == w.reads.n


@pytest.mark.parametrize("seed,rl,err,ts", [(3, 2500, 0.13, 100), (5, 9000, 0.13, 126), (9, 6000, 0.05, 100),
                                            (13, 4000, 0.20, 100), (17, 3000, 0.13, 64), (19, 3000, 0.13, 128)])
def test_lengths_error_rates_and_trace_spacings(gpu_ctx, seed, rl, err, ts):
    w = sim.Workload(250_000, 3, 250, rl, seed=seed, err=err, spacing=15000)
    run_both(gpu_ctx, w.contigs, w.reads, tspace=ts, **T)


@pytest.mark.parametrize("mod", [8, 4])
def test_bench_options_of_the_mapping_pass(gpu_ctx, mod):
    """k = 20, modimer sampling 1 / 8 (bench.py's default) and 1 / 4 (its --kmer-mod 4 line), x-drop 60: the options
    bench.py maps configs[2] with, HIP against the oracle bit for bit and against the truth placement."""
    w = sim.Workload(1_000_000, 8, 3000, 10_000, seed=23)
    las, _ = run_both(gpu_ctx, w.contigs, w.reads, k=20, kmer_mod=mod, xdrop=60, **T)
    s, e = w.read_truth[las["bread"], 0], w.read_truth[las["bread"], 1]
    cs = w.contig_start[las["aread"]]
    ok = ((las["flags"] & 1) == w.read_truth[las["bread"], 2]) & (cs + las["abpos"] >= s - 80) & (cs + las["aepos"] <= e + 80)
    assert ok.mean() > 0.999 and len(set(las["bread"].tolist())) == w.reads.n


def test_short_and_ragged_inputs(gpu_ctx):
    """Reads shorter than a tile / than the band, reads hanging over both contig ends, a read equal to
    its contig, tiny contigs, an empty read."""
    rng = np.random.default_rng(11)
    g = rng.integers(0, 4, 6000).astype(np.uint8)
    contigs = sim.SeqDb.from_list([g[:3000], g[3100:3160], g[3200:6000], g[100:140]])
    reads = sim.SeqDb.from_list([g[2900:3000], g[2950:3160], g[0:3000], g[3150:3300], g[10:70], sim.revcomp(g[3300:5900]),
                                 g[3100:3160], g[2990:3110], g[20:52], g[0:0], g[5:12]])
    las, _ = run_both(gpu_ctx, contigs, reads, k=12, hmin=20, min_len=20, **T)
    assert len(las) >= 8


@pytest.mark.parametrize("kw", [dict(strands=1), dict(strands=2), dict(pen=4, xdrop=60), dict(max_cand=2, max_la=1),
                                dict(min_len=2000), dict(max_err_ppm=200000), dict(k=12, tcap=8)])
def test_option_sweep(gpu_ctx, kw):
    w = sim.Workload(150_000, 2, 250, 3000, seed=61, spacing=15000)
    run_both(gpu_ctx, w.contigs, w.reads, **kw, **T)


def test_ont_like_error_profile(gpu_ctx):
    g = sim.genome(97, 600_000)
    gb, ge = sim.gaps(98, len(g), 3, 50, 3000, 20000)
    contigs, _ = sim.contigs_from_gaps(g, gb, ge)
    reads, truth = sim.reads(99, g, 240, 20000, 0, err=0.10, p_ins=0.30, p_del=0.40)
    las, _ = run_both(gpu_ctx, contigs, reads, k=20, kmer_mod=4, **T)
    assert len(set(las["bread"].tolist())) == reads.n


def test_blocks_and_chunks_give_the_same_bits(gpu_ctx, monkeypatch):
    """One call, a loop over read blocks (`damapper <ref> <reads>.<block>`, Snakefile:1143-1170) and a
    small internal chunk size: the same records."""
    w = sim.Workload(600_000, 5, 2400, 6000, seed=29)
    A, B = gpu_ctx.db(w.contigs), gpu_ctx.db(w.reads)
    g = dentist_amd.default_align_opts(k=20, kmer_mod=4, **T)
    las, trace = gpu_ctx.align_db(A, B, g, select_best=True)
    handles = [gpu_ctx.align_db_block(A, B, f, 800, g, select_best=True, raw=True) for f in (0, 800, 1600)]
    las2, trace2 = dentist_amd.merge_las(handles)
    assert_same_las((las2, trace2), (las, trace))
    monkeypatch.setenv("DH_ALIGN_CHUNK", "700")
    las3, trace3 = gpu_ctx.align_db(A, B, g, select_best=True)
    assert_same_las((las3, trace3), (las, trace))
    monkeypatch.setenv("DH_TILE_WAVES_PER_CU", "1")
    las4, trace4 = gpu_ctx.align_db(A, B, g, select_best=True)
    assert_same_las((las4, trace4), (las, trace))


@pytest.mark.parametrize("seed,grouped", [(21, False), (57, True), (33, True)])
def test_symmetric_all_vs_all(gpu_ctx, seed, grouped):
    """skip_self = 2 (daligner -s126 pile x pile, processPileUps/package.d:478-482): every unordered pair
    seeded once, DH-2 aligns the pair and the transposed pair through the same seed; both records."""
    g = sim.genome(seed, 20000)
    reads, _ = sim.reads(seed + 1, g, 30, 6000)
    if grouped:
        reads = sim.SeqDb(reads.bases, reads.off, group=np.arange(reads.n) % 3)
    las, trace = run_both(gpu_ctx, reads, reads, same=True, tspace=126, skip_self=2, min_len=500, max_la=64, max_cand=128, **T)
    assert len(las) > reads.n and not np.any(las["aread"] == las["bread"])
    key = set(zip(las["aread"].tolist(), las["bread"].tolist(), (las["flags"] & 1).tolist()))
    assert all((b, a, c) in key for a, b, c in key)


@pytest.mark.parametrize("qbatch", [None, "1", "7"])
def test_symmetric_items_with_more_candidates_than_the_attempt_cap(gpu_ctx, monkeypatch, qbatch):
    """An item (read, strand) attempts at most 64 alignments (oracle/align.c: `nd < 64`; dh_tile.h: MAXREG).  In a pile-up of
    166 reads -- the reference's behaviour, no read cap: processPileUps/package.d:283-374 -- an item meets more partners than
    that, and k_units_fat splits it into work units without changing which candidates are aligned: lone candidates below
    index 64 on their own, the rest of the item in order with the lone ones counted.  180 reads of one locus, all on one
    strand: ~90 candidates per forward item, some pairs with two candidate band pairs (the `rest` unit).  Also with one work
    unit per queue atomic and with batches of seven (DH_TILE_QBATCH: a wavefront takes 64 per atomic by default)."""
    if qbatch:
        monkeypatch.setenv("DH_TILE_QBATCH", qbatch)
    g = sim.genome(91, 2600)
    reads, truth = sim.reads(92, g, 180, 2000, min_len=1500)
    seqs = [reads.seq(i) if truth[i, 2] == 0 else sim.revcomp(reads.seq(i)) for i in range(reads.n)]
    reads = sim.SeqDb.from_list(seqs)
    las, trace = run_both(gpu_ctx, reads, reads, same=True, tspace=126, skip_self=2, min_len=500, max_la=256, max_cand=256, **T)
    st = gpu_ctx.align_stats()
    assert st.cands > 64 * reads.n // 2 and st.alignments < st.cands   # items above the cap exist, and the cap dropped candidates
    per_item = np.bincount(las["bread"][(las["flags"] & 1) == 0], minlength=reads.n)
    assert per_item.max() <= 2 * 64 + 64


def test_tandem_self_alignments(gpu_ctx):
    """skip_self = 3 (`datander <block>`, DAMASKER; DENTIST's call commandline.d:2866-2876, Snakefile:1056-1076): every read
    against itself, below the main diagonal -- seeds with A position > B position in the same read (k_seed), cells in which
    B's base does not come before A's barred from matching (k_tile<TAN>: a row mask per tile).  Bit-exact against the
    oracle: records, traces, counters; the planted arrays are all found, also the one of period 24 (inside half a band of
    the main diagonal)."""
    db, truth = tandem_reads()
    kw = dict(skip_self=3, strands=1, k=12, band_shift=4, min_len=500, tspace=126, **T)
    las, trace = run_both(gpu_ctx, db, db, same=True, **kw)
    assert len(las) > 0 and (las["aread"] == las["bread"]).all() and (las["abpos"] > las["bbpos"]).all()
    assert set(las["aread"].tolist()) == {t[0] for t in truth}
    for t in truth:
        mine = las[las["aread"] == t[0]]
        assert (mine["aepos"] - mine["bbpos"]).max() >= 0.8 * (t[2] - t[1])
    # whole-DB call in chunks of two reads: the same bits
    d = gpu_ctx.db(db)
    g = dentist_amd.default_align_opts(**kw)
    for bad in (dict(algo=0, width=30), dict(strands=3)):
        with pytest.raises(dentist_amd.DhError):
            gpu_ctx.align_db(d, d, dentist_amd.default_align_opts(**{**kw, **bad}))
    with pytest.raises(dentist_amd.DhError):
        gpu_ctx.align_db(d, gpu_ctx.db(db), g)     # two DBs


@pytest.mark.parametrize("case", ["mapping", "ragged", "symmetric", "tandem", "noisy"])
def test_band_of_32_rows(gpu_ctx, case):
    """dh_align_opts.width = 32 with algo 1: the band of DH-2 on 32-bit vectors (k_tile<., 32>: half the instructions per
    column; the band re-centres at every tile boundary, so +- 16 diagonals of drift per tile are what it must hold).
    Bit-exact against the oracle's plain DP at W = 32 in every mode the band of 64 is tested in."""
    T32 = dict(algo=1, width=32)
    if case == "mapping":
        w = sim.Workload(1_000_000, 8, 3000, 10_000, seed=23)
        las, _ = run_both(gpu_ctx, w.contigs, w.reads, k=20, kmer_mod=8, xdrop=60, **T32)
        assert len(set(las["bread"].tolist())) >= 0.995 * w.reads.n
    elif case == "noisy":
        w = sim.Workload(250_000, 3, 250, 4000, seed=13, err=0.20, spacing=15000)
        run_both(gpu_ctx, w.contigs, w.reads, tspace=126, **T32)
    elif case == "ragged":
        rng = np.random.default_rng(11)
        g = rng.integers(0, 4, 6000).astype(np.uint8)
        contigs = sim.SeqDb.from_list([g[:3000], g[3100:3160], g[3200:6000], g[100:140]])
        reads = sim.SeqDb.from_list([g[2900:3000], g[2950:3160], g[0:3000], g[3150:3300], g[10:70], sim.revcomp(g[3300:5900]),
                                     g[3100:3160], g[2990:3110], g[20:52]])
        run_both(gpu_ctx, contigs, reads, k=12, hmin=20, min_len=20, **T32)
    elif case == "symmetric":
        g = sim.genome(21, 20000)
        reads, _ = sim.reads(22, g, 30, 6000)
        las, _ = run_both(gpu_ctx, reads, reads, same=True, tspace=126, skip_self=2, min_len=500, max_la=64, max_cand=128, **T32)
        assert len(las) > reads.n
    else:
        db, truth = tandem_reads()
        las, _ = run_both(gpu_ctx, db, db, same=True, skip_self=3, strands=1, k=12, band_shift=4, min_len=500, tspace=126, **T32)
        assert set(las["aread"].tolist()) == {t[0] for t in truth}


def test_rejections(gpu_ctx):
    """DH-2 has one band width, tiles of at most 128 columns, 2-bit sequences only."""
    rng = np.random.default_rng(5)
    g = rng.integers(0, 4, 4000).astype(np.uint8)
    db = gpu_ctx.db(sim.SeqDb.from_list([g, g[100:3000]]))
    for kw in (dict(width=30), dict(width=48), dict(tspace=200)):
        with pytest.raises(dentist_amd.DhError):
            gpu_ctx.align_db(db, db, dentist_amd.default_align_opts(**{**T, **kw}))
    gn = g.copy()
    gn[50] = 4
    dn = gpu_ctx.db(sim.SeqDb.from_list([gn]))
    with pytest.raises(dentist_amd.DhError):
        gpu_ctx.align_db(dn, db, dentist_amd.default_align_opts(**T))


def test_damapper_chains_of_reads_with_a_long_indel(gpu_ctx):
    """damapper's chain flags (dazzler.d:1728-1758, 1991-1998): a read that carries a 2-5 kb indel against its contig
    maps as two collinear local alignments -- one chain: START|BEST on the first, NEXT on the second; an alternate
    placement of a read is a chain of its own (START without BEST = alternateChain); -n (dh_set_near_best) disables
    alternates far below the best chain.  Product == oracle."""
    rng = np.random.default_rng(77)
    g = sim.genome(101, 400_000)
    contigs = sim.SeqDb.from_list([g[:200_000], g[200_000:]])
    reads, truth = sim.reads(102, g, 60, 12_000)
    seqs = [reads.seq(i) for i in range(reads.n)]
    planted = []
    for i in range(0, 24, 2):      # deletions in the read (contig bases missing) and insertions (foreign bases in the read)
        s = seqs[i]
        cut, ln = len(s) // 2, int(rng.integers(2000, 5000))
        if i % 4 == 0 and len(s) > cut + ln + 3000:
            seqs[i] = np.concatenate([s[:cut], s[cut + ln:]])
        else:
            seqs[i] = np.concatenate([s[:cut], rng.integers(0, 4, ln).astype(np.uint8), s[cut:]])
        planted.append(i)
    reads2 = sim.SeqDb.from_list(seqs)
    g_, o = __import__("test_parity_map_gpu").both_opts(k=20, kmer_mod=2, **T)
    A, B = gpu_ctx.db(contigs), gpu_ctx.db(reads2)
    for ppm in (0, 850000):
        lib = dentist_amd.lib()
        lib.dh_set_near_best(ppm)
        oz.lib().oz_set_near_best(ppm)
        try:
            las, trace = gpu_ctx.align_db(A, B, g_, select_best=True)
            exp = oz.align_db(contigs, reads2, o, nthreads=os.cpu_count() or 1, select_best=True)
        finally:
            lib.dh_set_near_best(0)
            oz.lib().oz_set_near_best(0)
        assert_same_las((las, trace), exp[:2])
        chained = 0
        for i in planted:
            mine = las[las["bread"] == i]
            st = mine[(mine["flags"] & 0x4) != 0]
            nx = mine[(mine["flags"] & 0x8) != 0]
            if truth[i][0] < 200_000 - 12_000 or truth[i][0] > 200_000:   # the read lies inside one contig
                if len(nx) >= 1:
                    chained += 1
                    assert np.all((nx["flags"] & 0x10) != 0) and np.any((st["flags"] & 0x10) != 0)
                    assert np.all((mine["flags"] & (0x4 | 0x8)) != (0x4 | 0x8))
        assert chained >= 6
        plain = las[~np.isin(las["bread"], planted)]
        assert np.all((plain["flags"] & 0x4) != 0) or np.any((plain["flags"] & 0x8) != 0)


@pytest.mark.parametrize("chunk", [None, "300"])
def test_transposed_records_of_a_mapping(gpu_ctx, monkeypatch, chunk):
    """dh_align_db_transposed (`damapper -C`): the mapping and, from the same pass, the records (read, contig) of the
    transposed pairs -- A'' = the read forward, B'' = the contig (complemented for reverse-strand mappings), through
    the same seed, accepted on its own, trace on the read's grid -- against oz_align_db2, bit-exact, also when the reads
    come in several chunks; the first set equals the plain call."""
    if chunk:
        monkeypatch.setenv("DH_ALIGN_CHUNK", chunk)
    w = sim.Workload(500_000, 5, 900, 6000, seed=29, spacing=20000, gap_max=1200)
    o = dentist_amd.default_align_opts(k=16, kmer_mod=2, **T)
    A, B = gpu_ctx.db(w.contigs), gpu_ctx.db(w.reads)
    (las, trace), (las2, trace2) = gpu_ctx.align_db_transposed(A, B, o)
    plain = gpu_ctx.align_db(A, B, o)
    assert_same_las((las, trace), plain)
    oo = oz.default_opts(k=16, kmer_mod=2, **T)
    (exp, exp_t), (exp2, exp2_t) = oz.align_db_transposed(w.contigs, w.reads, oo, nthreads=8)
    assert len(exp) > 800 and len(exp2) > 0.95 * len(exp) and ((exp2["flags"] & 1) != 0).any()
    assert_same_las((las, trace), (exp, exp_t))
    assert_same_las((las2, trace2), (exp2, exp2_t))
    check_trace_invariants(las2, trace2, o.tspace)
    # chain flags on both sets; every transposed record belongs to a chain of its read
    (_, _), (best2, _) = gpu_ctx.align_db_transposed(A, B, o, select_best=True)
    assert len(best2) == len(las2) and np.all((best2["flags"] & (0x4 | 0x8)) != 0) and ((best2["flags"] & 0x10) != 0).any()
    # refused where it is not defined
    with pytest.raises(dentist_amd.DhError):
        gpu_ctx.align_db_transposed(A, B, dentist_amd.default_align_opts(k=16, kmer_mod=2))   # DH-1
