"""Alignment-coverage mask of `dentist mask-repetitive-regions`: the reference's unittest
vectors against the oracle (CPU) and against the product's device path (dh_db_mask_coverage), plus
product == oracle on seeded random alignments."""
import numpy as np
import pytest

import dentist_amd
from dentist_amd import sim
from oracle import maskcov as mc

# maskRepetitiveRegions.d:299-333, 336-343
ALIGNMENTS = [(1, 5, 18), (1, 5, 18), (1, 5, 20), (1, 10, 20), (1, 10, 30), (1, 10, 30), (1, 13, 30), (1, 20, 30),
              (1, 20, 30), (1, 20, 30), (1, 24, 30), (2, 0, 3), (2, 0, 3), (2, 0, 5), (2, 0, 5), (2, 0, 15), (2, 0, 15),
              (2, 0, 15), (2, 5, 15), (2, 5, 15), (2, 5, 15), (2, 9, 15), (3, 1, 4), (3, 2, 5), (3, 3, 6), (3, 4, 7),
              (3, 5, 8), (3, 6, 9), (3, 7, 10), (3, 8, 11), (3, 9, 12), (3, 10, 13), (3, 11, 14)]
CONTIGS = [(1, 0, 30), (2, 0, 15), (3, 0, 15)]
# :394-410
MASK_3_5 = [(1, 0, 5), (1, 10, 18), (1, 20, 30), (2, 0, 3), (2, 5, 15), (3, 0, 3), (3, 12, 15)]
# :603-631
CHANGES = [(1, 0, 0, 0), (1, 5, 0, 3), (1, 10, 3, 6), (1, 13, 6, 7), (1, 18, 7, 5), (1, 20, 5, 6), (1, 24, 6, 7),
           (1, 30, 7, 0), (2, 0, 0, 7), (2, 3, 7, 5), (2, 5, 5, 6), (2, 9, 6, 7), (2, 15, 7, 0), (3, 0, 0, 0), (3, 1, 0, 1),
           (3, 2, 1, 2), (3, 3, 2, 3), (3, 4, 3, 3), (3, 5, 3, 3), (3, 6, 3, 3), (3, 7, 3, 3), (3, 8, 3, 3), (3, 9, 3, 3),
           (3, 10, 3, 3), (3, 11, 3, 3), (3, 12, 3, 2), (3, 13, 2, 1), (3, 14, 1, 0), (3, 15, 0, 0)]


def test_oracle_coverage_changes_vector():
    assert mc.coverage_changes(ALIGNMENTS, CONTIGS) == CHANGES


def test_oracle_assessor_vector():
    assert mc.bad_coverage_mask(ALIGNMENTS, CONTIGS, 3, 5) == MASK_3_5
    assert mc.bad_coverage_mask([], CONTIGS, 3, 5) == []


def test_coverage_bounds_from_read_coverage():
    for x in (10.0, 25.0, 50.0, 87.5, 200.0):
        assert dentist_amd.max_coverage_reads(x) == mc.max_coverage_reads(x)
        assert dentist_amd.max_improper_coverage_reads(x) == mc.max_improper_coverage_reads(x)
    assert mc.max_improper_coverage_reads(50.0) == 25 and mc.max_improper_coverage_reads(4.0) >= 4


def _las_of(intervals, rlen=None):
    las = np.zeros(len(intervals), dtype=dentist_amd.LA_DTYPE)
    for i, (c, b, e) in enumerate(intervals):
        las[i]["aread"], las[i]["abpos"], las[i]["aepos"], las[i]["bread"] = c - 1, b, e, i
        las[i]["bbpos"], las[i]["bepos"] = 0, e - b
    return las


def _product_mask(ctx, lens, las, lower, upper, **kw):
    db = ctx.db(sim.SeqDb.from_list([np.zeros(n, dtype=np.uint8) for n in lens]))
    db.mask_coverage(las, lower, upper, **kw)
    ptr, iv = db.get_mask()
    iv = np.asarray(iv).reshape(-1, 2)
    return [(c + 1, int(iv[j][0]), int(iv[j][1])) for c in range(len(lens)) for j in range(ptr[c], ptr[c + 1])]


@pytest.mark.gpu
def test_product_assessor_vector(gpu_ctx):
    assert _product_mask(gpu_ctx, [30, 15, 15], _las_of(ALIGNMENTS), 3, 5) == MASK_3_5
    assert _product_mask(gpu_ctx, [30, 15, 15], _las_of([]), 3, 5) == []


@pytest.mark.gpu
@pytest.mark.parametrize("seed,lower,upper", [(1, 0, 8), (2, 2, 6), (3, 0, 3), (4, 5, 40)])
def test_product_equals_oracle_on_random_alignments(gpu_ctx, seed, lower, upper):
    rng = np.random.default_rng(seed)
    lens = rng.integers(50, 6000, 37).tolist()
    iv = []
    for _ in range(4000):
        c = int(rng.integers(0, len(lens)))
        b = int(rng.integers(0, lens[c]))
        e = int(min(lens[c], b + rng.integers(1, 1500)))
        iv.append((c + 1, b, e))
    contigs = [(c + 1, 0, n) for c, n in enumerate(lens)]
    assert _product_mask(gpu_ctx, lens, _las_of(iv), lower, upper) == mc.bad_coverage_mask(iv, contigs, lower, upper)


@pytest.mark.gpu
def test_improper_only_counts_improper_alignments(gpu_ctx):
    """The second assessor of the reads case (maskRepetitiveRegions.d:157-176): only alignments that are
    not proper within the allowance (base.d:537-557) enter the coverage."""
    lens, rlen = [1000], 400
    # (abpos, aepos, bbpos, bepos): proper ones reach a read end or a contig end on both sides
    rows = [(0, 300, 100, 400), (100, 500, 0, 400), (200, 350, 50, 200), (220, 380, 100, 260), (240, 360, 150, 270),
            (700, 1000, 0, 300)]
    las = np.zeros(len(rows), dtype=dentist_amd.LA_DTYPE)
    for i, (ab, ae, bb, be) in enumerate(rows):
        las[i]["aread"], las[i]["bread"] = 0, i
        las[i]["abpos"], las[i]["aepos"], las[i]["bbpos"], las[i]["bepos"] = ab, ae, bb, be
    ro = np.arange(len(rows) + 1, dtype=np.int64) * rlen
    got = _product_mask(gpu_ctx, lens, las, 0, 1, read_off=ro, improper_only=True, allowance=0)
    improper = [(1, ab, ae) for ab, ae, bb, be in rows if not ((ab <= 0 or bb <= 0) and (ae >= 1000 or be >= rlen))]
    assert improper == [(1, 200, 350), (1, 220, 380), (1, 240, 360)]
    assert got == mc.bad_coverage_mask(improper, [(1, 0, 1000)], 0, 1) == [(1, 220, 360)]


@pytest.mark.gpu
def test_propagate_mask_matches_the_oracle_on_a_mapping(gpu_ctx):
    """`dentist propagate-mask` (propagateMask.d:136-305): a contig mask carried to the reads through
    the trace points of a real mapping; product == oracle, and against the truth of the simulation every
    propagated interval covers the read bases that came from the masked contig bases (within a trace
    tile on either side)."""
    w = sim.Workload(400_000, 4, 3000, 6000, seed=5)
    A, B = gpu_ctx.db(w.contigs), gpu_ctx.db(w.reads)
    las, trace = gpu_ctx.align_db(A, B, dentist_amd.default_align_opts(kmer_mod=4, k=20), select_best=True)
    rng = np.random.default_rng(9)
    ptr, iv = [0], []
    for c in range(w.contigs.n):
        n = int(w.contigs.off[c + 1] - w.contigs.off[c])
        cuts = np.sort(rng.choice(np.arange(1, n), size=24, replace=False))
        for b, e in cuts.reshape(-1, 2):
            iv.append((int(b), int(min(e, b + 900))))
        ptr.append(len(iv))
    ptr, iv = np.array(ptr, dtype=np.int64), np.array(iv, dtype=np.int32)
    rlen = np.diff(w.reads.off)
    optr, oiv = dentist_amd.propagate_mask(las, trace, 100, (ptr, iv), w.contigs.n, w.reads.off)
    exp = mc.propagate_mask(las, trace, 100, ptr, iv, rlen)
    got = {r: [tuple(x) for x in oiv[optr[r]:optr[r + 1]].tolist()] for r in range(w.reads.n) if optr[r + 1] > optr[r]}
    assert got == exp and len(got) > 100
    # truth: read r covers genome [s, e) on strand st; forward-read position of genome position x is x - s
    # (st = 0) or e - x (st = 1), up to the indels of the read
    checked = 0
    for la in las[:: max(1, len(las) // 400)]:
        r, c = int(la["bread"]), int(la["aread"])
        s, e, st = (int(x) for x in w.read_truth[r][:3])
        for b, en in iv[ptr[c]:ptr[c + 1]]:
            gb, ge = max(int(w.contig_start[c]) + max(b, la["abpos"]), s), min(int(w.contig_start[c]) + min(en, la["aepos"]), e)
            if ge - gb < 50:
                continue
            mid = (gb + ge) // 2
            pos = mid - s if st == 0 else e - mid
            pos = pos * rlen[r] / max(e - s, 1)
            assert any(x0 - 150 <= pos <= x1 + 150 for x0, x1 in got.get(r, [])), (r, c, b, en)
            checked += 1
    assert checked > 50


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_validate_regions_matches_the_literal_restatement(seed):
    """`dentist validate-regions` (validateRegions.d:325-512): the prefix-sum formulation of the product
    against the literal window loop of the oracle on random alignments around random regions."""
    rng = np.random.default_rng(seed)
    lens = [9000, 400, 15000, 2500]
    co = np.concatenate([[0], np.cumsum(lens)])
    rows = []
    for c, n in enumerate(lens):
        for _ in range(int(rng.integers(0, 120))):
            b = int(rng.integers(0, n))
            rows.append((c, b, int(min(n, b + rng.integers(1, 6000)))))
    rows.sort(key=lambda r: r[0])
    las = np.zeros(len(rows), dtype=dentist_amd.LA_DTYPE)
    for i, (c, b, e) in enumerate(rows):
        las[i]["aread"], las[i]["abpos"], las[i]["aepos"], las[i]["bread"] = c, b, e, i
    regions = []
    for c, n in enumerate(lens):
        for _ in range(4):
            b = int(rng.integers(0, n))
            regions.append((c, b, int(min(n, b + rng.integers(0, 800)))))
    for ctx, win, mincov, minspan in [(1000, 500, 3, 3), (200, 50, 1, 1), (0, 700, 5, 2), (300, 5000, 2, 1)]:
        rep, weak = dentist_amd.validate_regions(las, co, regions, mincov, min_spanning_reads=minspan, region_context=ctx,
                                                 weak_coverage_window=win)
        exp_weak = []
        for r, (c, b, e) in enumerate(regions):
            al = [(x[1], x[2]) for x in rows if x[0] == c]
            sp, wk, ok, (cb, ce) = mc.validate_region(al, (b, e), lens[c], ctx, win, mincov, minspan)
            assert (rep[r]["num_spanning_reads"], bool(rep[r]["is_valid"]), rep[r]["ctx_begin"], rep[r]["ctx_end"]) == (sp, ok, cb, ce)
            assert rep[r]["weak_bp"] == sum(y - x for x, y in wk)
            exp_weak += [(c, x, y) for x, y in wk]
        assert [tuple(x) for x in weak.tolist()] == exp_weak


def test_validate_regions_on_a_closed_gap_layout():
    """A well covered closed gap is valid; a gap that only two reads cross is not (too few spanning reads
    and a weakly covered stretch that contains the gap)."""
    co = np.array([0, 20000], dtype=np.int64)
    good = [(0, 500 * i, 500 * i + 9000) for i in range(12)]
    las = np.zeros(len(good) + 2, dtype=dentist_amd.LA_DTYPE)
    for i, (c, b, e) in enumerate(good + [(0, 11000, 19000), (0, 11500, 19500)]):
        las[i]["aread"], las[i]["abpos"], las[i]["aepos"], las[i]["bread"] = c, b, e, i
    rep, weak = dentist_amd.validate_regions(las, co, [(0, 6000, 6100), (0, 17000, 17100)], 3)
    assert rep[0]["is_valid"] == 1 and rep[0]["num_spanning_reads"] >= 3 and rep[0]["weak_bp"] == 0
    assert rep[1]["is_valid"] == 0 and rep[1]["num_spanning_reads"] == 2 and rep[1]["weak_bp"] > 0
    assert all(c == 0 and b <= 17000 and e >= 17100 for c, b, e in weak.tolist())
