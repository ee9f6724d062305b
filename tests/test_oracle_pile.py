"""The two oracle drivers of `collect` + `process` agree: oracle/pile.c (C, OpenMP over pile-ups; the
checker at larger sizes and the CPU baseline of bench.py) against oracle/process.py (Python, the
independent restatement the small parity cases use).  CPU only."""
import numpy as np
import pytest

from dentist_amd import sim
from oracle import process as pr
from oracle import pyoracle as oz


@pytest.mark.parametrize("seed,max_reads,rounds", [(17, 20, 3), (19, 60, 2)])
def test_c_and_python_oracle_drivers_agree(seed, max_reads, rounds):
    w = sim.Workload(300_000, 3, 1200, 6000, seed=seed, spacing=20000, gap_max=800)
    las, tr, _ = oz.align_db(w.contigs, w.reads, oz.default_opts(width=30), nthreads=4)
    po = oz.default_process_opts(max_reads=max_reads, rounds=rounds)
    gaps, tris = oz.collect_spanning_c(las, w.contigs, po)
    exp = pr.collect_spanning(las, tr, w.contigs, w.reads, max_reads=max_reads)
    assert sorted(exp) == gaps.tolist() and len(gaps) >= 2
    for g, t3 in zip(gaps, tris):
        assert [tuple(x) for x in t3.tolist()] == [tuple(int(v) for v in e) for e in exp[int(g)]]
        assert len(set(t3[:, 0].tolist())) == len(t3), "a read enters a pile-up once"
    rec, bases = oz.process_piles_c(w.contigs, w.reads, las, tr, gaps, tris, po, nthreads=4)
    closed = 0
    for r, g in zip(rec, gaps):
        e = pr.process_pile(exp[int(g)], las, tr, w.contigs, w.reads, int(g), rounds=rounds, nthreads=1)
        assert (e["status"] == "ok") == (r["status"] == 0), (g, e["status"], int(r["status"]))
        if r["status"] != 0:
            continue
        closed += 1
        assert np.array_equal(bases[r["cons_off"]:r["cons_off"] + r["cons_len"]], e["consensus"])
        assert (r["left_aepos"], r["right_abpos"], r["ins_begin"], r["ins_end"]) == \
               (e["left_aepos"], e["right_abpos"], e["ins_begin"], e["ins_end"])
        assert r["ref_idx"] == e["ref_idx"] and r["ref_read_id"] == e["read_ids"][e["ref_idx"]]
        assert (r["crop_left"], r["crop_right"]) == (e["cropL"], e["cropR"]) and r["nreads"] == e["pile"].n
    assert closed >= 2


def test_quality_cut_keeps_the_cleanest_reads():
    """More spanning reads than max_reads: the kept ones are those whose anchoring LAs have the lowest
    error rate, listed in read order (dh_collect_spanning does the same, tests/test_parity_process_gpu)."""
    w = sim.Workload(200_000, 1, 1500, 6000, seed=23, spacing=20000, gap_max=500)
    las, tr, _ = oz.align_db(w.contigs, w.reads, oz.default_opts(width=30), nthreads=4)
    allr = pr.collect_spanning(las, tr, w.contigs, w.reads, max_reads=10**6)
    cut = pr.collect_spanning(las, tr, w.contigs, w.reads, max_reads=8)
    (g, full), = allr.items()
    assert len(full) > 16 and len(cut[g]) == 8

    def err(e):
        L, R = las[e[1]], las[e[2]]
        return (int(L["diffs"]) + int(R["diffs"])) / (int(L["aepos"] - L["abpos"]) + int(R["aepos"] - R["abpos"]))
    kept = set(cut[g])
    assert max(err(e) for e in cut[g]) <= min(err(e) for e in full if e not in kept) + 1e-6
    assert [e[0] for e in cut[g]] == sorted(e[0] for e in cut[g])


def _las(rows):
    a = np.zeros(len(rows), dtype=oz.LA_DTYPE)
    for i, (ar, br, ab, ae, bb, be, fl) in enumerate(rows):
        a[i] = (0, 0, ab, bb, ae, be, fl, ar, br, 0, i)
    return a


def test_chains_below_the_default_min_relative_score():
    """buildAlignmentChains (common/alignments/chaining.d:151-312) with --min-relative-score below 1: four LAs of one
    pair -- a, then b and c which both continue a (b scores more), and d far away in a component of its own.
      1.0 / 0.9: the best chain a -> b only (a -> c scores 1800 of 2000, d 600).
      0.5: a -> c is within 0.5 of the best and shares a with it: an ALTERNATE chain (chain start without BEST, dazzler.d:
           2063-2068) written with its whole path -- a appears twice (toff tells the copies apart); d (600 < 1000) is dropped.
      0.0: d as well, a best chain of its own component (START | BEST).
    The C and the Python restatement agree on these and on random pairs."""
    rows = [(0, 1, 0, 1000, 0, 1000, 0), (0, 1, 1100, 2100, 1100, 2100, 0), (0, 1, 1100, 1900, 1150, 1950, 0),
            (0, 1, 5000, 5600, 100, 700, 0)]
    want = {1.0: [(0, 0x14), (1, 0x8), (2, 0x20), (3, 0x20)], 0.9: [(0, 0x14), (1, 0x8), (2, 0x20), (3, 0x20)],
            0.5: [(0, 0x14), (0, 0x4), (1, 0x8), (2, 0x8), (3, 0x20)], 0.0: [(0, 0x14), (0, 0x4), (1, 0x8), (2, 0x8), (3, 0x14)]}
    for rel, exp in want.items():
        p = pr.chain_pile_las(_las(rows), min_rel_score=rel, min_score=126)
        c = oz.chain_las_c(_las(rows), 126, int(rel * 1e6))
        assert [(int(x["toff"]), int(x["flags"])) for x in p] == exp
        assert p.tobytes() == c.tobytes()
    rng = np.random.default_rng(1)
    ndup = 0
    for it in range(200):
        n = int(rng.integers(1, 9))
        rows = []
        for _ in range(n):
            ab, ln, d = int(rng.integers(0, 20) * 300), int(rng.integers(2, 8) * 150), int(rng.integers(-3, 4) * 100)
            bb = max(0, ab + d)
            rows.append((0, 1, ab, ab + ln, bb, bb + ln + int(rng.integers(-20, 20)), int(rng.integers(0, 2)) if it % 3 == 0 else 0))
        rows.sort(key=lambda r: (r[2], r[4]))
        for rel in (1.0, 0.8, 0.3):
            p = pr.chain_pile_las(_las(rows), min_rel_score=rel, min_score=126)
            c = oz.chain_las_c(_las(rows), 126, int(rel * 1e6))
            assert p.tobytes() == c.tobytes(), (rows, rel)
            ndup += len(p) - n
            live = p[(p["flags"] & 0x20) == 0]
            assert len(live) >= 1 and ((live["flags"] & 0x1c) != 0).all()
    assert ndup > 50   # (alternate chains do occur in the sample)


def test_c_and_python_drivers_agree_below_the_default_min_relative_score():
    """`process` with --min-relative-score 0.3 through both drivers, on pile-ups in which half of the reads carry 1.5 kb of
    foreign bases inside the gap: such a read and a plain one align as two local alignments that cannot be chained -- at
    the default only the better one survives, at 0.3 both do (and the consensus is built from more records)."""
    from helpers import plant_gap_insertions
    w = sim.Workload(300_000, 2, 1500, 7000, seed=53, spacing=20000, gap_min=1500, gap_max=2500)
    reads, planted = plant_gap_insertions(w, np.random.default_rng(3))
    assert len(planted) >= 6
    las, tr, _ = oz.align_db(w.contigs, reads, oz.default_opts(width=30), nthreads=4)
    po = oz.default_process_opts(max_reads=30, rounds=2, min_relative_score_ppm=300000)
    gaps, tris = oz.collect_spanning_c(las, w.contigs, po)
    exp = pr.collect_spanning(las, tr, w.contigs, reads, max_reads=30)
    rec, bases = oz.process_piles_c(w.contigs, reads, las, tr, gaps, tris, po, nthreads=4)
    closed = extra = 0
    for r, g in zip(rec, gaps):
        e = pr.process_pile(exp[int(g)], las, tr, w.contigs, reads, int(g), rounds=2, nthreads=1, min_rel_score=0.3)
        e1 = pr.process_pile(exp[int(g)], las, tr, w.contigs, reads, int(g), rounds=1, nthreads=1)
        assert (e["status"] == "ok") == (r["status"] == 0)
        if r["status"] != 0:
            continue
        closed += 1
        # (the records the tile QVs see, computeQVs package.d:486-505; the parts of such a pair are improper overlaps and
        # leave before the consensus, :507-512)
        extra += int(((e["chained_las"]["flags"] & 0x20) == 0).sum()) - int(((e1["chained_las"]["flags"] & 0x20) == 0).sum())
        assert np.array_equal(bases[r["cons_off"]:r["cons_off"] + r["cons_len"]], e["consensus"])
    assert closed >= 1 and extra > 0   # (the lower threshold keeps more records)


def test_tandem_self_alignments_of_the_oracle():
    """oz_opts.skip_self = 3 (the role of `datander <block>`, DAMASKER; DENTIST's call commandline.d:2866-2876): a read against
    itself, below the main diagonal.  Every planted tandem array is found -- records with aread == bread, A after B, on a
    diagonal that is a multiple of the period, inside the array -- also the one whose period (24) is smaller than half the
    band of DH-2: the rule that B's base must come before A's keeps the extension off the main diagonal, where everything
    matches.  Reads without an array give no record."""
    from helpers import tandem_reads
    db, truth = tandem_reads()
    o = oz.default_opts(skip_self=3, strands=1, algo=1, width=64, k=12, band_shift=4, min_len=500, tspace=126)
    las, tr, _ = oz.align_db(db, db, o, nthreads=4)
    assert len(las) > 0 and (las["aread"] == las["bread"]).all() and (las["abpos"] > las["bbpos"]).all()
    assert (las["flags"] & 1 == 0).all()
    planted = {t[0]: t for t in truth}
    assert set(las["aread"].tolist()) == set(planted)
    for la in las:
        _, b, e, per = planted[int(la["aread"])]
        d0, d1 = int(la["abpos"] - la["bbpos"]), int(la["aepos"] - la["bepos"])
        assert b - 60 <= la["bbpos"] and la["aepos"] <= e + 60
        m = max(1, round(d0 / per))
        assert abs(d0 - m * per) <= 0.15 * m * per + 8 and abs(d1 - m * per) <= 0.15 * m * per + 8
    for t in truth:   # the first off-diagonal (one period) covers the array but for its first copy
        mine = las[las["aread"] == t[0]]
        assert (mine["aepos"] - mine["bbpos"]).max() >= 0.8 * (t[2] - t[1])
