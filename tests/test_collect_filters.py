"""The six alignment filters of `dentist collect` (filter.d:122-356): unit cases of the reference's
predicates (base.d:600-640 isFullyContained), the product (dh_collect_filter, host code through the C
ABI) against the oracle restatement on random alignment sets, and a planted scenario.  CPU only."""
import numpy as np
import pytest

import dentist_amd
from oracle import collect_filters as cf

LA = dentist_amd.LA_DTYPE


def la(aread, bread, abpos, aepos, bbpos, bepos, diffs=0, comp=0):
    r = np.zeros(1, dtype=LA)[0]
    r["aread"], r["bread"], r["abpos"], r["aepos"], r["bbpos"], r["bepos"], r["diffs"], r["flags"] = \
        aread, bread, abpos, aepos, bbpos, bepos, diffs, comp
    return r


def test_is_fully_contained_cases_of_the_reference():
    """base.d:600-640: contig A of 50, read B of 15."""
    assert cf.is_fully_contained(la(0, 0, 30, 35, 5, 10), 50, 15)        # read with extension on A from 25 to 40
    assert not cf.is_fully_contained(la(0, 0, 0, 10, 5, 10), 50, 15)     # from -5 to 15
    assert not cf.is_fully_contained(la(0, 0, 40, 50, 5, 10), 50, 15)    # from 35 to 55
    # the product agrees: the redundant filter drops exactly the first read
    las = np.stack([la(0, 0, 30, 35, 5, 10), la(0, 1, 0, 10, 5, 10), la(0, 2, 40, 50, 5, 10)])
    po = dentist_amd.default_process_opts(allowance=100, min_anchor=0)
    out, dropped, used = dentist_amd.collect_filter(las, [0, 50], [0, 15, 30, 45], po)
    assert used.tolist() == [0, 1, 1] and dropped[5] == 1 and (out["flags"] & 0x20 != 0).tolist() == [True, False, False]


def random_las(rng, nreads, ncontigs, clen, rlen):
    """1-3 alignments per read; most are proper (touch a contig or read end), some low quality, some
    shrunk copies of another alignment (contained), some reads short enough to fit inside a contig."""
    rows = []
    for r in range(nreads):
        for _ in range(int(rng.integers(1, 4))):
            c, comp = int(rng.integers(0, ncontigs)), int(rng.integers(0, 2))
            ln = int(rng.integers(600, 5000))
            kind = rng.random()
            if kind < 0.35:      # read hangs over the contig begin
                ab, bb = 0, rlen - ln
            elif kind < 0.7:     # ... over the contig end
                ab, bb = clen - ln, 0
            elif kind < 0.85:    # inside the contig, whole read aligned (short read): redundant
                ab, bb, ln = int(rng.integers(1000, clen - 7000)), 0, rlen
            else:                # somewhere: improper
                ab, bb = int(rng.integers(100, clen - 6000)), int(rng.integers(100, rlen - ln))
            d = int(ln * rng.choice([0.05, 0.13, 0.2, 0.35]))
            rows.append(la(c, r, ab, ab + ln, bb, bb + ln, d, comp))
            if rng.random() < 0.15:   # a contained copy
                rows.append(la(c, r, ab + 50, ab + ln - 50, bb + 50, bb + ln - 50, d // 2, comp))
    out = np.stack(rows)
    return out[rng.permutation(len(out))]


def test_product_filters_equal_the_oracle_on_random_sets():
    total = np.zeros(6, dtype=np.int64)
    for seed in range(1, 7):
        rng = np.random.default_rng(seed)
        nc, nr, clen, rlen = 6, 150, 20000, 6000
        las = random_las(rng, nr, nc, clen, rlen)
        coff = np.arange(nc + 1, dtype=np.int64) * clen
        roff = np.arange(nr + 1, dtype=np.int64) * rlen
        ptr = np.arange(nc + 1, dtype=np.int64) * 2
        iv = np.tile(np.asarray([0, 4000, 15000, 19990], dtype=np.int32), nc)
        po = dentist_amd.default_process_opts()
        got, gd, gu = dentist_amd.collect_filter(las, coff, roff, po, repeat_mask=(ptr, iv))
        exp, ed, eu = cf.collect_filter(las, coff, roff, repeat_mask=(ptr, iv))
        assert np.array_equal(got["flags"], exp["flags"]) and np.array_equal(gd, ed) and np.array_equal(gu, eu)
        assert (got["flags"] & 0x20 == 0).sum() > 0
        total += ed
        # the same records grouped by read, as the aligner emits them: the product takes its per-read path (no global
        # regrouping), the oracle does not care
        byread = las[np.argsort(las["bread"], kind="stable")]
        got2, gd2, gu2 = dentist_amd.collect_filter(byread, coff, roff, po, repeat_mask=(ptr, iv))
        exp2, ed2, eu2 = cf.collect_filter(byread, coff, roff, repeat_mask=(ptr, iv))
        assert np.array_equal(got2["flags"], exp2["flags"]) and np.array_equal(gd2, ed2) and np.array_equal(gu2, eu2)
        assert np.array_equal(gd2, gd) and np.array_equal(gu2, gu)
    assert all(d > 0 for d in total), total     # every stage is exercised


def test_planted_scenario():
    """One contig pair; reads: a clean spanning read, a low-quality one, an improper one, one anchored
    only in a repeat, one contained alignment, an ambiguous read and a redundant one."""
    coff, roff = [0, 10000, 20000], [0] + [6000 * (i + 1) for i in range(7)]
    las = np.stack([
        la(0, 0, 7000, 10000, 0, 3000, diffs=300), la(1, 0, 0, 2900, 3100, 6000, diffs=300),     # spanning read 0: kept
        la(0, 1, 7000, 10000, 0, 3000, diffs=1200),                                               # LQ (40 %)
        la(0, 2, 3000, 5000, 2000, 4000, diffs=100),                                              # improper
        la(0, 3, 9400, 10000, 0, 600, diffs=50),                                                  # weakly anchored (repeat)
        la(0, 4, 6000, 10000, 0, 4000, diffs=300), la(0, 4, 6050, 10000, 50, 4000, diffs=290),    # second is contained
        la(0, 5, 7000, 10000, 0, 3000, diffs=300), la(1, 5, 0, 4000, 2000, 6000, diffs=300),      # overlap on the read: ambiguous
        la(0, 6, 1000, 7000, 0, 6000, diffs=500),                                                 # fits inside contig 0: redundant
    ])
    ptr, iv = np.asarray([0, 1, 1], dtype=np.int64), np.asarray([9300, 10000], dtype=np.int32)
    po = dentist_amd.default_process_opts()
    out, dropped, used = dentist_amd.collect_filter(las, coff, roff, po, repeat_mask=(ptr, iv))
    assert dropped.tolist() == [1, 1, 1, 1, 2, 1]
    assert (out["flags"] & 0x20 == 0).tolist() == [True, True, False, False, False, True, False, False, False, False]
    assert used.tolist() == [1, 1, 1, 1, 1, 0, 0]
    exp, ed, eu = cf.collect_filter(las, np.asarray(coff), np.asarray(roff), repeat_mask=(ptr, iv))
    assert np.array_equal(out["flags"], exp["flags"]) and np.array_equal(dropped, ed) and np.array_equal(used, eu)


# ------------------------------------------------------------------ alignment chains as the unit (base.d:306-421)
def chained(rows):
    """rows of one chain, in order: START on the first, NEXT on the others (dazzler.d:1728-1758)."""
    out = np.stack(rows)
    out["flags"] |= 0x8
    out["flags"][0] = (int(out["flags"][0]) & 0xFFFFFFF7) | 0x4
    return out


def test_chain_predicates_of_the_reference():
    """base.d:608-660 isFullyContained on chains of two, :662-680 coveredBases, :683-715 totalDiffs / averageErrorRate."""
    two = chained([la(0, 0, 10, 20, 5, 10, 1), la(0, 0, 30, 40, 5, 10, 1)])       # read with extension on A from 5 to 45
    u, cov, _ = cf.chain_units(two)
    assert len(u) == 1 and cf.is_fully_contained(u[0], 50, 15)
    wide = chained([la(0, 0, 0, 20, 5, 10, 1), la(0, 0, 30, 50, 5, 10, 1)])       # from -5 to 55
    assert not cf.is_fully_contained(cf.chain_units(wide)[0][0], 50, 15)
    c3 = chained([la(0, 0, 1, 3, 1, 3, 1), la(0, 0, 5, 10, 5, 10, 2)])
    u, cov, _ = cf.chain_units(c3)
    assert cov == [7] and int(u[0]["diffs"]) == 3                                  # averageErrorRate = 3 / 7
    # the product: the redundant filter judges the chain (members 10..20 and 30..40 of a read of 15: inside the contig of 50)
    po = dentist_amd.default_process_opts(allowance=100, min_anchor=0)
    out, dropped, used = dentist_amd.collect_filter(np.concatenate([two, wide.copy()]), [0, 50], [0, 15], po)
    exp, ed, eu = cf.collect_filter(np.concatenate([two, wide.copy()]), np.asarray([0, 50]), np.asarray([0, 15]), min_anchor=0)
    assert np.array_equal(out["flags"], exp["flags"]) and np.array_equal(dropped, ed) and np.array_equal(used, eu)


def test_chains_are_the_unit_of_every_filter():
    """A read with a 3 kb insertion maps as two collinear records: as ONE chain it is a proper alignment with the error
    rate of both parts together and it is dropped or kept as a whole; judged record by record (NEXT flags cleared) either
    part would be improper (it ends or begins in the middle of read and contig).  Product == oracle on random chained sets."""
    coff, roff = np.asarray([0, 20000]), np.asarray([0, 12000, 24000])
    a = la(0, 0, 12000, 15000, 0, 3000, diffs=300)      # read 0 enters the contig at 12000 ...
    b = la(0, 0, 15050, 20000, 6050, 11000, diffs=500)  # ... skips 3 kb of its own and runs to the contig's end
    las = np.concatenate([chained([a, b]), np.stack([la(0, 1, 14000, 20000, 0, 6000, diffs=600)])])
    po = dentist_amd.default_process_opts()
    out, dropped, used = dentist_amd.collect_filter(las, coff, roff, po)
    assert dropped.tolist() == [0, 0, 0, 0, 0, 0] and not np.any(out["flags"] & 0x20)
    single = las.copy()
    single["flags"] &= ~np.uint32(0xC)
    out1, dropped1, _ = dentist_amd.collect_filter(single, coff, roff, po)
    assert dropped1[1] == 2 and bool(out1["flags"][0] & 0x20) and bool(out1["flags"][1] & 0x20)   # either part alone is improper
    # low quality is decided on totalDiffs / coveredBases: 2400 + 100 differences over 3000 + 4950 bases = 31 %
    lq = np.concatenate([chained([la(0, 0, 12000, 15000, 0, 3000, diffs=2400), la(0, 0, 15050, 20000, 6050, 11000, diffs=100)])])
    out2, dropped2, _ = dentist_amd.collect_filter(lq, coff, roff, po)
    assert dropped2[0] == 1 and np.all(out2["flags"] & 0x20)
    total = np.zeros(6, dtype=np.int64)
    for seed in range(1, 7):
        rng = np.random.default_rng(100 + seed)
        nc, nr, clen, rlen = 5, 120, 20000, 9000
        base = random_las(rng, nr, nc, clen, rlen)
        base = base[np.argsort(base["bread"], kind="stable")]
        rows = []
        for r in base:   # a third of the records are split into a chain of two around a gap of 0 .. 2 kb
            ln = int(r["aepos"] - r["abpos"])
            if rng.random() < 0.33 and ln > 1500:
                cut, gap = int(rng.integers(500, ln - 500)), int(rng.integers(0, 2000))
                p, q = r.copy(), r.copy()
                p["aepos"], p["bepos"], p["diffs"] = r["abpos"] + cut, r["bbpos"] + cut, r["diffs"] // 2
                q["abpos"], q["bbpos"], q["diffs"] = r["abpos"] + cut + 20, min(int(r["bepos"]) - 10, int(r["bbpos"]) + cut + 20 + gap), r["diffs"] - r["diffs"] // 2
                if q["abpos"] < q["aepos"] and q["bbpos"] < q["bepos"]:
                    rows += list(chained([p, q]))
                    continue
            r = r.copy()
            r["flags"] |= 0x4
            rows.append(r)
        las = np.stack(rows)
        coff2 = np.arange(nc + 1, dtype=np.int64) * clen
        roff2 = np.arange(nr + 1, dtype=np.int64) * rlen
        ptr = np.arange(nc + 1, dtype=np.int64) * 2
        iv = np.tile(np.asarray([0, 4000, 15000, 19990], dtype=np.int32), nc)
        got, gd, gu = dentist_amd.collect_filter(las, coff2, roff2, po, repeat_mask=(ptr, iv))
        exp, ed, eu = cf.collect_filter(las, coff2, roff2, repeat_mask=(ptr, iv))
        assert np.array_equal(got["flags"], exp["flags"]) and np.array_equal(gd, ed) and np.array_equal(gu, eu)
        for i, j in cf.chain_ranges(las):   # a chain is kept or dropped as a whole
            assert len(set((got["flags"][i:j] & 0x20).tolist())) == 1
        total += ed
    assert all(d > 0 for d in total), total


def test_common_trace_point_over_regions_with_holes_and_a_mask():
    """getCommonTracePoint (cropper.d:446-500) on alignment chains: the common region of a flank is the intersection of
    the chains' A regions (a chain = the union of its members' A intervals, common/package.d:228-241), so it has a hole
    where a chained read misses contig bases; candidates come from the inner side for front seeds; trace points inside
    the repeat mask are avoided as long as one outside exists.  Product (dh_common_trace_point) == oracle on hand-made
    and on random cases."""
    from oracle import process as pr
    plain = np.stack([la(0, 0, 1000, 5000, 0, 4000), la(0, 1, 1250, 4800, 0, 3550)])
    hole = chained([la(0, 2, 900, 2030, 0, 1130), la(0, 2, 3170, 5000, 1130, 2960)])
    las = np.concatenate([plain, hole])
    reg = pr.intersect_regions([pr.region_of(las, i) for i in (0, 1, 2)])
    assert reg == [[1250, 2030], [3170, 4800]]
    for front, exp in ((False, 1300), (True, 4700)):   # candidates: iota(ceil(min), ceil(sup), 100) -- 4800 is not one
        assert pr.common_trace_point([pr.region_of(las, i) for i in (0, 1, 2)], 5000, 100, front) == exp
        assert dentist_amd.common_trace_point(las, [0, 1, 2], 5000, 100, front) == exp
    # a candidate inside the hole is skipped: only chain 2 and the back seed -> 900 .. 2030 | 3170 .. 5000
    assert dentist_amd.common_trace_point(las, [2], 5000, 100, True) == 4900
    assert dentist_amd.common_trace_point(las, [2], 4950, 100, True) == 4950   # the contig end is a candidate when ceil(sup) lies beyond it
    only_tail = np.concatenate([np.stack([la(0, 0, 2000, 3200, 0, 1200)]), hole])
    assert pr.intersect_regions([pr.region_of(only_tail, i) for i in (0, 1)]) == [[2000, 2030], [3170, 3200]]
    assert dentist_amd.common_trace_point(only_tail, [0, 1], 5000, 100, False) == 2000
    assert dentist_amd.common_trace_point(only_tail, [0, 1], 5000, 100, True) == 2000   # 3100 .. 2100 lie in the hole
    # the mask: outside first, inside when nothing else is left
    assert dentist_amd.common_trace_point(las, [0, 1, 2], 5000, 100, False, mask=[(1200, 1950)]) == 2000
    assert dentist_amd.common_trace_point(las, [0, 1, 2], 5000, 100, False, mask=[(0, 5000)]) == 1300
    assert dentist_amd.common_trace_point(las, [0], 5000, 100, False, mask=[(900, 1000), (1000 + 1, 4950)]) == 1000
    # no common region at all
    far = np.concatenate([plain, np.stack([la(0, 3, 4900, 5000, 0, 100)])])
    assert dentist_amd.common_trace_point(far, [0, 1, 2], 5000, 100, False) == -1
    rng = np.random.default_rng(8)
    for _ in range(300):
        rows, firsts = [], []
        clen = int(rng.integers(3000, 9000))
        for r in range(int(rng.integers(1, 6))):
            b = int(rng.integers(0, clen // 2))
            e = int(rng.integers(b + 200, clen + 1))
            firsts.append(len(rows))
            if rng.random() < 0.5 and e - b > 900:
                m = int(rng.integers(b + 200, e - 400))
                rows += list(chained([la(0, r, b, m, 0, m - b), la(0, r, m + int(rng.integers(0, 300)), e, m - b, e - b)]))
            else:
                rows.append(la(0, r, b, e, 0, e - b))
        arr = np.stack(rows)
        mask = []
        x = 0
        while rng.random() < 0.6 and x < clen - 50:
            b = int(rng.integers(x, clen - 10))
            e = int(rng.integers(b + 1, min(clen, b + 2500) + 1))
            mask.append((b, e))
            x = e + 1
        for front in (False, True):
            exp = pr.common_trace_point([pr.region_of(arr, i) for i in firsts], clen, 100, front, mask=mask)
            assert dentist_amd.common_trace_point(arr, firsts, clen, 100, front, mask=mask) == exp
