"""Stage-level entry points (dh_tile_qv = DAScover + DASqv, dh_consensus = computeintrinsicqv +
daccord) through the C ABI against the oracle -- bit exact."""
import os

import numpy as np
import pytest

import dentist_amd
from dentist_amd import sim
from oracle import pyoracle as oz

pytestmark = pytest.mark.gpu
TS = 126


def pile_case(seed, nreads=24, rlen=4000):
    g = sim.genome(seed, 9000)
    reads, _ = sim.reads(seed + 1, g, nreads, rlen, 0, min_len=500)
    return reads


def pile_las(ctx, d):
    o = dentist_amd.default_align_opts(tspace=TS, skip_self=2, max_la=64, max_cand=128)
    return ctx.align_db(d, d, o)  # LAsort order: grouped by aread


@pytest.mark.parametrize("seed", [41, 43])
def test_tile_qv_matches_the_oracle(gpu_ctx, seed):
    reads = pile_case(seed)
    d = gpu_ctx.db(reads)
    las, trace = pile_las(gpu_ctx, d)
    assert len(las) > 50
    rlen = np.diff(reads.off).astype(np.int32)
    for cov in (4, reads.n):
        exp = oz.tile_qv(las, trace, rlen, TS, cov)
        got = dentist_amd.tile_qv(gpu_ctx, d, las, trace, TS, cov, exp.shape[1])
        assert np.array_equal(got, exp)
        assert got.min() <= 50 and (got[got != 255] <= 50).all()


def test_tile_qv_with_more_overlaps_than_fit_the_staging(gpu_ctx):
    """k_tile_qv keeps what a tile asks of an overlap in LDS for up to 512 overlaps of a read; deeper reads walk the
    records in memory.  Fabricated overlaps (no alignment is run: the stage reads records and trace pairs only): read 0
    has 700 of them, read 1 exactly 512, read 2 a handful, some disabled, some partial."""
    rng = np.random.default_rng(7)
    rl = np.array([3000, 2600, 1900], dtype=np.int64)
    reads = sim.SeqDb(np.zeros(int(rl.sum()), dtype=np.uint8), np.concatenate([[0], np.cumsum(rl)]))
    d = gpu_ctx.db(reads)
    recs, tr = [], []
    toff = 0
    for a, n in ((0, 700), (1, 512), (2, 9)):
        for i in range(n):
            ab = 0 if i % 3 else int(rng.integers(0, rl[a] // 2))
            ae = int(rl[a]) if i % 5 else int(rng.integers(ab + 300, rl[a]))
            ntp = (ae - 1) // TS - ab // TS + 1
            la = np.zeros(1, dtype=dentist_amd.LA_DTYPE)
            la["aread"], la["bread"] = a, (a + 1 + i) % 3
            la["abpos"], la["aepos"], la["bbpos"], la["bepos"] = ab, ae, 0, ae - ab
            la["tlen"], la["toff"] = 2 * ntp, toff
            la["flags"] = 0x20 if i % 11 == 0 else 0
            pairs = np.stack([rng.integers(0, 40, ntp), rng.integers(100, 140, ntp)], axis=1).astype(np.uint16)
            la["diffs"] = int(pairs[:, 0].sum())
            recs.append(la)
            tr.append(pairs.reshape(-1))
            toff += 2 * ntp
    las = np.ascontiguousarray(np.concatenate(recs))
    trace = np.ascontiguousarray(np.concatenate(tr))
    for cov in (4, 40, 1000):
        exp = oz.tile_qv(las, trace, rl.astype(np.int32), TS, cov)
        got = dentist_amd.tile_qv(gpu_ctx, d, las, trace, TS, cov, exp.shape[1])
        assert np.array_equal(got, exp)


@pytest.mark.parametrize("seed", [41, 47])
def test_consensus_of_one_read_matches_the_oracle(gpu_ctx, seed):
    reads = pile_case(seed)
    d = gpu_ctx.db(reads)
    las, trace = pile_las(gpu_ctx, d)
    for ref in (0, 5):
        exp = oz.consensus(reads.seq(ref), reads, las, trace, ref, TS)
        got = dentist_amd.consensus(gpu_ctx, d, las, trace, TS, ref, rounds=1)
        assert np.array_equal(got, exp)
    # more rounds only ever see the reads again: still a sequence close to the first round's
    two = dentist_amd.consensus(gpu_ctx, d, las, trace, TS, 0, rounds=2)
    ed, _ = oz.nw(dentist_amd.consensus(gpu_ctx, d, las, trace, TS, 0, rounds=1), two)
    assert ed <= 0.05 * len(two)


def test_stage_entry_points_reject_bad_input(gpu_ctx):
    reads = pile_case(51, nreads=8)
    d = gpu_ctx.db(reads)
    las, trace = pile_las(gpu_ctx, d)
    with pytest.raises(RuntimeError):
        dentist_amd.tile_qv(gpu_ctx, d, las[::-1].copy(), trace, TS, 4, 40)  # not grouped by aread
    with pytest.raises(RuntimeError):
        dentist_amd.consensus(gpu_ctx, d, las, trace, TS, reads.n + 3)


def test_consensus_known_answer_of_the_reference(gpu_ctx):
    """dazzler.d:4257-4299: the HIP consensus of the reference's own three-read pile equals the
    clean read (tests/golden/consensus_3reads.json, transcribed by scripts/make_golden_consensus3.py)."""
    import json
    d = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "consensus_3reads.json")))
    db = sim.SeqDb.from_list([sim.encode(r["sequence"].lower()) for r in d["reads"]])
    dd = gpu_ctx.db(db)
    o = dentist_amd.default_align_opts(skip_self=2, tspace=100, min_len=d["daligner_min_alignment_length"], max_la=64,
                                       max_cand=128)
    las, trace = gpu_ctx.align_db(dd, dd, o)
    assert len(las) == 6
    for ref in range(3):
        for rounds in (1, 3):
            cons = dentist_amd.consensus(gpu_ctx, dd, las, trace, 100, ref, rounds=rounds)
            assert sim.decode(cons) == d["expected_consensus"].lower()


def test_remapping_call_of_the_bubble_resolver(gpu_ctx):
    """dh_remap_skipping_reads = getReadAlignmentsOnContigs of resolveBubbles (collectPileUps/pileups.d:1316-1385):
    the skipping reads against the intermediate contigs alone, no mask, chains that do not cover their contig within the
    allowance disabled, ids of the full DBs -- against the same steps done by hand (host subsets, dh_align_db with chain
    flags, the completelyCovers rule of base.d:562-566)."""
    import dentist_amd
    from dentist_amd import sim
    from helpers import assert_same_las
    g = sim.genome(5, 60000)
    contigs = sim.SeqDb.from_list([g[:20000], g[20080:20700], g[20780:45000], g[45060:45900], g[46000:60000]])
    reads, truth = sim.reads(6, g, 260, 6000)
    # reads that touch one of the two short intermediate contigs (1 and 3)
    touch = lambda s, e: (truth[:, 0] < e - 100) & (truth[:, 1] > s + 100)   # noqa: E731
    read_ids = np.nonzero(touch(20080, 20700) | touch(45060, 45900))[0].astype(np.int32)
    contig_ids = np.asarray([1, 3], dtype=np.int32)
    assert len(read_ids) >= 20
    A, B = gpu_ctx.db(contigs), gpu_ctx.db(reads)
    o = dentist_amd.default_align_opts(algo=1, width=64, k=14, min_len=300)
    allowance = 100
    las, trace = gpu_ctx.remap_skipping_reads(A, B, contig_ids, read_ids, o, allowance)
    subA = sim.SeqDb.from_list([contigs.seq(int(c)) for c in contig_ids])
    subB = sim.SeqDb.from_list([reads.seq(int(r)) for r in read_ids])
    el, et = gpu_ctx.align_db(gpu_ctx.db(subA), gpu_ctx.db(subB), o, select_best=True)
    el = el.copy()
    i = 0
    while i < len(el):
        j = i + 1
        while j < len(el) and (el["flags"][j] & 0x8) and not (el["flags"][j] & 0x4):
            j += 1
        alen = subA.length(int(el["aread"][i]))
        if not (el["abpos"][i] <= allowance and el["aepos"][j - 1] >= alen - allowance):
            el["flags"][i:j] |= 0x20
        i = j
    el["aread"] = contig_ids[el["aread"]]
    el["bread"] = read_ids[el["bread"]]
    assert_same_las((las, trace), (el, et))
    on = las[(las["flags"] & 0x20) == 0]
    assert len(on) >= 10 and len(on) < len(las) and set(on["aread"].tolist()) == {1, 3}
    # what stays enabled really spans its contig in the read's truth
    cs = np.asarray([0, 20080, 20780, 45060, 46000])
    for r in on:
        s, e = truth[r["bread"], 0], truth[r["bread"], 1]
        assert s <= cs[r["aread"]] + allowance + 60 and e >= cs[r["aread"]] + contigs.length(int(r["aread"])) - allowance - 60
    with pytest.raises(dentist_amd.DhError):
        gpu_ctx.remap_skipping_reads(A, B, [3, 1], read_ids, o, allowance)   # ids must ascend
