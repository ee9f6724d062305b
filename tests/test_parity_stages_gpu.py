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
