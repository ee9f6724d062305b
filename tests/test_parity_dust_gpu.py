"""DBdust's role on the device (k_dust through dh_db_dust) against the oracle (oracle/dust.c), and the
-mdust seeding exclusion in the alignment pass and in the fused `process` path -- bit exact."""
import os

import numpy as np
import pytest

import dentist_amd
from dentist_amd import sim
from helpers import assert_same_las
from oracle import process as pr
from oracle import pyoracle as oz

pytestmark = pytest.mark.gpu


def plant_low_complexity(seq, rng, every=4000):
    """Homopolymer runs, di-/tri-nucleotide microsatellites and short tandem repeats."""
    seq = seq.copy()
    pos = 500
    while pos + 400 < len(seq):
        kind = int(rng.integers(0, 4))
        ln = int(rng.integers(12, 300))
        if kind == 0:
            unit = rng.integers(0, 4, 1)
        elif kind == 1:
            unit = rng.integers(0, 4, 2)
        elif kind == 2:
            unit = rng.integers(0, 4, 3)
        else:
            unit = rng.integers(0, 4, int(rng.integers(4, 9)))
        rep = np.tile(unit, ln // len(unit) + 1)[:ln].astype(np.uint8)
        seq[pos:pos + ln] = rep
        pos += ln + int(rng.integers(every // 2, every * 2))
    return seq


def test_dust_mask_matches_the_oracle(gpu_ctx):
    rng = np.random.default_rng(11)
    g = plant_low_complexity(sim.genome(3, 400_000), rng, every=1500)
    g[1000:1010] = 4                      # a run of N: windows touching it are skipped
    g[200_000:200_003] = 4
    seqs = [g[:150_000], g[150_000:150_015], g[150_015:150_030], g[150_030:150_100], g[150_100:], sim.revcomp(g[:150_000])]
    db = sim.SeqDb.from_list(seqs)
    ptr, iv = oz.dust(db)
    d = gpu_ctx.db(db)
    d.dust()
    gptr, giv = d.get_mask()
    assert np.array_equal(gptr, ptr) and np.array_equal(giv, iv)
    assert ptr[-1] > 100
    # symmetric: the reverse complement gets the mirrored intervals
    n0 = len(seqs[0])
    fwd = iv[2 * ptr[0]:2 * ptr[1]].reshape(-1, 2)
    rev = iv[2 * ptr[5]:2 * ptr[6]].reshape(-1, 2)
    assert np.array_equal(np.sort(n0 - fwd[:, ::-1], axis=0), np.sort(rev, axis=0))
    # an explicit track is a layer of its own: the effective mask is its OR with the dust bits ...
    extra_ptr = np.zeros(db.n + 1, dtype=np.int64)
    extra_ptr[1:] = 1
    d.set_mask(extra_ptr, np.asarray([20_000, 20_500], dtype=np.int32))
    p2, i2 = d.get_mask()
    assert p2[-1] >= ptr[-1] and any(b <= 20_000 and e >= 20_500 for b, e in i2[:2 * p2[1]].reshape(-1, 2))
    # ... and a second dh_db_set_mask REPLACES the first track (set semantics), the dust layer stays
    d.set_mask(extra_ptr, np.asarray([30_000, 30_100], dtype=np.int32))
    p3, i3 = d.get_mask()
    first = i3[:2 * p3[1]].reshape(-1, 2)
    assert any(b <= 30_000 and e >= 30_100 for b, e in first)
    dust0 = iv[2 * ptr[0]:2 * ptr[1]].reshape(-1, 2)
    if not any(b < 20_500 and e > 20_000 for b, e in dust0):   # the old track is gone unless dust covers it anyway
        assert not any(b < 20_500 and e > 20_000 for b, e in first)
    empty = np.zeros(db.n + 1, dtype=np.int64)
    d.set_mask(empty, np.zeros(0, dtype=np.int32))
    p4, i4 = d.get_mask()
    assert np.array_equal(p4, ptr) and np.array_equal(i4, iv)   # only the dust layer is left
    d.set_mask(None, None)
    p5, _ = d.get_mask()
    assert p5[-1] == 0


def test_alignment_with_dust_masks_matches_the_oracle(gpu_ctx):
    """daligner / damapper -mdust: dusted k-mers are neither indexed nor looked up (both strands)."""
    rng = np.random.default_rng(5)
    w = sim.Workload(300_000, 3, 500, 6000, seed=71, spacing=15000)
    w.truth = plant_low_complexity(w.truth, rng)
    w.contigs, w.contig_start = sim.contigs_from_gaps(w.truth, w.gap_begin, w.gap_end)
    w.reads, w.read_truth = sim.reads(73, w.truth, 500, 6000)
    g = dentist_amd.default_align_opts()
    o = oz.default_opts(width=g.width)
    exp = oz.align_db(oz.with_dust(w.contigs), oz.with_dust(w.reads), o, nthreads=os.cpu_count() or 1)
    plain = oz.align_db(w.contigs, w.reads, o, nthreads=os.cpu_count() or 1)
    A, B = gpu_ctx.db(w.contigs), gpu_ctx.db(w.reads)
    A.dust()
    B.dust()
    got = gpu_ctx.align_db(A, B, g)
    st = gpu_ctx.align_stats()
    assert (st.hits, st.cands, st.alignments, st.wave_cells) == tuple(int(x) for x in exp[2])
    assert_same_las(got, exp[:2])
    assert exp[2][0] < plain[2][0], "the dust mask must remove k-mer hits"


def test_process_with_low_complexity_flanks_matches_the_oracle(gpu_ctx):
    """The fused path dusts the pile-up DB and the flank DB (package.d:476-482, 655-667); the oracle
    driver does the same with its own DUST -- consensus and splice coordinates stay bit exact."""
    rng = np.random.default_rng(9)
    w = sim.Workload(300_000, 3, 1200, 6000, seed=75, spacing=20000, gap_max=800)
    w.truth = plant_low_complexity(w.truth, rng, every=900)
    w.contigs, w.contig_start = sim.contigs_from_gaps(w.truth, w.gap_begin, w.gap_end)
    w.reads, w.read_truth = sim.reads(77, w.truth, 1200, 6000)
    g = dentist_amd.default_align_opts()
    A, B = gpu_ctx.db(w.contigs), gpu_ctx.db(w.reads)
    las, trace = gpu_ctx.align_db(A, B, g)
    po = dentist_amd.default_process_opts(rounds=2)
    assert po.dust == 1
    piles = dentist_amd.Pileups(las, w.contigs.off, po)
    rec, bases = dentist_amd.process_pileups(gpu_ctx, A, B, las, trace, piles, po)
    ndust = 0
    closed = 0
    for i in range(len(piles)):
        gap, tri = piles.get(i)
        e = pr.process_pile([tuple(t) for t in tri.tolist()], las, trace, w.contigs, w.reads, gap, rounds=2, nthreads=4)
        r = rec[i]
        assert (r["status"] == 0) == (e["status"] == "ok"), (gap, int(r["status"]), e["status"])
        if "pile" in e and getattr(e["pile"], "mask", None) is not None:
            ndust += int(e["pile"].mask[0][-1])
        if r["status"] != 0:
            continue
        closed += 1
        assert np.array_equal(bases[r["cons_off"]:r["cons_off"] + r["cons_len"]], e["consensus"]), gap
        assert (r["left_aepos"], r["right_abpos"], r["ins_begin"], r["ins_end"]) == \
               (e["left_aepos"], e["right_abpos"], e["ins_begin"], e["ins_end"])
    assert closed >= 2 and ndust > 0
    # without dust the pile-up alignment sees more k-mer hits (the option is honoured)
    po2 = dentist_amd.default_process_opts(rounds=2, dust=0)
    rec2, _ = dentist_amd.process_pileups(gpu_ctx, A, B, las, trace, piles, po2)
    assert len(rec2) == len(rec)
