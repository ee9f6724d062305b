"""Parity of the HIP alignment pass (through the C ABI) with the CPU oracle -- bit exact.

Coordinates, read ids, flags, diffs and every trace value must be identical (integer work).
"""
import os

import numpy as np
import pytest

import dentist_amd
from dentist_amd import sim
from helpers import assert_same_las, check_trace_invariants, la_rows
from oracle import pyoracle as oz

pytestmark = pytest.mark.gpu

OPT_FIELDS = ("k", "hmin", "band_shift", "tspace", "min_len", "pen", "xdrop", "max_err_ppm", "max_cand",
              "max_la", "tcap", "strands", "skip_self", "dmax", "width", "kmer_mod", "algo")


def both_opts(**kw):
    g = dentist_amd.default_align_opts(**kw)
    o = oz.default_opts()
    for f in OPT_FIELDS:
        setattr(o, f, getattr(g, f))
    return g, o


def run_both(ctx, A, B, same=False, **kw):
    g, o = both_opts(**kw)
    exp = oz.align_db(A, B, o, nthreads=os.cpu_count() or 1)
    dA = ctx.db(A)
    dB = dA if same else ctx.db(B)
    got = ctx.align_db(dA, dB, g)
    st = ctx.align_stats()
    assert (st.hits, st.cands, st.alignments, st.wave_cells) == tuple(int(x) for x in exp[2])
    assert_same_las(got, exp[:2])
    check_trace_invariants(got[0], got[1], g.tspace)
    return got


def test_mapping_reads_to_contigs(gpu_ctx):
    w = sim.Workload(400_000, 4, 800, 5000, seed=7, spacing=20000)
    las, _ = run_both(gpu_ctx, w.contigs, w.reads)
    assert len(set(las["bread"].tolist())) == w.reads.n


@pytest.mark.parametrize("seed,rl,err", [(3, 2500, 0.13), (5, 9000, 0.13), (9, 6000, 0.05), (13, 4000, 0.20)])
def test_mapping_various_lengths_and_error_rates(gpu_ctx, seed, rl, err):
    w = sim.Workload(250_000, 3, 250, rl, seed=seed, err=err, spacing=15000)
    run_both(gpu_ctx, w.contigs, w.reads)


def test_lognormal_read_lengths_like_the_reference_fixture(gpu_ctx):
    """simulator -m25000 -s12500 -e.13 (tests/test-commands.sh:7-13) on the 4 097 bp fixture."""
    raw = open(os.path.join(os.path.dirname(__file__), "golden", "test_commands_assembly_reference.fasta")).read()
    seq = sim.encode("".join(raw.split("\n")[1:]))
    contigs = sim.SeqDb.from_list([seq[:2000], seq[2097:]])
    reads, _ = sim.reads(1724161952, seq, 33, 25000, 12500, min_len=500)
    las, _ = run_both(gpu_ctx, contigs, reads, min_len=500)
    assert len(las) > 0


@pytest.mark.parametrize("mod", [2, 4])
def test_modimer_sampling(gpu_ctx, mod):
    """daligner -%: index and query only k-mers whose hash is 0 modulo kmer_mod."""
    w = sim.Workload(300_000, 3, 400, 6000, seed=29, spacing=15000)
    las, _ = run_both(gpu_ctx, w.contigs, w.reads, kmer_mod=mod)
    assert len(set(las["bread"].tolist())) == w.reads.n


@pytest.mark.parametrize("k,mod", [(20, 1), (20, 4), (17, 2), (24, 4)])
def test_long_kmers_like_damapper(gpu_ctx, k, mod):
    """damapper's own default is -k20 (DENTIST passes no -k, commandline.d:2943-2955): keys beyond
    32 bits take the wide hash path of the sampler and a deeper directory shift."""
    w = sim.Workload(300_000, 3, 400, 8000, seed=31, spacing=15000)
    las, _ = run_both(gpu_ctx, w.contigs, w.reads, k=k, kmer_mod=mod)
    assert len(set(las["bread"].tolist())) >= 0.99 * w.reads.n


def test_read_blocks_merged_equal_the_single_call(gpu_ctx, monkeypatch):
    """dh_align_db_block per DBsplit-style block + dh_la_set_merge (LAmerge) == dh_align_db of the
    whole DB; with a small DH_ALIGN_CHUNK the single call runs its own chunk loop, whose derived
    copies (reverse complement, 2-bit packed) are made per chunk in scratch memory."""
    w = sim.Workload(300_000, 3, 900, 5000, seed=53, spacing=15000)
    g, o = both_opts(kmer_mod=2)
    exp = oz.align_db(w.contigs, w.reads, o, nthreads=os.cpu_count() or 1)
    A, B = gpu_ctx.db(w.contigs), gpu_ctx.db(w.reads)
    bounds = [0, 1, 250, 251, 600, 900]
    hs = [gpu_ctx.align_db_block(A, B, bounds[i], bounds[i + 1] - bounds[i], g, raw=True) for i in range(len(bounds) - 1)]
    assert_same_las(dentist_amd.merge_las(hs), exp[:2])
    monkeypatch.setenv("DH_ALIGN_CHUNK", "300")
    B.drop_cache()
    assert_same_las(gpu_ctx.align_db(A, B, g), exp[:2])
    with pytest.raises(dentist_amd.DhError):
        gpu_ctx.align_db_block(A, B, 800, 200, g)


def test_long_b_sequences_use_the_hbm_staged_seed_path(gpu_ctx):
    """Contigs as B against an index of the reads (the transposed `damapper -C` file): far more
    than 16384 k-mer hits per sequence -> hits are staged in HBM, results still bit-exact."""
    w = sim.Workload(150_000, 1, 600, 4000, seed=37, spacing=15000)
    las, _ = run_both(gpu_ctx, w.reads, w.contigs, max_la=64, max_cand=256)
    assert gpu_ctx.align_stats().big_items > 0
    assert len(las) > 100


def test_soft_masks_exclude_seeds_on_both_sides(gpu_ctx):
    """daligner/damapper -m<track>: k-mers touching a masked interval are neither indexed (A) nor
    looked up (B, mirrored for the complement strand); alignments still extend through the mask."""
    w = sim.Workload(200_000, 2, 300, 5000, seed=43, spacing=15000)
    rng = np.random.default_rng(5)

    def random_mask(db, frac):
        ptr, iv = [0], []
        for i in range(db.n):
            n, pos = db.length(i), 0
            while True:
                pos += int(rng.integers(200, 3000))
                ln = int(rng.integers(20, int(3000 * frac) + 30))
                if pos + ln >= n:
                    break
                iv += [pos, pos + ln]
                pos += ln
            ptr.append(len(iv) // 2)
        return np.asarray(ptr, dtype=np.int64), np.asarray(iv + [0, 0], dtype=np.int32)

    w.contigs.mask = random_mask(w.contigs, 0.5)
    w.reads.mask = random_mask(w.reads, 0.3)
    plain_hits = None
    las, _ = run_both(gpu_ctx, w.contigs, w.reads)
    masked_hits = gpu_ctx.align_stats().hits
    w.contigs.mask = w.reads.mask = None
    run_both(gpu_ctx, w.contigs, w.reads)
    assert masked_hits < 0.8 * gpu_ctx.align_stats().hits
    assert len(set(las["bread"].tolist())) >= 0.95 * w.reads.n


def pile(seed, glen=20000, n=30, rl=6000):
    g = sim.genome(seed, glen)
    reads, _ = sim.reads(seed + 1, g, n, rl)
    return reads


def test_pile_all_vs_all_tspace_126(gpu_ctx):
    """daligner -s126 -l500 on a pile-up DB against itself (commandline.d:2886-2902)."""
    p = pile(21)
    las, _ = run_both(gpu_ctx, p, p, same=True, tspace=126, skip_self=1, min_len=500)
    assert len(las) > p.n
    assert not np.any(las["aread"] == las["bread"])


@pytest.mark.parametrize("seed", [21, 57])
def test_symmetric_all_vs_all_emits_both_records_from_one_alignment(gpu_ctx, seed):
    """skip_self = 2: each unordered pair is aligned once (aread < bread) and the transposed record
    with its own trace (grid of the other read, mirrored for complemented overlaps) is emitted too."""
    p = pile(seed)
    las, trace = run_both(gpu_ctx, p, p, same=True, tspace=126, skip_self=2, min_len=500, max_la=64, max_cand=128)
    key = lambda x: (int(x["aread"]), int(x["bread"]), int(x["flags"]) & 1)
    recs = {key(x): x for x in las}
    assert len(recs) == len(las) and len(las) > p.n
    for (a, b, c), x in recs.items():
        y = recs[(b, a, c)]
        la_, lb_ = p.length(a), p.length(b)
        if c == 0:
            assert (x["abpos"], x["aepos"], x["bbpos"], x["bepos"]) == (y["bbpos"], y["bepos"], y["abpos"], y["aepos"])
        else:
            assert (x["abpos"], x["aepos"], x["bbpos"], x["bepos"]) == (la_ - y["bepos"], la_ - y["bbpos"],
                                                                          lb_ - y["aepos"], lb_ - y["abpos"])
        assert x["diffs"] == y["diffs"]
    # about half the wave work of the two-sided mode
    sym_cells = gpu_ctx.align_stats().wave_cells
    g, _ = both_opts(tspace=126, skip_self=1, min_len=500, max_la=64, max_cand=128)
    d = gpu_ctx.db(p)
    gpu_ctx.align_db(d, d, g)
    assert sym_cells < 0.6 * gpu_ctx.align_stats().wave_cells


def test_grouped_piles_never_cross_groups(gpu_ctx):
    p1, p2 = pile(31, n=12), pile(41, n=9)
    both = sim.SeqDb(np.concatenate([p1.bases, p2.bases]), np.concatenate([p1.off, p2.off[1:] + p1.off[-1]]),
                     group=np.asarray([0] * p1.n + [1] * p2.n, dtype=np.int32))
    las, trace = run_both(gpu_ctx, both, both, same=True, tspace=126, skip_self=1)
    grp = both.group
    assert np.all(grp[las["aread"]] == grp[las["bread"]])
    # property: the batched result is the union of the per-pile results
    g, _ = both_opts(tspace=126, skip_self=1)
    d1 = gpu_ctx.db(p1)
    l1, _ = gpu_ctx.align_db(d1, d1, g)
    d2 = gpu_ctx.db(p2)
    l2, _ = gpu_ctx.align_db(d2, d2, g)
    assert len(l1) + len(l2) == len(las)
    sel = las[grp[las["aread"]] == 0]
    for f in ("abpos", "aepos", "bbpos", "bepos", "diffs", "aread", "bread"):
        assert np.array_equal(sel[f], l1[f])


@pytest.mark.parametrize("k,mod", [(14, 1), (12, 1), (16, 2)])
def test_grouped_index_counted_in_lds_equals_the_atomic_passes(gpu_ctx, monkeypatch, k, mod):
    """The index of a grouped DB is built group by group with LDS counters (k_group_index); the generic passes with
    global atomics (DH_INDEX_ATOMICS=1) and the oracle give the same hits, candidates and records.  Group ids with no
    sequence (2) and a group of one short read are part of the input."""
    ps = [pile(51, n=10), pile(52, n=7), pile(53, n=14), pile(54, n=1)]
    gid = [0, 1, 3, 4]
    off = [np.zeros(1, dtype=np.int64)]
    for p in ps:
        off.append(p.off[1:] + off[-1][-1])
    both = sim.SeqDb(np.concatenate([p.bases for p in ps]), np.concatenate(off),
                     group=np.concatenate([np.full(p.n, g, dtype=np.int32) for p, g in zip(ps, gid)]))
    kw = dict(tspace=126, skip_self=2, min_len=500, max_la=64, max_cand=128, k=k, kmer_mod=mod)
    monkeypatch.setenv("DH_NO_JOIN", "1")         # the directory path (a grouped DB against itself takes the k-mer join otherwise)
    monkeypatch.setenv("DH_INDEX_LDS_MIN", "0")   # ... with the LDS-counted passes whatever the size of the DB
    las, trace = run_both(gpu_ctx, both, both, same=True, **kw)
    assert len(las) > 0 and np.all(both.group[las["aread"]] == both.group[las["bread"]])
    monkeypatch.setenv("DH_INDEX_ATOMICS", "1")
    las2, trace2 = run_both(gpu_ctx, both, both, same=True, **kw)
    assert_same_las((las2, trace2), (las, trace))


def grouped_piles(seeds=(51, 52, 53, 54), sizes=(10, 7, 14, 1), gid=(0, 1, 3, 4), extra=None):
    ps = [pile(sd, n=n) for sd, n in zip(seeds, sizes)]
    if extra is not None:
        ps.append(extra[0])
        gid = tuple(gid) + (extra[1],)
    off = [np.zeros(1, dtype=np.int64)]
    for p in ps:
        off.append(p.off[1:] + off[-1][-1])
    return sim.SeqDb(np.concatenate([p.bases for p in ps]), np.concatenate(off),
                     group=np.concatenate([np.full(p.n, g, dtype=np.int32) for p, g in zip(ps, gid)]))


@pytest.mark.parametrize("k,mod,ss", [(14, 1, 2), (12, 1, 2), (16, 2, 2), (14, 1, 1), (14, 3, 0), (9, 1, 2)])
def test_pile_up_kmer_join_equals_the_directory_lookups(gpu_ctx, monkeypatch, k, mod, ss):
    """A grouped DB against itself is seeded by the per-pile-up k-mer join (k_join_part / k_join: entries binned per
    group and slice, chained in an LDS table, hits written per B read) -- the same multiset of hits as the directory
    lookups (DH_NO_JOIN=1) and as the oracle: hit / candidate / cell counters and every record and trace value.
    skip_self 0 / 1 / 2, sampled k-mers, k = 16 (32-bit k-mers filled), empty group ids, a group of one read."""
    both = grouped_piles()
    kw = dict(tspace=126, skip_self=ss, min_len=500, max_la=64, max_cand=128, k=k, kmer_mod=mod)
    if k == 9:
        kw["tcap"] = 20  # short k-mers repeat inside a pile-up: the -t cap decides per orientation class
    las, trace = run_both(gpu_ctx, both, both, same=True, **kw)
    assert len(las) > 0 and np.all(both.group[las["aread"]] == both.group[las["bread"]])
    monkeypatch.setenv("DH_NO_JOIN", "1")
    las2, trace2 = run_both(gpu_ctx, both, both, same=True, **kw)
    assert_same_las((las2, trace2), (las, trace))


def test_kmer_join_reruns_with_a_larger_hit_buffer_and_masks_and_strands(gpu_ctx, monkeypatch):
    """The hit buffer is sized from an estimate; when it is too small the join reports the size it needs and runs
    again (DH_JOIN_HITCAP forces that).  Soft masks (DBdust's role) exclude k-mers on both sides; strands = 1 / 2
    restrict the hits to one orientation class."""
    both = grouped_piles()
    rng = np.random.default_rng(11)
    ptr, iv = [0], []
    for i in range(both.n):
        n, pos = both.length(i), 0
        while True:
            pos += int(rng.integers(100, 1500))
            ln = int(rng.integers(5, 200))
            if pos + ln >= n:
                break
            iv += [pos, pos + ln]
            pos += ln
        ptr.append(len(iv) // 2)
    both.mask = (np.asarray(ptr, dtype=np.int64), np.asarray(iv + [0, 0], dtype=np.int32))
    monkeypatch.setenv("DH_JOIN_HITCAP", "1000")
    for strands in (3, 1, 2):
        run_both(gpu_ctx, both, both, same=True, tspace=126, skip_self=2, min_len=500, max_la=64, max_cand=128,
                 strands=strands)


def test_kmer_join_falls_back_when_a_slice_overflows(gpu_ctx, monkeypatch, capfd):
    """A k-mer with thousands of copies in one pile-up (a satellite the low-complexity mask does not catch) lands in
    ONE slice and overflows its LDS table: the call takes the directory path, results as the oracle's."""
    rng = np.random.default_rng(3)
    unit = np.asarray([0, 1, 2, 3, 1, 0, 2, 2, 3, 0, 1], dtype=np.uint8)
    sat = np.tile(unit, 560)  # 6 160 bases of period 11: every one of its 11 k-mers 560 times per read, 4 480 per pile-up
    reads = []
    for i in range(8):
        r = rng.integers(0, 4, 9000).astype(np.uint8)
        r[1000:1000 + len(sat)] = sat
        reads.append(r)
    extra = sim.SeqDb.from_list(reads)
    both = grouped_piles(seeds=(61,), sizes=(6,), gid=(0,), extra=(extra, 1))
    monkeypatch.setenv("DH_TRACE", "1")
    run_both(gpu_ctx, both, both, same=True, tspace=126, skip_self=2, min_len=500, max_la=64, max_cand=128, tcap=64)
    assert "falling back to the k-mer directory" in capfd.readouterr().err


@pytest.mark.parametrize("nreads,rlen,copies,unit,cap", [(9, 5200, 1, 300, None), (3, 2600, 1, 300, None), (4, 5200, 1, 300, "2048"), (24, 5200, 1, 300, None), (64, 1400, 1, 300, None),
                                                         (9, 5200, 8, 250, None), (9, 5200, 20, 120, "2048")])
def test_hit_sort_by_diagonal_buckets_and_its_fall_backs(gpu_ctx, monkeypatch, capfd, nreads, rlen, copies, unit, cap):
    """The seed back end sorts a read's hits by diagonal buckets and ranks inside a bucket (LDS variants: through
    registers; HBM variant: through its slab); a bucket above its limit (256 / 2048 hits) sends the read through the bitonic
    network.  Reads that differ by substitutions only put ALL hits of an overlap on one diagonal: a few of them fit LDS
    with buckets beyond its limit (network); 64 short ones have more hits than LDS holds, in buckets the HBM variant still
    ranks; two dozen long ones or a tandem repeat in all of them overflow the HBM variant's buckets too -- records, trace values and the
    hit / candidate counters as the oracle's every time (pytest -s with a -DDH_SEED_PROF build prints the paths taken)."""
    rng = np.random.default_rng(17 + copies + nreads)
    u = rng.integers(0, 4, unit).astype(np.uint8)
    rep = np.concatenate([u if i == 0 else np.where(rng.random(unit) < 0.03, rng.integers(0, 4, unit), u).astype(np.uint8)
                          for i in range(copies)])
    base = rng.integers(0, 4, rlen + 800).astype(np.uint8)
    if copies > 1:
        base[1500:1500 + min(len(rep), 4000)] = rep[:4000]
    reads = []
    for i in range(nreads):
        a = int(rng.integers(0, 600))
        r = base[a:a + rlen].copy()
        e = rng.random(len(r)) < (0.07 if copies == 1 else 0.04)
        r[e] = rng.integers(0, 4, int(e.sum()))
        reads.append(r if i % 3 else (3 - r[::-1]).astype(np.uint8))
    both = grouped_piles(seeds=(71,), sizes=(5,), gid=(0,), extra=(sim.SeqDb.from_list(reads), 2))
    if cap:
        monkeypatch.setenv("DH_SEED_CAP", cap)
    monkeypatch.setenv("DH_TRACE", "1")
    las, _ = run_both(gpu_ctx, both, both, same=True, tspace=126, skip_self=2, min_len=500, max_la=256, max_cand=256, tcap=200)
    assert len(las) > 10
    err = capfd.readouterr().err
    print(err)
    if nreads >= 24 or copies > 1:
        # (the 16384-entry tier of the segment-fed back end takes these reads whole; without it they go on to the HBM variant)
        monkeypatch.setenv("DH_SEED_NO16K", "1")
        run_both(gpu_ctx, both, both, same=True, tspace=126, skip_self=2, min_len=500, max_la=256, max_cand=256, tcap=200)
        err = capfd.readouterr().err
        assert "overflow" in err  # reads with more hits than the LDS tiers hold went to the HBM variant


def test_edge_cases_empty_short_and_n_reads(gpu_ctx):
    g = sim.genome(5, 30000)
    rd, _ = sim.reads(6, g, 20, 3000)
    seqs = [rd.seq(i).copy() for i in range(rd.n)]
    seqs[3] = seqs[3][:9]                      # shorter than k
    seqs[5] = np.zeros(0, dtype=np.uint8)      # empty
    seqs[7][100:140] = 4                       # run of N
    seqs[8] = np.full(500, 4, dtype=np.uint8)  # all N
    B = sim.SeqDb.from_list(seqs)
    A = sim.SeqDb.from_list([g[:14000], g[14100:], np.zeros(0, dtype=np.uint8), g[:5]])
    las, _ = run_both(gpu_ctx, A, B)
    assert 3 not in las["bread"] and 5 not in las["bread"] and 8 not in las["bread"]


def test_no_alignments_at_all(gpu_ctx):
    A = sim.SeqDb.from_list([sim.genome(1, 20000)])
    B = sim.SeqDb.from_list([sim.genome(2, 3000), sim.genome(3, 3000)])
    las, trace = run_both(gpu_ctx, A, B)
    assert len(las) == 0 and len(trace) == 0


def test_seed_on_trace_boundary_and_sequence_ends(gpu_ctx):
    """Reads that start exactly on a multiple of tspace and reach both contig ends."""
    g = sim.genome(77, 6000)
    A = sim.SeqDb.from_list([g])
    B = sim.SeqDb.from_list([g[0:3000].copy(), g[3000:6000].copy(), g[1400:4400].copy(), sim.revcomp(g[200:5800])])
    las, trace = run_both(gpu_ctx, A, B, min_len=100)
    assert len(las) == 4 and int(las["diffs"].sum()) == 0
    full = las[las["bread"] == 3][0]
    assert (full["abpos"], full["aepos"], full["flags"] & 1) == (200, 5800, 1)


def test_chunked_launches_give_identical_results(gpu_ctx, monkeypatch):
    w = sim.Workload(200_000, 2, 300, 3000, seed=19, spacing=15000)
    g, _ = both_opts()
    A, B = gpu_ctx.db(w.contigs), gpu_ctx.db(w.reads)
    one = gpu_ctx.align_db(A, B, g)
    monkeypatch.setenv("DH_ALIGN_CHUNK", "64")
    many = gpu_ctx.align_db(A, B, g)
    assert gpu_ctx.align_stats().wave_launches > 1
    assert_same_las(many, one)


@pytest.mark.parametrize("sym", [False, True])
def test_wide_windows_use_one_alignment_per_wavefront(gpu_ctx, sym):
    """width <= 14 runs four alignments per wavefront (k_wave2 with 16-lane groups), width <= 30 two
    (32-lane groups), wider windows one (k_wave): all against the oracle at their own width."""
    w = sim.Workload(200_000, 2, 300, 4000, seed=37, spacing=15000)
    if sym:
        sub = sim.SeqDb.from_list([w.reads.seq(i) for i in range(80)])
        for width in (62, 30, 14, 12, 5):
            run_both(gpu_ctx, sub, sub, same=True, skip_self=2, tspace=126, max_la=64, max_cand=128, width=width)
    else:
        for width in (62, 30, 14, 12, 5):
            run_both(gpu_ctx, w.contigs, w.reads, width=width)


@pytest.mark.parametrize("sym", [False, True])
def test_packed_and_byte_wave_paths_agree(gpu_ctx, monkeypatch, sym):
    """ACGT-only DBs are aligned from 2-bit packed copies (32 bases per load), DBs with other codes
    from the byte arrays; both instantiations of the wave kernel must give the same bits."""
    w = sim.Workload(200_000, 2, 400, 3000, seed=29, spacing=15000)
    if sym:
        g, _ = both_opts(skip_self=2, tspace=126, max_la=64, max_cand=128)
        sub = sim.SeqDb.from_list([w.reads.seq(i) for i in range(60)])
        A = B = gpu_ctx.db(sub)
    else:
        g, _ = both_opts()
        A, B = gpu_ctx.db(w.contigs), gpu_ctx.db(w.reads)
    packed = gpu_ctx.align_db(A, B, g)
    monkeypatch.setenv("DH_WAVE_BYTES", "1")
    plain = gpu_ctx.align_db(A, B, g)
    assert len(packed[0]) > 0
    assert_same_las(plain, packed)


def test_select_best_flags(gpu_ctx):
    w = sim.Workload(200_000, 2, 200, 4000, seed=23, spacing=15000)
    g, _ = both_opts()
    A, B = gpu_ctx.db(w.contigs), gpu_ctx.db(w.reads)
    las, _ = gpu_ctx.align_db(A, B, g, select_best=True)
    assert np.all(las["flags"] & 0x4) and np.any(las["flags"] & 0x10)


def test_larger_run_properties(gpu_ctx):
    """At a size the oracle would take long for: size-independent properties only."""
    w = sim.Workload(3_000_000, 20, 20000, 8000, seed=101)
    g, _ = both_opts()
    A, B = gpu_ctx.db(w.contigs), gpu_ctx.db(w.reads)
    las, trace = gpu_ctx.align_db(A, B, g)
    check_trace_invariants(las[:: max(1, len(las) // 2000)], trace, 100)
    assert len(set(las["bread"].tolist())) >= 0.995 * w.reads.n
    s = w.read_truth[las["bread"], 0]
    e = w.read_truth[las["bread"], 1]
    cs = w.contig_start[las["aread"]]
    ok = ((las["flags"] & 1) == w.read_truth[las["bread"], 2]) & (cs + las["abpos"] >= s - 80) & (cs + las["aepos"] <= e + 80)
    assert ok.mean() > 0.999
    # idempotence: a second run (index rebuilt) is bit-identical
    A.drop_cache()
    las2, trace2 = gpu_ctx.align_db(A, B, g)
    assert_same_las((las2, trace2), (las, trace))


@pytest.mark.parametrize("kw", [dict(k=12), dict(k=16, kmer_mod=3), dict(k=10, hmin=50), dict(band_shift=5),
                                dict(strands=1), dict(strands=2), dict(tspace=64, min_len=300), dict(pen=4, xdrop=60),
                                dict(max_cand=2, max_la=1), dict(tcap=1), dict(dmax=200)])
def test_option_sweep(gpu_ctx, kw):
    """Every option of dh_align_opts away from its default (short k-mers flood the hit buffers, odd
    modimer moduli take the rotate-free divisibility test, ...): still bit-exact."""
    w = sim.Workload(150_000, 2, 250, 3000, seed=61, spacing=15000)
    run_both(gpu_ctx, w.contigs, w.reads, **kw)


def test_per_item_overflow_is_reported_not_fatal(gpu_ctx):
    """More overlaps of one read and strand than max_la slots (symmetric mode): the excess pairs are
    dropped, the items are counted in overflow_items and the call succeeds (one read must not abort
    a batch -- the reference skips a failing pile-up and carries on, processPileUps/package.d:319-363).
    Everything that is returned is an alignment the oracle (with room for all) finds too."""
    p = pile(41, n=40)
    g, o = both_opts(skip_self=2, tspace=126, max_la=8, max_cand=128)
    o.max_la = 64
    exp = oz.align_db(p, p, o, nthreads=os.cpu_count() or 1)
    d = gpu_ctx.db(p)
    las, trace = gpu_ctx.align_db(d, d, g)
    st = gpu_ctx.align_stats()
    assert st.overflow_items > 0 and 0 < len(las) < len(exp[0])
    want = set(la_rows(*exp[:2]))
    assert all(r in want for r in la_rows(las, trace))
    per_item = np.bincount(las["aread"] * 2 + (las["flags"] & 1))
    assert per_item.max() <= 8


@pytest.mark.parametrize("err,width,xdrop", [(0.10, 30, 120), (0.15, 14, 60)])
def test_ont_like_error_profile_of_configs4(gpu_ctx, err, width, xdrop):
    """BASELINE configs[4] reads: 20 kb, deletion-biased ONT-like errors (ins .3 / del .4 / sub .3,
    SURVEY 8(d)).  The narrow windows do not break on this profile: every read maps end to end, and the
    HIP path equals the oracle bit for bit."""
    g = sim.genome(97, 600_000)
    gb, ge = sim.gaps(98, len(g), 3, 50, 3000, 20000)
    contigs, _ = sim.contigs_from_gaps(g, gb, ge)
    reads, truth = sim.reads(99, g, 240, 20000, 0, err=err, p_ins=0.30, p_del=0.40)
    las, _ = run_both(gpu_ctx, contigs, reads, k=20, kmer_mod=4, width=width, xdrop=xdrop)
    assert len(set(las["bread"].tolist())) == reads.n
    covered = np.zeros(reads.n, dtype=np.int64)
    np.add.at(covered, las["bread"], las["bepos"] - las["bbpos"])
    lens = np.diff(reads.off)
    assert np.mean(covered >= 0.9 * lens) > 0.93   # reads across a gap lose the gap itself


@pytest.mark.gpu
def test_map_reads_equals_mapping_followed_by_the_collect_filters(gpu_ctx, monkeypatch):
    """dh_map_reads filters the records of every finished chunk on a host thread while the device maps
    the next chunk; records, flags, traces and per-stage counts equal dh_align_db(select_best) followed
    by dh_collect_filter, however the reads are chunked."""
    w = sim.Workload(1_500_000, 12, 6000, 8000, seed=31)
    A, B = gpu_ctx.db(w.contigs), gpu_ctx.db(w.reads)
    mo = dentist_amd.default_align_opts(kmer_mod=4, k=20, width=14, xdrop=60)
    po = dentist_amd.default_process_opts()
    las, trace = gpu_ctx.align_db(A, B, mo, select_best=True)
    exp, exp_dropped, _ = dentist_amd.collect_filter(las, w.contigs.off, w.reads.off, po)
    assert exp_dropped.sum() > 0
    for chunk in ("512", "3000", None):
        if chunk is None:
            monkeypatch.delenv("DH_ALIGN_CHUNK", raising=False)
        else:
            monkeypatch.setenv("DH_ALIGN_CHUNK", chunk)
        got, gtrace, dropped = gpu_ctx.map_reads(A, B, mo, po)
        assert np.array_equal(dropped, exp_dropped)
        assert np.array_equal(got, exp) and np.array_equal(gtrace, trace)
    # a block of the reads: ids of the whole DB, filters see the whole DB's read lengths
    monkeypatch.setenv("DH_ALIGN_CHUNK", "1024")
    first, count = 1500, 2500
    got, gtrace, dropped = gpu_ctx.map_reads(A, B, mo, po, first=first, count=count)
    sel = (exp["bread"] >= first) & (exp["bread"] < first + count)
    for f in ("aread", "bread", "abpos", "aepos", "bbpos", "bepos", "diffs", "flags", "tlen"):
        assert np.array_equal(got[f], exp[f][sel]), f
