"""dev: k_wave2 (two alignments per wavefront, width <= 30) vs the oracle and vs k_wave."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import dentist_amd
from dentist_amd import sim
from helpers import assert_same_las, check_trace_invariants
from oracle import pyoracle as oz

F = ("k", "hmin", "band_shift", "tspace", "min_len", "pen", "xdrop", "max_err_ppm", "max_cand",
     "max_la", "tcap", "strands", "skip_self", "dmax", "width", "kmer_mod")
ctx = dentist_amd.Context(0)

def both(**kw):
    g = dentist_amd.default_align_opts(**kw)
    o = oz.default_opts()
    for f in F:
        setattr(o, f, getattr(g, f))
    return g, o

def case(name, A, B, same=False, **kw):
    g, o = both(width=30, **kw)
    exp = oz.align_db(A, B, o, nthreads=os.cpu_count() or 1)
    dA = ctx.db(A)
    dB = dA if same else ctx.db(B)
    os.environ.pop("DH_WAVE_SINGLE", None)
    t = time.time(); got = ctx.align_db(dA, dB, g); t = time.time() - t
    st = ctx.align_stats()
    assert (st.hits, st.cands, st.alignments, st.wave_cells) == tuple(int(x) for x in exp[2]), (st.as_dict(), exp[2])
    assert_same_las(got, exp[:2])
    check_trace_invariants(got[0], got[1], g.tspace)
    os.environ["DH_WAVE_SINGLE"] = "1"
    one = ctx.align_db(dA, dB, g)
    os.environ.pop("DH_WAVE_SINGLE", None)
    assert_same_las(one, exp[:2])
    print(f"{name}: ok, {len(got[0])} LAs, wave {st.ms_wave:.2f} ms")

w = sim.Workload(300_000, 3, 600, 5000, seed=31, spacing=20000)
case("map", w.contigs, w.reads)
case("map-mod4", w.contigs, w.reads, kmer_mod=4)
sub = sim.SeqDb.from_list([w.reads.seq(i) for i in range(120)])
case("pile-sym", sub, sub, same=True, skip_self=2, tspace=126, max_la=64, max_cand=128)
case("pile-skip1", sub, sub, same=True, skip_self=1, tspace=126)
# N runs -> byte path
g = sim.genome(7, 60_000)
g[1000:1100] = 4
A = sim.SeqDb.from_list([g[:30000].copy(), g[30000:].copy()])
rd, _ = sim.reads(9, g, 200, 3000, 0, min_len=500)
case("with-N", A, rd)
print("all ok")
