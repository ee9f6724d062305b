import os, sys, time
sys.path.insert(0, ".")
import dentist_amd
from dentist_amd import sim
w = sim.Workload(100_000_000, 1000, 1_000_000, 15_000, seed=20260929, read_range=(0, 300_000))
ctx = dentist_amd.Context(0)
A, B = ctx.db(w.contigs), ctx.db(w.reads)
mo = dentist_amd.default_align_opts(kmer_mod=int(os.environ.get("KMER_MOD", "8")), k=20, width=64, xdrop=60, algo=1)
po = dentist_amd.default_process_opts(algo=1)
for rep in range(2):
    las, trace, dropped, cands = ctx.map_reads(A, B, mo, po, sorted=False, candidates=True)
    st = ctx.align_stats()
    print("seeds %.1f ms tiles %.1f" % (st.ms_seed, st.ms_wave), flush=True)
