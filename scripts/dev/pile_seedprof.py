"""dev: phase split of the seed kernel in the process stage (build with -DDH_SEED_PROF, run with DH_TRACE=1)."""
import sys
sys.path.insert(0, ".")
import dentist_amd
from dentist_amd import sim
w = sim.Workload(100_000_000, 1000, 1_000_000, 15_000, seed=20260929, read_range=(0, 1_000_000))
ctx = dentist_amd.Context(0)
A, B = ctx.db(w.contigs), ctx.db(w.reads)
mo = dentist_amd.default_align_opts(kmer_mod=4, k=20, width=64, xdrop=60, algo=1)
po = dentist_amd.default_process_opts(algo=1)
las, trace, dropped, cands = ctx.map_reads(A, B, mo, po, sorted=False, candidates=True)
piles = cands.select(las, po)
print("PROCESS", flush=True)
rec, bases = dentist_amd.process_pileups(ctx, A, B, las, trace, piles, po)
