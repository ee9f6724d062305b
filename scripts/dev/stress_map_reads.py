"""dev: stress the chunk hooks / copy stream of dh_map_reads: many small chunks, repeated, results compared."""
import os, sys
sys.path.insert(0, '.')
import numpy as np, dentist_amd
from dentist_amd import sim
w = sim.Workload(3_000_000, 20, 20000, 8000, seed=41)
ctx = dentist_amd.Context(0)
A, B = ctx.db(w.contigs), ctx.db(w.reads)
mo = dentist_amd.default_align_opts(kmer_mod=4, k=20, width=14, xdrop=60)
po = dentist_amd.default_process_opts()
las, trace = ctx.align_db(A, B, mo, select_best=True)
exp, expd, _ = dentist_amd.collect_filter(las, w.contigs.off, w.reads.off, po)
exp, trace = exp.copy(), trace.copy()
bad = 0
for it in range(40):
    os.environ["DH_ALIGN_CHUNK"] = str([256, 1000, 4096, 10000, 40000][it % 5])
    got, gt, d = ctx.map_reads(A, B, mo, po)
    ok = np.array_equal(got, exp) and np.array_equal(gt, trace) and np.array_equal(d, expd)
    bad += 0 if ok else 1
    del got, gt
print('iterations 40, mismatches', bad)
