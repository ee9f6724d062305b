"""dev: host-glue cost of the sharded collect + process at configs[2] size (world = 1: collectives are no-ops)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, dentist_amd, cProfile, pstats
from dentist_amd import sim, parallel
w = sim.Workload(100_000_000, 1000, 1_000_000, 15_000, seed=20260929)
ctx = dentist_amd.Context(0)
A, B = ctx.db(w.contigs), ctx.db(w.reads)
mo = dentist_amd.default_align_opts(kmer_mod=4, k=20, width=14, xdrop=60)
po = dentist_amd.default_process_opts()
las, trace = ctx.align_db(A, B, mo, select_best=True)
for rep in range(2):
    t0 = time.perf_counter()
    piles = dentist_amd.Pileups(las, w.contigs.off, po)
    rec, bases = dentist_amd.process_pileups(ctx, A, B, las, trace, piles, po)
    t1 = time.perf_counter()
    pr = cProfile.Profile(); pr.enable()
    rec2, bases2, info = parallel.sharded_process(ctx, A, B, 0, w.contigs.off, las, trace, po, 0, 1)
    pr.disable()
    t2 = time.perf_counter()
    print('direct %.1f ms  sharded %.1f ms  kernels %.1f' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, dentist_amd.process_stats(ctx)['ms_total']), np.array_equal(rec2['ins_end'], rec['ins_end']), flush=True)
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
