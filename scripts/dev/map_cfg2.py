"""dev: mapping + collect of configs[2], host timeline (DH_TRACE=1)."""
import sys, time, os
sys.path.insert(0, '.')
import numpy as np, dentist_amd
from dentist_amd import sim
import bench
w = sim.Workload(seed=20260929, **bench.WORKLOADS["cfg2_100Mb_1000gaps_1Mx15kb"])
ctx = dentist_amd.Context(0)
A, B = ctx.db(w.contigs), ctx.db(w.reads)
mo = dentist_amd.default_align_opts(kmer_mod=4, k=20, width=14, xdrop=60)
popts = dentist_amd.default_process_opts()
for rep in range(3):
    A.drop_cache(); B.drop_cache()
    t0 = time.perf_counter()
    las, tr = ctx.align_db(A, B, mo, select_best=True)
    t1 = time.perf_counter()
    las2, dropped, _ = dentist_amd.collect_filter(las, w.contigs.off, w.reads.off, popts, inplace=True)
    t2 = time.perf_counter()
    piles = dentist_amd.Pileups(las2, w.contigs.off, popts)
    t3 = time.perf_counter()
    print('map %.1f filter %.1f collect %.1f ms; las %d trace %d' % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, len(las), len(tr)), flush=True)
