"""dev: the mapping pass of configs[2] with the band of 64 / 32 rows: python scripts/dev/map_band.py <band> [reps] [kmer_mod]"""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, dentist_amd
from dentist_amd import sim
band = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
mod = int(sys.argv[3]) if len(sys.argv) > 3 else 8
w = sim.Workload(seed=20260929, genome_len=100_000_000, ngaps=1000, nreads=1_000_000, read_len=15_000)
ctx = dentist_amd.Context(0)
A, B = ctx.db(w.contigs), ctx.db(w.reads)
mo = dentist_amd.default_align_opts(kmer_mod=mod, k=20, width=band, xdrop=60, algo=1)
po = dentist_amd.default_process_opts(algo=1)
for rep in range(reps):
    A.drop_cache(); B.drop_cache()
    t = time.perf_counter()
    m = ctx.map_reads(A, B, mo, po, sorted=False, candidates=False)
    dt = time.perf_counter() - t
    st = ctx.align_stats().as_dict()
    print('band', band, 'las', len(m[0]), 'wall %.1f ms' % (dt * 1e3), {k: (round(v, 1) if isinstance(v, float) else v) for k, v in st.items()}, flush=True)
    del m
