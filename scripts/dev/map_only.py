"""dev: mapping of configs[1] at given (width, xdrop) pairs, kernel times only."""
import sys
sys.path.insert(0, '.')
import numpy as np, dentist_amd
from dentist_amd import sim
w = sim.Workload(10_000_000, 100, 100_000, 10_000, seed=20260929)
ctx = dentist_amd.Context(0)
A, B = ctx.db(w.contigs), ctx.db(w.reads)
for arg in sys.argv[1:]:
    wd, xd = (int(x) for x in arg.split(':'))
    mo = dentist_amd.default_align_opts(kmer_mod=4, k=20, width=wd, xdrop=xd)
    for rep in range(3):
        las, tr = ctx.align_db(A, B, mo, select_best=True)
    st = ctx.align_stats().as_dict()
    print('width', wd, 'xdrop', xd, 'las', len(las), 'aligned', int((las['aepos']-las['abpos']).sum()), 'wave ms %.2f' % st['ms_wave'], 'cells', st['wave_cells'], flush=True)
