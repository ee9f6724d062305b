import sys, os
sys.path.insert(0, '.')
import numpy as np, dentist_amd
from dentist_amd import sim
from oracle import pyoracle as oz
w = sim.Workload(100_000, 1, int(sys.argv[1]), 3000, seed=37, spacing=15000)
ctx = dentist_amd.Context(0)
A, B = ctx.db(w.contigs), ctx.db(w.reads)
mo = dentist_amd.default_align_opts(width=int(sys.argv[2]))
try:
    las, tr = ctx.align_db(A, B, mo)
except Exception as e:
    print('error', e); sys.exit(0)
o = oz.default_opts(width=int(sys.argv[2]))
el, et, _ = oz.align_db(w.contigs, w.reads, o, nthreads=8)
print('n', sys.argv[1], 'width', sys.argv[2], 'las', len(las), len(el), 'same', len(las)==len(el) and all(np.array_equal(las[f], el[f]) for f in ('abpos','aepos','bbpos','bepos','diffs')), flush=True)
