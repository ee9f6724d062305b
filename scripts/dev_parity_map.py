import sys, time
sys.path.insert(0, '.')
import numpy as np
import dentist_amd
from dentist_amd import sim
from oracle import pyoracle as oz
sys.path.insert(0, 'tests')
from helpers import *
w = sim.Workload(400_000, 4, 800, 5000, seed=7, spacing=20000)
oo = oz.default_opts(width=62)
t=time.time(); exp = oz.align_db(w.contigs, w.reads, oo, nthreads=8); print("oracle s", time.time()-t, exp[2])
ctx = dentist_amd.Context(0)
A = ctx.db(w.contigs); B = ctx.db(w.reads)
o = dentist_amd.default_align_opts()
t=time.time(); got = ctx.align_db(A, B, o); print("gpu s", time.time()-t)
print(ctx.align_stats().as_dict())
print(len(got[0]), len(exp[0]))
assert_same_las(got, exp[:2])
print("PARITY OK")
for i in range(3):
    A.drop_cache(); t=time.time(); got = ctx.align_db(A, B, o); print("gpu s", time.time()-t, ctx.align_stats().as_dict())
