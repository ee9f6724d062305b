"""dev: mapping of one 1/8 shard of configs[2] (125 000 reads) in steady state, default chunking vs one chunk."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np, dentist_amd
from dentist_amd import sim
import bench
spec = bench.WORKLOADS["cfg2_100Mb_1000gaps_1Mx15kb"]
w = sim.Workload(seed=20260929, read_range=(0, 125000), **spec)
ctx = dentist_amd.Context(0)
A, B = ctx.db(w.contigs), ctx.db(w.reads)
mo = dentist_amd.default_align_opts(kmer_mod=4, k=20, width=14, xdrop=60)
po = dentist_amd.default_process_opts()
for mode in ("default", "262144", "default", "262144"):
    if mode == "default":
        os.environ.pop("DH_ALIGN_CHUNK", None)
    else:
        os.environ["DH_ALIGN_CHUNK"] = mode
    ts = []
    for it in range(4):
        A.drop_cache(); B.drop_cache()
        t0 = time.perf_counter()
        las, tr, d, c = ctx.map_reads(A, B, mo, po, sorted=False, candidates=True)
        ts.append((time.perf_counter() - t0) * 1e3)
        del las, tr, c
    print(mode, ['%.1f' % t for t in ts], flush=True)
