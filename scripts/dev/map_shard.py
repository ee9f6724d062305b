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
mo = dentist_amd.default_align_opts(kmer_mod=8, k=20, width=64, xdrop=60, algo=1)
po = dentist_amd.default_process_opts(algo=1)
for mode in os.environ.get("SHARD_MAP_MODES", "default 65536 default 65536 32768").split():
    if mode == "default":
        os.environ.pop("DH_ALIGN_CHUNK", None)
    else:
        os.environ["DH_ALIGN_CHUNK"] = mode
    ts = []
    for it in range(4):
        t0 = time.perf_counter()
        las, tr, d = ctx.map_reads(A, B, mo, po)
        ts.append((time.perf_counter() - t0) * 1e3)
        del las, tr
    print(mode, ['%.1f' % t for t in ts], flush=True)
