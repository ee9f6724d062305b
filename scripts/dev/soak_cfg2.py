"""dev: N whole steps of configs[2]; every step must give byte-identical records and consensus bases."""
import sys, time, hashlib
sys.path.insert(0, '.')
import numpy as np, dentist_amd
from dentist_amd import sim
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
w = sim.Workload(seed=20260929, **bench.WORKLOADS["cfg2_100Mb_1000gaps_1Mx15kb"])
ctx = dentist_amd.Context(0)
A, B = ctx.db(w.contigs), ctx.db(w.reads)
mo = dentist_amd.default_align_opts(kmer_mod=4, k=20, width=14, xdrop=60)
po = dentist_amd.default_process_opts()
sigs = []
for it in range(n):
    A.drop_cache(); B.drop_cache()
    t0 = time.perf_counter()
    las, trace, dropped, cands = ctx.map_reads(A, B, mo, po, sorted=False, candidates=True)
    piles = cands.select(las, po)
    rec, bases = dentist_amd.process_pileups(ctx, A, B, las, trace, piles, po)
    dt = time.perf_counter() - t0
    h = hashlib.md5()
    h.update(np.ascontiguousarray(las).tobytes()); h.update(np.ascontiguousarray(trace).tobytes())
    h.update(np.ascontiguousarray(rec).tobytes()); h.update(np.ascontiguousarray(bases).tobytes())
    sigs.append(h.hexdigest())
    print(it, '%.1f ms' % (dt * 1e3), 'closed', int((rec["status"] == 0).sum()), sigs[-1][:12], flush=True)
    del las, trace, rec, bases, cands, piles
print('distinct signatures', len(set(sigs)))
