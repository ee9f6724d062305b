"""dev: many steps in one process -- results stay bit-identical, memory does not grow."""
import os, sys, time, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dentist_amd
from dentist_amd import sim
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
w = sim.Workload(10_000_000, 100, 100_000, 10_000, seed=20260929)
ctx = dentist_amd.Context(0)
A, B = ctx.db(w.contigs), ctx.db(w.reads)
mo = dentist_amd.default_align_opts(kmer_mod=4); po = dentist_amd.default_process_opts()
ref = None
for it in range(n):
    A.drop_cache(); B.drop_cache()
    t = time.time()
    las, trace = ctx.align_db(A, B, mo, select_best=True)
    piles = dentist_amd.Pileups(las, w.contigs.off, po)
    rec, bases = dentist_amd.process_pileups(ctx, A, B, las, trace, piles, po)
    dt = time.time() - t
    sig = (len(las), int(las["abpos"].sum()), int(trace.astype(np.int64).sum()), rec.tobytes(), bases.tobytes())
    if ref is None: ref = sig
    assert sig == ref, f"step {it}: results differ"
    if it % 10 == 0 or it == n - 1:
        free, total = torch.cuda.mem_get_info()
        print(f"step {it}: {dt*1e3:.1f} ms, HBM used {(total-free)/2**30:.2f} GiB, host RSS {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss/2**20:.2f} GiB", flush=True)
print("soak ok")
