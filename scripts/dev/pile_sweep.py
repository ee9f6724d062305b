"""dev: the uncapped process stage of configs[2] under variants of the symmetric k_tile launch (environment knobs are read per
call).  Usage: python scripts/dev/pile_sweep.py "K1=V1,K2=V2" "K1=V3" ...  (each argument one variant; '-' = defaults)"""
import os, sys, time, hashlib
sys.path.insert(0, ".")
import numpy as np
import dentist_amd
from dentist_amd import sim
import bench
spec = bench.WORKLOADS[os.environ.get("DH_WORKLOAD", "cfg2_100Mb_1000gaps_1Mx15kb")]
w = sim.Workload(seed=20260929, **spec)
ctx = dentist_amd.Context(0)
A, B = ctx.db(w.contigs), ctx.db(w.reads)
mo = dentist_amd.default_align_opts(kmer_mod=8, k=20, width=64, xdrop=60, algo=1)
po = dentist_amd.default_process_opts(algo=1)
po.max_reads = int(os.environ.get("DH_MAX_READS", "0"))
gaps = np.stack([np.arange(w.contigs.n - 1), np.arange(1, w.contigs.n)], axis=1).astype(np.int32)
las, trace, dropped = ctx.map_reads(A, B, mo, po, sorted=False, candidates=False)[:3]
gp, _ = dentist_amd.scaffold_spanning_pileups(las, w.contigs.off, w.reads.off, gaps, with_extensions=True, min_spanning_reads=po.min_reads)
piles = gp.select(las, po)
if not os.environ.get("DH_PARTS"):
    os.environ["DH_PROCESS_SERIAL"] = "1"
dentist_amd.process_pileups(ctx, A, B, las, trace, piles, po)   # warm-up (allocations)
for var in sys.argv[1:] or ["-"]:
    kv = dict(x.split("=") for x in var.split(",")) if var != "-" else {}
    os.environ.update(kv)
    best = None
    for rep in range(2):
        t0 = time.perf_counter()
        rec, bases = dentist_amd.process_pileups(ctx, A, B, las, trace, piles, po)
        dt = (time.perf_counter() - t0) * 1e3
        st = {k: round(v, 1) for k, v in dentist_amd.process_stats(ctx).items() if k.startswith("ms_")}
        if best is None or dt < best[0]:
            best = (dt, st)
    cum = ctx.cum_stats().as_dict()
    print("%-50s process %.1f ms %s md5 %s" % (var, best[0], best[1], hashlib.md5(rec.tobytes() + bases.tobytes()).hexdigest()[:8]), flush=True)
    for k in kv:
        del os.environ[k]
