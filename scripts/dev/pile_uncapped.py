"""dev: the process stage of configs[2] at the reference's behaviour (no read cap: 166 reads per pile-up), parts one after
the other, timings from the stats.  Usage: python scripts/dev/pile_uncapped.py [reps]; DH_DEV_LIB / DH_TRACE / knobs apply."""
import os, sys, time
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
tr = os.environ.pop("DH_TRACE", None)
las, trace, dropped = ctx.map_reads(A, B, mo, po, sorted=False, candidates=False)[:3]
gp, _ = dentist_amd.scaffold_spanning_pileups(las, w.contigs.off, w.reads.off, gaps, with_extensions=True, min_spanning_reads=po.min_reads)
piles = gp.select(las, po)
if not os.environ.get("DH_PARTS"):
    os.environ["DH_PROCESS_SERIAL"] = "1"
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    if tr and rep > 0:
        os.environ["DH_TRACE"] = tr
    t0 = time.perf_counter()
    rec, bases = dentist_amd.process_pileups(ctx, A, B, las, trace, piles, po)
    print("process %.1f ms" % ((time.perf_counter() - t0) * 1e3), {k: round(v, 1) for k, v in dentist_amd.process_stats(ctx).items() if k.startswith("ms_")},
          "closed", int((rec["status"] == 0).sum()), flush=True)
import hashlib
print("result md5", hashlib.md5(rec.tobytes() + bases.tobytes()).hexdigest())
