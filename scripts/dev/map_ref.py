"""dev: the mapping pass of configs[2] at the reference's behaviour (kmer_mod 1 unless DH_KMER_MOD), stats of the last of
[reps] calls.  Usage: python scripts/dev/map_ref.py [reps]; DH_DEV_LIB / DH_TRACE / knobs apply."""
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
mo = dentist_amd.default_align_opts(kmer_mod=int(os.environ.get("DH_KMER_MOD", "1")), k=20, width=64, xdrop=60, algo=1)
po = dentist_amd.default_process_opts(algo=1)
tr = os.environ.pop("DH_TRACE", None)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for rep in range(reps):
    if tr and rep == reps - 1:
        os.environ["DH_TRACE"] = tr
    A.drop_cache(); B.drop_cache()
    t0 = time.perf_counter()
    las, trace, dropped = ctx.map_reads(A, B, mo, po, sorted=False, candidates=False)[:3]
    dt = (time.perf_counter() - t0) * 1e3
    st = ctx.align_stats()
    print("map %.1f ms: index %.1f seed %.1f wave %.1f; %d las, md5 %s" % (dt, st.ms_index, st.ms_seed, st.ms_wave, len(las),
          hashlib.md5(las.tobytes()).hexdigest()[:8]), "mjoin", ctx.mjoin_counts(reset=True), flush=True)
