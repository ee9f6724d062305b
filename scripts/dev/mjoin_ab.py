"""dev: mapping seeds of configs[2], partitioned join against directory lookups, per sampling rate: python scripts/dev/mjoin_ab.py 1 2 4 8"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, dentist_amd
from dentist_amd import sim
w = sim.Workload(seed=20260929, genome_len=100_000_000, ngaps=1000, nreads=1_000_000, read_len=15_000)
ctx = dentist_amd.Context(0)
A, B = ctx.db(w.contigs), ctx.db(w.reads)
po = dentist_amd.default_process_opts(algo=1)
os.environ["DH_MJOIN_MIN"] = "0"
for mod in [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8]:
    mo = dentist_amd.default_align_opts(kmer_mod=mod, k=20, width=64, xdrop=60, algo=1)
    for path in ("join", "directory"):
        os.environ.pop("DH_NO_MJOIN", None)
        if path == "directory":
            os.environ["DH_NO_MJOIN"] = "1"
        best = None
        for rep in range(3):
            A.drop_cache(); B.drop_cache()
            m = ctx.map_reads(A, B, mo, po, sorted=False, candidates=False)
            st = ctx.align_stats().as_dict()
            best = st if best is None or st["ms_seed"] < best["ms_seed"] else best
            n = len(m[0]); del m
        print(f"kmer_mod {mod} {path:9s}: seeds {best['ms_seed']:.1f} ms, index {best['ms_index']:.1f}, tiles {best['ms_wave']:.1f}, records {n}, mjoin {ctx.mjoin_counts(reset=True)}", flush=True)
