"""dev: the symmetric all-vs-all of a pile-up-like grouped DB (NG groups x 60 reads x 2.8 kb @13 %): k-mer join against
the directory path (DH_NO_JOIN=1), timings from DH_TRACE=1."""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
import dentist_amd
from dentist_amd import sim
NG = int(sys.argv[1]) if len(sys.argv) > 1 else 500
seqs, grp = [], []
for g in range(NG):
    region = sim.genome(1000 + g, 3400)
    r, _ = sim.reads(5000 + g, region, 60, 2840)
    for i in range(r.n):
        seqs.append(r.seq(i)); grp.append(g)
db = sim.SeqDb.from_list(seqs)
db = sim.SeqDb(db.bases, db.off, group=np.asarray(grp, dtype=np.int32))
ctx = dentist_amd.Context(0)
D = ctx.db(db)
o = dentist_amd.default_align_opts(algo=1, width=64, tspace=126, skip_self=2, max_la=64, max_cand=128, min_len=500)
for rep in range(3):
    D.drop_cache()
    las, tr = ctx.align_db(D, D, o)
    st = ctx.align_stats()
    print(f"rep {rep}: index {st.ms_index:.1f} seeds {st.ms_seed:.1f} tiles {st.ms_wave:.1f} gather {st.ms_gather:.1f} ms; hits {st.hits} las {len(las)}", flush=True)
