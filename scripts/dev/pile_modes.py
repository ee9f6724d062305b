"""dev: k_tile on a pile-up-like DB (1000 groups x 60 reads x 2.6 kb @13 %) in symmetric mode and as a plain
all-vs-all (every pair aligned from both sides, no claimed slots): what the symmetric machinery costs."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import dentist_amd
from dentist_amd import sim
seqs, grp = [], []
NG = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
t0 = time.time()
for g in range(NG):
    region = sim.genome(1000 + g, 3200)
    r, _ = sim.reads(5000 + g, region, 60, 2600)
    for i in range(r.n):
        seqs.append(r.seq(i)); grp.append(g)
db = sim.SeqDb.from_list(seqs)
db = sim.SeqDb(db.bases, db.off, group=np.asarray(grp, dtype=np.int32))
print('generated', NG, 'groups in %.1f s' % (time.time() - t0), flush=True)
ctx = dentist_amd.Context(0)
D = ctx.db(db)
for ss in (2, 1, 2, 1):
    o = dentist_amd.default_align_opts(algo=1, width=64, tspace=126, skip_self=ss, max_la=64, max_cand=128)
    t0 = time.perf_counter()
    print('aligning, skip_self', ss, flush=True)
    las, tr = ctx.align_db(D, D, o)
    st = ctx.align_stats()
    print(f"skip_self {ss}: k_tile {st.ms_wave:.1f} ms seeds {st.ms_seed:.1f} alignments {st.alignments} las {len(las)} cells {st.wave_cells/1e9:.1f} G -> {st.wave_cells/st.ms_wave/1e9:.1f} Tcells/s", flush=True)
