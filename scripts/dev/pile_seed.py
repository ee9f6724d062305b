"""dev: pile-up all-vs-all of a configs[1]-like batch at different LDS hit capacities."""
import os, sys
sys.path.insert(0, '.')
import numpy as np, dentist_amd
from dentist_amd import sim
g = sim.genome(5, 3_000_000)
seqs, grp = [], []
for p in range(600):
    s = 1000 + p * 4500
    rd, _ = sim.reads(100 + p, g[s:s + 2600], 60, 2600, 0, err=0.13)
    for i in range(rd.n):
        seqs.append(rd.seq(i)); grp.append(p)
db = sim.SeqDb.from_list(seqs, np.array(grp, dtype=np.int32))
ctx = dentist_amd.Context(0)
d = ctx.db(db)
o = dentist_amd.default_align_opts(tspace=126, skip_self=2, max_la=64, max_cand=128)
for cap in sys.argv[1:]:
    os.environ["DH_SEED_CAP"] = cap
    for rep in range(2):
        d.drop_cache()
        las, tr = ctx.align_db(d, d, o)
    st = ctx.align_stats().as_dict()
    print('cap', cap, 'las', len(las), 'seed ms %.2f wave %.2f index %.2f' % (st['ms_seed'], st['ms_wave'], st['ms_index']), 'big', st['big_items'], 'hits', st['hits'], flush=True)
