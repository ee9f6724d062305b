"""dev: scaffold-graph builder vs the spanning collector on configs[1]."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, dentist_amd
from dentist_amd import sim
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "cfg1_10Mb_100gaps_100kx10kb"
w = sim.Workload(seed=20260929, **bench.WORKLOADS[name])
ctx = dentist_amd.Context(0)
A, B = ctx.db(w.contigs), ctx.db(w.reads)
mo = dentist_amd.default_align_opts(kmer_mod=4, k=20, width=14, xdrop=60)
po = dentist_amd.default_process_opts()
las, tr = ctx.align_db(A, B, mo, select_best=True)
las, dropped, _ = dentist_amd.collect_filter(las, w.contigs.off, w.reads.off, po, inplace=True)
t0 = time.perf_counter()
cand = dentist_amd.Pileups(las, w.contigs.off, po, candidates=True)
t1 = time.perf_counter()
ig = np.stack([np.arange(w.contigs.n - 1), np.arange(1, w.contigs.n)], axis=1)
gp, skipped = dentist_amd.scaffold_spanning_pileups(las, w.contigs.off, w.reads.off, ig, min_spanning_reads=po.min_reads)
t2 = time.perf_counter()
joins, ent = dentist_amd.scaffold_pileups(las, w.contigs.off, w.reads.off, ig, min_spanning_reads=po.min_reads)
print('spanning collect %.1f ms, graph %.1f ms; piles %d vs %d, skipped %d; joins by type %s' % ((t1-t0)*1e3, (t2-t1)*1e3, len(cand), len(gp), skipped, np.bincount(joins['type'], minlength=3)))
a, b = cand.flat(), gp.flat()
same = sum(1 for _ in [0] if np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]))
print('identical', bool(same), 'entries', len(a[2]) // 3, len(b[2]) // 3)
if not same:
    ta, tb = a[2].reshape(-1, 3), b[2].reshape(-1, 3)
    sa, sb = set(map(tuple, ta.tolist())), set(map(tuple, tb.tolist()))
    print('only spanning', len(sa - sb), 'only graph', len(sb - sa))
    print(sorted(sa - sb)[:5], sorted(sb - sa)[:5])
n1 = joins[joins['type'] == 1]
print('gap joins', len(n1), 'entries in gap piles', int(n1['count'].sum()), 'extension reads merged', int(sum((ent[j['first']:j['first']+j['count']]['n'] == 1).sum() for j in n1)))
