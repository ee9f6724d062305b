"""dev: the process stage of configs[2] as one call against two half-batches of pile-ups processed concurrently by two
contexts (own streams / scratch) from two host threads: does one half's host work hide behind the other's kernels?"""
import sys, threading, time
sys.path.insert(0, ".")
import numpy as np
import dentist_amd
from dentist_amd import sim
import bench
spec = bench.WORKLOADS["cfg2_100Mb_1000gaps_1Mx15kb"]
w = sim.Workload(seed=20260929, **spec)
mo = dentist_amd.default_align_opts(kmer_mod=4, k=20, width=64, xdrop=60, algo=1)
po = dentist_amd.default_process_opts(algo=1)
gaps = np.stack([np.arange(w.contigs.n - 1), np.arange(1, w.contigs.n)], axis=1).astype(np.int32)
ctxs = [dentist_amd.Context(0), dentist_amd.Context(0)]
dbs = [(c.db(w.contigs), c.db(w.reads)) for c in ctxs]
las, trace, dropped, cands = ctxs[0].map_reads(dbs[0][0], dbs[0][1], mo, po, sorted=False, candidates=True)
gp, _ = dentist_amd.scaffold_spanning_pileups(las, w.contigs.off, w.reads.off, gaps, with_extensions=True, min_spanning_reads=po.min_reads)
piles = gp.select(las, po)
cl, cnt, tri = piles.flat()
h = len(cl) // 2
ofs = np.concatenate([[0], np.cumsum(cnt)])
halves = [dentist_amd.Pileups.from_flat(cl[:h], cnt[:h], tri[:ofs[h]]), dentist_amd.Pileups.from_flat(cl[h:], cnt[h:], tri[ofs[h]:])]

def run(i, p, out):
    c, (A, B) = ctxs[i], dbs[i]
    t0 = time.perf_counter()
    out[i] = dentist_amd.process_pileups(c, A, B, las, trace, p, po) + (time.perf_counter() - t0,)

for rep in range(3):
    out = [None, None]
    t0 = time.perf_counter(); run(0, piles, out); t_all = time.perf_counter() - t0
    ref = out[0]
    out = [None, None]
    t0 = time.perf_counter(); run(0, halves[0], out); run(1, halves[1], out); t_seq = time.perf_counter() - t0
    out = [None, None]
    th = [threading.Thread(target=run, args=(i, halves[i], out)) for i in range(2)]
    t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; t_con = time.perf_counter() - t0
    same = np.array_equal(np.concatenate([out[0][0]["cons_len"], out[1][0]["cons_len"]]), ref[0]["cons_len"])
    print(f"one call {t_all*1e3:.1f} ms; halves one after the other {t_seq*1e3:.1f}; halves concurrently {t_con*1e3:.1f} (per half {out[0][2]*1e3:.1f} / {out[1][2]*1e3:.1f}); same lengths {same}", flush=True)
