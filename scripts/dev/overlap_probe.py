"""dev: can seeds (memory-bound) and tiles (VALU-bound) share the chip?  Two contexts (own streams, own scratch) map the
two halves of configs[2]'s reads from two host threads, with both persistent grids sized for co-residency
(DH_SEED_BLOCKS_PER_CU, DH_TILE_WAVES_PER_CU), against the same two calls one after the other."""
import os, sys, threading, time
sys.path.insert(0, ".")
import numpy as np
import dentist_amd
from dentist_amd import sim
import bench
spec = bench.WORKLOADS["cfg2_100Mb_1000gaps_1Mx15kb"]
w = sim.Workload(seed=20260929, **spec)
n = w.reads.n
halves = []
for lo, hi in ((0, n // 2), (n // 2, n)):
    halves.append(sim.SeqDb(w.reads.bases[w.reads.off[lo]:w.reads.off[hi]], w.reads.off[lo:hi + 1] - w.reads.off[lo]))
mo = dentist_amd.default_align_opts(kmer_mod=4, k=20, width=64, xdrop=60, algo=1)
po = dentist_amd.default_process_opts(algo=1)
ctxs = [dentist_amd.Context(0), dentist_amd.Context(0)]
dbs = [(c.db(w.contigs), c.db(h)) for c, h in zip(ctxs, halves)]

def run(i, out):
    c, (A, B) = ctxs[i], dbs[i]
    t0 = time.perf_counter()
    las, trace, dropped, cands = c.map_reads(A, B, mo, po, sorted=False, candidates=True)
    out[i] = (time.perf_counter() - t0, len(las), c.align_stats().ms_seed, c.align_stats().ms_wave)

for sb, tw in ((0, 0), (2, 4), (2, 6), (2, 8), (1, 8)):
    if sb:
        os.environ["DH_SEED_BLOCKS_PER_CU"] = str(sb); os.environ["DH_TILE_WAVES_PER_CU"] = str(tw)
    for mode in ("warm", "sequential", "concurrent"):
        out = [None, None]
        t0 = time.perf_counter()
        if mode == "concurrent":
            th = [threading.Thread(target=run, args=(i, out)) for i in range(2)]
            [t.start() for t in th]; [t.join() for t in th]
        else:
            run(0, out); run(1, out)
        dt = time.perf_counter() - t0
        if mode != "warm":
            print(f"seed blocks/CU {sb or 'max'} tile waves/CU {tw or 12}: {mode:10s} {dt*1e3:7.1f} ms  per call {[round(o[0]*1e3,1) for o in out]} seeds {[round(o[2],1) for o in out]} tiles {[round(o[3],1) for o in out]} las {[o[1] for o in out]}", flush=True)
