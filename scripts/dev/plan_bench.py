"""dev (CPU only): the host work of one rank between the collectives of the sharded collector at the size of configs[2]
-- fabricated alignments of 1 M x 15 kb reads laid over 1 001 contigs of 98.8 kb with 1.1 kb gaps (no sequence, no GPU):
dh_shard_read_joins of one shard, dh_shard_graph_plan_create on the 8 gathered blobs.  DH_TRACE=1 prints the laps.

usage: python scripts/dev/plan_bench.py [world] [reps]"""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import dentist_amd
from dentist_amd._lib import LA_DTYPE, ShardPlan, shard_read_joins

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
NC, LC, GAP, NR, LR = 1001, 98_800, 1_100, 1_000_000, 15_000
S = LC + GAP
rng = np.random.default_rng(5)
start = rng.integers(-LR // 2, NC * S - LR // 2, NR)   # (read ids say nothing about positions, as in sim.Workload)
recs = []
for shift in (0, 1):   # a read touches at most two contigs
    c = np.clip(start // S + shift, 0, NC - 1)
    c0 = c * S
    lo = np.maximum(start, c0)
    hi = np.minimum(start + LR, c0 + LC)
    m = hi - lo >= 500
    la = np.zeros(int(m.sum()), dtype=LA_DTYPE)
    la["aread"], la["bread"] = c[m], np.nonzero(m)[0]
    la["abpos"], la["aepos"] = (lo - c0)[m], (hi - c0)[m]
    la["bbpos"], la["bepos"] = (lo - start)[m], (hi - start)[m]
    la["diffs"] = (hi - lo)[m] // 8
    recs.append(la)
las = np.concatenate(recs)
las = las[np.lexsort((las["abpos"], las["aread"], las["bread"]))]
_, first = np.unique(las[["aread", "bread"]], return_index=True)
las = np.ascontiguousarray(las[np.sort(first)])
coff = np.arange(NC + 1, dtype=np.int64) * LC
roff = np.arange(NR + 1, dtype=np.int64) * LR
gaps = np.stack([np.arange(NC - 1), np.arange(1, NC)], axis=1).astype(np.int32)
po = dentist_amd.default_process_opts(algo=1)
print("alignments", len(las))
blobs = []
for r in range(world):
    lo, hi = NR * r // world, NR * (r + 1) // world
    mine = np.ascontiguousarray(las[(las["bread"] >= lo) & (las["bread"] < hi)])
    t = time.perf_counter()
    b = shard_read_joins(mine, coff, roff[lo:hi + 1] - roff[lo], lo)
    if r == 1:
        print("read joins of rank 1: %.2f ms, %d alignments, blob %d bytes" % ((time.perf_counter() - t) * 1e3, len(mine), len(b)))
    blobs.append(b)
for it in range(reps):
    t = time.perf_counter()
    plan = ShardPlan(blobs, po, graph=(NC, gaps, {"min_spanning_reads": po.min_reads}))
    dt = (time.perf_counter() - t) * 1e3
    cl, cnt, tri = plan.piles.flat()
    print("plan %.2f ms: %d pile-ups, %d entries, %d records" % (dt, len(cl), int(cnt.sum()), len(plan.las)))
    plan.close()
