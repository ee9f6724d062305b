"""dev: one pass at 4x config[1] (4.2 Gbp of reads: total bases beyond 2^31) -- size-independent checks."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dentist_amd
from dentist_amd import sim
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 4
t = time.time()
rl = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
w = sim.Workload(10_000_000 * scale, 100 * scale, 100_000 * scale, rl, seed=77)
print(f"workload: {len(w.reads.bases)/1e9:.2f} Gbp reads, {w.contigs.n} contigs, {time.time()-t:.1f} s", flush=True)
ctx = dentist_amd.Context(0)
A, B = ctx.db(w.contigs), ctx.db(w.reads)
mo = dentist_amd.default_align_opts(kmer_mod=4)
po = dentist_amd.default_process_opts()
for it in range(2):
    A.drop_cache(); B.drop_cache()
    t = time.time()
    las, trace = ctx.align_db(A, B, mo, select_best=True)
    t1 = time.time()
    piles = dentist_amd.Pileups(las, w.contigs.off, po)
    rec, bases = dentist_amd.process_pileups(ctx, A, B, las, trace, piles, po)
    t2 = time.time()
    ok = rec[rec["status"] == 0]
    print(f"pass {it}: map {1e3*(t1-t):.0f} ms ({len(las)} LAs), process {1e3*(t2-t1):.0f} ms, piles {len(piles)}, closed {len(ok)}", flush=True)
# mapped positions agree with the truth
s = w.read_truth[las["bread"], 0]; e = w.read_truth[las["bread"], 1]
cs = w.contig_start[las["aread"]]
good = ((las["flags"] & 1) == w.read_truth[las["bread"], 2]) & (cs + las["abpos"] >= s - 80) & (cs + las["aepos"] <= e + 80)
print("LAs consistent with truth:", float(good.mean()))
assert good.mean() > 0.999 and len(ok) >= 0.97 * 100 * scale
from oracle import pyoracle as oz
bad = 0
for r in ok[:: max(1, len(ok) // 40)]:
    g = int(r["contig_left"])
    cons = bases[r["cons_off"]:r["cons_off"] + r["cons_len"]]
    cseq = sim.revcomp(cons) if r["comp"] else cons
    ins = cseq[r["ins_begin"]:r["ins_end"]]
    truth = w.truth[w.contig_start[g] + r["left_aepos"]: w.gap_end[g] + r["right_abpos"]]
    ed, _ = oz.nw(truth, ins)
    bad += ed > max(3, 0.01 * len(truth))
print("sampled insertions off by > 1 %:", bad)
assert bad <= 1
print("scale ok")
