import sys, time
sys.path.insert(0, '.')
import numpy as np
import dentist_amd
from dentist_amd import sim
w = sim.Workload(1_000_000, 10, 5000, 10000, seed=20260929)
ctx = dentist_amd.Context(0)
A, B = ctx.db(w.contigs), ctx.db(w.reads)
g = dentist_amd.default_align_opts()
t=time.time(); las, trace = ctx.align_db(A, B, g); print("map s", time.time()-t, ctx.align_stats().as_dict())
po = dentist_amd.default_process_opts()
t=time.time(); piles = dentist_amd.Pileups(las, w.contigs.off, po); print("collect s", time.time()-t, len(piles))
for it in range(2):
    t=time.time(); rec, bases = dentist_amd.process_pileups(ctx, A, B, las, trace, piles, po); print("process s", time.time()-t)
    print(dentist_amd.process_stats(ctx))
print(rec[["contig_left","status","nreads","ref_read","ins_begin","ins_end","cons_len"]])
from oracle import pyoracle as oz
tot=0; err=0
for r in rec:
    if r["status"]: continue
    gap=r["contig_left"]
    cons = bases[r["cons_off"]:r["cons_off"]+r["cons_len"]]
    cseq = sim.revcomp(cons) if r["comp"] else cons
    ins = cseq[r["ins_begin"]:r["ins_end"]]
    truth = w.truth[w.contig_start[gap] + r["left_aepos"]: w.gap_end[gap] + r["right_abpos"]]
    ed,_ = oz.nw(truth, ins); tot+=len(truth); err+=ed
    print(gap, "gap len", w.gap_end[gap]-w.gap_begin[gap], "ins", len(ins), "truth", len(truth), "edit", ed)
print("total", tot, "err", err, err/max(tot,1))
