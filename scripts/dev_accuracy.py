import sys, time
sys.path.insert(0, '.')
import numpy as np
import dentist_amd
from dentist_amd import sim
from oracle import pyoracle as oz
w = sim.Workload(10_000_000, 100, 100_000, 10_000, seed=20260929)
ctx = dentist_amd.Context(0)
A, B = ctx.db(w.contigs), ctx.db(w.reads)
las, trace = ctx.align_db(A, B, dentist_amd.default_align_opts(kmer_mod=4), select_best=True)
for rounds, maxr in ((1,60),(2,60),(3,60),(4,60),(2,30),(3,30),(2,20)):
    po = dentist_amd.default_process_opts(rounds=rounds, max_reads=maxr)
    piles = dentist_amd.Pileups(las, w.contigs.off, po)
    t=time.time(); rec, bases = dentist_amd.process_pileups(ctx, A, B, las, trace, piles, po); dt=time.time()-t
    tot=err=0; nclosed=0; bad=[]
    for r in rec:
        if r["status"]: continue
        nclosed+=1
        g=r["contig_left"]; cons = bases[r["cons_off"]:r["cons_off"]+r["cons_len"]]
        cseq = sim.revcomp(cons) if r["comp"] else cons
        ins = cseq[r["ins_begin"]:r["ins_end"]]
        truth = w.truth[w.contig_start[g] + r["left_aepos"]: w.gap_end[g] + r["right_abpos"]]
        ed,_ = oz.nw(truth, ins); tot+=len(truth); err+=ed
        if ed: bad.append((int(g), int(ed), len(truth)))
    print(f"rounds {rounds} max_reads {maxr}: closed {nclosed} err {err}/{tot} = {err/tot:.5f}  time {dt*1e3:.0f} ms; gaps with errors {len(bad)}")
