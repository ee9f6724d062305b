"""dev: mapping + process of configs[1] at a given wave width (quality and kernel times)."""
import sys
sys.path.insert(0, '.')
import numpy as np, dentist_amd
from dentist_amd import sim
wd, pw = int(sys.argv[1]), int(sys.argv[2])
w = sim.Workload(10_000_000, 100, 100_000, 10_000, seed=20260929)
ctx = dentist_amd.Context(0)
A, B = ctx.db(w.contigs), ctx.db(w.reads)
mo = dentist_amd.default_align_opts(kmer_mod=4, k=20, width=wd)
for rep in range(2):
    las, tr = ctx.align_db(A, B, mo, select_best=True)
st = ctx.align_stats().as_dict()
po = dentist_amd.default_process_opts()
piles = dentist_amd.Pileups(las, w.contigs.off, po)
import os
os.environ["DH_PILE_WIDTH"] = str(pw)
rec, bases = dentist_amd.process_pileups(ctx, A, B, las, tr, piles, po)
ps = dentist_amd.process_stats(ctx)
from oracle import pyoracle as oz
ed = tot = 0
for r in rec[rec["status"] == 0]:
    g = int(r["contig_left"]); cons = bases[r["cons_off"]:r["cons_off"] + r["cons_len"]]
    cseq = sim.revcomp(cons) if r["comp"] else cons
    t = w.truth[w.contig_start[g] + r["left_aepos"]: w.gap_end[g] + r["right_abpos"]]
    e, _ = oz.nw(t, cseq[r["ins_begin"]:r["ins_end"]]); ed += e; tot += len(t)
print('map width', wd, 'pile width', pw, 'las', len(las), 'aligned', int((las['aepos'] - las['abpos']).sum()),
      'map wave ms %.2f seed %.2f' % (st['ms_wave'], st['ms_seed']), 'piles', len(piles), 'closed', int((rec['status'] == 0).sum()),
      'pile_align %.1f' % ps['ms_pile_align'], 'pile_las', ps['pile_las'], 'edits', ed, tot, flush=True)
