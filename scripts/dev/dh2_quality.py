"""DH-2 (tiled banded extension) against DH-1 (O(ND) wave) on the CPU oracle: LA counts, placement
against the truth, diffs, and what the pile-up consensus path makes of the mappings.

usage: python scripts/dev/dh2_quality.py [genome_len ngaps nreads read_len [err]]
"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from dentist_amd import sim  # noqa: E402
from oracle import pyoracle as oz  # noqa: E402

args = [int(x) for x in sys.argv[1:5]] or [1_000_000, 10, 5000, 10_000]
err = float(sys.argv[5]) if len(sys.argv) > 5 else 0.13
w = sim.Workload(*args, seed=20260929, err=err)
print("workload", args, "err", err, "read bp", int(w.reads.off[-1]))


def evaluate(name, **kw):
    o = oz.default_opts(k=20, kmer_mod=4, **kw)
    t0 = time.time()
    las, trace, stats = oz.align_db(w.contigs, w.reads, o, nthreads=8, select_best=True)
    dt = time.time() - t0
    s, e = w.read_truth[las["bread"], 0], w.read_truth[las["bread"], 1]
    cs = w.contig_start[las["aread"]]
    strand_ok = (las["flags"] & 1) == w.read_truth[las["bread"], 2]
    ok = strand_ok & (cs + las["abpos"] >= s - 80) & (cs + las["aepos"] <= e + 80)
    # how far the alignment's ends are from where the truth says the overlap ends: contig interval
    # intersected with the read's truth interval
    clen = (w.contigs.off[1:] - w.contigs.off[:-1])[las["aread"]]
    tb = np.maximum(s, cs) - cs
    te = np.minimum(e, cs + clen) - cs
    db, de = las["abpos"] - tb, las["aepos"] - te
    alen = las["aepos"] - las["abpos"]
    print(f"{name:28s} {dt:6.1f}s las {len(las):6d} reads mapped {len(set(las['bread'].tolist())):6d} placed {ok.mean():.4f} "
          f"aligned bp {int(alen.sum()):11d} diffs/alen {las['diffs'].sum() / alen.sum():.4f} "
          f"|begin off| mean {np.abs(db).mean():.1f} p99 {np.percentile(np.abs(db), 99):.0f}  "
          f"|end off| mean {np.abs(de).mean():.1f} p99 {np.percentile(np.abs(de), 99):.0f} cells {stats[3]}")
    po = oz.default_process_opts()
    gaps, tri = oz.collect_spanning_c(las, w.contigs, po)
    t0 = time.time()
    rec, bases = oz.process_piles_c(w.contigs, w.reads, las, trace, gaps, tri, po, nthreads=8)
    closed = rec[rec["status"] == 0]
    edits = total = 0
    for r in closed:
        g = int(r["gap"])
        cons = bases[r["cons_off"]:r["cons_off"] + r["cons_len"]]
        cseq = sim.revcomp(cons) if r["comp"] else cons
        ins = cseq[r["ins_begin"]:r["ins_end"]]
        truth = w.truth[w.contig_start[g] + r["left_aepos"]: w.gap_end[g] + r["right_abpos"]]
        ed, _ = oz.nw(truth, ins)
        edits += ed
        total += len(truth)
    print(f"{'':28s} piles {len(gaps)} (reads {sum(len(t) for t in tri)}) closed {len(closed)} consensus edits {edits} / {total} "
          f"= {edits / max(total, 1):.5f}  ({time.time() - t0:.1f}s)")
    return las, trace


evaluate("DH-1 width 14 xdrop 60", width=14, xdrop=60)


evaluate("DH-2 band 32 xdrop 120", width=32, xdrop=120, algo=1)
evaluate("DH-2 band 64 xdrop 120", width=64, xdrop=120, algo=1)
