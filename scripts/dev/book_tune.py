"""k_tile: lanes that must wait before a bookkeeping pass (DH_TILE_BOOK_MIN) x waves per CU, on the cfg2 step."""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
import dentist_amd
from dentist_amd import sim
w = sim.Workload(100_000_000, 1000, 1_000_000, 15_000, seed=20260929)
ctx = dentist_amd.Context(0)
A, B = ctx.db(w.contigs), ctx.db(w.reads)
mo = dentist_amd.default_align_opts(kmer_mod=4, k=20, width=64, xdrop=60, algo=1)
po = dentist_amd.default_process_opts(algo=1)
ref = None
for bm, wpc in [(1, 12), (1, 16)]:
    os.environ["DH_TILE_BOOK_MIN"] = str(bm)
    os.environ["DH_TILE_WAVES_PER_CU"] = str(wpc)
    for rep in range(2):
        A.drop_cache(); B.drop_cache()
        ctx.cum_stats(reset=True)
        t0 = time.perf_counter()
        las, trace, dropped, cands = ctx.map_reads(A, B, mo, po, sorted=False, candidates=True)
        st = ctx.align_stats()
        t1 = time.perf_counter()
        piles = cands.select(las, po)
        rec, bases = dentist_amd.process_pileups(ctx, A, B, las, trace, piles, po)
        t2 = time.perf_counter()
        pst = dentist_amd.process_stats(ctx)
        cum = ctx.cum_stats().as_dict()
    if ref is None:
        ref = (rec.copy(), bases.copy())
    same = np.array_equal(ref[0], rec) and np.array_equal(ref[1], bases)
    print(f"book_min {bm:2d} waves/CU {wpc:2d}: map {1e3*(t1-t0):6.1f} (index {st.ms_index:4.1f} seeds {st.ms_seed:5.1f} k_tile {st.ms_wave:5.1f}) process {1e3*(t2-t1):6.1f} "
          f"(pile align {pst['ms_pile_align']:6.1f} [{' '.join('%s=%.1f' % (k[3:], v) for k, v in pst.items() if k.startswith('ms_pa_'))}] realign {pst['ms_realign']:5.1f} flank {pst['ms_flank_align']:4.1f}) all k_tile {cum['ms_wave']:5.1f} all seeds {cum['ms_seed']:5.1f} same {same}", flush=True)
