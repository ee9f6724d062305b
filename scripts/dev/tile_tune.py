"""k_tile tuning on the configs[2] mapping pass: resident waves per CU x reads per launch."""
import os
import sys
import time

sys.path.insert(0, ".")
import numpy as np  # noqa: E402

import dentist_amd  # noqa: E402
from dentist_amd import sim  # noqa: E402

nreads = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
w = sim.Workload(100_000_000, 1000, nreads, 15_000, seed=20260929)
ctx = dentist_amd.Context(0)
A, B = ctx.db(w.contigs), ctx.db(w.reads)
mo = dentist_amd.default_align_opts(kmer_mod=4, k=20, width=64, xdrop=60, algo=1)
po = dentist_amd.default_process_opts()
settings = [(16, 1 << 18), (8, 1 << 18), (4, 1 << 18), (16, 1 << 19), (8, 1 << 19), (16, 1 << 20), (8, 1 << 20), (4, 1 << 20),
            (16, 1 << 21), (8, 1 << 21), (12, 1 << 20)]
if len(sys.argv) > 2:
    settings = [tuple(int(x) for x in s.split(":")) for s in sys.argv[2:]]
for wpc, chunk in settings:
    os.environ["DH_TILE_WAVES_PER_CU"] = str(wpc)
    os.environ["DH_ALIGN_CHUNK"] = str(chunk)
    for rep in range(2):
        A.drop_cache()
        B.drop_cache()
        t0 = time.perf_counter()
        las, trace, dropped, cands = ctx.map_reads(A, B, mo, po, sorted=False, candidates=True)
        dt = time.perf_counter() - t0
        st = ctx.align_stats()
    print(f"waves/CU {wpc:3d} items/launch {chunk:8d}: wall {dt * 1e3:7.1f} ms  seed {st.ms_seed:6.1f} wave {st.ms_wave:6.1f} "
          f"gather {st.ms_gather:5.1f} launches {st.wave_launches} las {len(las)}", flush=True)
