#!/usr/bin/env python3
"""BASELINE configs[4] (3 Gb assembly, 10 000 gaps, 10 M x 20 kb ONT-like reads on 8 GPUs): ONE rank's share on one GPU,
as a measured line for profiles/ (not the headline): the whole assembly and its index, the rank's block of 1.25 M reads
mapped against it (dh_map_reads, bench.py's options), and the 1 250 pile-ups it owns processed with the spanning reads of
all ranks.  Prints one JSON line.  tests/test_configs_gpu.py::test_config4_one_rank_of_eight checks the properties."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dentist_amd
from dentist_amd import sim

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
rebuild = len(sys.argv) > 2 and sys.argv[2] == "rebuild"   # the index is dropped and rebuilt in every pass (as bench.py does)
s = sim.RankShare(3_000_000_000, 10_000, 10_000_000, 20_000, rank=0, world=8, seed=20260929)
ctx = dentist_amd.Context(0)
KMER_MOD = int(os.environ.get("KMER_MOD", "4"))   # the stress configuration samples 1/4; 8 = bench.py's configs[2] default
mo = dentist_amd.default_align_opts(kmer_mod=KMER_MOD, k=20, width=64, xdrop=60, algo=1)
po = dentist_amd.default_process_opts(algo=1)
po.max_reads = int(os.environ.get("MAX_READS", "60"))   # 0 = no cap, the reference's behaviour
A, B, P = ctx.db(s.contigs), ctx.db(s.reads), ctx.db(s.pile_reads)
rows = []
for step in range(steps + 1):   # the first pass builds the index and sizes the buffers
    ctx.cum_stats(reset=True)
    if rebuild:
        A.drop_cache()
    t0 = time.perf_counter()
    las, trace, dropped, cands = ctx.map_reads(A, B, mo, po, sorted=False, candidates=True)
    t1 = time.perf_counter()
    ast = ctx.align_stats()
    pl, pt, _, pc = ctx.map_reads(A, P, mo, po, sorted=False, candidates=True)
    piles = pc.select(pl, po)
    t2 = time.perf_counter()
    rec, bases = dentist_amd.process_pileups(ctx, A, P, pl, pt, piles, po)
    t3 = time.perf_counter()
    rows.append(dict(map_ms=(t1 - t0) * 1e3, seeds_ms=ast.ms_seed, tiles_ms=ast.ms_wave, index_ms=ast.ms_index,
                     pile_map_ms=(t2 - t1) * 1e3, process_ms=(t3 - t2) * 1e3))
    del las, trace
mine = rec[np.isin(rec["contig_left"], s.owned_gaps)]
read_bp = int(s.reads.off[-1])
steady = rows[1:]
mean = lambda k: float(np.mean([r[k] for r in steady]))  # noqa: E731
seed_bytes = read_bp * (1.0 + 64.0 / KMER_MOD)
placed = None
if hasattr(s, "read_truth"):   # fraction of the records inside their read's true interval (+- 80 bp), right strand
    las, _, _, _ = ctx.map_reads(A, B, mo, po, sorted=False, candidates=True)
    t, e = s.read_truth[las["bread"], 0], s.read_truth[las["bread"], 1]
    cs = s.contig_start[las["aread"]]
    ok = ((las["flags"] & 1) == s.read_truth[las["bread"], 2]) & (cs + las["abpos"] >= t - 80) & (cs + las["aepos"] <= e + 80)
    placed = [float(ok.mean()), int(len(las)), int(len(set(las["bread"].tolist())))]
print(json.dumps({"workload": "cfg4_3Gb_10000gaps_10Mx20kb_ONT, rank 0 of 8 on one GPU", "read_bp_mapped": read_bp,
                  "mapping_kmer_mod": KMER_MOD, "placed_frac_records_reads": placed,
                  "first_pass": rows[0], "steady": {k: mean(k) for k in steady[0]},
                  "read_bp_mapped_per_sec": read_bp / (mean("map_ms") * 1e-3),
                  "owned_gaps": int(len(s.owned_gaps)), "closed": int((mine["status"] == 0).sum()),
                  "k_seed_algorithmic_GBs": seed_bytes / (mean("seeds_ms") * 1e-3) / 1e9,
                  "k_seed_hbm_frac": seed_bytes / (mean("seeds_ms") * 1e-3) / 1e9 / 8000.0}))
