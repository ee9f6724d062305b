#!/usr/bin/env python3
"""bench.py -- the hot path of DENTIST (alignment + consensus) on N MI355X, one JSON line.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched through
torch.distributed.run, one rank per GPU (RCCL).

Workload (default): BASELINE.json configs[2] -- 100 Mb synthetic assembly, 1 000 gaps, 1 M x 15 kb
reads at 13 % PacBio-like error (15.7 Gbp) -- the configuration north_star quotes its target on.
A "step" is one pass of the whole hot path over that input, resident in HBM when the clock starts:
  1. every read is aligned to the assembly: k-mer index of the contigs, then a loop over read
     chunks (seed filter, wave alignment with trace points, LAs back on the host)  -- damapper's
     role, one call per read block against the persistent index (snakemake/Snakefile:1143-1170),
  2. the alignments pass the six filters of `dentist collect` and the spanning reads of every gap are
     collected (host)                                                           -- `dentist collect`,
  3. every pile-up goes through crop -> pile-up all-vs-all alignment -> filters -> tile QV ->
     reference read -> consensus rounds -> flank re-alignment -> insertion      -- `dentist process`
     with daligner/DASqv/daccord replaced by kernels.
metric = gap-bases closed / second over the WHOLE step (mapping included); read-bp aligned/sec is
reported next to it.

N > 1 is STRONG scaling of the same workload (configs[3]): the reads are block-sharded over the
ranks (contigs and their index replicated), candidate spanning reads are all-gathered, pile-ups
are bin-packed over the ranks by n^2 L, cropped reads travel to the owners with one all-to-all(v)
and the closed gaps are gathered (dentist_amd/parallel.py) -- bit-identical to the 1-GPU result.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

WORKLOADS = {
    # BASELINE.json configs[2] / [3]: 100 Mb assembly, 1 000 gaps, 1 M x 15 kb PacBio-error reads (150x)
    "cfg2_100Mb_1000gaps_1Mx15kb": dict(genome_len=100_000_000, ngaps=1000, nreads=1_000_000, read_len=15_000),
    # BASELINE.json configs[1]: 10 Mb assembly, 100 gaps, 100 k x 10 kb PacBio-error reads
    "cfg1_10Mb_100gaps_100kx10kb": dict(genome_len=10_000_000, ngaps=100, nreads=100_000, read_len=10_000),
    # reduced shape for quick checks (not a bench line)
    "dev_1Mb_10gaps_5kx10kb": dict(genome_len=1_000_000, ngaps=10, nreads=5_000, read_len=10_000),
}
def load_constants():
    """Ceilings the roofline entries are priced against, with the run each one comes from (profiles/constants.json)."""
    with open(os.path.join(ROOT, "profiles", "constants.json")) as f:
        return {k: v["value"] for k, v in json.load(f).items() if isinstance(v, dict)}


CONST = load_constants()
HBM_PEAK_GBS = CONST["hbm_peak_GBs"]  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def kernel_traffic(args, world):
    """Measured HBM bytes per launch (rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE, scripts/profile_round.sh) of THIS build of
    the kernels on THIS configuration, or (None, None, reason): the newest profiles/*_kernel_traffic.json whose build
    id (sha1 of the kernel sources) and configuration match.  A figure of another build is never reported."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from traffic_json import kernel_build_id
    bid = kernel_build_id()
    if world != 1:
        return None, None, "N > 1"
    cands = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_kernel_traffic.json"))
    stale = []
    for name in reversed(cands):
        tj = json.load(open(os.path.join(ROOT, "profiles", name)))
        if not (tj.get("workload") == args.workload and tj.get("mapping_kmer_mod") == args.kmer_mod and
                tj.get("mapping_k") == args.map_k and tj.get("mapping_algo") == args.map_algo):
            continue
        if tj.get("kernel_build_id") == bid:
            return tj.get("k_seed_hbm_bytes_per_launch"), tj.get("k_tile_hbm_bytes_per_launch"), "profiles/" + name
        stale.append(name)
    msg = f"no HBM-counter profile of kernel build {bid} for this configuration" + (f" (stale: {', '.join(stale)})" if stale else "")
    print("bench.py: roofline.traffic = null: " + msg + "; run scripts/profile_round.sh and commit its kernel_traffic.json",
          file=sys.stderr)
    return None, None, msg


def closed_gap_stats(w, rec, bases):
    """Gap bases closed (insertions that passed every gate) and, outside the timed region, their
    edit distance to the truth."""
    from dentist_amd import sim
    from oracle import pyoracle as oz  # checker only, never timed
    closed = rec[rec["status"] == 0]
    gap_bases = int((closed["ins_end"] - closed["ins_begin"]).sum())
    edits = truth_bases = 0
    for r in closed:
        g = int(r["contig_left"])
        cons = bases[r["cons_off"]:r["cons_off"] + r["cons_len"]]
        cseq = sim.revcomp(cons) if r["comp"] else cons
        ins = cseq[r["ins_begin"]:r["ins_end"]]
        truth = w.truth[w.contig_start[g] + r["left_aepos"]: w.gap_end[g] + r["right_abpos"]]
        ed, _ = oz.nw(truth, ins)
        edits += ed
        truth_bases += len(truth)
    return gap_bases, len(closed), edits, truth_bases


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg2_100Mb_1000gaps_1Mx15kb")
    ap.add_argument("--cpu-seconds", type=float, default=40.0, help="CPU time budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kmer-mod", type=int, default=1,
                    help="modimer sampling of the mapping index: one canonical k-mer in kmer_mod is indexed / looked up.  "
                         "1 (default) = every k-mer, the reference's behaviour (damapper has no sampling, "
                         "commandline.d:2943-2955); 8 is the labelled fast mode (fast_mode_timed)")
    ap.add_argument("--map-k", type=int, default=20, help="k-mer length of the mapping pass (damapper's default)")
    ap.add_argument("--map-algo", type=int, default=1,
                    help="extension algorithm of the mapping pass: 1 = DH-2 (tiled banded bit-parallel DP, one "
                         "alignment per lane, k_tile), 0 = DH-1 (O(ND) wave, k_wave2)")
    ap.add_argument("--map-width", type=int, default=None,
                    help="DH-2: the band (64); DH-1: live diagonals of the wave (default 14: four alignments per wavefront)")
    ap.add_argument("--band", type=int, choices=(64, 32), default=64,
                    help="DH-2: rows of the band of every alignment stage (mapping, pile-up all-vs-all, re-alignment, flanks): 64, "
                         "or 32 = the same recurrence on 32-bit vectors (half the instructions per column, +- 16 diagonals of "
                         "drift per tile; bit-exact against the oracle at W = 32 as well)")
    ap.add_argument("--map-xdrop", type=int, default=60, help="score lag that trims the mapping waves")
    ap.add_argument("--process-algo", type=int, default=1,
                    help="alignments of the process stages (pile-up all-vs-all, re-alignment, flanks): 1 = DH-2, 0 = DH-1")
    ap.add_argument("--max-reads", type=int, default=0,
                    help="reads kept per pile-up: 0 (default) = no cap, the reference's behaviour "
                         "(processPileUps/package.d:283-374 processes every read alignment); 60 is the labelled fast mode")
    ap.add_argument("--collect", choices=("graph", "spanning"), default=None,
                    help="pile-up membership: 'graph' = the scaffold-graph builder of `dentist collect` with the extension "
                         "entries it merges into a gap (pileups.d:173-208; the default for every N: at N > 1 the raw joins "
                         "of each rank's reads are all-gathered), 'spanning' = one entry per spanning read, collected per "
                         "chunk while mapping")
    ap.add_argument("--device-trace", action="store_true",
                    help="leave the mapping's trace values on the device (dh_map_reads want_sorted & 8) for `process` to gather "
                         "from, instead of copying them to the host chunk by chunk (330 MB per step of configs[2])")
    ap.add_argument("--fast-steps", "--ref-steps", dest="fast_steps", type=int, default=3,
                    help="steps of the FAST MODE (modimers 1/8 in the mapping index, 60 reads per pile-up: two work-reducing "
                         "knobs the reference does not have) timed after the headline loop in the same process and reported "
                         "as fast_mode_timed; 0 = skip; N = 1 only")
    ap.add_argument("--fast-kmer-mod", type=int, default=8)
    ap.add_argument("--fast-max-reads", type=int, default=60)
    ap.add_argument("--ref-partners", type=int, default=60,
                    help="partner reads per read of the pile-up all-vs-all in the third timed configuration (the headline's knobs "
                         "plus dh_process_opts.max_partners = this): reported as uncapped_partner_cut_timed; 0 = skip")
    ap.add_argument("--max-partners", type=int, default=None, help="dh_process_opts.max_partners of the headline run (default 0: every pair)")
    ap.add_argument("--dev-share-gpu", action="store_true",
                    help="development only: all ranks on cuda:0 with gloo collectives (exercises the N > 1 "
                         "code path on a 1-GPU box; not a measurement)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.dev_share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dev_share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))

    import dentist_amd
    from dentist_amd import sim
    from dentist_amd.parallel import shard_range, sharded_process

    spec = WORKLOADS[args.workload]
    # one workload for every N (strong scaling): the assembly on every rank, the reads block-sharded
    lo, hi = shard_range(spec["nreads"], rank, world)
    w = sim.Workload(seed=20260929, read_range=(lo, hi), **spec)
    stream = torch.cuda.current_stream().cuda_stream
    ctx = dentist_amd.Context(local_rank, stream=stream)
    A, B = ctx.db(w.contigs), ctx.db(w.reads)
    # mapping pass: damapper's k-mer length, modimer sampling 1/8 (--kmer-mod), every other option at its default
    if args.map_width is None:
        args.map_width = args.band if args.map_algo == 1 else 14
    mopts = dentist_amd.default_align_opts(kmer_mod=args.kmer_mod, k=args.map_k, width=args.map_width,
                                           xdrop=args.map_xdrop, algo=args.map_algo)
    band_kw = dict(width=32) if (args.band == 32 and args.process_algo == 1) else {}
    popts = dentist_amd.default_process_opts(algo=args.process_algo, **band_kw)
    popts.max_reads = args.max_reads
    if args.max_partners is not None:
        popts.max_partners = args.max_partners
    read_bp = int(len(w.reads.bases))
    if args.collect is None:
        args.collect = "graph"
    input_gaps = np.stack([np.arange(w.contigs.n - 1), np.arange(1, w.contigs.n)], axis=1).astype(np.int32)

    def step(mopts=mopts, popts=popts):
        A.drop_cache()  # the k-mer index and the derived copies are rebuilt every step
        B.drop_cache()
        ctx.cum_stats(reset=True)
        t0 = time.perf_counter()
        # mapping + the six alignment filters of `dentist collect` (filter.d:122-356): per-read decisions,
        # so every rank filters the alignments of its own reads, chunk by chunk on the host while the
        # device maps the next chunk (dh_map_reads)
        # and lists the spanning-read candidates of the chunk; records stay in mapping order (by read)
        # (the spanning-read candidates only when that collector is asked for: the scaffold graph does not use them)
        # (--device-trace: the trace values stay in HBM and `process` fetches the ones the cropper reads -- measured slower by
        # 5 ms per step than copying all of them chunk by chunk beside the next chunk's kernels: LABNOTES 10)
        mapped = ctx.map_reads(A, B, mopts, popts, sorted=False, candidates=args.collect != "graph",
                               trace_on_device=(world == 1 and args.device_trace))
        las, trace, dropped = mapped[:3]
        cands = mapped[3] if len(mapped) > 3 else None
        ast = ctx.align_stats()
        mj = ctx.mjoin_counts(reset=True)   # (chunks seeded by the partitioned join, chunks redone by the directory)
        t1 = time.perf_counter()
        if world == 1:
            if args.collect == "graph":
                # `dentist collect` on the filtered alignments: scaffold graph, forks by read support, min spanning
                # reads, extensions merged into their gap; then the optional read cap (by alignment quality)
                gp, _ = dentist_amd.scaffold_spanning_pileups(las, w.contigs.off, w.reads.off, input_gaps,
                                                              with_extensions=True, min_spanning_reads=popts.min_reads)
                piles = gp.select(las, popts)
            else:
                piles = cands.select(las, popts)   # min / max reads per pile-up (choice by alignment quality)
            t2 = time.perf_counter()
            rec, bases = dentist_amd.process_pileups(ctx, A, B, las, trace, piles, popts)
            info = {"piles": len(piles), "entries": int(piles.flat()[1].sum())}
        else:
            las["bread"] += lo   # read ids of the whole reads DB, as in the .las of a block
            t2 = t1
            # graph collector: the raw joins of a rank's reads are all-gathered and every rank builds the same scaffold
            graph = dict(read_off=w.reads.off, input_gaps=input_gaps) if args.collect == "graph" else None
            rec, bases, info = sharded_process(ctx, A, B, lo, w.contigs.off, las, trace, popts, rank, world, cands=cands,
                                               graph=graph)
        t3 = time.perf_counter()
        pst = dentist_amd.process_stats(ctx)
        cum = ctx.cum_stats().as_dict()  # every k_wave / k_seed launch of the step
        info["filtered"] = dropped.tolist()
        return dict(las=las, rec=rec, bases=bases, ast=ast, pst=pst, cum=cum, info=info, mj=mj,
                    t_map=t1 - t0, t_collect=t2 - t1, t_process=t3 - t2)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    runs = []
    for _ in range(args.steps):
        if runs:  # only the last step's results are inspected: the earlier buffers go back to the pools
            for key in ("las", "rec", "bases"):  # before the next step allocates its own
                runs[-1].pop(key, None)
        runs.append(step())
    barrier()
    dt = time.perf_counter() - t0

    # The headline loop above runs at the reference's behaviour by default (every k-mer looked up, every read of a pile-up
    # processed).  Two more operating points are timed here, inside the same run, the same way, and labelled as what they are:
    # the FAST MODE (modimer sampling + read cap: work the reference does not skip) and the headline's knobs with the pile-up
    # all-vs-all bounded to --ref-partners partner reads per read (dh_process_opts.max_partners: every read still votes in the
    # consensus) -- neither is the reference's behaviour
    def timed_variant(vm, vp, nsteps):
        step(vm, vp)   # one untimed step (first-use allocations of differently sized buffers)
        barrier()
        t0_ = time.perf_counter()
        vruns = []
        for _ in range(nsteps):
            if vruns:
                for key in ("las", "rec", "bases"):
                    vruns[-1].pop(key, None)
            vruns.append(step(vm, vp))
        barrier()
        return vruns, time.perf_counter() - t0_

    fast_runs, fast_dt, cut_runs, cut_dt = [], 0.0, [], 0.0
    if args.fast_steps > 0 and world == 1:
        for r in runs[:-1]:
            for key in ("las", "rec", "bases"):
                r.pop(key, None)
        if args.ref_partners > 0 and args.process_algo == 1 and not popts.max_partners:
            cpopts = dentist_amd.default_process_opts(algo=args.process_algo, **band_kw)
            cpopts.max_reads = popts.max_reads
            cpopts.max_partners = args.ref_partners
            cut_runs, cut_dt = timed_variant(mopts, cpopts, args.fast_steps)
        if args.kmer_mod != args.fast_kmer_mod or popts.max_reads != args.fast_max_reads:
            fmopts = dentist_amd.default_align_opts(kmer_mod=args.fast_kmer_mod, k=args.map_k, width=args.map_width,
                                                    xdrop=args.map_xdrop, algo=args.map_algo)
            fpopts = dentist_amd.default_process_opts(algo=args.process_algo, **band_kw)
            fpopts.max_reads = args.fast_max_reads
            fast_runs, fast_dt = timed_variant(fmopts, fpopts, args.fast_steps)

    last = runs[-1]
    aligned_bp = int((last["las"]["aepos"] - last["las"]["abpos"]).sum())
    mean = lambda f: float(np.mean([f(r) for r in runs]))  # noqa: E731
    cum = last["cum"]
    wave_ms = mean(lambda r: r["cum"]["ms_wave"])
    alg_bytes = 2.0 * cum["aligned_bp"] + 2.0 * cum["trace_values"]
    if world > 1:
        cdev = "cpu" if args.dev_share_gpu else "cuda"
        tt = torch.tensor([dt], device=cdev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        tot = torch.tensor([aligned_bp, read_bp], device=cdev, dtype=torch.int64)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        aligned_all, read_all = (int(x) for x in tot.tolist())
    else:
        aligned_all, read_all = aligned_bp, read_bp

    if rank == 0:
        # the closed-gap records are the gathered result of all ranks (identical on every rank)
        gap_all, nclosed_all, edits_all, truth_all = closed_gap_stats(w, last["rec"], last["bases"])
        ms_step = dt / args.steps * 1e3
        # extension kernels of the step (k_tile: mapping launches, pile-up all-vs-all, re-alignment, flanks; k_wave2 when
        # an algo is 0).  Algorithmic bytes of a launch = both sequences of every alignment it emits streamed once
        # (2 B per aligned A base at one byte per base) + its trace (2 B per trace value)
        achieved = alg_bytes / (wave_ms * 1e-3) / 1e9
        seed_traffic, traffic, traffic_src = kernel_traffic(args, world)
        # Mapping seeds.  Directory path (k_seed<cap, false>): per read base 1 B of sequence, per sampled canonical k-mer one
        # random 64-byte directory line of which 16 bytes are the directory word.  Partitioned join (round 5, csrc/dh_mjoin.h:
        # k_mj_part -> k_mj_filter -> k_mj_hits -> k_seed<cap, JOIN>): per read base 1 B read, per sampled k-mer an 8-byte
        # entry written once and read once -- the same 1 + 16 / kmer_mod bytes per base; that figure prices both paths.
        joined = last["mj"][0] > 0
        seed_line = 1.0 * read_bp * (1.0 + 64.0 / max(1, args.kmer_mod))
        seed_useful = 1.0 * read_bp * (1.0 + 16.0 / max(1, args.kmer_mod))
        seed_bytes = seed_useful if joined else seed_line
        seed_ms = mean(lambda r: r["ast"].ms_seed)
        out = {
            "metric": "gap-bases closed/sec",
            "value": gap_all * args.steps / dt,
            "unit": "gap-bp/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": args.workload, "shape": spec, "mapping_k": args.map_k,
                       "mapping_kmer_mod": args.kmer_mod, "mapping_algo": "DH-2 tiled band (k_tile)" if args.map_algo == 1 else "DH-1 wave (k_wave2)",
                       "mapping_width": args.map_width,
                       "mapping_xdrop": args.map_xdrop,
                       "collect": "scaffold-graph builder, extension entries merged into the gap pile-ups" if args.collect == "graph"
                                  else "spanning reads, one entry per read",
                       "pile_up_entries": int(last["info"].get("entries", 0)),
                       "process": {"algo": "DH-2 tiled band (k_tile)" if popts.algo == 1 else "DH-1 wave (k_wave2)",
                                   "max_reads_per_pile_up": popts.max_reads, "min_reads_per_pile_up": popts.min_reads,
                                   "consensus_rounds": popts.rounds, "width": (32 if popts.width == 32 else 64) if popts.algo == 1 else (popts.width or 30),
                                   "xdrop": 120, "tspace": popts.tspace_pile, "dust": popts.dust},
                       "parallelism": f"reads and gaps sharded over {world} GPU(s)",
                       "collectives": last["info"].get("collectives", "none (one rank)"),
                       "read_bp_total": read_all, "pile_ups": int(last["info"]["piles"]),
                       "collect_filters_dropped_las": dict(zip(("lq", "improper", "weakly_anchored", "contained",
                                                                "ambiguous", "redundant"), last["info"]["filtered"])),
                       "gaps_closed": nclosed_all, "gap_bases_closed": gap_all,
                       "consensus_edit_distance_vs_truth": edits_all, "consensus_truth_bases": truth_all,
                       "consensus_error_rate": (edits_all / truth_all) if truth_all else None},
            "read_bp_aligned_per_sec": aligned_all * args.steps / dt,
            "read_bp_aligned_per_sec_mapping_stage": aligned_all / mean(lambda r: r["t_map"]),
            # dominant kernel of the step: the seed filter (k_seed) of the mapping launches.  It is bound by random
            # 64-byte lines (LABNOTES 5): per read base 1 B of sequence, per sampled canonical k-mer one 64 B
            # directory line, one pass for both strands; the ceiling of random 64 B lines measured on this part is
            # 55 G lines/s = 3.5 TB/s at working sets of 128 MB - 2 GB, 3.2 at 4 GB, 3.06 at 16 GB (scripts/rand_access_probe.cpp)
            "roofline": {"bound": "hbm",
                         "kernel": "mapping seeds per chunk of reads: k_mj_part + k_mj_filter + k_mj_hits + k_seed<cap, JOIN> (partitioned k-mer join)"
                                   if joined else "k_seed (mapping launches, directory lookups)",
                         "mapping_chunks_by_join_and_by_directory": list(last["mj"]),
                         "achieved": seed_bytes / (seed_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": seed_bytes / (seed_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": seed_traffic,
                         "launches_per_step": int(last["ast"].wave_launches), "kernel_ms_per_step": seed_ms,
                         "avg_launch_ms": seed_ms / max(1, int(last["ast"].wave_launches)),
                         "algorithmic_bytes_per_step": seed_bytes, "traffic_source": traffic_src,
                         "useful_bytes_per_step": seed_useful,
                         "useful_frac": seed_useful / (seed_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "line_priced_bytes_per_step": seed_line,
                         "line_priced_frac": seed_line / (seed_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "note": "a `launch` is one chunk's seed kernels together (HIP events around them on the context's stream); "
                                 "line_priced_* keeps round 4's figure (64 B per looked-up k-mer) for comparison",
                         "measured_random_line_ceiling_GBs": CONST["random_line_ceiling_GBs"]},
            # second: the extension kernel, one alignment per lane.  VALU bound (scripts/valu_probe.cpp: 2-cycle class
            # 0.90-0.94, 4-cycle class 0.56-0.58 G wave-instructions/s per SIMD); the HBM fraction is reported as asked
            "roofline_tile": {"bound": "hbm", "kernel": "k_tile" if args.map_algo == 1 else "k_wave2", "achieved": achieved,
                              "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                              "launches_per_step": cum["wave_launches"], "kernel_ms_per_step": wave_ms,
                              "avg_launch_ms": wave_ms / max(1, cum["wave_launches"]),
                              "algorithmic_bytes_per_step": alg_bytes,
                              "band_cells_per_s": cum["wave_cells"] / (wave_ms * 1e-3),
                              # VALU issue: 65 wave-instructions per 64 lanes x 64 band cells (SQ_INSTS_VALU of the mapping
                              # launches, profiles/constants.json) against the measured ceiling of
                              # 0.57 G wave-instructions/s per SIMD (scripts/valu_probe.cpp), 1024 SIMDs
                              "valu_frac": (CONST["k_tile_valu_wave_instr_per_4096_cells"] * cum["wave_cells"] / 4096.0) /
                                           (wave_ms * 1e-3) / (CONST["simds"] * CONST["valu_wave_instr_per_s_per_simd"]),
                              "note": "rank 0's launches; integer VALU-issue bound: 65 wave-instructions per 64 band "
                                      "columns (45-48 of them the column step), DP cell updates/s is the honest secondary"},
            # third: the stage the metric is named after.  SURVEY 8(d): per closed gap with n reads of mean cropped length L
            # the pile-up alignment reads every read once per partner (n (n - 1) L), the consensus once (n L), flanks and
            # output 2 L -- (n^2 + 2) L bytes; summed over the pile-ups by the library (dh_get_process_work) and divided by
            # the wall time of the whole stage (crop .. insertions, host phases included).  The stage is integer-VALU work
            # (k_tile, the seed back end, the bit-parallel consensus NW): the HBM fraction is reported as asked, cells/s
            # of its k_tile launches is the honest secondary
            "roofline_process": (lambda pw, pb: {
                "bound": "hbm", "kernel": "process stage (k_join, k_seed back end, k_tile symmetric, k_seg_vote*)",
                "achieved": pb / (pw * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": pb / (pw * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                "algorithmic_bytes_per_step": pb, "wall_ms_per_step": pw,
                "pile_ups": last["pst"]["pile_ups"], "entries": last["pst"]["entries"],
                "mean_entries_per_pile_up": last["pst"]["entries"] / max(1, last["pst"]["pile_ups"]),
                "mean_cropped_length": last["pst"]["cropped_bases"] / max(1, last["pst"]["entries"]),
                "k_tile_process_band_cells_per_s": (cum["wave_cells"] - last["ast"].wave_cells) /
                                                   max(1e-9, (wave_ms - mean(lambda r: r["ast"].ms_wave)) * 1e-3),
                "note": "rank 0's pile-ups; k_tile time of the concurrent parts of the batch is summed (it overstates the time)"})(
                    mean(lambda r: r["t_process"]) * 1e3, float(last["pst"]["algorithmic_bytes"])),
            "stages_ms": {"map_wall": mean(lambda r: r["t_map"]) * 1e3,
                          "map_index": mean(lambda r: r["ast"].ms_index),
                          "map_seed": seed_ms,
                          "map_wave": mean(lambda r: r["ast"].ms_wave),
                          "all_wave": wave_ms, "all_seed": mean(lambda r: r["cum"]["ms_seed"]),
                          "map_gather": mean(lambda r: r["ast"].ms_gather),
                          "collect_wall": mean(lambda r: r["t_collect"]) * 1e3,
                          "process_wall": mean(lambda r: r["t_process"]) * 1e3,
                          **{"process_" + k[3:]: mean(lambda r, k=k: r["pst"][k]) for k in last["pst"] if k.startswith("ms_")}},
        }
        # the other operating points timed above in this very run (labelled: not the reference's behaviour)
        def variant_block(vruns, vdt, what, vkmod):
            rl = vruns[-1]
            rgap, rclosed, redits, rtruth = closed_gap_stats(w, rl["rec"], rl["bases"])
            rmean = lambda f: float(np.mean([f(r) for r in vruns]))  # noqa: E731
            rseed_ms = rmean(lambda r: r["ast"].ms_seed)
            return {
                "what": what,
                "steps": args.fast_steps, "warmup": 1, "ms_per_step": vdt / args.fast_steps * 1e3,
                "value": rgap * args.fast_steps / vdt, "unit": "gap-bp/s",
                "gaps_closed": rclosed, "gap_bases_closed": rgap, "pile_up_entries": int(rl["info"].get("entries", 0)),
                "consensus_error_rate": (redits / rtruth) if rtruth else None,
                "read_bp_aligned_per_sec_mapping_stage": int((rl["las"]["aepos"] - rl["las"]["abpos"]).sum()) / rmean(lambda r: r["t_map"]),
                "seeds_frac_of_hbm_peak": read_bp * (1.0 + 16.0 / vkmod) / (rseed_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "stages_ms": {"map_wall": rmean(lambda r: r["t_map"]) * 1e3, "map_index": rmean(lambda r: r["ast"].ms_index),
                              "map_seed": rseed_ms, "map_wave": rmean(lambda r: r["ast"].ms_wave),
                              "collect_wall": rmean(lambda r: r["t_collect"]) * 1e3,
                              "process_wall": rmean(lambda r: r["t_process"]) * 1e3,
                              **{"process_" + k[3:]: rmean(lambda r, k=k: r["pst"][k]) for k in rl["pst"] if k.startswith("ms_")}}}

        out["operating_point"] = ("reference behaviour: every k-mer of the reads looked up (kmer_mod 1; damapper has no sampling, "
                                  "commandline.d:2943-2955), every read alignment of a pile-up processed (max_reads 0; "
                                  "processPileUps/package.d:283-374), every pair of a pile-up aligned (max_partners 0; package.d:474-485)"
                                  if (args.kmer_mod == 1 and popts.max_reads == 0 and not popts.max_partners) else
                                  f"NOT the reference's behaviour: kmer_mod {args.kmer_mod}, max_reads {popts.max_reads}, "
                                  f"max_partners {popts.max_partners}")
        out["fast_mode_timed"] = variant_block(
            fast_runs, fast_dt, f"FAST MODE, not the reference's behaviour: modimer sampling 1/{args.fast_kmer_mod} of the mapping k-mers "
            f"and at most {args.fast_max_reads} reads per pile-up (two work-reducing knobs the reference does not have)",
            args.fast_kmer_mod) if fast_runs else None
        out["uncapped_partner_cut_timed"] = variant_block(
            cut_runs, cut_dt, f"the headline's knobs with the pile-up all-vs-all bounded to {args.ref_partners} partner reads per "
            "read (dh_process_opts.max_partners; every read still votes in the consensus rounds) -- NOT the reference's "
            "behaviour: daligner aligns every pair of a pile-up (package.d:474-485)", args.kmer_mod) if cut_runs else None
        if not args.no_cpu_baseline and world == 1:   # rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(w, last, mopts, popts, args, gap_all, read_all)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(w, last, mopts, popts, args, gap_bases, read_bp_total):
    """The CPU restatement of the same path (oracle/align.c + oracle/pile.c: C with OpenMP, every
    host core) timed on a bounded sample of rank 0's share of the workload -- kind "port": NOT the
    reference binaries (daligner / damapper / daccord are not on this box).  Legs, sized
    to about --cpu-seconds in total:
      index    k-mer index of the whole assembly (what every damapper job builds first),
      map      a sample of reads against it (marginal cost per read-bp, extrapolated to all reads),
      process  `collect` + `process` of a sample of pile-ups, OpenMP over pile-ups (cost per
               pile-up, extrapolated to all).  Its input LAs come from the oracle's own mapping of
               the reads around the sampled gaps (not timed: it stands in for the mapping output)."""
    from dentist_amd import sim
    from oracle import pyoracle as oz
    cores = os.cpu_count() or 1
    o = oz.default_opts(width=mopts.width, kmer_mod=mopts.kmer_mod, k=mopts.k, xdrop=mopts.xdrop, algo=mopts.algo)

    def map_reads(n):
        sub = sim.SeqDb(w.reads.bases[:w.reads.off[n]], w.reads.off[:n + 1])
        t = time.perf_counter()
        oz.align_db(w.contigs, sub, o, nthreads=cores)
        return time.perf_counter() - t, int(sub.off[-1])

    # index build: a call with a single read is the index build plus one alignment
    t_index, _ = map_reads(1)
    t_probe, bp_probe = map_reads(min(w.reads.n, 4 * cores))
    rate = bp_probe / max(t_probe - t_index, 1e-3)
    n_map = int(min(w.reads.n, max(4 * cores, 0.15 * args.cpu_seconds * rate / (read_bp_total / max(w.nreads_total, 1)))))
    t_map, bp_map = map_reads(n_map)
    map_bp_s = bp_map / max(t_map - t_index, 1e-3)   # marginal rate, index build excluded
    # spread of the mapping rate: two more samples of a third of the size (different reads would need another DB:
    # the same reads are re-mapped, so this is the timing noise of the box, not of the data)
    rates = [map_bp_s]
    for _ in range(2 if t_index < 3.0 else 0):   # every call rebuilds the index: not repeated when that alone takes seconds
        t_s, bp_s = map_reads(max(4 * cores, n_map // 3))
        rates.append(bp_s / max(t_s - t_index, 1e-3))

    rec, las_all = last["rec"], last["las"]
    npiles = int(last["info"]["piles"])
    po = oz.default_process_opts(rounds=popts.rounds, max_reads=popts.max_reads, min_reads=popts.min_reads, algo=popts.algo,
                                 max_partners=popts.max_partners,
                                 **(dict(width=32) if (popts.algo == 1 and popts.width == 32) else {}))
    gaps_sorted = [int(r["contig_left"]) for r in rec]
    # the untimed stand-in for the mapping output may sample its k-mers (it only has to place the reads around the sampled gaps)
    o_standin = oz.default_opts(width=mopts.width, kmer_mod=max(8, mopts.kmer_mod), k=mopts.k, xdrop=mopts.xdrop, algo=mopts.algo)
    # a pile-up per OpenMP thread and batch; uncapped pile-ups (166 reads: ~100 core-seconds each) go 8 threads to a pile-up
    # (oracle/pile.c nests the all-vs-all's OpenMP loop), so that one batch stays within the budget
    budget, batch = 0.5 * args.cpu_seconds, max(cores, 8) if popts.max_reads else max(cores // 8, 4)
    done, t_proc, used = 0, 0.0, 0
    while done < len(gaps_sorted) and t_proc < budget:
        gs = gaps_sorted[done:done + batch]
        sel = np.isin(las_all["aread"], gs) | np.isin(las_all["aread"], [g + 1 for g in gs])
        rids = np.unique(las_all["bread"][sel]) - w.read_first
        rids = rids[(rids >= 0) & (rids < w.reads.n)]
        remap = sim.SeqDb.from_list([w.reads.seq(int(r)) for r in rids])
        ol, ot, _ = oz.align_db(w.contigs, remap, o_standin, nthreads=cores)   # not timed (mapping output stand-in)
        t = time.perf_counter()
        g2, tris = oz.collect_spanning_c(ol, w.contigs, po)
        keep = [i for i, g in enumerate(g2) if int(g) in set(gs)]
        oz.process_piles_c(w.contigs, remap, ol, ot, g2[keep], [tris[i] for i in keep], po, nthreads=cores)
        t_proc += time.perf_counter() - t
        used += len(keep)
        done += len(gs)
    per_pile = t_proc / max(used, 1)
    est_map, est_proc = t_index + read_bp_total / map_bp_s, per_pile * npiles
    return {"value": gap_bases / (est_map + est_proc), "unit": "gap-bp/s", "cores": cores, "kind": "port",
            "label": "CPU restatement of the same algorithm (C + OpenMP) at the same knobs as `value` "
                     f"(kmer_mod {mopts.kmer_mod}, max_reads {popts.max_reads}, max_partners {popts.max_partners}) -- not the reference binaries",
            "read_bp_mapped_per_sec": map_bp_s, "read_bp_mapped_per_sec_per_core": map_bp_s / cores,
            "read_bp_mapped_per_sec_samples": rates, "pile_ups_per_sec_per_core": 1.0 / max(per_pile, 1e-9) / cores,
            "index_build_s": t_index,
            "pile_ups_per_sec": 1.0 / max(per_pile, 1e-9),
            "estimated_seconds": {"mapping": est_map, "process": est_proc},
            "sample": f"index of the whole assembly ({t_index:.1f} s); {n_map} reads / {bp_map} bp mapped in "
                      f"{t_map:.1f} s incl. index; {used} pile-ups through collect + process in {t_proc:.1f} s "
                      f"(OpenMP over pile-ups, {cores} threads); extrapolated to {read_bp_total} read-bp and "
                      f"{npiles} pile-ups"}


if __name__ == "__main__":
    main()
