#!/usr/bin/env python3
"""bench.py -- the hot path of DENTIST (alignment + consensus) on N MI355X, one JSON line.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched through
torch.distributed.run, one rank per GPU (RCCL).

A "step" is one pass of the whole hot path over one batch of synthetic input resident in HBM:
  1. every read of the rank's read block is aligned to the rank's assembly (k-mer index build,
     seed filter, wave alignment with trace points, LAs back on the host)  -- damapper's role,
  2. the spanning reads of every gap are collected (host),
  3. every pile-up goes through crop -> pile-up all-vs-all alignment -> filters -> tile QV ->
     reference read -> consensus rounds -> flank re-alignment -> insertion      -- `dentist process`
     with daligner/DASqv/daccord replaced by kernels.
metric = gap-bases closed / second over the WHOLE step (mapping included), read-bp aligned/sec is
reported next to it.  Weak scaling: every rank owns one BASELINE configs[1]-sized block (its own
assembly region + reads), the way the reference shards (one damapper job per read block,
snakemake/Snakefile:1143-1170; one `process` job per pile-up batch, :1315-1334); the closed-gap
records are exchanged with one all-gather over RCCL (merge-insertions, Snakefile:1347-1358).
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
    # BASELINE.json configs[2] -- the configuration north_star quotes the target on: 100 Mb assembly,
    # 1 000 gaps, 1 M x 15 kb PacBio-error reads (15 Gbp, 150x)
    "cfg2_100Mb_1000gaps_1Mx15kb": dict(genome_len=100_000_000, ngaps=1000, nreads=1_000_000, read_len=15_000),
    # BASELINE.json configs[1]: 10 Mb assembly, 100 gaps, 100 k x 10 kb PacBio-error reads
    "cfg1_10Mb_100gaps_100kx10kb": dict(genome_len=10_000_000, ngaps=100, nreads=100_000, read_len=10_000),
    # reduced shape for quick checks (not a bench line)
    "dev_1Mb_10gaps_5kx10kb": dict(genome_len=1_000_000, ngaps=10, nreads=5_000, read_len=10_000),
}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def closed_gap_stats(w, rec, bases, check_identity):
    """Gap bases closed (insertions that passed every gate) and, outside the timed region, their
    edit distance to the truth."""
    from dentist_amd import sim
    closed = rec[rec["status"] == 0]
    gap_bases = int((closed["ins_end"] - closed["ins_begin"]).sum())
    edits = truth_bases = 0
    if check_identity:
        from oracle import pyoracle as oz  # checker only, never timed
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
    ap.add_argument("--cpu-sample-reads", type=int, default=3000)
    ap.add_argument("--cpu-sample-gaps", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kmer-mod", type=int, default=4)
    ap.add_argument("--map-k", type=int, default=20, help="k-mer length of the mapping pass (damapper's default)")
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
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    import dentist_amd
    from dentist_amd import sim
    from dentist_amd.parallel import all_gather_closed_gaps

    spec = WORKLOADS[args.workload]
    # every rank owns its own block: assembly region + reads (SURVEY 8(d) seeds, shifted by rank)
    w = sim.Workload(seed=20260929 + 1000 * rank, **spec)
    stream = torch.cuda.current_stream().cuda_stream
    ctx = dentist_amd.Context(local_rank, stream=stream)
    A, B = ctx.db(w.contigs), ctx.db(w.reads)
    # mapping pass: modimer sampling 1/4 (daligner's -%), every other option at its default
    mopts = dentist_amd.default_align_opts(kmer_mod=args.kmer_mod, k=args.map_k)
    popts = dentist_amd.default_process_opts()
    read_bp = int(len(w.reads.bases))

    def step():
        A.drop_cache()  # the k-mer index and the reverse complement are rebuilt every step
        B.drop_cache()
        ctx.cum_stats(reset=True)
        t0 = time.perf_counter()
        las, trace = ctx.align_db(A, B, mopts, select_best=True)
        ast = ctx.align_stats()
        t1 = time.perf_counter()
        piles = dentist_amd.Pileups(las, w.contigs.off, popts)
        t2 = time.perf_counter()
        rec, bases = dentist_amd.process_pileups(ctx, A, B, las, trace, piles, popts)
        t3 = time.perf_counter()
        pst = dentist_amd.process_stats(ctx)
        cum = ctx.cum_stats().as_dict()  # every k_wave / k_seed launch of the step
        gathered = all_gather_closed_gaps(rec, bases, rank, world) if world > 1 else None
        return dict(las=las, rec=rec, bases=bases, ast=ast, pst=pst, cum=cum, npiles=len(piles), gathered=gathered,
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
        runs.append(step())
        if len(runs) > 1:  # only the last step's results are inspected: release the earlier buffers
            for key in ("las", "rec", "bases", "gathered"):
                runs[-2].pop(key, None)
    barrier()
    dt = time.perf_counter() - t0

    last = runs[-1]
    gap_bases, nclosed, edits, truth_bases = closed_gap_stats(w, last["rec"], last["bases"], True)
    aligned_bp = int((last["las"]["aepos"] - last["las"]["abpos"]).sum())
    if world > 1:
        tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        tot = torch.tensor([gap_bases, nclosed, edits, truth_bases, aligned_bp, read_bp, len(last["rec"])],
                           device="cuda", dtype=torch.int64)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        gap_all, nclosed_all, edits_all, truth_all, aligned_all, read_all, npiles_all = (int(x) for x in tot.tolist())
    else:
        gap_all, nclosed_all, edits_all, truth_all, aligned_all, read_all, npiles_all = (
            gap_bases, nclosed, edits, truth_bases, aligned_bp, read_bp, len(last["rec"]))

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        mean = lambda f: float(np.mean([f(r) for r in runs]))  # noqa: E731
        # dominant kernel of the step: k_wave2 (mapping launch + pile-up all-vs-all launch + the small
        # re-alignment and flank launches).  Algorithmic bytes of a launch = both sequences of every
        # alignment it emits streamed once (2 B per aligned A base at one byte per base) + its trace
        # (2 B per trace value); summed over the step's launches and divided by their summed
        # HIP-event durations, i.e. the per-launch average weighted by work.
        cum = last["cum"]
        wave_ms = mean(lambda r: r["cum"]["ms_wave"])
        alg_bytes = 2.0 * cum["aligned_bp"] + 2.0 * cum["trace_values"]
        achieved = alg_bytes / (wave_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "k_wave_traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if tj.get("workload") == args.workload and tj.get("mapping_kmer_mod") == args.kmer_mod:
                traffic = tj["hbm_bytes_per_step"] / max(1, cum["wave_launches"])
        out = {
            "metric": "gap-bases closed/sec",
            "value": gap_all * args.steps / dt,
            "unit": "gap-bp/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": args.workload, "per_gpu": spec, "mapping_kmer_mod": args.kmer_mod, "mapping_k": args.map_k,
                       "read_bp_total": read_all,
                       "pile_ups": npiles_all, "gaps_closed": nclosed_all, "gap_bases_closed": gap_all,
                       "consensus_edit_distance_vs_truth": edits_all, "consensus_truth_bases": truth_all,
                       "consensus_error_rate": (edits_all / truth_all) if truth_all else None},
            "read_bp_aligned_per_sec": aligned_all * args.steps / dt,
            "read_bp_aligned_per_sec_mapping_stage": aligned_bp / mean(lambda r: r["t_map"]),
            "roofline": {"bound": "hbm", "kernel": "k_wave2", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "launches_per_step": cum["wave_launches"], "kernel_ms_per_step": wave_ms,
                         "avg_launch_ms": wave_ms / max(1, cum["wave_launches"]),
                         "algorithmic_bytes_per_step": alg_bytes,
                         "wave_cells_per_s": cum["wave_cells"] / (wave_ms * 1e-3),
                         "note": "integer VALU/latency-bound by nature (SURVEY 7d); cell updates/s is the "
                                 "honest secondary"},
            # second kernel of the step: the seed filter of the mapping launch is bound by random
            # 64-byte directory lines: per read base and strand 1 B of sequence
            # + one 64 B line per sampled k-mer
            "roofline_seed": (lambda ms, b: {"bound": "hbm", "kernel": "k_seed (mapping launch)",
                                            "achieved": b / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                            "frac": b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                            "algorithmic_bytes_per_launch": b, "avg_launch_ms": ms,
                                            "note": "random 64 B accesses: ~45 % of peak is the practical ceiling"})(
                mean(lambda r: r["ast"].ms_seed), 2.0 * read_bp * (1.0 + 64.0 / max(1, args.kmer_mod))),
            "stages_ms": {"map_wall": mean(lambda r: r["t_map"]) * 1e3,
                          "map_index": mean(lambda r: r["ast"].ms_index),
                          "map_seed": mean(lambda r: r["ast"].ms_seed),
                          "map_wave": mean(lambda r: r["ast"].ms_wave),
                          "all_wave": wave_ms, "all_seed": mean(lambda r: r["cum"]["ms_seed"]),
                          "map_gather": mean(lambda r: r["ast"].ms_gather),
                          "collect_wall": mean(lambda r: r["t_collect"]) * 1e3,
                          "process_wall": mean(lambda r: r["t_process"]) * 1e3,
                          **{"process_" + k[3:]: mean(lambda r, k=k: r["pst"][k]) for k in last["pst"] if k.startswith("ms_")}},
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w, last, mopts, popts, args)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(w, last, mopts, popts, args):
    """The oracle ("port") timed on this box's host cores on a bounded sample of the same block:
    mapping of the first `cpu_sample_reads` reads against the whole assembly (index build
    included) and `process` of the first `cpu_sample_gaps` pile-ups; both legs are extrapolated to
    the block (reads: by read-bp, pile-ups: by count) to give gap-bases closed per second."""
    from dentist_amd import sim
    from oracle import process as pr
    from oracle import pyoracle as oz
    cores = os.cpu_count() or 1
    n = min(args.cpu_sample_reads, w.reads.n)
    sub = sim.SeqDb(w.reads.bases[:w.reads.off[n]], w.reads.off[:n + 1])
    o = oz.default_opts(width=mopts.width, kmer_mod=mopts.kmer_mod, k=mopts.k)
    t0 = time.perf_counter()
    oz.align_db(w.contigs, sub, o, nthreads=cores)
    t_map = time.perf_counter() - t0
    map_bp_s = float(sub.off[-1]) / t_map
    las = last["las"]
    trace_dummy = None
    # pile-ups of the sample gaps from the (bit-identical) LAs; traces are needed: redo the mapping of
    # just those reads with the oracle so the baseline is self-contained
    gaps = sorted(set(int(g) for g in last["rec"]["contig_left"][:args.cpu_sample_gaps]))
    rids = sorted(set(int(r) for r in las["bread"][np.isin(las["aread"], gaps) | np.isin(las["aread"], [g + 1 for g in gaps])]))
    remap = sim.SeqDb.from_list([w.reads.seq(r) for r in rids])
    ol, ot, _ = oz.align_db(w.contigs, remap, o, nthreads=cores)
    t1 = time.perf_counter()
    piles = pr.collect_spanning(ol, ot, w.contigs, remap)
    done = closed = 0
    for g in gaps:
        if g not in piles:
            continue
        r = pr.process_pile(piles[g], ol, ot, w.contigs, remap, g, rounds=popts.rounds, nthreads=cores)
        done += 1
        if r["status"] == "ok":
            closed += len(r["insertion"])
    t_proc = time.perf_counter() - t1
    npiles = len(last["rec"])
    gap_bases = int((last["rec"]["ins_end"] - last["rec"]["ins_begin"])[last["rec"]["status"] == 0].sum())
    est_total = float(len(w.reads.bases)) / map_bp_s + (t_proc / max(done, 1)) * npiles
    return {"value": gap_bases / est_total, "unit": "gap-bp/s", "cores": cores, "kind": "port",
            "read_bp_mapped_per_sec": map_bp_s,
            "sample": f"mapping: first {n} reads vs the whole assembly incl. index build ({t_map:.1f} s); "
                      f"process: {done} pile-ups ({t_proc:.1f} s); extrapolated to the block "
                      f"({len(w.reads.bases)} read-bp, {npiles} pile-ups)"}


if __name__ == "__main__":
    main()
