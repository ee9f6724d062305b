#!/usr/bin/env python3
"""bench.py -- the hot path of DENTIST (alignment + consensus) on N MI355X, one JSON line.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched through
torch.distributed.run, one rank per GPU (RCCL).  A "step" is one pass of the hot path over one
batch of synthetic input resident in HBM: every read of this rank's read block is aligned to the
rank's assembly (k-mer index build included), results copied back to the host.  Weak scaling:
every rank owns one BASELINE configs[1]-sized block (its own assembly region + reads), exactly
how the reference shards (one damapper job per read block, snakemake/Snakefile:1143-1170).
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
    # BASELINE.json configs[1]: 10 Mb assembly, 100 gaps, 100 k x 10 kb PacBio-error reads
    "cfg1_10Mb_100gaps_100kx10kb": dict(genome_len=10_000_000, ngaps=100, nreads=100_000, read_len=10_000),
    # reduced shape for quick checks (not a bench line)
    "dev_1Mb_10gaps_5kx10kb": dict(genome_len=1_000_000, ngaps=10, nreads=5_000, read_len=10_000),
}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg1_10Mb_100gaps_100kx10kb")
    ap.add_argument("--cpu-sample-reads", type=int, default=4000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
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

    spec = WORKLOADS[args.workload]
    # every rank owns its own block: assembly region + reads (SURVEY 8(d) seeds, shifted by rank)
    w = sim.Workload(seed=20260929 + 1000 * rank, **spec)
    stream = torch.cuda.current_stream().cuda_stream
    ctx = dentist_amd.Context(local_rank, stream=stream)
    A, B = ctx.db(w.contigs), ctx.db(w.reads)
    opts = dentist_amd.default_align_opts()
    read_bp = int(len(w.reads.bases))

    def step():
        A.drop_cache()  # the k-mer index and the reverse complement are rebuilt every step
        B.drop_cache()
        las, trace = ctx.align_db(A, B, opts, select_best=True)
        return las, trace, ctx.align_stats()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    wave_ms, seed_ms, index_ms, gather_ms, cells, aligned_bp = [], [], [], [], 0, 0
    for _ in range(args.steps):
        las, trace, st = step()
        wave_ms.append(st.ms_wave)
        seed_ms.append(st.ms_seed)
        index_ms.append(st.ms_index)
        gather_ms.append(st.ms_gather)
        cells = st.wave_cells
        aligned_bp = int((las["aepos"] - las["abpos"]).sum())
        nla = len(las)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        tot = torch.tensor([aligned_bp, read_bp, nla], device="cuda", dtype=torch.int64)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        aligned_all, read_all, nla_all = (int(x) for x in tot.tolist())
    else:
        aligned_all, read_all, nla_all = aligned_bp, read_bp, nla

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = aligned_all * args.steps / dt
        # dominant kernel: k_wave (one launch per step at this size). Algorithmic bytes: both
        # sequences of every alignment streamed once (2 B per aligned A base) + its trace
        # (2 x u16 per tspace bases) -- DESIGN.md "Roofline".
        wave_s = float(np.mean(wave_ms)) * 1e-3
        alg_bytes = aligned_bp * 2.0 + aligned_bp / opts.tspace * 4.0
        achieved = alg_bytes / wave_s / 1e9
        out = {
            "metric": "read-bp aligned/sec",
            "value": value,
            "unit": "bp/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": args.workload, "per_gpu": spec, "read_bp_per_gpu": read_bp,
                       "local_alignments": nla_all, "read_bp_total": read_all},
            "roofline": {"bound": "hbm", "kernel": "k_wave", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "wave_cells_per_s": cells / wave_s, "kernel_ms": wave_s * 1e3},
            "stages_ms": {"index": float(np.mean(index_ms)), "seed": float(np.mean(seed_ms)),
                          "wave": float(np.mean(wave_ms)), "gather": float(np.mean(gather_ms))},
        }
        if not args.no_cpu_baseline:
            from oracle import pyoracle as oz
            n = min(args.cpu_sample_reads, w.reads.n)
            sub = sim.SeqDb(w.reads.bases[:w.reads.off[n]], w.reads.off[:n + 1])
            o = oz.default_opts(width=opts.width)
            cores = os.cpu_count() or 1
            t1 = time.perf_counter()
            cl, _, _ = oz.align_db(w.contigs, sub, o, nthreads=cores)
            ct = time.perf_counter() - t1
            out["cpu_baseline"] = {"value": float((cl["aepos"] - cl["abpos"]).sum()) / ct, "unit": "bp/s",
                                   "cores": cores, "kind": "port",
                                   "sample": f"first {n} reads of the same block against the same assembly, "
                                             f"index build included, {ct:.1f} s"}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
