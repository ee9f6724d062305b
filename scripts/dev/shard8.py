"""dev: 8 emulated ranks of the sharded collect + process on configs[2] on ONE GPU; prints the host time
every rank spends between the collectives (the Python / host glue that does not shrink with N)."""
import sys, time
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, dentist_amd
from dentist_amd import sim
from dentist_amd.parallel import shard_range, sharded_process_steps
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
spec = bench.WORKLOADS["cfg2_100Mb_1000gaps_1Mx15kb"]
ctx = dentist_amd.Context(0)
# SHARD8_KMER_MOD / SHARD8_MAX_READS: 8 / 60 = the fast mode (round 5's emulation); 1 / 0 = the reference's behaviour
mo = dentist_amd.default_align_opts(kmer_mod=int(os.environ.get("SHARD8_KMER_MOD", "1")), k=20, width=64, xdrop=60, algo=1)
po = dentist_amd.default_process_opts(algo=1)
po.max_reads = int(os.environ.get("SHARD8_MAX_READS", "0"))
print("knobs: kmer_mod", mo.kmer_mod, "max_reads", po.max_reads, flush=True)
ranks = []
A = None
for r in range(N):
    lo, hi = shard_range(spec["nreads"], r, N)
    w = sim.Workload(seed=20260929, read_range=(lo, hi), **spec)
    if A is None:
        A = ctx.db(w.contigs)
    B = ctx.db(w.reads)
    t0 = time.perf_counter()
    ctx.map_reads(A, B, mo, po)
    t0 = time.perf_counter()
    las, trace, dropped = ctx.map_reads(A, B, mo, po)
    t1 = time.perf_counter()
    las["bread"] += lo
    ranks.append(dict(lo=lo, B=B, las=las, trace=trace, w=w, t_map=t1 - t0))
    print('rank', r, 'map+filter %.1f ms, las %d' % ((t1 - t0) * 1e3, len(las)), flush=True)
ncg = ranks[0]["w"].contigs.n
ROUNDS = int(os.environ.get("SHARD8_ROUNDS", "2"))   # the last round is reported: a process in steady state, as bench.py times it
gaps = np.stack([np.arange(ncg - 1), np.arange(1, ncg)], axis=1).astype(np.int32)
graph = len(sys.argv) <= 2 or sys.argv[2] != "spanning"
for rnd in range(ROUNDS):
  gens = [sharded_process_steps(ctx, A, R["B"], R["lo"], R["w"].contigs.off, R["las"], R["trace"], po, r, N,
                                graph=dict(read_off=R["w"].reads.off, input_gaps=gaps) if graph else None) for r, R in enumerate(ranks)]
  phase_t = [[] for _ in range(N)]
  reqs = []
  for r, g in enumerate(gens):
      t = time.perf_counter(); reqs.append(next(g)); phase_t[r].append(time.perf_counter() - t)
  results = [None] * N
  while any(q is not None for q in reqs):
      kind = next(q[0] for q in reqs if q is not None)
      if kind == "all_gather":
          answers = [[np.asarray(reqs[s][1], dtype=np.uint8) for s in range(N)] for _ in range(N)]
      else:
          answers = [[np.asarray(reqs[s][1][d], dtype=np.uint8) for s in range(N)] for d in range(N)]
      print(kind, 'bytes per rank', [int(np.asarray(q[1]).nbytes) if kind == "all_gather" else int(sum(len(x) for x in q[1])) for q in reqs][:3], flush=True)
      nxt = []
      for r, g in enumerate(gens):
          t = time.perf_counter()
          try:
              nxt.append(g.send(answers[r]))
              if kind == "all_to_all":
                  print('rank', r, 'process stages', {k: round(v, 1) for k, v in dentist_amd.process_stats(ctx).items() if k.startswith('ms_')}, flush=True)
          except StopIteration as done:
              results[r] = done.value; nxt.append(None)
          phase_t[r].append(time.perf_counter() - t)
      reqs = nxt
for r in range(N):
    print('rank', r, 'map %.1f' % (ranks[r]["t_map"] * 1e3), 'phases ms', [round(x * 1e3, 1) for x in phase_t[r]], 'total %.1f' % (1e3 * (ranks[r]["t_map"] + sum(phase_t[r]))))
rec = results[0][0]
print('closed', int((rec["status"] == 0).sum()), 'of', len(rec))
