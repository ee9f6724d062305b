"""The N > 1 path on CPU: world_size 2, gloo backend (the same code runs over RCCL on GPUs)."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from dentist_amd import INSERTION_DTYPE
    from dentist_amd.parallel import all_gather_closed_gaps, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # rank r "closed" r + 2 gaps with consensus lengths 10 * (i + 1) + r
    n = rank + 2
    rec = np.zeros(n, dtype=INSERTION_DTYPE)
    bases, off = [], 0
    for i in range(n):
        ln = 10 * (i + 1) + rank
        rec[i]["contig_left"] = 100 * rank + i
        rec[i]["cons_len"] = ln
        rec[i]["cons_off"] = off
        bases.append(np.full(ln, (rank + i) % 4, dtype=np.uint8))
        off += ln
    allrec, allbases, origin = all_gather_closed_gaps(rec, np.concatenate(bases), rank, world)
    lo, hi = shard_range(10, rank, world)
    q.put((rank, allrec.tobytes(), allbases.tobytes(), origin.tolist(), (lo, hi)))
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_closed_gaps_world2():
    from dentist_amd import INSERTION_DTYPE
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    out.sort()
    assert out[0][1] == out[1][1] and out[0][2] == out[1][2], "ranks disagree on the gathered result"
    rec = np.frombuffer(out[0][1], dtype=INSERTION_DTYPE)
    bases = np.frombuffer(out[0][2], dtype=np.uint8)
    assert out[0][3] == [0, 0, 1, 1, 1]
    assert rec["contig_left"].tolist() == [0, 1, 100, 101, 102]
    for r, org in zip(rec, out[0][3]):
        i = int(r["contig_left"]) % 100
        seq = bases[r["cons_off"]:r["cons_off"] + r["cons_len"]]
        assert len(seq) == 10 * (i + 1) + org and np.all(seq == (org + i) % 4)
    assert out[0][4] == (0, 5) and out[1][4] == (5, 10)


def _worker_bytes(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from dentist_amd.parallel import all_gather_bytes, all_to_all_bytes
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = np.arange(7 * rank + 3, dtype=np.uint8) + rank          # ragged; rank 0 sends 3, rank 1 sends 10 bytes
    gathered = all_gather_bytes(mine, world)
    per_dest = [np.full(5 * rank + dst, 10 * rank + dst, dtype=np.uint8) for dst in range(world)]  # (0,0) is empty
    got = all_to_all_bytes(per_dest, world)
    q.put((rank, [g.tolist() for g in gathered], [g.tolist() for g in got]))
    dist.barrier()
    dist.destroy_process_group()


def test_ragged_all_gather_and_all_to_all_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bytes, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, gathered, got in out:
        assert gathered == [list(range(3)), [x + 1 for x in range(10)]]
        assert got == [[10 * src + rank] * (5 * src + rank) for src in range(2)]


def test_owner_assignment_and_candidate_merge_are_deterministic():
    from dentist_amd import LA_DTYPE
    from dentist_amd.parallel import CAND_DTYPE, assign_owners, emulate_ranks, merge_candidates
    owner = assign_owners([5, 9, 9, 1, 4, 4], 3)
    assert owner.tolist() == [2, 0, 1, 1, 2, 0]     # largest first: 9->r0, 9->r1, 5->r2, 4->r2, 4->r0, 1->r1
    loads = [sum(c for c, o in zip([5, 9, 9, 1, 4, 4], owner) if o == r) for r in range(3)]
    assert max(loads) - min(loads) <= 4
    # candidates of two ranks (rank 1 holds the higher read ids); gap 7 appears on both
    def cand(gap, read, ab):
        c = np.zeros(1, dtype=CAND_DTYPE)
        c["gap"], c["read"] = gap, read
        c["L"]["abpos"], c["R"]["abpos"] = ab, ab + 1
        return c
    r0 = np.concatenate([cand(3, 1, 10), cand(7, 0, 20), cand(7, 2, 30)])
    r1 = np.concatenate([cand(7, 5, 40), cand(9, 6, 50)])
    las, gaps, counts, triples = merge_candidates([r0, r1])
    assert gaps.tolist() == [3, 7, 9] and counts.tolist() == [1, 3, 1] and las.dtype == LA_DTYPE
    assert triples[1:4, 0].tolist() == [0, 2, 5]     # read-id order across the ranks
    for read, il, ir in triples.tolist():
        assert ir == il + 1 and las[ir]["abpos"] == las[il]["abpos"] + 1
    # the lockstep driver: two generators exchanging through "collectives"
    def gen(rank):
        got = yield ("all_gather", np.asarray([rank + 1], dtype=np.uint8))
        s = sum(int(g[0]) for g in got)
        got = yield ("all_to_all", [np.asarray([10 * rank + d], dtype=np.uint8) for d in range(2)])
        return s, [int(g[0]) for g in got]
    assert emulate_ranks([gen(0), gen(1)]) == [(3, [0, 10]), (3, [1, 11])]
