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
