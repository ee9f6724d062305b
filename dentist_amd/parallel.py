"""Multi-GPU layer: one process per GPU, pile-ups / read blocks sharded with no data-path
collective; the only exchange is the gather of the closed-gap records at the end -- the role of
`dentist merge-insertions` (source/dentist/commands/mergeInsertions.d:60-164, workflow rule
snakemake/Snakefile:1347-1358).  Payload is a few MB, so it is latency-bound: one all-gather of
the sizes, one all-gather of the padded bytes (RCCL over xGMI on GPUs, gloo in the CPU tests).
"""
import numpy as np

from ._lib import INSERTION_DTYPE


def shard_range(n, rank, world):
    """Contiguous block partition of n units (read blocks / pile-up batches) over the ranks."""
    return n * rank // world, n * (rank + 1) // world


def all_gather_closed_gaps(rec, bases, rank, world, device=None):
    """Every rank receives all ranks' insertion records and consensus bases.

    Returns (records, bases, origin): records concatenated in rank order with ``cons_off``
    rebased into the concatenated ``bases``; ``origin[i]`` = rank that produced record i.
    """
    import torch
    import torch.distributed as dist

    if world == 1 or not dist.is_initialized():
        return rec.copy(), bases.copy(), np.zeros(len(rec), dtype=np.int32)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    rb = np.frombuffer(np.ascontiguousarray(rec, dtype=INSERTION_DTYPE).tobytes(), dtype=np.uint8)
    bb = np.ascontiguousarray(bases, dtype=np.uint8)
    sizes = torch.tensor([len(rb), len(bb)], dtype=torch.int64, device=device)
    all_sizes = [torch.zeros(2, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(all_sizes, sizes)
    all_sizes = [tuple(int(x) for x in s.tolist()) for s in all_sizes]
    cap = max(1, max(a + b for a, b in all_sizes))
    payload = torch.zeros(cap, dtype=torch.uint8, device=device)
    mine = np.concatenate([rb, bb])
    if len(mine):
        payload[:len(mine)] = torch.from_numpy(mine.copy()).to(device)
    gathered = [torch.zeros(cap, dtype=torch.uint8, device=device) for _ in range(world)]
    dist.all_gather(gathered, payload)
    recs, seqs, origin, base_off = [], [], [], 0
    for r, (nr, nb) in enumerate(all_sizes):
        buf = gathered[r].cpu().numpy()
        rr = np.frombuffer(buf[:nr].tobytes(), dtype=INSERTION_DTYPE).copy()
        rr["cons_off"] += base_off
        recs.append(rr)
        seqs.append(buf[nr:nr + nb].copy())
        origin.append(np.full(len(rr), r, dtype=np.int32))
        base_off += nb
    return np.concatenate(recs), np.concatenate(seqs), np.concatenate(origin)
